import sys, os
sys.argv=[sys.argv[0]]
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[os.path.join(R,'tools')]
import microbench as mb
print("LIB", os.environ.get("AVVAD_LIB","default"))
for n in (1024,):
    mb.conv_case(n, 128, 128, 9, 3, 1, 1, 'fwd')
    mb.conv_case(n, 256, 256, 5, 3, 1, 1, 'fwd')
    mb.conv_case(n, 512, 512, 3, 3, 1, 1, 'fwd')
