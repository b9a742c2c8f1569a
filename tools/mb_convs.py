import sys, os
sys.argv=[sys.argv[0]]
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[os.path.join(R,'tools')]
import microbench as mb
for what in ('fwd','dgrad','wgrad'):
    mb.conv_case(1024, 64, 64, 17, 3, 1, 1, what)
    mb.conv_case(1024, 128, 128, 9, 3, 1, 1, what)
    mb.conv_case(1024, 256, 256, 5, 3, 1, 1, what)
    mb.conv_case(1024, 512, 512, 3, 3, 1, 1, what)
    mb.conv_case(1024, 64, 128, 17, 3, 2, 1, what)
    mb.conv_case(1024, 128, 256, 9, 3, 2, 1, what)
    mb.conv_case(1024, 256, 512, 5, 3, 2, 1, what)
    mb.conv_case(1024, 64, 128, 17, 1, 2, 0, what)
