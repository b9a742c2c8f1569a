import sys, os
sys.argv=[sys.argv[0]]
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[os.path.join(R,'tools')]
import microbench as mb
for what in ('fwd','dgrad','wgrad'):
    mb.conv_case(4096, 256, 256, 5, 3, 1, 1, what)
mb.conv_case(1024, 64, 128, 17, 3, 2, 1, 'dgrad')
