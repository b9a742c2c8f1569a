import sys, os
sys.argv=[sys.argv[0]]
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[os.path.join(R,'tools')]
import microbench as mb
for var in ('w8',):
    os.environ['AVVAD_IGEMM_VARIANT']=var
    print('variant',var)
    for n in (1024,):
        for what in ('fwd','dgrad','wgrad'):
            mb.conv_case(n, 64, 64, 17, 3, 1, 1, what)
            mb.conv_case(n, 128, 128, 9, 3, 1, 1, what)
            mb.conv_case(n, 256, 256, 5, 3, 1, 1, what)
            mb.conv_case(n, 512, 512, 3, 3, 1, 1, what)
    mb.gemm_case(4096,4096,4096,0,1)
