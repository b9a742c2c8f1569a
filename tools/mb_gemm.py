import sys, os
sys.argv=[sys.argv[0]]
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[os.path.join(R,'tools')]
import microbench as mb
for sk in ('', 'all'):
    if sk: os.environ['AVVAD_NO_STREAMK'] = sk
    print('no_streamk=', sk or 'off')
    mb.gemm_case(4096,4096,4096,1,0)
    mb.gemm_case(4096,4096,4096,0,1)
    mb.gemm_case(4096,4096,4096,0,0)
