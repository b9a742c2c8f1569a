import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[R, os.path.join(R,'audio-visual-vad_amd')]
import numpy as np, torch
from avvad import ops
for (M,N,K) in [(54,512,4608),(648,128,1152),(1024,4096,768)]:
    rng=np.random.RandomState(1)
    A=rng.normal(size=(M,K)).astype(np.float32); B=rng.normal(size=(K,N)).astype(np.float32)
    ref=A.astype(np.float64)@B.astype(np.float64)
    a=torch.from_numpy(A).cuda(); b=torch.from_numpy(B).cuda()
    for mode in ('all','none'):
        os.environ['AVVAD_NO_STREAMK']=mode
        errs=[]
        for rep in range(3):
            c=torch.full((M,N), 7.0, device='cuda')
            ops.gemm(a,b,c,M,N,K,K,N,N)
            e=np.abs(c.cpu().numpy()-ref)
            errs.append((e.max(), np.sqrt((e**2).mean())))
        print(M,N,K,'no_streamk=',mode, ' max err %.3e rms %.3e | %.3e %.3e | %.3e %.3e'%(errs[0]+errs[1]+errs[2]), ' scale', np.abs(ref).max())
