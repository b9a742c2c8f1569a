import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[R, os.path.join(R,'audio-visual-vad_amd'), os.path.join(R,'tests')]
import numpy as np, torch
from conftest import load_golden, wn_cfg_from
from packages.models.wavenet_autoencoder import wavenet_autoencoder
T=torch.from_numpy
for name in ['wn_w0_t16','wn_w0']:
    g=load_golden(name); cfg=wn_cfg_from(g)
    m=wavenet_autoencoder(**cfg); m.load_state_dict({k[2:]:T(v) for k,v in g.items() if k.startswith('p.')}); m=m.cuda()
    x=T(g['x']).cuda().requires_grad_(True)
    y=m(x); (y*T(g['G']).cuda()).sum().backward()
    e=np.abs(x.grad.cpu().numpy()-g['dx'])
    print(name,'dx err max',e.max(),'n>1e-5',(e>1e-5).sum(),'of',e.size)
    idx=np.argwhere(e>1e-5)
    print(' first/last bad idx', idx[:3].tolist(), idx[-3:].tolist())
    for k,p in m.named_parameters():
        r=g['g.'+k]; ee=np.abs(p.grad.cpu().numpy()-r)
        if ee.max()>1e-5*max(1,np.abs(r).max()): print('  ',k,'err',ee.max(),'ref max',np.abs(r).max())
