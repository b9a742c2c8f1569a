import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[R, os.path.join(R,'audio-visual-vad_amd'), os.path.join(R,'tests')]
import numpy as np, torch, stategen
from avvad import nn as avnn, ops
from packages.models.Video_Net import DeepVAD_video
import test_gpu_parity as tg
sd0 = tg._video_state()
N=6
x = stategen.rand(21, N, 67, 67).cuda(); G = stategen.rand(22, N, 512).cuda()
saved=[]
orig=ops._ws
def rec(n,d):
    t=orig(n,d); t.fill_(1.0); saved.append(t); return t
ops._ws=rec
res={}
for mode in ('sk','nosk'):
    if mode=='nosk': os.environ['AVVAD_NO_STREAMK']='all'
    else: os.environ.pop('AVVAD_NO_STREAMK',None)
    m = DeepVAD_video(2, 16, 1); m.load_state_dict(sd0); m = m.cuda().train()
    f = avnn.trunk_forward(m.features, x, True)
    torch.cuda.synchronize()
    ws_fwd = saved[-1].clone()
    (f * G).sum().backward()
    torch.cuda.synchronize()
    res[mode]=(ws_fwd, saved[-1].clone(), {k:p.grad.clone() for k,p in m.features.named_parameters()})
a,b=res['nosk'],res['sk']

def al(n): return (n+63)//64*64
off=0
convs=[(1,64,7)]
cin=64
for st,c in enumerate((64,128,256,512)):
    for bb in range(2):
        convs.append((cin,c,3)); convs.append((c,c,3))
        if bb==0 and st>0: convs.append((cin,c,1))
        cin=c
for i,(ci,co,ks) in enumerate(convs):
    off+=al(ks*ks*ci*co)
    if i>0: off+=al(ks*ks*ci*co)
off+=4*al(20*512); off+=al(3*512); off+=al(256*2*512*2)
names=[]
hs=[67,34,17,9,5,3]
names.append(('c0',off,N*34*34*64)); off+=al(N*34*34*64)
names.append(('p0',off,N*17*17*64)); off+=al(N*17*17*64)
for st,c in enumerate((64,128,256,512)):
    for bb in range(2):
        n=N*hs[st+2]**2*c
        for j,nm in enumerate(('c1','a1','c2','cd','out')):
            if j==3 and not (bb==0 and st>0): continue
            names.append(('s%db%d.%s'%(st,bb,nm),off,n)); off+=al(n)
for nm,o,n in names:
    x=a[0][o:o+n]; y=b[0][o:o+n]
    print("%-10s maxabs %.3e  maxdiff %.3e  rel %.2e  flips %d"%(nm, float(x.abs().max()), float((x-y).abs().max()), float((x-y).abs().max()/x.abs().max()), int(((x>0)!=(y>0)).sum())))

nm,o,n=[t for t in names if t[0]=='s3b1.out'][0]
for tag,rr in (('nosk',a),('sk',b)):
    out=rr[0][o:o+n].view(N,9,512)
    dbeta=((G.view(N,1,512)/9.0)*(out>0).float()).sum((0,1))
    got=rr[2]['7.1.bn2.bias']
    print(tag,'7.1.bn2.bias rel err vs direct:', float((got-dbeta).norm()/dbeta.norm()), ' max', float((got-dbeta).abs().max()), float(dbeta.abs().max()))
cnt=0
for k in a[2]:
    r=a[2][k]; g=b[2][k]
    rel=float((r-g).norm()/r.norm())
    if rel>1e-5:
        cnt+=1
        if cnt<4 or k.startswith('7.1'): print(k, '%.1e'%rel, end=' | ')
print("ndiff", cnt)
k='7.1.bn2.bias'
print("a", a[2][k][:6].tolist()); print("b", b[2][k][:6].tolist())
d=(a[2][k]-b[2][k]).abs(); print("maxdiff", float(d.max()), int(d.argmax()), float(a[2][k].norm()), float(b[2][k].norm()), float((a[2][k]-b[2][k]).norm()))
