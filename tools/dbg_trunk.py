import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[R, os.path.join(R,'audio-visual-vad_amd'), os.path.join(R,'tests')]
import numpy as np, torch, stategen
from oracle import resnet18
from avvad import nn as avnn
from packages.models.Video_Net import DeepVAD_video
import test_gpu_parity as tg
sd0 = tg._video_state()
N=6
x = stategen.rand(21, N, 67, 67); G = stategen.rand(22, N, 512)
sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone()) for k, v in sd0.items() if k.startswith("features.")}
ref = resnet18.trunk_forward(sd, x[:, None].repeat(1, 3, 1, 1), True)
(ref * G).sum().backward()
for mode in ['none','all','none','all','none','1','1','0','0','none','all']:
    os.environ['AVVAD_NO_STREAMK']=mode
    m = DeepVAD_video(2, 16, 1); m.load_state_dict(sd0); m = m.cuda().train()
    f = avnn.trunk_forward(m.features, x.cuda(), True)
    (f * G.cuda()).sum().backward()
    bad=[]
    for k, p in m.features.named_parameters():
        r = sd["features." + k].grad.numpy().astype(np.float64); g = p.grad.cpu().numpy().astype(np.float64)
        rel = np.linalg.norm(g-r)/np.linalg.norm(r)
        if rel > 1e-5: bad.append((k, float(rel), float(np.abs(g-r).max())))
    print("no-streamk-for-mode", mode, "fwd err", float((f.detach().cpu()-ref).abs().max()), "nbad", len(bad), [b[0] for b in bad[-6:]])
