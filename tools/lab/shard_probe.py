"""Lab: where does grad(full batch) != grad(shard 0) + grad(shard 1) come from under option bf16 (eval-mode BatchNorm)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "audio-visual-vad_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from avvad import _lib as L, nn as avnn
from packages.models.Video_Net import DeepVAD_video
DEV = "cuda:0"
mode = int(os.environ.get("BF16", "1"))
L.set_option("bf16", mode)
torch.manual_seed(0)
m = DeepVAD_video(1, 8, 1).to(DEV).eval()
N = int(os.environ.get("N", "256"))
x = torch.randn(N, 67, 67, device=DEV)
Gd = torch.randn(N, 512, device=DEV) * 1e-3
params = dict(m.features.named_parameters())
def grads(sl):
    for p in params.values():
        p.grad = None
    f = avnn.trunk_forward(m.features, x[sl], False)
    (f * Gd[sl]).sum().backward()
    return f.detach().clone(), {k: p.grad.clone() for k, p in params.items()}
ff, gf = grads(slice(0, N))
f0, g0 = grads(slice(0, N // 2))
f1, g1 = grads(slice(N // 2, N))
print("bf16 option", mode, "N", N)
print("features full vs shards: max rel", float((ff - torch.cat([f0, f1])).abs().max() / ff.abs().max()))
rels = sorted(((float((g0[k] + g1[k] - gf[k]).norm() / gf[k].norm().clamp_min(1e-30)), k) for k in gf), reverse=True)
print("worst:", rels[:5])
print("best:", rels[-3:])
