"""debug: stream-K (slab + fix-up) vs whole-tile schedule, component by component, same inputs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-visual-vad_amd"), os.path.join(ROOT, "tests")]
import torch
from avvad import _lib as L, ops, nn as avnn
from packages.models.Video_Net import DeepVAD_video
DEV = "cuda:0"
def rel(a, b): return float((a - b).norm() / b.norm().clamp_min(1e-30))
# ---- plain GEMMs of the head shapes
torch.manual_seed(0)
for (M, N, K, tA, tB, acc, split) in [(1024, 768, 4096, 0, 0, 0, 1), (512, 768, 4096, 0, 0, 0, 1), (4096, 768, 1024, 1, 0, 1, 8), (4096, 1024, 1024, 1, 0, 1, 8),
                                       (64, 1024, 4096, 0, 0, 1, 16), (1024, 4096, 768, 0, 1, 0, 1), (32, 1024, 4096, 0, 0, 1, 16)]:
    A = torch.randn((K, M) if tA else (M, K), device=DEV); B = torch.randn((N, K) if tB else (K, N), device=DEV)
    outs = []
    for nsk in (1, 0, 0):
        L.set_option("no_streamk", nsk)
        C = torch.ones(M, N, device=DEV) if acc else torch.empty(M, N, device=DEV)
        ops.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, bool(tA), bool(tB), accumulate=bool(acc), split_k=split)
        torch.cuda.synchronize(); outs.append(C)
    ref = (A.t() if tA else A).double() @ (B.t() if tB else B).double() + (1.0 if acc else 0.0)
    print("gemm %5dx%5dx%5d tA%d tB%d acc%d: whole-tile vs f64 %.2e | stream-K vs f64 %.2e | run-to-run %s" % (
        M, N, K, tA, tB, acc, rel(outs[0].double(), ref), rel(outs[1].double(), ref), torch.equal(outs[1], outs[2])))
# ---- trunk backward, eval mode, N = 1024 / 512
m = DeepVAD_video(1, 8, 1).to(DEV).eval()
for N in (1024, 512, 96):
    x = torch.randn(N, 67, 67, device=DEV); G = torch.randn(N, 512, device=DEV)
    res = []
    for nsk in (1, 0):
        L.set_option("no_streamk", nsk)
        for p in m.features.parameters(): p.grad = None
        f = avnn.trunk_forward(m.features, x, False)
        (f * G).sum().backward(); torch.cuda.synchronize()
        res.append((f.detach().clone(), {k: p.grad.clone() for k, p in m.features.named_parameters()}))
    print("trunk N=%d fwd stream-K vs whole-tile %.2e" % (N, rel(res[1][0], res[0][0])))
    bad = [(k, rel(res[1][1][k], res[0][1][k])) for k in res[0][1] if rel(res[1][1][k], res[0][1][k]) > 1e-4]
    print("   grads differing > 1e-4:", len(bad), bad[:6], bad[-3:])
