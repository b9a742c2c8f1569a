"""debug: which tensors break the C4 shard-sum property, with and without stream-K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-visual-vad_amd"), os.path.join(ROOT, "tests")]
import torch
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from avvad import _lib as L
from packages.models.AV_Net import DeepVAD_AV
from packages.models.utils import batch_binary_cross_entropy
DEV = "cuda:0"
torch.manual_seed(0)
m = DeepVAD_AV(2, 1024, 1, wavenet_params=bench.W0).to(DEV).eval()
wave, video, target, lengths = bench.make_inputs(torch, 64, 1234, torch.device(DEV))
lengths = lengths.clone(); lengths[::3] = 11
named = [(n, p) for n, p in m.named_parameters() if not n.startswith("bn.")]
def grads(sl):
    for _, p in named: p.grad = None
    y = m(wave[sl], video[sl], lengths[sl])
    loss = batch_binary_cross_entropy(y, target[sl], lengths[sl], 1e-8)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), [p.grad.clone() for _, p in named]
for nsk in (0, 1):
    L.set_option("no_streamk", nsk)
    lf, gf = grads(slice(0, 64)); l0, g0 = grads(slice(0, 32)); l1, g1 = grads(slice(32, 64))
    lf2, gf2 = grads(slice(0, 64))
    print("no_streamk=%d loss %.4f = %.4f + %.4f" % (nsk, lf, l0, l1))
    for (n, _), f, a, b, f2 in zip(named, gf, g0, g1, gf2):
        rel = float(((a + b) - f).norm() / f.norm().clamp_min(1e-30))
        rep = float((f2 - f).norm() / f.norm().clamp_min(1e-30))
        if rel > 1e-4 or rep > 0:
            print("   %-50s shard-sum relL2 %.2e   run-to-run %.2e  |g| %.3e" % (n, rel, rep, float(f.norm())))
