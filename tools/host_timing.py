"""Host-side cost of the big C-ABI calls of one training step (is the single autograd thread the bottleneck?):
   python tools/host_timing.py"""
import os, sys, time, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "audio-visual-vad_amd")]
import torch
from avvad import _lib as L
lib = L.lib()
acc = {}
def wrap(name):
    f = getattr(lib, name)
    def g(*a):
        t0 = time.perf_counter(); r = f(*a); acc.setdefault(name, []).append((time.perf_counter() - t0, time.perf_counter()))
        return r
    return g
class Proxy:
    def __init__(self, lib): self._lib = lib; self._w = {}
    def __getattr__(self, n):
        if n not in self._w: self._w[n] = wrap(n) if n.startswith("avvad_") else getattr(self._lib, n)
        return self._w[n]
prox = Proxy(lib)
L.lib = lambda: prox
import bench
from avvad.optim import FlatAdam
from packages.models.AV_Net import DeepVAD_AV
from packages.models.utils import batch_binary_cross_entropy
torch.manual_seed(0)
model = DeepVAD_AV(2, 1024, 1, wavenet_params=bench.W0).cuda().train()
wave, video, target, lengths = bench.make_inputs(torch, bench.N_SEQ, 1234, torch.device("cuda"))
opt = FlatAdam(model.parameters(), lr=1e-4)
def step():
    loss = batch_binary_cross_entropy(model(wave, video, lengths), target, lengths, 1e-8)
    loss.backward(); opt.step(); opt.zero_grad()
for _ in range(5): step()
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("10 steps: host enqueue %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
for k, v in sorted(acc.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    print("  %-28s calls/step %5.1f  host %.3f ms/step" % (k, len(v) / 10, sum(x[0] for x in v) * 100))
