"""GPU-side timeline of the two backward branches WITHOUT a profiler (HIP events around the two big backward calls):
   python tools/overlap_timeline.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "audio-visual-vad_amd")]
import torch
import bench
from avvad import ops
from avvad.optim import FlatAdam
from packages.models.AV_Net import DeepVAD_AV
from packages.models.utils import batch_binary_cross_entropy
marks = {}
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks[name] = e
def wrap(fn, tag):
    orig = fn.backward
    def b(ctx, *g):
        ev(tag + "_bwd_start"); r = orig(ctx, *g); ev(tag + "_bwd_end"); return r
    fn.backward = staticmethod(b)
    orig_f = fn.forward
    def f(ctx, *a):
        ev(tag + "_fwd_start"); r = orig_f(ctx, *a); ev(tag + "_fwd_end"); return r
    fn.forward = staticmethod(f)
wrap(ops.TrunkFn, "trunk"); wrap(ops.WavenetFn, "enc")
torch.manual_seed(0)
model = DeepVAD_AV(2, 1024, 1, wavenet_params=bench.W0).cuda().train()
wave, video, target, lengths = bench.make_inputs(torch, bench.N_SEQ, 1234, torch.device("cuda"))
opt = FlatAdam(model.parameters(), lr=1e-4)
def step(mark=False):
    if mark: ev("step_start")
    loss = batch_binary_cross_entropy(model(wave, video, lengths), target, lengths, 1e-8)
    loss.backward(); opt.step(); opt.zero_grad()
    if mark: ev("step_end")
for _ in range(6): step()
torch.cuda.synchronize()
step(True); torch.cuda.synchronize()
t0 = marks["step_start"]
for k in ("trunk_fwd_start", "trunk_fwd_end", "enc_fwd_start", "enc_fwd_end", "trunk_bwd_start", "trunk_bwd_end", "enc_bwd_start", "enc_bwd_end", "step_end"):
    print("%-18s %7.2f ms" % (k, t0.elapsed_time(marks[k])))
