"""bench.py's parity probe (ragged slice of the benched model after 63 training steps vs the CPU oracle) under the kernel-path
options: does cpu_ref_max_abs_delta depend on which convolution kernels ran?   python tools/lab/probe_paths.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-visual-vad_amd"))
import torch
import bench
from avvad import _lib as L
from avvad.optim import FlatAdam
from packages.models.AV_Net import DeepVAD_AV
from packages.models.utils import batch_binary_cross_entropy

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = DeepVAD_AV(2, 1024, 1, use_mcb=False, eps=1e-8, wavenet_params=bench.w0(16)).to(dev).train()
wave, video, target, lengths = bench.make_inputs(torch, 64, 1234, dev, T=16, L=6143, kind="av")
opt = FlatAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.999))
for _ in range(63):
    loss = batch_binary_cross_entropy(model(wave, video, lengths), target, lengths, 1e-8)
    loss.backward()
    opt.step()
    opt.zero_grad()
torch.cuda.synchronize()
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
for opts in ({}, {"no_conv64": 1}, {"no_s2_cls": 1}, {"no_cls": 1}, {"no_conv64": 1, "no_cls": 1, "no_fused_stats": 1, "no_streamk": 1}):
    for k, v in opts.items():
        L.set_option(k, v)
    d, m = bench.parity_probe(torch, model)
    print("%-70s max|delta| %.3e on max|logit| %.2f" % (opts or "default", d, m), flush=True)
    for k in opts:
        L.set_option(k, 0)
