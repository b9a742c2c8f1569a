"""Lab: per-launch time of one encoder residual block (avvad_wavenet_block_fwd) under each kernel form, bench and C2 shapes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "audio-visual-vad_amd")):
    sys.path.insert(0, p)
import torch
from avvad import _lib as L
lib = L.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(B, Lin, dil, form, reps=20):
    L.set_option("wn_flat", form)
    s_in = torch.randn(B, 32, Lin, device="cuda"); out = torch.empty(B, 32, Lin - dil, device="cuda")
    wd, bd = torch.randn(32, 32, 2, device="cuda") * .1, torch.randn(32, device="cuda") * .1
    we, be = torch.randn(32, 32, 1, device="cuda") * .1, torch.randn(32, device="cuda") * .1
    f = lambda: lib.avvad_wavenet_block_fwd(L.ptr(s_in), L.ptr(wd), L.ptr(bd), L.ptr(we), L.ptr(be), L.ptr(out), B, Lin, dil, st)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    byts = 4.0 * B * 32 * (2 * Lin - dil)
    return us, byts / us / 1e6
for (B, Lin) in ((64, 6079), (256, 15936)):
    for dil in (64, 512):
        row = []
        for form in (4, 3, 5):
            for g in ((0,) if form != 5 else (0, 256, 1024)):
                L.set_option("wn_grid", g)
                us, gbs = run(B, Lin, dil, form)
                row.append("form%d%s %.1f us %.0f GB/s" % (form, "" if not g else "/g%d" % g, us, gbs))
        L.set_option("wn_grid", 0)
        print("B=%d Lin=%d d=%d: " % (B, Lin, dil) + " | ".join(row), flush=True)
