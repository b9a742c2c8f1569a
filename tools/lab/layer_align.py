"""Lab: does the row alignment of the [B][32][L] planes matter?  Same layer, L a multiple of 32 samples vs odd."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "audio-visual-vad_amd")):
    sys.path.insert(0, p)
import torch
from avvad import _lib as L
lib = L.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(B, Lin, dil, form, reps=30):
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
    return us, 4.0 * B * 32 * (2 * Lin - dil) / us / 1e6
for (B, Lin, dil) in ((64, 6144, 64), (64, 6143, 64), (64, 6079, 64), (64, 6144 + 448, 512), (64, 6079, 512), (256, 16000 + 64, 64), (256, 15936, 64), (256, 15937, 64), (256, 15999, 1), (256, 15487, 512)):
    print("B=%d Lin=%d Lo=%d (Lin%%32=%d, Lo%%32=%d): " % (B, Lin, Lin - dil, Lin % 32, (Lin - dil) % 32) +
          " | ".join("form%d %.1f us %.2f TB/s" % ((f,) + run(B, Lin, dil, f)) for f in (4, 3, 5)), flush=True)
