"""conv64 kernels vs the engine under a CU cap, at the two-rank test's layer-1 sizes (32 and 16 frames of 17x17)."""
import os, sys, ctypes as Ct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-vad_amd"))
import torch
from avvad import _lib as L, ops
lib = L.lib()
DEV = torch.device("cuda", 0)
st = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
ews = ops.engine_ws(DEV); wsz = ews.numel() * 4
for N in (32, 16, 24):
    d = L.ConvDesc(N, 17, 17, 64, 64, 3, 1, 1)
    torch.manual_seed(N)
    x = torch.randn(N, 17, 17, 64, device=DEV); gy = torch.randn(N, 17, 17, 64, device=DEV)
    w = torch.randn(64, 64, 3, 3, device=DEV) / 24.0
    wf = torch.empty(9 * 64 * 64, device=DEV); wdg = torch.empty(9 * 64 * 64, device=DEV)
    L.check(lib.avvad_conv2d_pack_weights(L.ptr(w), L.ptr(wf), L.ptr(wdg), Ct.byref(d), st), "pack")
    def run():
        y = torch.full((N, 17, 17, 64), float("nan"), device=DEV); dx = torch.full_like(y, float("nan")); dw = torch.full((576, 64), float("nan"), device=DEV)
        L.check(lib.avvad_conv2d_fwd(L.ptr(x), L.ptr(wf), L.ptr(y), Ct.byref(d), L.ptr(ews), wsz, st), "fwd")
        L.check(lib.avvad_conv2d_dgrad(L.ptr(gy), L.ptr(wdg), L.ptr(dx), Ct.byref(d), 0, L.ptr(ews), wsz, st), "dgrad")
        L.check(lib.avvad_conv2d_wgrad(L.ptr(x), L.ptr(gy), L.ptr(dw), Ct.byref(d), L.ptr(ews), wsz, st), "wgrad")
        torch.cuda.synchronize()
        return y, dx, dw
    L.set_option("no_conv64", 1); ref = run(); L.set_option("no_conv64", 0)
    for cap in (0, 240, 200, 128):
        L.set_option("max_cus", cap)
        a = run(); b = run()
        L.set_option("max_cus", 0)
        print("N=%d cap=%d: vs engine fwd %.2e dgrad %.2e wgrad %.2e | rerun equal %s | nan %s" % (
            N, cap, *[float((p - q).abs().max() / q.abs().max()) for p, q in zip(a, ref)],
            all(torch.equal(p, q) for p, q in zip(a, b)), any(bool(torch.isnan(p).any()) for p in a)), flush=True)
