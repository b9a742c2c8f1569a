#!/usr/bin/env python3
"""GPU micro-benchmarks of the implicit-GEMM engine (run on the MI355X box): python tools/microbench.py"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'audio-visual-vad_amd')]
import torch
from avvad import _lib as L, ops
lib = L.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def timeit(fn, reps=10):
    """Sustained timing: the chip's clock ramps for tens of ms after idle, so a 3+10-launch burst of sub-ms kernels
    under-reports by 10-20 %.  Warm for AVVAD_MB_WARM seconds (default 0.5) of back-to-back launches, then time >= 0.25 s."""
    import time
    warm = float(os.environ.get("AVVAD_MB_WARM", "0.5"))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < warm:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    per = max((time.time() - t0) / max(n, 1), 1e-6)
    reps = max(reps, int(0.25 / per))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def gemm_case(M, N, K, tA, tB):
    A = torch.randn((K, M) if tA else (M, K), device='cuda'); B = torch.randn((N, K) if tB else (K, N), device='cuda')
    Cc = torch.empty(M, N, device='cuda')
    ms = timeit(lambda: ops.gemm(A, B, Cc, M, N, K, A.shape[1], B.shape[1], N, bool(tA), bool(tB)))
    print("gemm %5dx%5dx%5d tA%d tB%d : %8.3f ms  %6.1f TFLOP/s" % (M, N, K, tA, tB, ms, 2.0 * M * N * K / ms / 1e9), flush=True)

def conv_case(n, c, co, h, ks, stride, pad, what='fwd'):
    ho = (h + 2 * pad - ks) // stride + 1
    x = torch.randn(n, h, h, c, device='cuda'); wf = torch.randn(ks * ks * c, co, device='cuda') * 0.05
    wd = torch.randn(ks * ks * co, c, device='cuda') * 0.05
    y = torch.randn(n, ho, ho, co, device='cuda'); dx = torch.empty_like(x); dw = torch.empty(ks * ks * c, co, device='cuda')
    d = L.ConvDesc(n, h, h, c, co, ks, stride, pad)
    ews = torch.empty(lib.avvad_engine_workspace() // 4, device='cuda')
    fns = {'fwd': lambda: lib.avvad_conv2d_fwd(L.ptr(x), L.ptr(wf), L.ptr(y), C.byref(d), L.ptr(ews), ews.numel() * 4, st),
           'dgrad': lambda: lib.avvad_conv2d_dgrad(L.ptr(y), L.ptr(wd), L.ptr(dx), C.byref(d), 0, L.ptr(ews), ews.numel() * 4, st),
           'wgrad': lambda: lib.avvad_conv2d_wgrad(L.ptr(x), L.ptr(y), L.ptr(dw), C.byref(d), L.ptr(ews), ews.numel() * 4, st)}
    ms = timeit(fns[what])
    fl = 2.0 * n * ho * ho * co * ks * ks * c
    print("conv %-5s n=%5d %3d->%3d @%2d k%d s%d : %8.3f ms  %6.1f TFLOP/s" % (what, n, c, co, h, ks, stride, ms, fl / ms / 1e9), flush=True)

if __name__ == '__main__':
    which = sys.argv[1:] or ['gemm', 'conv']
    if 'gemm' in which:
        gemm_case(8192, 8192, 4096, 1, 0)
        gemm_case(8192, 8192, 4096, 0, 1)
        gemm_case(4096, 4096, 4096, 1, 0)
    if 'conv' in which:
        for n in (1024, 4096):
            for what in ('fwd', 'dgrad', 'wgrad'):
                conv_case(n, 64, 64, 17, 3, 1, 1, what)
                conv_case(n, 128, 128, 9, 3, 1, 1, what)
                conv_case(n, 256, 256, 5, 3, 1, 1, what)
                conv_case(n, 512, 512, 3, 3, 1, 1, what)
                conv_case(n, 64, 128, 17, 3, 2, 1, what)
