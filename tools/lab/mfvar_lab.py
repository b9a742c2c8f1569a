"""What keeps the block's MFMA chain below the pipe rate: python tools/lab/mfvar_lab.py"""
import ctypes as C, os, time
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libs", "mem_lab.so"))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
y = torch.empty(1024, device="cuda")
ntw = 64
for blocks in (256, 512, 768):
    wps = blocks * 4 / 1024.0
    ideal = ntw * 48 * 64 * wps / 2.39e3
    for name in ("v_base", "v_regw", "v_lds_norelu", "v_regw_norelu", "v_two", "v_pure", "v_two_pure"):
        fn = getattr(lib, "launch_" + name)
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        t_end = time.time() + 0.5
        while time.time() < t_end:
            fn(y.data_ptr(), ntw, blocks, st); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn(y.data_ptr(), ntw, blocks, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        print("blocks %4d (%.0f waves/SIMD) %-14s %.1f us  (MFMA pipe time %.1f us -> %.2f)" % (blocks, wps, name, us, ideal, ideal / us), flush=True)
