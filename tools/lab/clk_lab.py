"""In-kernel shader clock and sustained rate of a back-to-back v_mfma_f32_32x32x2_f32 loop (every SIMD of the chip busy,
no memory traffic): python tools/lab/clk_lab.py   -> profiles/r02_fp32_mfma_clock.txt"""
import ctypes as C, os, time
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libs", "mem_lab.so"))
lib.launch_clk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
lib.launch_clk_rand.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
rnd = torch.randn(64 * 33, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for blocks, label in ((256, "1 wave/SIMD"), (512, "2 waves/SIMD"), (1024, "4 waves/SIMD"), (-512, "2 waves/SIMD, random operands"), (-1024, "4 waves/SIMD, random operands")):
    rand = blocks < 0
    blocks = abs(blocks)
    launch = (lambda o, k, it: lib.launch_clk_rand(o, k, rnd.data_ptr(), it, blocks, st)) if rand else (lambda o, k, it: lib.launch_clk(o, k, it, 0.5, blocks, st))
    out = torch.zeros(2 * blocks, dtype=torch.int64, device="cuda"); sink = torch.zeros(256, device="cuda")
    iters = 20000
    t_end = time.time() + 2.0                 # >= 2 s of back-to-back launches before the stamped one
    while time.time() < t_end:
        launch(out.data_ptr(), sink.data_ptr(), iters)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(out.data_ptr(), sink.data_ptr(), iters); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    v = out.view(blocks, 2).double()
    ghz = (v[:, 0] / v[:, 1] * 0.1).median().item()
    mfma_per_wave = iters * 32
    waves = blocks * 4
    tflops = waves * mfma_per_wave * 4096 / (ms * 1e-3) / 1e12
    cyc = (v[:, 0].median().item()) / mfma_per_wave / (waves / 1024.0)
    print("%-30s in-kernel clock %.2f GHz | %.1f TFLOP/s sustained = %.2f of the 157.3 nominal | %.1f shader cycles per MFMA per SIMD"
          % (label, ghz, tflops, tflops / 157.3, cyc), flush=True)
