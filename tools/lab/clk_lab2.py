"""In-kernel clock, CU/SIMD placement and per-wave duration of the forward block's work mix:
python tools/lab/clk_lab2.py [blocks]"""
import ctypes as C, os, sys, time
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libs", "mem_lab.so"))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, L, dil = 256, 16000, 64
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 768
nw = blocks * 4
x = torch.randn(B, 32, L, device="cuda"); y = torch.empty(B, 32, L - dil, device="cuda")
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ("clk_full", "clk2_full", "clk_mfmaonly", "clk2_mfmaonly", "clk_memonly", "clk2_memonly")
if len(sys.argv) > 5: B, L, dil = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
x = torch.randn(B, 32, L, device="cuda"); y = torch.empty(B, 32, L - dil, device="cuda")
for name in names:
    fn = getattr(lib, "launch_" + name)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    stamp = torch.zeros(4 * nw, dtype=torch.int64, device="cuda")
    t_end = time.time() + 1.0
    while time.time() < t_end:
        fn(x.data_ptr(), y.data_ptr(), stamp.data_ptr(), B, L, dil, blocks, st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(x.data_ptr(), y.data_ptr(), stamp.data_ptr(), B, L, dil, blocks, st); e1.record(); torch.cuda.synchronize()
    v = stamp.view(nw, 4).cpu()
    ghz = (v[:, 0].double() / v[:, 1].double() * 0.1).median().item()
    hw = v[:, 3]
    xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
    cu = (xcc << 8) | (((hwid >> 13) & 7) << 5) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xf)
    simd = (hwid >> 4) & 3
    key = cu * 4 + simd
    uniq, inv, cnt = torch.unique(key, return_inverse=True, return_counts=True)
    dur = v[:, 1].double() * 0.01
    print("%-13s kernel %.1f us, clock %.2f GHz, CUs %d, (CU,SIMD) slots used %d" % (name, e0.elapsed_time(e1) * 1e3, ghz, torch.unique(cu).numel(), uniq.numel()))
    print("   waves per SIMD histogram:", dict(zip(*[t.tolist() for t in torch.unique(cnt, return_counts=True)])))
    per = cnt[inv]
    for c in torch.unique(cnt).tolist():
        d = dur[per == c]
        print("   waves sharing a SIMD with %d waves in total: n=%d, duration median %.1f us, max %.1f us" % (c, d.numel(), d.median().item(), d.max().item()))
    sys.stdout.flush()
