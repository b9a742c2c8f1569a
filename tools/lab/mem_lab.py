"""python tools/lab/mem_lab.py  (needs tools/lab/libs/mem_lab.so built by hipcc -shared)"""
import ctypes as C, os, sys
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libs", "mem_lab.so"))
hip = C.CDLL("libamdhip64.so")
def run(name, B, L, dil, blocks):
    x = torch.randn(B, 32, L, device="cuda"); y = torch.empty(B, 32, L - dil, device="cuda")
    import ctypes
    # launch through hipModuleLaunchKernel is clumsy from ctypes: use the <<<>>> wrappers exported below
    w = getattr(lib, "launch_" + name)
    w.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): w(x.data_ptr(), y.data_ptr(), B, L, dil, blocks, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): w(x.data_ptr(), y.data_ptr(), B, L, dil, blocks, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byts = 4.0 * B * 32 * (L + L - dil)
    print("%s B=%d L=%d dil=%d blocks=%d: %.1f us  %.2f TB/s (read+write of one plane each)" % (name, B, L, dil, blocks, ms * 1e3, byts / ms / 1e9), flush=True)
import sys
if len(sys.argv) > 1 and sys.argv[1] == "mfma":
    for name, blocks in (("m_lds_relu", 768), ("m_reg_relu", 768), ("m_reg_norelu", 768), ("m_reg_relu_dual", 512), ("m_reg_norelu_dual", 512),
                         ("m_lds_relu_dual", 512), ("m_reg_relu_2w", 512), ("m_reg_relu_1w", 256), ("m_nostore", 768), ("m_nostore_dual", 512)):
        run(name, 64, 6144, 64, blocks)
    for name, blocks in (("m_reg_relu", 768), ("m_nostore", 768), ("m_nostore_dual", 512)):
        run(name, 256, 16000, 64, blocks)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "fwd":
    for (B, L) in ((64, 6144), (256, 16000)):
        for dil in (64,):
            for name, blocks in (("full3", 768), ("full2", 512), ("nomfma3", 768), ("noload3", 768), ("nores3", 768), ("noresnomfma3", 768), ("pat0", 768)):
                run(name, B, L, dil, blocks)
    sys.exit(0)
for (B, L) in ((64, 6144), (256, 16000)):
    for dil in (64, 512):
        for name, blocks in (("pat0", 512), ("pat0", 1024), ("pat0", 2048), ("pat2", 512), ("pat2", 1024), ("pat1", 1024), ("pat1", 2048), ("pat1", 4096)):
            run(name, B, L, dil, blocks)
