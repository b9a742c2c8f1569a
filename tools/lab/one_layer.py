"""One residual block of the encoder through avvad_wavenet_block_fwd, N launches (for rocprofv3 --pmc / --kernel-trace):
python tools/lab/one_layer.py [B] [L] [dil] [launches]"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, 'audio-visual-vad_amd')]
import torch
from avvad import _lib as L_
B, L, dil, n = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (256, 16000, 64, 5)
lib = L_.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
s_in = torch.randn(B, 32, L, device="cuda")
s_out = torch.empty(B, 32, L - dil, device="cuda")
wd, bd = torch.randn(32, 32, 2, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.1
we, be = torch.randn(32, 32, 1, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.1
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(n):
    if i == n - 1: e0.record()
    L_.check(lib.avvad_wavenet_block_fwd(L_.ptr(s_in), L_.ptr(wd), L_.ptr(bd), L_.ptr(we), L_.ptr(be), L_.ptr(s_out), B, L, dil, st), "block")
e1.record(); torch.cuda.synchronize()
print("B %d L %d dil %d: last launch %.1f us  (%.0f GB/s algorithmic)" % (B, L, dil, e0.elapsed_time(e1) * 1e3, 4.0 * B * 32 * (2 * L - dil) / e0.elapsed_time(e1) / 1e6))
