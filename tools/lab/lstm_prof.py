"""Phase times of the persistent LSTM forward (needs a -DAVVAD_LSTM_PROF build given by AVVAD_LIB):
A = h loads + recurrent MFMAs, B = reduce + gates + stores + drain, C = step barrier."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, 'audio-visual-vad_amd')]
import torch
from avvad import _lib as L
lib = L.lib()
B, T, In, H = 64, 16, 768, 1024
torch.manual_seed(0)
x = torch.randn(B, T, In, device='cuda')
w_ih, w_hh = torch.randn(4 * H, In, device='cuda') * 0.02, torch.randn(4 * H, H, device='cuda') * 0.02
b_ih, b_hh = torch.zeros(4 * H, device='cuda'), torch.zeros(4 * H, device='cuda')
lens = torch.full((B,), T, dtype=torch.int32, device='cuda')
d = L.LstmDesc(B, T, In, H, lens.data_ptr(), 1)
nbytes = lib.avvad_lstm_workspace(C.byref(d))
ws = torch.zeros(nbytes // 4, device='cuda')
y = torch.empty(B, T, H, device='cuda')
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
al = lambda n: (n + 63) // 64 * 64
slab = al(B * T * 4 * H) + al(B * T * H) + al(4 * H) + 2 * al(B * H) + al(B * T * H)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.avvad_lstm_layer_fwd(L.ptr(x), L.ptr(w_ih), L.ptr(w_hh), L.ptr(b_ih), L.ptr(b_hh), L.ptr(y), C.byref(d), L.ptr(ws), nbytes, st), "fwd")
    e1.record(); torch.cuda.synchronize()
    raw = ws[slab + 16: slab + 16 + 48].view(torch.int64).cpu().view(8, 3).double() * 0.01 / (T - 2)   # us per step (100 MHz), 14 timed steps
    print("layer fwd %.1f us | per step us (blocks 0..7): loads+MFMA %s | reduce+gates+drain %s | barrier %s | status %d" % (
        e0.elapsed_time(e1) * 1e3, [round(v, 1) for v in raw[:, 0].tolist()][:4], [round(v, 1) for v in raw[:, 1].tolist()][:4],
        [round(v, 1) for v in raw[:, 2].tolist()][:4], int(ws[slab + 1].view(torch.int32).item())))
