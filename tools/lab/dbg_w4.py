import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, 'audio-visual-vad_amd')]
import torch
from avvad import _lib as L_
lib = L_.lib()
B, Lin, dil = 64, 14977, 256
torch.manual_seed(1)
s_in = torch.randn(B, 32, Lin, device="cuda")
wd, bd = torch.randn(32, 32, 2, device="cuda") * 0.2, torch.randn(32, device="cuda") * 0.1
we, be = torch.randn(32, 32, 1, device="cuda") * 0.2, torch.randn(32, device="cuda") * 0.1
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(form, grid):
    L_.set_option("wn_flat", form); L_.set_option("wn_grid", grid)
    out = torch.full((B, 32, Lin - dil), float("nan"), device="cuda")
    L_.check(lib.avvad_wavenet_block_fwd(L_.ptr(s_in), L_.ptr(wd), L_.ptr(bd), L_.ptr(we), L_.ptr(be), L_.ptr(out), B, Lin, dil, st), "blk")
    torch.cuda.synchronize(); return out
ref = run(1, 0)
for grid in (256, 512, 520, 768, 1024):
    o = run(3, grid)
    bad = (o - ref).abs() > 1e-4
    print("grid", grid, "bad elements", int(bad.sum()), "nan", int(torch.isnan(o).sum()))
    if bad.any():
        idx = bad.nonzero()
        print("  b range", idx[:, 0].min().item(), idx[:, 0].max().item(), " rows", sorted(set(idx[:, 1].tolist()))[:40])
        t = idx[:, 2]
        tiles = torch.unique(idx[:, 0] * 116 + t // 128)
        print("  bad super-tiles:", tiles.numel(), tiles[:24].tolist())
        print("  t mod 128 hist of bad:", torch.bincount(t % 128, minlength=128).tolist()[:16], '...')
o = run(3, 512)
bad = (o - ref).abs() > 1e-4
idx = bad.nonzero()
b0, r0, t0 = idx[0].tolist()
T0 = t0 // 128 * 128
print("first bad: b", b0, "row", r0, "t", t0, "tile start", T0)
sub = bad[b0, :, T0:T0 + 128]
print("bad count per row:", sub.sum(1).tolist())
print("bad count per (t mod 4):", [int(sub[:, j::4].sum()) for j in range(4)])
print("bad t offsets (row %d):" % r0, sub[r0].nonzero().flatten().tolist())
d = (o - ref)[b0, r0, T0:T0 + 128]
res = s_in[b0, r0, T0 + dil:T0 + dil + 128]
tt = sub[r0].nonzero().flatten()[:6]
for t in tt.tolist():
    print("  t+%d: out %.5f ref %.5f diff %.5f  residual %.5f  residual(row^4) %.5f  res next row %.5f" % (t, o[b0, r0, T0 + t], ref[b0, r0, T0 + t], d[t], res[t], s_in[b0, r0 ^ 4, T0 + dil + t], s_in[b0, (r0 + 1) % 32, T0 + dil + t]))
# does out - ref + residual match some other residual sample?
