"""Phase-time breakdown of the engine on conv shapes (needs a -DAVVAD_PROF build given by AVVAD_LIB)."""
import sys, os, ctypes as C
sys.argv = [sys.argv[0]]
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, 'tools')]
import microbench as mb
import torch
lib = mb.lib
lib.avvad_debug_prof.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 8)()
def run(what, *a):
    lib.avvad_debug_prof(None, 1)
    # microbench does 3 warm-up + 10 timed calls = 13 launches
    mb.conv_case(*a, what)
    torch.cuda.synchronize()
    lib.avvad_debug_prof(buf, 0)
    seg, pro, loop, bar, stg, epi, its = [buf[i] / 13.0 for i in range(7)]
    # s_memtime ticks at 100 MHz on gfx9 (constant clock) -> us
    f = 1e-2
    print("   per launch: segments %.0f  ktile-iters %.0f | per segment us: prologue %.2f loop %.2f (barrier %.2f staging %.2f) epilogue %.2f | loop us per ktile %.3f"
          % (seg, its, pro / seg * f, loop / seg * f, bar / seg * f, stg / seg * f, epi / seg * f, loop / its * f), flush=True)
import numpy as np
lib.avvad_debug_prof_blocks.argtypes = [C.c_void_p, C.c_int]
_run = run
def run(what, *a):
    _run(what, *a)
    nb = 512
    blk = (C.c_ulonglong * (2 * nb))()
    lib.avvad_debug_prof_blocks(blk, nb)
    v = np.array(list(blk), dtype=np.float64).reshape(nb, 2) * 0.01   # us
    t0 = v[:, 0].min()
    st, en = v[:, 0] - t0, v[:, 1] - t0
    d = en - st
    hw = (C.c_ulonglong * nb)()
    lib.avvad_debug_prof_hw(hw, nb)
    h = np.array(list(hw), dtype=np.uint64)
    hid = (h & np.uint64(0xffffffff)).astype(np.int64); xcc = (h >> np.uint64(32)).astype(np.int64)
    wave_id, simd, cu, sh, se = hid & 15, (hid >> 4) & 3, (hid >> 8) & 15, (hid >> 12) & 1, (hid >> 13) & 7
    key = xcc * 10000 + se * 1000 + sh * 100 + cu
    groups = {}
    for b in range(nb): groups.setdefault(int(key[b]), []).append(b)
    sizes = sorted(set(len(v_) for v_ in groups.values()))
    pairs = [sorted(v_, key=lambda b: en[b]) for v_ in groups.values() if len(v_) == 2]
    if what == 'fwd' and a[1] == 128:
        print("   distinct CUs %d, workgroups per CU %s; first 6 pairs (block ids, wave slot of thread 0, end us):" % (len(groups), sizes))
        for pr in pairs[:6]: print("      ", [(b, int(wave_id[b]), round(float(en[b]), 1)) for b in pr])
        dif = np.array([pr[1] - pr[0] for pr in pairs]); print("   block-id difference within a CU pair: ", sorted(set(np.abs(dif).tolist()))[:10])
        early_is_low_slot = np.mean([wave_id[pr[0]] < wave_id[pr[1]] for pr in pairs]); print("   early finisher has the lower wave slot in %.0f %% of CUs" % (100 * early_is_low_slot))
        early_is_low_id = np.mean([pr[0] < pr[1] for pr in pairs]); print("   early finisher has the lower block id in %.0f %% of CUs" % (100 * early_is_low_id))
        pe = np.array([[en[pr[0]], en[pr[1]]] for pr in pairs]); print("   per-CU end of first / second workgroup: mean %.1f / %.1f, CU finish (max) min %.1f median %.1f max %.1f" % (pe[:,0].mean(), pe[:,1].mean(), pe[:,1].min(), np.median(pe[:,1]), pe[:,1].max()))
    cu_end = {}
    for b in range(nb): cu_end[int(key[b])] = max(cu_end.get(int(key[b]), 0.0), float(en[b]))
    byx = {}
    for k_, e_ in cu_end.items(): byx.setdefault(k_ // 10000, []).append(e_)
    ck = (C.c_ulonglong * nb)()
    lib.avvad_debug_prof_clk.argtypes = [C.c_void_p, C.c_int]
    lib.avvad_debug_prof_clk(ck, nb)
    cyc = np.array(list(ck), dtype=np.float64)
    ghz = cyc / (d * 1e3)              # cycles / (us * 1000) = GHz
    byc = {}
    for b in range(nb): byc.setdefault(int(xcc[b]), []).append(ghz[b])
    print("   in-kernel clock by XCD (GHz): " + "  ".join("x%d %.3f" % (x_, sum(v_) / len(v_)) for x_, v_ in sorted(byc.items())))
    print("   CU finish time by XCD (min/mean/max): " + "  ".join("x%d %.0f/%.0f/%.0f" % (x_, min(v_), sum(v_) / len(v_), max(v_)) for x_, v_ in sorted(byx.items())))
    print("   last launch, 512 workers (us): start spread %.1f | end min %.1f median %.1f max %.1f | busy min %.1f median %.1f max %.1f"
          % (st.max(), en.min(), np.median(en), en.max(), d.min(), np.median(d), d.max()), flush=True)
for what in ('fwd', 'wgrad'):
    run(what, 1024, 128, 128, 9, 3, 1, 1)
    run(what, 1024, 256, 256, 5, 3, 1, 1)
    run(what, 1024, 512, 512, 3, 3, 1, 1)
