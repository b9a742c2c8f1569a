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
for what in ('fwd', 'wgrad'):
    run(what, 1024, 128, 128, 9, 3, 1, 1)
    run(what, 1024, 256, 256, 5, 3, 1, 1)
    run(what, 1024, 512, 512, 3, 3, 1, 1)
