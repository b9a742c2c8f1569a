"""Encoder micro-benchmark: W0 forward + backward at the bench shape (and the C2 shape with `c2`), A/B over a library
option:  python tools/mb_wn.py [c2] [option=value ...]   e.g.  python tools/mb_wn.py wn_bwd_t=1"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'audio-visual-vad_amd')]
import torch
from avvad import _lib as L
from packages.models.wavenet_autoencoder import wavenet_autoencoder
c2 = 'c2' in sys.argv[1:]
opts = [a.split('=') for a in sys.argv[1:] if '=' in a]
P = 60 if c2 else 16
cfg = dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2, en_residual_channel=32,
           en_dilation_channel=32, en_bottleneck_width=256, en_pool_kernel_size=P, use_bias=True)
torch.manual_seed(0)
m = wavenet_autoencoder(**cfg).cuda()
x = (torch.rand(256, 1, 16000, device='cuda') * 2 - 1) if c2 else (torch.rand(64, 1, 6143, device='cuda') * 2 - 1)
G = torch.randn(x.shape[0], 256, P, device='cuda')


def step():
    for p in m.parameters():
        p.grad = None
    y = m(x)
    (y * G).sum().backward()


def fwd():
    with torch.no_grad():
        m(x)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


variants = [("default", [])] + ([("+".join("%s=%s" % tuple(o) for o in opts), opts)] if opts else [])
for rep in range(2):                      # A/B/A/B inside one process: devices differ by several %
    for name, ov in variants:
        for k, v in ov:
            L.set_option(k, int(v))
        print("%-28s fwd %.3f ms   fwd+bwd %.3f ms" % (name, timeit(fwd), timeit(step)), flush=True)
        for k, v in ov:
            L.set_option(k, 0)
