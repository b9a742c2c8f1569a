"""LSTM stack micro-benchmark at the bench shape (64 sequences x 16 steps, 2 x 1024), forward and forward+backward,
A/B over a library option:  python tools/mb_lstm.py [option=value]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'audio-visual-vad_amd')]
import torch
import torch.nn as nn
from avvad import _lib as L, ops
opts = [a.split('=') for a in sys.argv[1:] if '=' in a]
torch.manual_seed(0)
B, T, In, H = 64, 16, 768, 1024
lstm = nn.LSTM(In, H, 2).cuda()
x = torch.randn(B, T, In, device='cuda', requires_grad=True)
lens = [T] * B


def fwd():
    with torch.no_grad():
        ops.lstm_stack(x, lens, lstm)


def step():
    lstm.zero_grad(set_to_none=True)
    y = ops.lstm_stack(x, lens, lstm)
    y.sum().backward()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


variants = [("default", [])] + ([("+".join("%s=%s" % tuple(o) for o in opts), opts)] if opts else [])
for rep in range(2):
    for name, ov in variants:
        for k, v in ov:
            L.set_option(k, int(v))
        print("%-28s fwd %.3f ms   fwd+bwd %.3f ms" % (name, timeit(fwd), timeit(step)), flush=True)
        for k, v in ov:
            L.set_option(k, 0)
