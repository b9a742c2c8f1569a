import sys, os, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[R, os.path.join(R,'audio-visual-vad_amd')]
import torch
from packages.models.wavenet_autoencoder import wavenet_autoencoder
cfg = dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2, en_residual_channel=32,
           en_dilation_channel=32, en_bottleneck_width=256, en_pool_kernel_size=16, use_bias=True)
torch.manual_seed(0)
m = wavenet_autoencoder(**cfg).cuda()
x = (torch.rand(64, 1, 6143, device='cuda') * 2 - 1)
G = torch.randn(64, 256, 16, device='cuda')
def step():
    for p in m.parameters(): p.grad = None
    y = m(x); (y * G).sum().backward()
for _ in range(3): step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
print("LIB", os.environ.get("AVVAD_LIB", "default"), " wavenet fwd+bwd: %.3f ms" % (e0.elapsed_time(e1) / 10))
