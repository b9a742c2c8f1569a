"""Drop-in containers for ``packages/models/compact_bilinear_pooling.py`` (``CountSketch`` ``:59-114``,
``CompactBilinearPooling`` ``:222-263``): same constructor arguments, buffers ``h`` (int64 bucket of each
input channel) / ``s`` (+-1 sign) and sub-module names ``sketch1`` / ``sketch2`` so MCB checkpoints load.
The fusion arithmetic (count sketch -> FFT circular convolution) is the next hot-path row:
there is no HIP kernel for it yet and no fallback, so ``forward`` raises."""
import torch
import torch.nn as nn

from avvad import AvvadError


class CountSketch(nn.Module):
    def __init__(self, input_size, output_size, h=None, s=None):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        if h is None:
            h = torch.randint(0, output_size, (input_size,), dtype=torch.long)
        if s is None:
            s = 2.0 * torch.randint(0, 2, (input_size,)).float() - 1.0
        self.register_buffer("h", h)
        self.register_buffer("s", s)

    def _apply(self, fn, recurse=True):
        # `h` must stay int64 whatever dtype the module is cast to (reference :96-104)
        h = self.h
        super()._apply(fn, recurse)
        if self.h.dtype != torch.long:
            self.h = h.to(self.h.device)
        return self

    def forward(self, x):
        raise AvvadError("CountSketch is a buffer container here: the sketch runs inside avvad.ops.McbFusionFn (no PyTorch fallback)")


class CompactBilinearPooling(nn.Module):
    def __init__(self, input1_size, input2_size, output_size, h1=None, s1=None, h2=None, s2=None,
                 force_cpu_scatter_add=False):
        super().__init__()
        self.add_module("sketch1", CountSketch(input1_size, output_size, h1, s1))
        self.add_module("sketch2", CountSketch(input2_size, output_size, h2, s2))
        self.output_size = output_size
        self.force_cpu_scatter_add = force_cpu_scatter_add

    def forward(self, x, y=None):
        raise AvvadError("CompactBilinearPooling is a buffer container here: use DeepVAD_AV(use_mcb=True) (fused HIP path, no fallback)")
