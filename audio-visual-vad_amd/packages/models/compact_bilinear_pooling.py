"""Drop-in containers for ``packages/models/compact_bilinear_pooling.py`` (``CountSketch`` ``:59-114``,
``CompactBilinearPooling`` ``:222-263``): same constructor arguments, buffers ``h`` (int64 bucket of each
input channel) / ``s`` (+-1 sign) and sub-module names ``sketch1`` / ``sketch2`` so MCB checkpoints load.
``forward`` of both modules runs the stand-alone HIP kernels (``avvad_count_sketch_fwd/bwd``,
``avvad_mcb_fwd/bwd``: csrc/mcb.hip -- the FFT product of the reference is evaluated as the circular
convolution it equals) and returns the raw sketch / pooled vector like the reference; inside ``DeepVAD_AV``
the same arithmetic runs fused with the signed sqrt, L2 norm and BatchNorm1d (``ops.McbFusionFn``).
GPU tensors only: there is no PyTorch fallback."""
import torch
import torch.nn as nn

from avvad import ops


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
        assert x.shape[-1] == self.input_size
        return ops.CountSketchFn.apply(self.h, self.s, self.output_size, x)


class CompactBilinearPooling(nn.Module):
    def __init__(self, input1_size, input2_size, output_size, h1=None, s1=None, h2=None, s2=None,
                 force_cpu_scatter_add=False):
        super().__init__()
        self.add_module("sketch1", CountSketch(input1_size, output_size, h1, s1))
        self.add_module("sketch2", CountSketch(input2_size, output_size, h2, s2))
        self.output_size = output_size
        self.force_cpu_scatter_add = force_cpu_scatter_add

    def forward(self, x, y=None):
        if y is None:
            y = x
        return ops.CompactBilinearPoolingFn.apply(self.sketch1.h, self.sketch1.s, self.sketch2.h, self.sketch2.s,
                                                  self.output_size, x, y)
