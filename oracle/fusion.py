"""Oracle: compact bilinear pooling (count sketch + circular convolution) and the
MCB post-processing of ``DeepVAD_AV`` -- CPU fp32.

Restates:
  * ``CountSketchFn_forward`` ``packages/models/compact_bilinear_pooling.py:7-27``
      out[..., h[i]] += s[i] * x[..., i]
  * ``CompactBilinearPoolingFn.forward`` ``:140-173``
      irfft(rfft(psi(x,h1,s1)) * rfft(psi(y,h2,s2)), n=output_size)
    The reference uses ``torch.rfft/irfft`` (removed in torch 2.x; it raises
    AttributeError here), equivalent to ``torch.fft.rfft`` (unnormalised) and
    ``torch.fft.irfft(n=output_size)`` (1/n normalised).
  * signed sqrt, whole-tensor L2 normalisation (detached), BatchNorm1d on the
    (T, C, B) view  ``packages/models/AV_Net.py:109-121``

Test infrastructure only (see ``oracle/__init__.py``).
"""
import torch
import torch.nn.functional as F


def count_sketch(x, h, s, output_size):
    out = x.new_zeros(x.shape[:-1] + (output_size,))
    return out.scatter_add_(-1, h.view((1,) * (x.dim() - 1) + (-1,)).expand_as(x), x * s)


def mcb(x, y, h1, s1, h2, s2, output_size):
    fx = torch.fft.rfft(count_sketch(x, h1, s1, output_size), dim=-1)
    fy = torch.fft.rfft(count_sketch(y, h2, s2, output_size), dim=-1)
    return torch.fft.irfft(fx * fy, n=output_size, dim=-1)


def mcb_naive(x, y, h1, s1, h2, s2, output_size):
    """Independent definition: the sketch of the outer product,
    out[(h1[i] + h2[j]) mod d] += s1[i] s2[j] x[i] y[j]  (Pham & Pagh / Gao et al.)."""
    out = torch.zeros(x.shape[:-1] + (output_size,), dtype=torch.float64)
    xs = (x * s1).double()
    ys = (y * s2).double()
    idx = (h1[:, None] + h2[None, :]) % output_size
    outer = xs[..., :, None] * ys[..., None, :]
    out.view(-1, output_size).index_add_(1, idx.reshape(-1), outer.reshape(-1, idx.numel()))
    return out


def mcb_post(y, bn_weight, bn_bias, running_mean, running_var, eps, training, momentum=0.1):
    """``AV_Net.py:113-121``: y (B,T,C)."""
    y = torch.sign(y) * torch.sqrt(torch.abs(y) + eps)
    y = y / torch.norm(y, p=2).detach()
    y = y.permute(1, 2, 0).contiguous()              # (T, C, B): BN1d stats over T and B
    y = F.batch_norm(y, running_mean, running_var, bn_weight, bn_bias, training, momentum, eps)
    return y.permute(2, 0, 1).contiguous()
