"""Oracle: WaveNet-style encoder (valid dilated Conv1d stack), CPU fp32.

Restates ``packages/models/wavenet_autoencoder.py`` of the reference:
  * parameter set / shapes  -> ``:40-45`` (causal + bottleneck), ``:51-72`` (stacks)
  * forward                 -> ``_encode`` ``:74-93``
  * receptive field         -> ``:47-49``

Test infrastructure only (see ``oracle/__init__.py``).
"""
import math

import torch
import torch.nn.functional as F


def receptive_field(filter_width, dilations):
    """``wavenet_autoencoder._calc_receptive_field`` (``:47-49``)."""
    return (filter_width - 1) * (sum(dilations) + 1) + 1


def init_params(cfg, generator=None, dtype=torch.float32):
    """Parameters with the reference's state_dict key names and default torch
    Conv1d init (kaiming-uniform a=sqrt(5) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    for both weight and bias)."""
    fw, qc = cfg["filter_width"], cfg["quantization_channel"]
    R, D, Bn = cfg["en_residual_channel"], cfg["en_dilation_channel"], cfg["en_bottleneck_width"]
    use_bias = cfg["use_bias"]
    p = {}

    def conv(name, cout, cin, k):
        bound = 1.0 / math.sqrt(cin * k)
        p[name + ".weight"] = (torch.rand(cout, cin, k, generator=generator, dtype=dtype) * 2 - 1) * bound
        if use_bias:
            p[name + ".bias"] = (torch.rand(cout, generator=generator, dtype=dtype) * 2 - 1) * bound

    for i, _ in enumerate(cfg["dilations"]):
        conv("en_dilation_layer_stack.%d" % i, D, R, fw)
        conv("en_dense_layer_stack.%d" % i, R, D, 1)
    conv("en_causal_layer", R, qc, fw)
    conv("bottleneck_layer", Bn, R, 1)
    return p


def encode(params, wave, cfg, return_intermediates=False):
    """``_encode`` (``:74-93``): wave (B,qc,L) -> (B,Bn,P).

    s0 = causal(wave)                                   :75
    s_{i+1} = dense_i(relu(dil_i(relu(s_i)))) + s_i[..., -len:]   :78-86
    out = AdaptiveAvgPool1d(P)(relu(bottleneck(s_N)))   :88-92
    (the ctor argument ``en_pool_kernel_size`` is used as the pool OUTPUT size.)
    """
    g = lambda n: params.get(n)
    s = F.conv1d(wave, params["en_causal_layer.weight"], g("en_causal_layer.bias"))
    inter = [s]
    for i, d in enumerate(cfg["dilations"]):
        cur = s
        z = F.conv1d(F.relu(s), params["en_dilation_layer_stack.%d.weight" % i],
                     g("en_dilation_layer_stack.%d.bias" % i), dilation=d)
        s = F.conv1d(F.relu(z), params["en_dense_layer_stack.%d.weight" % i],
                     g("en_dense_layer_stack.%d.bias" % i))
        s = s + cur[:, :, -s.shape[2]:]
        inter.append(s)
    s = F.relu(F.conv1d(s, params["bottleneck_layer.weight"], g("bottleneck_layer.bias")))
    out = F.adaptive_avg_pool1d(s, cfg["en_pool_kernel_size"])
    if return_intermediates:
        return out, inter
    return out


def encode_loops(params, wave, cfg):
    """Same arithmetic with explicit index loops (no conv library call) -- small
    cases only.  Pins the tap order (cross-correlation: tap k multiplies
    x[t + k*d]) and the LEFT-crop of the residual independently of F.conv1d."""
    fw = cfg["filter_width"]
    B, qc, L = wave.shape

    def conv(x, w, b, d):
        co, ci, k = w.shape
        lo = x.shape[2] - d * (k - 1)
        y = torch.zeros(x.shape[0], co, lo, dtype=x.dtype)
        for kk in range(k):
            y += torch.einsum("oc,bct->bot", w[:, :, kk], x[:, :, kk * d: kk * d + lo])
        if b is not None:
            y += b[None, :, None]
        return y

    g = lambda n: params.get(n)
    s = conv(wave, params["en_causal_layer.weight"], g("en_causal_layer.bias"), 1)
    for i, d in enumerate(cfg["dilations"]):
        z = conv(torch.relu(s), params["en_dilation_layer_stack.%d.weight" % i],
                 g("en_dilation_layer_stack.%d.bias" % i), d)
        y = conv(torch.relu(z), params["en_dense_layer_stack.%d.weight" % i],
                 g("en_dense_layer_stack.%d.bias" % i), 1)
        s = y + s[:, :, d * (fw - 1):]
    s = torch.relu(conv(s, params["bottleneck_layer.weight"], g("bottleneck_layer.bias"), 1))
    P = cfg["en_pool_kernel_size"]
    Lv = s.shape[2]
    out = torch.zeros(B, s.shape[1], P, dtype=s.dtype)
    for i in range(P):
        a = (i * Lv) // P
        e = -((-(i + 1) * Lv) // P)
        out[:, :, i] = s[:, :, a:e].mean(dim=2)
    return out


W0 = dict(filter_width=2, quantization_channel=1,
          dilations=[2 ** i for i in range(10)] * 2,
          en_residual_channel=32, en_dilation_channel=32,
          en_bottleneck_width=256, en_pool_kernel_size=60, use_bias=True)
"""Build-defined WaveNet config "W0" (SURVEY.md 8d): the reference ships none
(its ``params/model_params.json`` is git-ignored)."""
