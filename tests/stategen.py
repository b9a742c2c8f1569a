"""Deterministic state_dict recipes shared by ``tools/gen_golden.py`` (which loads
them into the REFERENCE's classes to produce golden outputs) and by the tests
(which rebuild the same weights on the GPU box, where the reference does not
exist).  numpy ``RandomState`` is used because its streams are stable across
numpy/torch versions; big models therefore need no committed weights."""
import numpy as np
import torch


def _fill(rng, name, shape):
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_mean":
        return torch.from_numpy(rng.normal(0.0, 0.1, shape).astype(np.float32))
    if leaf == "running_var":
        return torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
    if len(shape) == 1 and leaf == "weight":                      # norm gamma
        return torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
    if leaf.startswith("bias"):
        return torch.from_numpy(rng.normal(0.0, 0.1, shape).astype(np.float32))
    if len(shape) == 4:                                            # conv OIHW, He-like
        fan_in = shape[1] * shape[2] * shape[3]
        return torch.from_numpy(rng.normal(0.0, np.sqrt(2.0 / fan_in), shape).astype(np.float32))
    if len(shape) == 3:                                            # conv1d
        bound = 1.0 / np.sqrt(shape[1] * shape[2])
        return torch.from_numpy(rng.uniform(-bound, bound, shape).astype(np.float32))
    if len(shape) == 2:                                            # lstm / linear
        bound = 1.0 / np.sqrt(shape[1]) if "lstm" not in name else 1.0 / np.sqrt(shape[0] // 4)
        return torch.from_numpy(rng.uniform(-bound, bound, shape).astype(np.float32))
    raise ValueError((name, shape))


def make_state(spec, seed):
    """spec: iterable of (key, shape) in state_dict order -> {key: tensor}."""
    rng = np.random.RandomState(seed)
    return {k: _fill(rng, k, tuple(s)) for k, s in spec}


def lstm_spec(prefix, in_size, hidden, layers):
    spec = []
    for l in range(layers):
        i = in_size if l == 0 else hidden
        spec += [("%sweight_ih_l%d" % (prefix, l), (4 * hidden, i)),
                 ("%sweight_hh_l%d" % (prefix, l), (4 * hidden, hidden)),
                 ("%sbias_ih_l%d" % (prefix, l), (4 * hidden,)),
                 ("%sbias_hh_l%d" % (prefix, l), (4 * hidden,))]
    return spec


def linear_spec(name, in_size, out_size):
    return [(name + ".weight", (out_size, in_size)), (name + ".bias", (out_size,))]


def wavenet_spec(cfg, prefix=""):
    fw, qc = cfg["filter_width"], cfg["quantization_channel"]
    R, D, Bn = cfg["en_residual_channel"], cfg["en_dilation_channel"], cfg["en_bottleneck_width"]
    spec = []

    def conv(n, co, ci, k):
        spec.append((prefix + n + ".weight", (co, ci, k)))
        if cfg["use_bias"]:
            spec.append((prefix + n + ".bias", (co,)))

    for i in range(len(cfg["dilations"])):
        conv("en_dilation_layer_stack.%d" % i, D, R, fw)
    for i in range(len(cfg["dilations"])):
        conv("en_dense_layer_stack.%d" % i, R, D, 1)
    conv("en_causal_layer", R, qc, fw)
    conv("bottleneck_layer", Bn, R, 1)
    return spec


def rand(seed, *shape, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).normal(0, 1, shape) * scale).astype(np.float32))
