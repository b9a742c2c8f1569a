import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "audio-visual-vad_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def lib_options():
    """Set schedule options of libavvad_hip.so for one test (avvad_set_option) and restore them afterwards.  The
    library reads AVVAD_* from the environment only once, so in-process switches go through this."""
    from avvad import _lib as L
    saved = {}

    def set_option(name, value):
        if name not in saved:
            saved[name] = L.get_option(name)
        L.set_option(name, value)
    yield set_option
    for k, v in saved.items():
        L.set_option(k, v)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture
def golden():
    return load_golden


def wn_cfg_from(g):
    fw, qc, R, D, Bn, P, ub = [int(v) for v in g["cfg_scalars"]]
    return dict(filter_width=fw, quantization_channel=qc, dilations=[int(d) for d in g["cfg_dilations"]],
                en_residual_channel=R, en_dilation_channel=D, en_bottleneck_width=Bn,
                en_pool_kernel_size=P, use_bias=bool(ub))
