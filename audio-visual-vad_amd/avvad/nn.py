"""Parameter containers + forward glue shared by the drop-in model classes.

torch.nn modules are used ONLY as parameter/buffer containers, so that
``state_dict()`` keys, shapes and default initialisation equal the reference's
(SURVEY.md 8b); their own ``forward`` is never called -- all arithmetic goes
through avvad.ops (HIP kernels).
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import ops

STAGE_WIDTHS = (64, 128, 256, 512)


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise L.AvvadError("this module is a parameter container; the trunk runs through avvad.ops.TrunkFn")


class BasicBlock(_NoForward):
    """Container mirroring torchvision's BasicBlock attribute names (conv1, bn1, relu, conv2, bn2, downsample)."""

    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))


def make_resnet18_trunk():
    """``nn.Sequential(*list(resnet18().children())[:-1])`` as a container: children 0 conv1, 1 bn1, 2 relu,
    3 maxpool, 4..7 layer1..4, 8 avgpool (Video_Net.py:35-37).  torchvision init: conv kaiming-normal
    (fan_out, relu), BN gamma 1 / beta 0."""
    layers = []
    cin = 64
    for s, c in enumerate(STAGE_WIDTHS):
        layers.append(nn.Sequential(BasicBlock(cin, c, 1 if s == 0 else 2), BasicBlock(c, c, 1)))
        cin = c
    trunk = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                          nn.MaxPool2d(3, 2, 1), *layers, nn.AdaptiveAvgPool2d((1, 1)))
    for m in trunk.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return trunk


def trunk_modules(features):
    """(conv, bn) pairs in the conv index order of include/avvad.h."""
    pairs = [(features[0], features[1])]
    for s in range(4):
        for b in range(2):
            blk = features[4 + s][b]
            pairs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
            if blk.downsample is not None:
                pairs.append((blk.downsample[0], blk.downsample[1]))
    assert len(pairs) == L.TRUNK_NCONV
    return pairs


def trunk_forward(features, frames, training):
    """frames (N,H,W) fp32 on the GPU -> (N,512)."""
    pairs = trunk_modules(features)
    bn0 = pairs[0][1]
    ts = [c.weight for c, _ in pairs] + [b.weight for _, b in pairs] + [b.bias for _, b in pairs] + \
         [b.running_mean for _, b in pairs] + [b.running_var for _, b in pairs]
    out = ops.TrunkFn.apply(frames, bool(training), bn0.momentum, bn0.eps, *ts)
    if training:
        torch._foreach_add_([b.num_batches_tracked for _, b in pairs], 1)
    return out


def video_features(features, video, training):
    """video (B,T,H,W) -> (B,T,512): the reference repeats the gray frame to 3 channels and runs
    ``self.features(...).squeeze()`` (Video_Net.py:60-81); the repeat is folded into the stem weights."""
    B, T, H, W = video.shape
    if not video.is_cuda:
        raise L.AvvadError("video must be on the GPU: no CPU fallback")
    return trunk_forward(features, video.reshape(B * T, H, W), training).view(B, T, 512)
