"""Oracle: ResNet-18 trunk, CPU fp32.

The reference does not contain this arithmetic: it calls
``torchvision.models.resnet18(pretrained=False)`` (``packages/models/Video_Net.py:18,35-37``,
``packages/models/AV_Net.py:25,28-30``) and keeps every child but the last
(``fc``).  torchvision is an un-vendored, un-pinned dependency that is not
installed in the build image, so this file restates torchvision's published
ResNet-18 structure:

  conv1 7x7/2 pad 3 (no bias) -> bn1 -> relu -> maxpool 3x3/2 pad 1
  -> layer1..layer4, each 2 BasicBlocks of [conv3x3(stride) -> bn -> relu ->
     conv3x3 -> bn ; (+ 1x1/stride conv + bn downsample on the identity when
     the shape changes) ; add ; relu], widths 64/128/256/512, stride 2 at the
     first block of layer2..4
  -> AdaptiveAvgPool2d(1) -> fc(512, 1000)
  init: conv kaiming-normal(fan_out, relu); BN gamma=1, beta=0; BN eps 1e-5,
  momentum 0.1.

PARITY UNPINNED by the reference for this file (no golden vector exists in the
reference for the tower).  What pins it: trunk parameter count 11,176,512
(= torchvision's documented 11,689,512 - fc 513,000), child order / state_dict
key names, the 67->34->17->9->5->3 shape chain, and the reference's own
``DeepVAD_video`` / ``DeepVAD_AV`` class bodies executing on top of it
(``tools/gen_golden.py``).

Test infrastructure only (see ``oracle/__init__.py``).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                nn.BatchNorm2d(planes))
        self.stride = stride

    def forward(self, x):
        idn = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            idn = self.downsample(x)
        return self.relu(out + idn)


class ResNet18(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(BasicBlock(64, 64, 1), BasicBlock(64, 64, 1))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128, 1))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256, 1))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512, 1))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet18(pretrained=False, **kw):
    """Stand-in for ``torchvision.models.resnet18`` (signature as called at
    ``Video_Net.py:18``)."""
    assert not pretrained, "no network / no weights in this image"
    return ResNet18(**kw)


# ---------------------------------------------------------------------------
# Functional trunk over a flat state_dict with the reference's key names
# (``features.0.weight`` = conv1, ``features.1.*`` = bn1, ``features.4.0.conv1.weight`` ...)
# ---------------------------------------------------------------------------
STAGES = [(64, 1), (128, 2), (256, 2), (512, 2)]


def trunk_keys(prefix="features."):
    """All parameter/buffer keys of the trunk in state_dict order, with shapes."""
    keys = []

    def bn(p, c):
        keys.extend([(p + ".weight", (c,)), (p + ".bias", (c,)), (p + ".running_mean", (c,)),
                     (p + ".running_var", (c,)), (p + ".num_batches_tracked", ())])

    keys.append((prefix + "0.weight", (64, 3, 7, 7)))
    bn(prefix + "1", 64)
    cin = 64
    for li, (c, stride) in enumerate(STAGES):
        for bi in range(2):
            p = "%s%d.%d" % (prefix, 4 + li, bi)
            s = stride if bi == 0 else 1
            keys.append((p + ".conv1.weight", (c, cin, 3, 3)))
            bn(p + ".bn1", c)
            keys.append((p + ".conv2.weight", (c, c, 3, 3)))
            bn(p + ".bn2", c)
            if s != 1 or cin != c:
                keys.append((p + ".downsample.0.weight", (c, cin, 1, 1)))
                bn(p + ".downsample.1", c)
            cin = c
    return keys


def _bn(x, sd, p, training, momentum=0.1, eps=1e-5):
    # F.batch_norm updates running stats in place when training (torch.nn.BatchNorm2d semantics)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training, momentum, eps)


def trunk_forward(sd, x, training=False, prefix="features.", return_intermediates=False):
    """x (N,3,H,W) -> (N,512).  ``sd`` maps reference key names to tensors;
    running stats are updated in place when ``training``."""
    inter = {}
    y = F.conv2d(x, sd[prefix + "0.weight"], None, 2, 3)
    inter["conv1"] = y
    y = F.relu(_bn(y, sd, prefix + "1", training))
    y = F.max_pool2d(y, 3, 2, 1)
    inter["pool"] = y
    cin = 64
    for li, (c, stride) in enumerate(STAGES):
        for bi in range(2):
            p = "%s%d.%d" % (prefix, 4 + li, bi)
            s = stride if bi == 0 else 1
            idn = y
            o = F.relu(_bn(F.conv2d(y, sd[p + ".conv1.weight"], None, s, 1), sd, p + ".bn1", training))
            inter["%d.%d.a1" % (4 + li, bi)] = o
            o = _bn(F.conv2d(o, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2", training)
            if s != 1 or cin != c:
                idn = _bn(F.conv2d(y, sd[p + ".downsample.0.weight"], None, s, 0), sd,
                          p + ".downsample.1", training)
            y = F.relu(o + idn)
            inter["%d.%d" % (4 + li, bi)] = y
            cin = c
    out = F.adaptive_avg_pool2d(y, 1).flatten(1)
    if return_intermediates:
        return out, inter
    return out
