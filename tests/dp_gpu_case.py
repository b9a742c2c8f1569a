"""Shared definition of the model / batch of the two-rank GPU data-parallel test."""
import torch

import stategen

WCFG = dict(filter_width=2, quantization_channel=1, dilations=[1, 2, 4, 8], en_residual_channel=32, en_dilation_channel=32,
            en_bottleneck_width=64, en_pool_kernel_size=8, use_bias=True)
RF = 16          # 1 + sum(dilations)
N_SEQ, T, HOP = 4, 8, 32
SGD_LR = 0.05


def make_model():
    from packages.models.AV_Net import DeepVAD_AV
    torch.manual_seed(5)
    return DeepVAD_AV(1, 32, 1, use_mcb=False, eps=1e-8, wavenet_params=WCFG)


def make_batch():
    wave = stategen.rand(301, N_SEQ, 1, T * HOP + RF - 1)
    video = stategen.rand(302, N_SEQ, T, 67, 67)
    target = (stategen.rand(303, N_SEQ, T, 1) > 0).float()
    lengths = torch.LongTensor([8, 6, 7, 5])
    return wave, video, target, lengths
