"""Drop-in for ``packages/models/wavenet_autoencoder.py`` of the reference (same class name,
constructor arguments, attribute / state_dict key names ``:8-45,51-72``); ``forward`` runs the
HIP encoder (csrc/wavenet.hip) instead of 2N+2 ATen conv calls (``_encode`` ``:74-93``)."""
import torch.nn as nn

from avvad import ops


class wavenet_autoencoder(nn.Module):
    def __init__(self, filter_width, quantization_channel, dilations, en_residual_channel, en_dilation_channel,
                 en_bottleneck_width, en_pool_kernel_size, use_bias):
        super().__init__()
        self.filter_width = filter_width
        self.quantization_channel = quantization_channel
        self.dilations = list(dilations)
        self.en_residual_channel = en_residual_channel
        self.en_dilation_channel = en_dilation_channel
        self.en_bottleneck_width = en_bottleneck_width
        self.en_pool_kernel_size = en_pool_kernel_size   # used as the pool OUTPUT size (reference :91)
        self.use_bias = use_bias
        self.receptive_field = (filter_width - 1) * (sum(self.dilations) + 1) + 1
        # registration order = reference order (stacks first, then causal + bottleneck) so that
        # state_dict() enumerates keys identically
        self.en_dilation_layer_stack = nn.ModuleList(
            nn.Conv1d(en_residual_channel, en_dilation_channel, filter_width, dilation=d, bias=use_bias)
            for d in self.dilations)
        self.en_dense_layer_stack = nn.ModuleList(
            nn.Conv1d(en_dilation_channel, en_residual_channel, 1, bias=use_bias) for _ in self.dilations)
        self.en_causal_layer = nn.Conv1d(quantization_channel, en_residual_channel, filter_width, bias=use_bias)
        self.bottleneck_layer = nn.Conv1d(en_residual_channel, en_bottleneck_width, 1, bias=use_bias)

    def config(self):
        return dict(filter_width=self.filter_width, quantization_channel=self.quantization_channel,
                    dilations=self.dilations, en_residual_channel=self.en_residual_channel,
                    en_dilation_channel=self.en_dilation_channel, en_bottleneck_width=self.en_bottleneck_width,
                    en_pool_kernel_size=self.en_pool_kernel_size, use_bias=self.use_bias)

    def forward(self, wave_sample):
        """wave_sample (B, quantization_channel, L) -> (B, en_bottleneck_width, en_pool_kernel_size)."""
        params = [self.en_causal_layer.weight, self.en_causal_layer.bias,
                  self.bottleneck_layer.weight, self.bottleneck_layer.bias]
        for dil, dense in zip(self.en_dilation_layer_stack, self.en_dense_layer_stack):
            params += [dil.weight, dil.bias, dense.weight, dense.bias]
        return ops.WavenetFn.apply(wave_sample, self.config(), *params)
