"""Drop-in for ``packages/models/AV_Net.py``: ``DeepVAD_AV(lstm_layers, lstm_hidden_size, y_dim,
use_mcb=False, eps=1e-8)`` with ``forward(audio, video, lengths)`` (reference ``:13-58,72-141``).

``wavenet_params`` re-opens the commented audio-encoder hook (``:102-106``): ``audio`` is then a raw
waveform (B, quantization_channel, L) whose encoder output (B,Bn,T) is fused with the video features."""
import torch
import torch.nn as nn

from avvad import nn as avnn
from avvad import ops
from packages.models.compact_bilinear_pooling import CompactBilinearPooling
from packages.models.utils import weights_init_normal
from packages.models.wavenet_autoencoder import wavenet_autoencoder


class DeepVAD_AV(nn.Module):
    def __init__(self, lstm_layers, lstm_hidden_size, y_dim, use_mcb=False, eps=1e-8, wavenet_params=None):
        super().__init__()
        self.lstm_layers = lstm_layers
        self.lstm_hidden_size = lstm_hidden_size
        self.y_dim = y_dim
        self.dropout = nn.Dropout(p=0.05)
        self.use_mcb = use_mcb
        self.eps = eps
        self.num_video_ftrs = 512
        self.features = avnn.make_resnet18_trunk()
        # constructed but never used in forward -- still part of every checkpoint (reference :33)
        self.bn = nn.BatchNorm1d(self.num_video_ftrs, eps=eps, momentum=0.1, affine=True)
        self.num_audio_ftrs = 513
        if wavenet_params is not None:
            self.wavenet_en = wavenet_autoencoder(**wavenet_params)
            self.num_audio_ftrs = wavenet_params["en_bottleneck_width"]
        if use_mcb:
            self.mcb_output_size = 1024
            self.lstm_input_size = self.mcb_output_size
            self.mcb = CompactBilinearPooling(self.num_audio_ftrs, self.num_video_ftrs, self.mcb_output_size)
            self.mcb_bn = nn.BatchNorm1d(self.mcb_output_size, eps=eps, momentum=0.1, affine=True)
        else:
            self.lstm_input_size = self.num_audio_ftrs + self.num_video_ftrs
        self.lstm_merged = nn.LSTM(input_size=self.lstm_input_size, hidden_size=lstm_hidden_size,
                                   num_layers=lstm_layers, bidirectional=False)
        self.vad_merged = nn.Linear(lstm_hidden_size, y_dim)

    def weight_init(self, mean=0.0, std=0.02):
        for m in self.named_parameters():
            weights_init_normal(m, mean=mean, std=std)

    def forward(self, audio, video, lengths):
        if hasattr(self, "wavenet_en") and audio.is_cuda and ops.overlap_enabled():
            # The audio encoder (HBM-bound Conv1d stack) and the video trunk (MFMA-bound Conv2d tower) are independent
            # until the fusion: run the encoder on a side HIP stream so the two kinds of kernel share the chip.
            # autograd replays each backward node on its forward stream, so the backward passes overlap the same way.
            main, side = torch.cuda.current_stream(), ops.side_stream()
            ops.note_fork(main)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                audio = ops.TransposeLast2Fn.apply(self.wavenet_en(audio))        # (B,T,Bn)
            feats = avnn.video_features(self.features, video, self.training)      # (B,T,512)
            main.wait_stream(side)
            audio.record_stream(main)
        else:
            feats = avnn.video_features(self.features, video, self.training)      # (B,T,512)
            if hasattr(self, "wavenet_en"):
                audio = ops.TransposeLast2Fn.apply(self.wavenet_en(audio))        # (B,T,Bn)
        if self.use_mcb:
            # sketch + circular convolution -> signed sqrt -> whole-tensor L2 norm -> BatchNorm1d, fused (csrc/mcb.hip)
            bn = self.mcb_bn
            y = ops.McbFusionFn.apply(audio, feats, self.mcb.sketch1.h, self.mcb.sketch1.s, self.mcb.sketch2.h,
                                      self.mcb.sketch2.s, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      self.eps, self.training, bn.momentum)
            if self.training:
                bn.num_batches_tracked += 1
        else:
            y = ops.ConcatColsFn.apply(audio, feats)
        out = ops.lstm_stack(y, lengths, self.lstm_merged)
        return ops.LinearFn.apply(out, self.vad_merged.weight, self.vad_merged.bias)
