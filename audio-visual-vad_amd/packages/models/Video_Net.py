"""Drop-in for ``packages/models/Video_Net.py``: ``DeepVAD_video(lstm_layers, lstm_hidden_size, y_dim)``
with ``forward(x, lengths, return_last=False)`` (reference ``:14-52,58-117``).  ``features`` keeps the
torchvision ResNet-18 child order / key names; its arithmetic runs in csrc/trunk.hip."""
import torch
import torch.nn as nn

from avvad import nn as avnn
from avvad import ops
from packages.models.utils import weights_init_normal


class DeepVAD_video(nn.Module):
    def __init__(self, lstm_layers, lstm_hidden_size, y_dim):
        super().__init__()
        self.lstm_input_size = 512
        self.lstm_layers = lstm_layers
        self.lstm_hidden_size = lstm_hidden_size
        self.y_dim = y_dim
        self.features = avnn.make_resnet18_trunk()
        self.mean = torch.as_tensor([0.485, 0.456, 0.406])   # plain attributes, unused (reference :40-41)
        self.std = torch.as_tensor([0.229, 0.224, 0.225])
        self.lstm_video = nn.LSTM(input_size=512, hidden_size=lstm_hidden_size, num_layers=lstm_layers,
                                  bidirectional=False)
        self.vad_video = nn.Linear(lstm_hidden_size, y_dim)
        self.dropout = nn.Dropout(p=0.5)

    def weight_init(self, mean=0.0, std=0.02):
        for m in self.named_parameters():
            weights_init_normal(m, mean=mean, std=std)

    def forward(self, x, lengths, return_last=False):
        feats = avnn.video_features(self.features, x, self.training)      # (B,T,512)
        out = ops.lstm_stack(feats, lengths, self.lstm_video)
        if return_last:                                                   # method3: last valid step
            idx = (ops.lengths_i32(lengths, out.device).long() - 1)
            out = out[torch.arange(out.shape[0], device=out.device), idx]
        return ops.LinearFn.apply(out, self.vad_video.weight, self.vad_video.bias)
