"""Drop-in for ``packages/models/Audio_Net.py``: ``DeepVAD_audio(lstm_layers, lstm_hidden_size, y_dim)``
with ``forward(x, lengths)`` (reference ``:12-36,43-60``).

``wavenet_params`` re-opens the hook the reference left commented out (``:22-29,44-45``): when given,
``x`` is a raw waveform (B, quantization_channel, L), the encoder output replaces the 513-bin
spectrogram features and the LSTM input size becomes ``en_bottleneck_width``."""
import torch.nn as nn

from avvad import ops
from packages.models.utils import weights_init_normal
from packages.models.wavenet_autoencoder import wavenet_autoencoder


class DeepVAD_audio(nn.Module):
    def __init__(self, lstm_layers, lstm_hidden_size, y_dim, wavenet_params=None):
        super().__init__()
        self.lstm_layers = lstm_layers
        self.lstm_hidden_size = lstm_hidden_size
        self.y_dim = y_dim
        self.lstm_input_size = 513
        if wavenet_params is not None:
            self.wavenet_en = wavenet_autoencoder(**wavenet_params)
            self.lstm_input_size = wavenet_params["en_bottleneck_width"]
        self.lstm_audio = nn.LSTM(input_size=self.lstm_input_size, hidden_size=lstm_hidden_size,
                                  num_layers=lstm_layers, bidirectional=False)
        self.vad_audio = nn.Linear(lstm_hidden_size, y_dim)
        self.dropout = nn.Dropout(p=0.5)      # constructed, unused -- as in the reference

    def weight_init(self, mean=0.0, std=0.02):
        for m in self.named_parameters():
            weights_init_normal(m, mean=mean, std=std)

    def forward(self, x, lengths):
        if hasattr(self, "wavenet_en"):
            x = ops.TransposeLast2Fn.apply(self.wavenet_en(x))      # (B,Bn,T) -> (B,T,Bn)
        out = ops.lstm_stack(x, lengths, self.lstm_audio)           # padded steps are zero
        return ops.LinearFn.apply(out, self.vad_audio.weight, self.vad_audio.bias)
