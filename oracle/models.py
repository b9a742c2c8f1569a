"""Oracle: whole-model forwards of the three reference classes -- CPU fp32.

Restates the ``forward`` bodies over a flat state_dict with the reference's keys:
  * ``DeepVAD_audio.forward(x, lengths)``                 ``packages/models/Audio_Net.py:43-60``
  * ``DeepVAD_video.forward(x, lengths, return_last)``    ``packages/models/Video_Net.py:58-117``
  * ``DeepVAD_AV.forward(audio, video, lengths)``         ``packages/models/AV_Net.py:72-141``
and the place where the WaveNet encoder plugs in (commented hook
``Audio_Net.py:26-29,44-45`` / ``AV_Net.py:102-106``): encoder output (B,Bn,T)
is moved to (B,T,Bn) and replaces the 513-bin spectrogram features.

Test infrastructure only (see ``oracle/__init__.py``).
"""
import torch

from . import fusion, head, resnet18, wavenet


def video_features(sd, x, training):
    """``Video_Net.py:60-81`` / ``AV_Net.py:78-94``: x (B,T,H,W) -> (B,T,512).
    The gray frame is repeated to 3 channels (``unsqueeze(2).repeat``)."""
    B, T, H, W = x.shape
    v = x.unsqueeze(2).repeat(1, 1, 3, 1, 1).view(B * T, 3, H, W)
    return resnet18.trunk_forward(sd, v, training).view(B, T, -1)


def audio_net(sd, x, lengths, num_layers, wavenet_cfg=None):
    if wavenet_cfg is not None:
        wp = {k[len("wavenet_en."):]: v for k, v in sd.items() if k.startswith("wavenet_en.")}
        x = wavenet.encode(wp, x, wavenet_cfg).permute(0, 2, 1)
    y = head.lstm_stack(x, lengths, sd, "lstm_audio.", num_layers)
    return head.linear(y, sd["vad_audio.weight"], sd["vad_audio.bias"])


def video_net(sd, x, lengths, num_layers, training=False, return_last=False):
    f = video_features(sd, x, training)
    y = head.lstm_stack(f, lengths, sd, "lstm_video.", num_layers)
    if return_last:
        y = head.last_valid(y, lengths)
    return head.linear(y, sd["vad_video.weight"], sd["vad_video.bias"])


def av_net(sd, audio, video, lengths, num_layers, use_mcb=False, eps=1e-8, training=False,
           wavenet_cfg=None):
    v = video_features(sd, video, training)
    if wavenet_cfg is not None:
        wp = {k[len("wavenet_en."):]: t for k, t in sd.items() if k.startswith("wavenet_en.")}
        audio = wavenet.encode(wp, audio, wavenet_cfg).permute(0, 2, 1)
    if use_mcb:
        y = fusion.mcb(audio, v, sd["mcb.sketch1.h"], sd["mcb.sketch1.s"], sd["mcb.sketch2.h"],
                       sd["mcb.sketch2.s"], sd["mcb_bn.weight"].numel())
        y = fusion.mcb_post(y, sd["mcb_bn.weight"], sd["mcb_bn.bias"], sd["mcb_bn.running_mean"],
                            sd["mcb_bn.running_var"], eps, training)
    else:
        y = torch.cat([audio, v], dim=2)
    y = head.lstm_stack(y, lengths, sd, "lstm_merged.", num_layers)
    return head.linear(y, sd["vad_merged.weight"], sd["vad_merged.bias"])
