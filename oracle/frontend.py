"""Oracle: STFT log-power front-end and the padding collates -- CPU fp32.

Restates:
  * ``stft_pytorch`` ``packages/processing/stft.py:102-151`` (legacy ``torch.stft``
    real-view output; that call raises on torch 2.x, so: ``return_complex=True`` +
    ``view_as_real``), and the power/log at its callers
    ``scripts/evaluate_audio_net.py:141-148`` / ``packages/data_handling.py:454-457``.
  * the collates ``packages/utils.py:9-226``: zero-pad the last (time) axis to the
    longest sample, stack, move time to axis 1, return ``(lengths, data..., target)``.

Test infrastructure only (see ``oracle/__init__.py``).
"""
import math

import numpy as np
import torch


def stft(x, fs=16e3, wlen_sec=50e-3, hop_percent=0.25, center=True, pad_mode="reflect", pad_at_end=True):
    """x (L,) -> (nfft/2+1, frames, 2).  Defaults as ``stft.py:102-110``; the
    callers pass wlen_sec=64e-3 (1024 samples), hop 0.25, center=False."""
    if wlen_sec * fs != int(wlen_sec * fs):
        raise ValueError("wlen_sample of STFT is not an integer.")
    nfft = int(wlen_sec * fs)
    hop = int(hop_percent * nfft)
    if pad_at_end:
        utt_len = len(x) / fs
        if math.ceil(utt_len / wlen_sec / hop_percent) != int(utt_len / wlen_sec / hop_percent):
            x = torch.nn.functional.pad(x, (0, hop), mode="constant")
    window = torch.hann_window(nfft)
    S = torch.stft(x, n_fft=nfft, hop_length=hop, win_length=None, window=window, center=center,
                   pad_mode=pad_mode, return_complex=True)
    return torch.view_as_real(S)


def stft_naive(x, nfft, hop):
    """Independent definition (center=False): explicit framing + float64 DFT."""
    x = np.asarray(x, dtype=np.float64)
    n = 1 + (len(x) - nfft) // hop
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(nfft) / nfft)        # periodic Hann
    frames = np.stack([x[i * hop: i * hop + nfft] * w for i in range(n)], 0)
    k = np.arange(nfft // 2 + 1)[:, None] * np.arange(nfft)[None, :]
    W = np.exp(-2j * np.pi * k / nfft)
    return (frames @ W.T).T                                         # (bins, frames)


def log_power(S, eps=1e-8):
    """|X|^2 = re^2 + im^2 -> log(. + eps)   (``evaluate_audio_net.py:141-148``)."""
    return torch.log(S[..., 0] ** 2 + S[..., 1] ** 2 + eps)


def audio_features(x_t, mean=None, std=None, n_label_frames=None, fs=16e3, wlen_sec=64e-3, hop_percent=0.25, eps=1e-8):
    """``process_utt`` up to the classifier input, ``scripts/evaluate_audio_net.py:122-163``:
    x / max|x| -> stft_pytorch(center=False, pad_at_end=True) -> re^2 + im^2 -> crop to the label's frame count ->
    log(. + eps) -> transpose to (frames, bins) -> ``x -= mean.T; x /= (std + eps).T`` -> [None].
    x_t (L,) -> (1, T, 513); mean / std are the (513, 1) train-set statistics (or None: ``std_norm`` off)."""
    x_t = x_t / torch.max(torch.abs(x_t))
    x_tf = stft(x_t, fs=fs, wlen_sec=wlen_sec, hop_percent=hop_percent, center=False, pad_at_end=True)
    x = x_tf[..., 0] ** 2 + x_tf[..., 1] ** 2
    if n_label_frames is not None and n_label_frames < x.shape[-1]:
        x = x[..., :n_label_frames]
    x = torch.log(x + eps).T
    if mean is not None:
        x = (x - mean.reshape(-1, 1).T) / (std.reshape(-1, 1) + eps).T
    return x[None]


def standardize(x, mean, std, eps=1e-8):
    """``x_norm = x - mean.T; x_norm /= (std + eps).T`` (``scripts/train_AV_net.py:286-291``): mean/std (F,1) or (1,1)."""
    return (x - mean.reshape(-1, 1).T) / (std.reshape(-1, 1) + eps).T


def pad_time_first(samples, max_len):
    """samples: list of (..., T_i) -> (B, T, ...) zero padded (``utils.py:51-72``)."""
    out = torch.zeros((len(samples),) + tuple(samples[0].shape[:-1]) + (max_len,))
    for i, s in enumerate(samples):
        out[i, ..., : s.shape[-1]] = s
    return out.movedim(-1, 1).contiguous()


def collate_many2many(batch, n_data):
    """Generic form of ``collate_many2many_{video,audio,AV}`` (``utils.py:42-110,148-185``):
    each item = (data_0 .. data_{n-1}, target, length)."""
    lengths = [it[-1] for it in batch]
    T = max(lengths)
    outs = [pad_time_first([it[j] for it in batch], T) for j in range(n_data + 1)]
    return (torch.LongTensor(lengths),) + tuple(outs)


def collate_many2many_waveform(batch, with_video):
    """``collate_many2many_audio_waveform`` / ``_AV_waveform`` (``utils.py:112-146,187-226``):
    item = (wave (L_i,), [video (H,W,T_i)], target (y,T_i), L_i, T_i); the waveform
    is padded to max L and comes back (B, Lmax) (its time axis IS axis 1)."""
    lengths = [it[-1] for it in batch]
    tl = [it[-2] for it in batch]
    T, Lm = max(lengths), max(tl)
    wave = torch.zeros(len(batch), Lm)
    for i, it in enumerate(batch):
        wave[i, : it[0].shape[-1]] = it[0]
    rest = [pad_time_first([it[j] for it in batch], T) for j in range(1, 3 if with_video else 2)]
    return (torch.LongTensor(lengths), wave) + tuple(rest)
