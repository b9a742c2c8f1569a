"""Drop-in for ``packages/utils.py``: ``count_parameters`` (``:5-6``) and the DataLoader collates
(``my_collate :9-40``, ``collate_many2many_video :42-77``, ``_audio :79-110``, ``_audio_waveform :112-146``,
``_AV :148-185``, ``_AV_waveform :187-226``).  Same call signatures and return tuples
``(lengths LongTensor, data..., target)``; every returned tensor is contiguous, batch-first with the time
axis second (B, T, ...), zero padded to the longest sample -- the layout the HIP kernels consume as is.
Built on one helper instead of the reference's per-function pad / unsqueeze / transpose sequence."""
import torch


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def _pad_stack_time_first(samples, max_len):
    """list of (..., T_i) -> (B, T, ...) zero padded, contiguous."""
    first = samples[0]
    out = first.new_zeros((len(samples), max_len) + tuple(first.shape[:-1]))
    for i, s in enumerate(samples):
        out[i, : s.shape[-1]] = s.movedim(-1, 0)
    return out


def _pad_rows(samples, max_len):
    out = samples[0].new_zeros((len(samples), max_len))
    for i, s in enumerate(samples):
        out[i, : s.shape[-1]] = s
    return out


def _collate(batch, n_streams):
    lengths = [item[-1] for item in batch]
    T = max(lengths)
    streams = tuple(_pad_stack_time_first([item[j] for item in batch], T) for j in range(n_streams))
    return (torch.LongTensor(lengths),) + streams


def collate_many2many_video(batch):
    """items (video (H,W,T_i), target (y,T_i), T_i) -> lengths, video (B,T,H,W), target (B,T,y)."""
    return _collate(batch, 2)


def collate_many2many_audio(batch):
    """items (audio (F,T_i), target (y,T_i), T_i) -> lengths, audio (B,T,F), target (B,T,y)."""
    return _collate(batch, 2)


def collate_many2many_AV(batch):
    """items (audio (F,T_i), video (H,W,T_i), target (y,T_i), T_i) -> lengths, audio, video, target."""
    return _collate(batch, 3)


def collate_many2many_audio_waveform(batch):
    """items (wave (L_i,), target (y,T_i), L_i, T_i) -> lengths, wave (B,Lmax), target (B,T,y)."""
    lengths = [item[-1] for item in batch]
    Lmax = max(item[-2] for item in batch)
    return (torch.LongTensor(lengths), _pad_rows([item[0] for item in batch], Lmax),
            _pad_stack_time_first([item[1] for item in batch], max(lengths)))


def collate_many2many_AV_waveform(batch):
    """items (wave (L_i,), video (H,W,T_i), target (y,T_i), L_i, T_i) -> lengths, wave, video, target."""
    lengths = [item[-1] for item in batch]
    T, Lmax = max(lengths), max(item[-2] for item in batch)
    return (torch.LongTensor(lengths), _pad_rows([item[0] for item in batch], Lmax),
            _pad_stack_time_first([item[1] for item in batch], T), _pad_stack_time_first([item[2] for item in batch], T))


def my_collate(batch):
    """items (clip (W,H,C,T_i), label, T_i) -> lengths, clips (B,T,C,H,W), labels (B,1)  (many-to-one)."""
    lengths = [item[2] for item in batch]
    T = max(lengths)
    clips = _pad_stack_time_first([item[0] for item in batch], T)       # (B,T,W,H,C)
    clips = torch.squeeze(clips.permute(0, 1, 4, 3, 2)).contiguous()     # (B,T,C,H,W); the reference squeezes too
    target = torch.zeros((len(batch), 1))
    for i, item in enumerate(batch):
        target[i] = item[1]
    return torch.LongTensor(lengths), clips, target
