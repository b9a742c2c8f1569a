"""Drop-in for ``stft_pytorch`` of ``packages/processing/stft.py:102-151`` (the librosa twins of that file are
off the hot path).  Same signature and legacy return layout (bins, frames, 2) = (re, im).

The reference calls ``torch.stft`` without ``return_complex`` (an error on torch >= 2); this version asks for
the complex result and returns its real view.  Framing/windowing: optional zero pad of one hop at the end
when the utterance is not a whole number of hops, periodic Hann of the FFT length, ``center`` as given."""
import math

import torch


def stft_pytorch(x, fs=16e3, wlen_sec=50e-3, win='hann', hop_percent=0.25, center=True, pad_mode='reflect',
                 pad_at_end=True):
    if wlen_sec * fs != int(wlen_sec * fs):
        raise ValueError("wlen_sample of STFT is not an integer.")
    nfft = int(wlen_sec * fs)
    hopsamp = int(hop_percent * nfft)
    if x.is_cuda:
        # GPU tensors take the HIP front-end (framing + Hann + DFT as one MFMA GEMM, csrc/stft.hip); a
        # configuration it does not implement is an error, never a silent library call.  center=True is the
        # reference's reflect padding of nfft/2 samples per side, applied (after its end padding) as plain data movement.
        from avvad import ops
        from avvad._lib import AvvadError
        if not (isinstance(win, str) and win == 'hann'):
            raise AvvadError("GPU stft_pytorch implements the reference's periodic Hann window only")
        if nfft % 32:
            raise AvvadError("GPU stft_pytorch needs an FFT length that is a multiple of 32 (got %d)" % nfft)
        if not center:
            return ops.stft(x, nfft, hopsamp, mode=2, pad_at_end=pad_at_end, fs=fs)
        x_ = x
        if pad_at_end:
            n_hops = len(x) / fs / wlen_sec / hop_percent
            if math.ceil(n_hops) != int(n_hops):
                x_ = torch.nn.functional.pad(x, (0, hopsamp), mode='constant')
        x_ = torch.nn.functional.pad(x_.view(1, 1, -1), (nfft // 2, nfft // 2), mode=pad_mode).view(-1)
        return ops.stft(x_, nfft, hopsamp, mode=2, pad_at_end=False, fs=fs)
    # host tensors: like every other op of the path there is no CPU / PyTorch fallback; the CPU restatement
    # of this function lives in oracle/frontend.py (test infrastructure)
    from avvad._lib import AvvadError
    raise AvvadError("stft_pytorch: x must be a GPU tensor -- the AV-VAD front-end has no CPU fallback")


def log_power_spectrogram(x, fs=16e3, wlen_sec=64e-3, hop_percent=0.25, eps=1e-8, pad_at_end=True):
    """Fused form of the callers' stft_pytorch -> re^2+im^2 -> log(.+eps) -> transpose chain
    (scripts/evaluate_audio_net.py:131-163): x (B,L) or (L,) on the GPU -> (B,T,513) log-power features."""
    from avvad import ops
    nfft = int(wlen_sec * fs)
    return ops.stft(x, nfft, int(hop_percent * nfft), mode=0, eps=eps, pad_at_end=pad_at_end, fs=fs)


def log_power(S, eps=1e-8):
    """|X|^2 -> log(. + eps): the callers' post-processing (scripts/evaluate_audio_net.py:141-148)."""
    return torch.log(S[..., 0] ** 2 + S[..., 1] ** 2 + eps)
