"""Evaluation entry point with the settings block of the reference's ``scripts/evaluate_audio_net.py``: every
utterance goes through the classifier, ``sigmoid``, the 0.5 threshold, and ``*_y_hat_soft.pt`` / ``*_y_hat_hard.pt`` are
written next to each other; utterances are split across ranks (one process per GPU under ``torch.distributed.run``)
like the reference's ``Pool(4)`` of per-GPU workers (``evaluate_audio_net.py`` ``main``).
With ``wav_list`` the reference's ``process_utt`` runs on real 16 kHz audio, all on the GPU: x / max|x| -> STFT -> power ->
log -> crop -> standardise (``avvad.train.process_utt``); without it a synthetic data source is used.
Run from the package root: ``python scripts/evaluate_audio_net.py``; then ``python scripts/run_metrics_dnn_classif.py``."""
import sys
sys.path.append('.')

from avvad.train import Stats, evaluate_main
from packages.models.Audio_Net import DeepVAD_audio

# Settings (names as in the reference script)
lstm_layers = 2
lstm_hidden_size = 1024
y_dim = 1
eps = 1e-8
std_norm = True
fs = int(16e3)            # STFT of the audio branch (evaluate_audio_net.py:40-47)
wlen_sec = 64e-3
hop_percent = 0.25
center = False
pad_at_end = True
classif_dir = None        # checkpoint written by scripts/train_audio_net.py or by the reference (same state_dict keys)
classif_data_dir = 'eval_out'
stats_dir = None          # directory holding trainset_audio_mean.npy / trainset_audio_std.npy (written by the training script)
wav_list = None           # e.g. sorted(glob.glob('data/subset/processed/ntcd_timit/Noisy/*/*/test/*/*.wav'))
WAVENET = False           # True: raw waveforms through the WaveNet encoder (the hook the reference left commented out)
wavenet_params = dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2,
                      en_residual_channel=32, en_dilation_channel=32, en_bottleneck_width=256,
                      en_pool_kernel_size=16, use_bias=True)


def make_model():
    return DeepVAD_audio(lstm_layers, lstm_hidden_size, y_dim, wavenet_params=wavenet_params if WAVENET else None)


if __name__ == '__main__':
    stats = Stats.load(stats_dir, eps) if (std_norm and stats_dir) else None
    evaluate_main('audio', make_model, checkpoint=classif_dir, waveform=WAVENET, out_dir=classif_data_dir, wav_list=wav_list, stats=stats)
