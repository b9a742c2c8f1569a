"""Evaluation entry point with the settings block of the reference's ``scripts/evaluate_video_net.py``: every
utterance goes through the classifier, ``sigmoid``, the 0.5 threshold, and ``*_y_hat_soft.pt`` / ``*_y_hat_hard.pt`` are
written next to each other; utterances are split across ranks (one process per GPU under ``torch.distributed.run``)
like the reference's ``Pool(4)`` of per-GPU workers (``evaluate_video_net.py`` ``main``).
Lip-crop sequences come from a synthetic source (the reference reads them from HDF5, evaluate_video_net.py:191-237).
Run from the package root: ``python scripts/evaluate_video_net.py``; then ``python scripts/run_metrics_dnn_classif.py``."""
import sys
sys.path.append('.')

from avvad.train import Stats, evaluate_main
from packages.models.Video_Net import DeepVAD_video

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
classif_dir = None        # checkpoint written by scripts/train_video_net.py or by the reference (same state_dict keys)
classif_data_dir = 'eval_out'
stats_dir = None          # directory holding trainset_video_mean.npy / trainset_video_std.npy


def make_model():
    return DeepVAD_video(lstm_layers, lstm_hidden_size, y_dim)


if __name__ == '__main__':
    stats = Stats.load(stats_dir, eps) if (std_norm and stats_dir) else None
    evaluate_main('video', make_model, checkpoint=classif_dir, out_dir=classif_data_dir, stats=stats)
