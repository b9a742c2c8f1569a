"""Training entry point with the settings block of the reference's ``scripts/train_video_net.py`` (module-level
constants are the configuration, as there).  Run from the package root -- ``python scripts/train_video_net.py`` -- or one
process per GPU under ``python -m torch.distributed.run --nproc-per-node N scripts/train_video_net.py``, which replaces
the reference's ``nn.DataParallel(model, device_ids=[0,1,2,3])`` by bucketed RCCL all-reduce.
Inputs: 67x67 gray lip crops (B,T,67,67); the ResNet-18 trunk is trained end to end (train_video_net.py:141,173).
The loop body (standardise -> forward -> summed masked BCE -> backward -> Adam -> per-sequence F1 -> checkpoint
``Video_Net_epoch_XXX_vloss_Y.pt``) is ``avvad.train.train_main``; a synthetic ragged data source stands in for the
reference's HDF5 datasets (h5py is not installed in this image).  AVVAD_EPOCHS / AVVAD_ITEMS / AVVAD_BATCH override sizes."""
import sys
sys.path.append('.')

from avvad.train import Stats, train_main
from packages.models.Video_Net import DeepVAD_video

# Settings (names as in the reference script)
lstm_layers = 2
lstm_hidden_size = 1024
y_dim = 1                 # 1: VAD labels; 513: IBM labels (train_AV_net.py:64-66)
batch_size = 16
learning_rate = 1e-4
end_epoch = 1
eps = 1e-8
std_norm = True           # standardise inputs with the train-set statistics when models/<model_name>/trainset_*.npy exist
model_name = 'video_Classif_synthetic'


def make_model():
    return DeepVAD_video(lstm_layers, lstm_hidden_size, y_dim)


if __name__ == '__main__':
    stats = Stats.load('models/' + model_name, eps) if std_norm else None
    train_main('video', make_model, model_name, waveform=False, epochs=end_epoch, batch_size=batch_size,
               lr=learning_rate, stats=stats)
