"""Training entry point with the settings block of the reference's ``scripts/train_audio_net.py`` (module-level
constants are the configuration, as there).  Run from the package root -- ``python scripts/train_audio_net.py`` -- or one
process per GPU under ``python -m torch.distributed.run --nproc-per-node N scripts/train_audio_net.py``, which replaces
the reference's ``nn.DataParallel(model, device_ids=[0,1,2,3])`` by bucketed RCCL all-reduce.
Inputs: 513-bin log-power spectrogram sequences (B,T,513) or, with WAVENET = True, raw waveforms through the WaveNet encoder.
The loop body (standardise -> forward -> summed masked BCE -> backward -> Adam -> per-sequence F1 -> checkpoint
``Video_Net_epoch_XXX_vloss_Y.pt``) is ``avvad.train.train_main``; a synthetic ragged data source stands in for the
reference's HDF5 datasets (h5py is not installed in this image).  AVVAD_EPOCHS / AVVAD_ITEMS / AVVAD_BATCH override sizes."""
import sys
sys.path.append('.')

from avvad.train import Stats, train_main
from packages.models.Audio_Net import DeepVAD_audio

# Settings (names as in the reference script)
lstm_layers = 2
lstm_hidden_size = 1024
y_dim = 1                 # 1: VAD labels; 513: IBM labels (train_AV_net.py:64-66)
batch_size = 16
learning_rate = 1e-4
end_epoch = 1
eps = 1e-8
std_norm = True           # standardise inputs with the train-set statistics when models/<model_name>/trainset_*.npy exist
model_name = 'audio_Classif_synthetic'
WAVENET = False           # True: raw waveforms through the WaveNet encoder (the hook the reference left commented out)
wavenet_params = dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2,
                      en_residual_channel=32, en_dilation_channel=32, en_bottleneck_width=256,
                      en_pool_kernel_size=16, use_bias=True)


def make_model():
    return DeepVAD_audio(lstm_layers, lstm_hidden_size, y_dim, wavenet_params=wavenet_params if WAVENET else None)


if __name__ == '__main__':
    stats = Stats.load('models/' + model_name, eps) if std_norm else None
    train_main('audio', make_model, model_name, waveform=WAVENET, epochs=end_epoch, batch_size=batch_size,
               lr=learning_rate, stats=stats)
