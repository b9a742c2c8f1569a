"""Entry point mirroring the reference's ``scripts/evaluate_AV_net.py`` (module-level constants = config; run from the
package root: ``python scripts/evaluate_AV_net.py``, or under ``torch.distributed.run`` for one process per GPU).
Synthetic data only (HDF5 / wav readers are out of scope: SURVEY.md 2.1); override sizes with AVVAD_EPOCHS /
AVVAD_ITEMS / AVVAD_BATCH.  Set WAVENET = True to train on raw waveforms through the WaveNet encoder."""
import sys
sys.path.append('.')

from avvad.train import evaluate_main
from packages.models.AV_Net import DeepVAD_AV

# Settings (names as in the reference script)
lstm_layers = 2
lstm_hidden_size = 1024
y_dim = 1
batch_size = 16
learning_rate = 1e-4
end_epoch = 1
classif_dir = None   # path of a checkpoint written by train_AV_net.py (or by the reference)
WAVENET = False
wavenet_params = dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2,
                      en_residual_channel=32, en_dilation_channel=32, en_bottleneck_width=256,
                      en_pool_kernel_size=16, use_bias=True)


def make_model():
    return DeepVAD_AV(lstm_layers, lstm_hidden_size, y_dim, use_mcb=False, eps=1e-8, wavenet_params=wavenet_params if WAVENET else None)


if __name__ == '__main__':
    evaluate_main('av', make_model, checkpoint=classif_dir, waveform=WAVENET)
