#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

    python tools/gen_golden.py            # needs /root/reference (absent on the GPU box)

The reference's own classes/functions are imported from /root/reference and run
on seeded inputs; only inputs, (small) weights and outputs are written -- data,
never reference source.  Two stubs are needed to import it on this image:
``torchvision`` is not installed, so ``torchvision.models.resnet18`` is provided
by ``oracle/resnet18.py`` (the build's restatement of torchvision's published
ResNet-18 -- see that file: tower parity is "unpinned" by the reference), and
``torchvision.transforms`` (imported, unused, ``Video_Net.py:9``) is empty.
Big weights are not stored: they come from the seeded recipe in
``tests/stategen.py``, which the tests replay.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import resnet18 as oracle_resnet  # noqa: E402
import stategen  # noqa: E402

tv = types.ModuleType("torchvision")
tvm = types.ModuleType("torchvision.models")
tvm.resnet18 = oracle_resnet.resnet18
tv.models = tvm
tv.transforms = types.ModuleType("torchvision.transforms")
sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.transforms": tv.transforms})
sys.path.insert(0, REF)          # reference's ``packages`` wins over anything else named so

from packages.models.wavenet_autoencoder import wavenet_autoencoder  # noqa: E402
from packages.models.Audio_Net import DeepVAD_audio  # noqa: E402
from packages.models.Video_Net import DeepVAD_video  # noqa: E402
from packages.models.AV_Net import DeepVAD_AV  # noqa: E402
from packages.models.compact_bilinear_pooling import CountSketch, CountSketchFn_backward  # noqa: E402
from packages.models import utils as ref_mutils  # noqa: E402
from packages import utils as ref_utils  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ WaveNet
WN_CFGS = {
    "wn_tiny": (dict(filter_width=2, quantization_channel=1, dilations=[1, 2, 4], en_residual_channel=4,
                     en_dilation_channel=4, en_bottleneck_width=3, en_pool_kernel_size=5, use_bias=True), 2, 64),
    "wn_fw3_qc2": (dict(filter_width=3, quantization_channel=2, dilations=[1, 2, 4, 1, 2], en_residual_channel=8,
                        en_dilation_channel=6, en_bottleneck_width=5, en_pool_kernel_size=7, use_bias=True), 3, 101),
    "wn_nobias": (dict(filter_width=2, quantization_channel=1, dilations=[1, 2, 4, 8], en_residual_channel=32,
                       en_dilation_channel=32, en_bottleneck_width=16, en_pool_kernel_size=4, use_bias=False), 2, 300),
    # the build-defined production config W0: RF 2048, 1 s chunk -> 60 frames
    "wn_w0": (dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2,
                   en_residual_channel=32, en_dilation_channel=32, en_bottleneck_width=256,
                   en_pool_kernel_size=60, use_bias=True), 1, 16000),
    # C4/C5 per-sequence shape: T=16 frames <-> 4096 valid samples
    "wn_w0_t16": (dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2,
                       en_residual_channel=32, en_dilation_channel=32, en_bottleneck_width=256,
                       en_pool_kernel_size=16, use_bias=True), 2, 6143),
}


def gen_wavenet():
    for i, (name, (cfg, B, L)) in enumerate(WN_CFGS.items()):
        m = wavenet_autoencoder(**cfg)
        sd = stategen.make_state(stategen.wavenet_spec(cfg), 100 + i)
        m.load_state_dict(sd)
        x = stategen.rand(200 + i, B, cfg["quantization_channel"], L, scale=0.5).requires_grad_(True)
        y = m(x)
        G = stategen.rand(300 + i, *y.shape)
        (y * G).sum().backward()
        arrs = dict(x=npy(x), y=npy(y), G=npy(G), dx=npy(x.grad))
        for k, p in m.named_parameters():
            arrs["p." + k] = npy(p)
            arrs["g." + k] = npy(p.grad)
        arrs["cfg_dilations"] = np.array(cfg["dilations"])
        arrs["cfg_scalars"] = np.array([cfg["filter_width"], cfg["quantization_channel"], cfg["en_residual_channel"],
                                        cfg["en_dilation_channel"], cfg["en_bottleneck_width"],
                                        cfg["en_pool_kernel_size"], int(cfg["use_bias"])])
        save(name, **arrs)


# ------------------------------------------------------------------ audio net (packed LSTM semantics)
def gen_audio():
    for name, (L, H, ydim, B, T, lens, seed) in {
        "audio_l2_h16": (2, 16, 1, 3, 7, [7, 5, 2], 1),
        "audio_l1_h32_y3": (1, 32, 3, 4, 6, [3, 6, 1, 6], 2),
    }.items():
        m = DeepVAD_audio(L, H, ydim)
        sd = stategen.make_state(stategen.lstm_spec("lstm_audio.", 513, H, L) +
                                 stategen.linear_spec("vad_audio", H, ydim), seed)
        m.load_state_dict(sd)
        x = stategen.rand(seed + 10, B, T, 513).requires_grad_(True)
        y = m(x, torch.LongTensor(lens))
        tgt = (stategen.rand(seed + 20, B, T, ydim) > 0).float()
        loss = 0.
        for (n, pred, t) in zip(lens, y, tgt):
            loss = loss + ref_mutils.binary_cross_entropy(pred[:n], t[:n], 1e-8)
        loss.backward()
        arrs = dict(x=npy(x), y=npy(y), lengths=np.array(lens), target=npy(tgt), loss=npy(loss), dx=npy(x.grad),
                    meta=np.array([L, H, ydim]))
        for k, p in m.named_parameters():
            arrs["g." + k] = npy(p.grad)
        save(name, **arrs)


# ------------------------------------------------------------------ video net / trunk (via the torchvision stub)
def trunk_spec():
    return oracle_resnet.trunk_keys("features.")


def gen_video():
    H = 16
    spec = trunk_spec() + stategen.lstm_spec("lstm_video.", 512, H, 2) + stategen.linear_spec("vad_video", H, 1)
    m = DeepVAD_video(2, H, 1)
    assert [k for k, _ in spec] == list(m.state_dict().keys()), "state_dict key order drifted"
    n_trunk = sum(p.numel() for k, p in m.named_parameters() if k.startswith("features."))
    assert n_trunk == 11176512, n_trunk
    sd = stategen.make_state(spec, 7)
    B, T = 2, 3
    lens = [3, 2]
    x = stategen.rand(8, B, T, 67, 67)
    arrs = dict(x=npy(x), lengths=np.array(lens))
    for mode in ("eval", "train"):
        m.load_state_dict(sd)
        m.train(mode == "train")
        feats = m.features(x.unsqueeze(2).repeat(1, 1, 3, 1, 1).view(B * T, 3, 67, 67)).squeeze()
        arrs["feat_" + mode] = npy(feats)
        m.load_state_dict(sd)
        y = m(x, torch.LongTensor(lens))
        arrs["y_" + mode] = npy(y)
        if mode == "train":
            new = m.state_dict()
            for k in ("features.1.running_mean", "features.1.running_var", "features.7.1.bn2.running_mean",
                      "features.7.1.bn2.running_var", "features.5.0.downsample.1.running_var"):
                arrs["rs." + k] = npy(new[k])
            arrs["nbt"] = npy(new["features.1.num_batches_tracked"])
    m.load_state_dict(sd)
    m.eval()
    arrs["y_last_eval"] = npy(m(x, torch.LongTensor(lens), return_last=True))
    # N == 1 .squeeze() hazard (Video_Net.py:79)
    arrs["y_single_eval"] = npy(m(x[:1, :1], torch.LongTensor([1])))
    save("video_h16", **arrs)


def gen_av():
    H = 16
    m = DeepVAD_AV(2, H, 1, use_mcb=False, eps=1e-8)
    keys = list(m.state_dict().keys())
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    spec = [(k, shapes[k]) for k in keys]
    sd = stategen.make_state(spec, 11)
    m.load_state_dict(sd)
    B, T, lens = 2, 3, [2, 3]
    a = stategen.rand(12, B, T, 513)
    v = stategen.rand(13, B, T, 67, 67)
    arrs = dict(audio=npy(a), video=npy(v), lengths=np.array(lens),
                keys=np.array(keys), shapes=np.array([str(shapes[k]) for k in keys]))
    m.eval()
    arrs["y_eval"] = npy(m(a, v, torch.LongTensor(lens)))
    m.train()
    arrs["y_train"] = npy(m(a, v, torch.LongTensor(lens)))
    save("av_concat_h16", **arrs)
    # key inventory of the MCB variant (its forward cannot run: torch.rfft is gone)
    mm = DeepVAD_AV(2, H, 1, use_mcb=True, eps=1e-8)
    save("av_mcb_keys", keys=np.array(list(mm.state_dict().keys())),
         shapes=np.array([str(tuple(v.shape)) for v in mm.state_dict().values()]),
         n_params=np.array(sum(p.numel() for p in DeepVAD_AV(2, 1024, 1, use_mcb=True).parameters())))


# ------------------------------------------------------------------ small functions
def gen_misc():
    arrs = {}
    r = stategen.rand(30, 9, 1)
    x = (stategen.rand(31, 9, 1) > 0).float()
    arrs["bce_r"], arrs["bce_x"] = npy(r), npy(x)
    arrs["bce"] = npy(ref_mutils.binary_cross_entropy(r, x, 1e-8))
    big = torch.tensor([[-40.0], [40.0], [0.0], [-120.0], [120.0]])
    bx = torch.tensor([[1.0], [0.0], [1.0], [0.0], [1.0]])
    arrs["bce_big_r"], arrs["bce_big_x"] = npy(big), npy(bx)
    arrs["bce_big"] = npy(ref_mutils.binary_cross_entropy(big, bx, 1e-8))
    yh = (stategen.rand(32, 50) > 0).int()
    yt = (stategen.rand(33, 50) > 0.3).long()
    arrs["f1_pred"], arrs["f1_true"] = npy(yh), npy(yt)
    arrs["f1"] = np.array([float(v) for v in ref_mutils.f1_loss(yh, yt, 1e-8)])
    z = torch.zeros(5, dtype=torch.int32)
    arrs["f1_zero"] = np.array([float(v) for v in ref_mutils.f1_loss(z, z.long(), 1e-8)])
    # count sketch module with fixed h, s
    rng = np.random.RandomState(40)
    h = torch.from_numpy(rng.randint(0, 1024, 513))
    s = torch.from_numpy((2 * rng.randint(0, 2, 513) - 1).astype(np.float32))
    cs = CountSketch(513, 1024, h.clone(), s.clone())
    cx = stategen.rand(41, 2, 3, 513)
    arrs["cs_h"], arrs["cs_s"], arrs["cs_x"], arrs["cs_y"] = npy(h), npy(s), npy(cx), npy(cs(cx))
    cg = stategen.rand(43, 2, 3, 1024)
    arrs["cs_g"], arrs["cs_dx"] = npy(cg), npy(CountSketchFn_backward(h, s, tuple(cx.shape), cg))
    # two-output-unit BCE on probabilities (imported by train_video_net.py:18)
    r1 = torch.sigmoid(stategen.rand(44, 6, 3)).requires_grad_(True)
    r2 = torch.sigmoid(stategen.rand(45, 6, 3)).requires_grad_(True)
    bx = (stategen.rand(46, 6, 3) > 0).float()
    l2 = ref_mutils.binary_cross_entropy_2classes(r1, r2, bx, 1e-8)
    l2.backward()
    arrs.update(bce2_r1=npy(r1), bce2_r2=npy(r2), bce2_x=npy(bx), bce2=npy(l2), bce2_d1=npy(r1.grad), bce2_d2=npy(r2.grad))
    # method3
    from torch.nn.utils.rnn import pack_padded_sequence
    seq = stategen.rand(42, 4, 5, 6)
    lens = torch.LongTensor([2, 5, 1, 3])
    pk = pack_padded_sequence(seq, lens, batch_first=True, enforce_sorted=False)
    arrs["m3_seq"], arrs["m3_len"], arrs["m3_out"] = npy(seq), npy(lens), npy(ref_mutils.method3(pk, lens))
    save("misc", **arrs)

    # collates on ragged toy batches
    carrs = {}
    lens = [4, 2, 5]
    items_av = [(stategen.rand(50 + i, 513, n), stategen.rand(60 + i, 67, 67, n), stategen.rand(70 + i, 1, n), n)
                for i, n in enumerate(lens)]
    for j, t in enumerate(ref_utils.collate_many2many_AV(items_av)):
        carrs["av_%d" % j] = npy(t)
    for j, t in enumerate(ref_utils.collate_many2many_audio([(a, y, n) for a, v, y, n in items_av])):
        carrs["audio_%d" % j] = npy(t)
    for j, t in enumerate(ref_utils.collate_many2many_video([(v, y, n) for a, v, y, n in items_av])):
        carrs["video_%d" % j] = npy(t)
    wl = [1000, 700, 1300]
    items_w = [(stategen.rand(80 + i, wl[i]), v, y, wl[i], n) for i, (a, v, y, n) in enumerate(items_av)]
    for j, t in enumerate(ref_utils.collate_many2many_AV_waveform(items_w)):
        carrs["avw_%d" % j] = npy(t)
    for j, t in enumerate(ref_utils.collate_many2many_audio_waveform([(w, y, L, n) for w, v, y, L, n in items_w])):
        carrs["aw_%d" % j] = npy(t)
    carrs["lens"] = np.array(lens)
    carrs["wlens"] = np.array(wl)
    save("collate", **carrs)
    # count_parameters / model sizes the survey quotes
    save("sizes", audio=np.array(ref_utils.count_parameters(DeepVAD_audio(2, 1024, 1))),
         video=np.array(ref_utils.count_parameters(DeepVAD_video(2, 1024, 1))),
         av=np.array(ref_utils.count_parameters(DeepVAD_AV(2, 1024, 1))))


# ------------------------------------------------------------------ N4 / C1: IBM head (y_dim = 513), a checkpoint written by the
# reference's own class, and one real utterance of data/subset through the evaluator's plumbing
def gen_eval():
    from scipy.io import wavfile
    from oracle import frontend
    fs, wav = wavfile.read(os.path.join(REF, "data/subset/processed/ntcd_timit/Noisy/Babble/-5/test/34M/sa1.wav"))
    assert fs == 16000 and wav.dtype == np.int16 and wav.ndim == 1
    wav = wav[:3 * 16000 + 100]                              # 3 s (+100 samples: the one-hop end-pad branch is taken)
    np.savez_compressed(os.path.join(OUT, "utt_sa1.npz"), samples=wav, fs=np.array(fs))
    x_t = torch.from_numpy(wav.astype(np.float32) / 32768.0)
    mean = stategen.rand(400, 513, 1) * 2.0 - 6.0
    std = stategen.rand(401, 513, 1).abs() + 1.5
    n_label = 180                                            # label shorter than the STFT: frames are cropped (:144-146)
    feats = frontend.audio_features(x_t, mean, std, n_label)  # oracle restatement (stft_pytorch cannot run on torch 2.x)
    arrs = dict(mean=npy(mean), std=npy(std), n_label=np.array(n_label), feats_shape=np.array(feats.shape))
    for tag, ydim in (("y1", 1), ("y513", 513)):
        torch.manual_seed(500 + ydim)
        m = DeepVAD_audio(2, 32, ydim)                       # the REFERENCE class; default torch init
        path = os.path.join(OUT, "audio_ref_h32_%s.pt" % tag)
        torch.save(m.state_dict(), path)                     # the checkpoint format of train_audio_net.py:367-372
        m.eval()
        y = m(feats, [feats.shape[1]])
        soft = torch.sigmoid(y[..., 0].detach())
        arrs["logits_" + tag] = npy(y)
        arrs["soft_" + tag] = npy(soft)
        arrs["hard_" + tag] = npy((soft > 0.5).int())
        print("%-28s %8.1f KB" % (os.path.basename(path), os.path.getsize(path) / 1024))
    save("eval_audio", **arrs)


# ------------------------------------------------------------------ reporting: confidence intervals / stats tables
def gen_metrics():
    import contextlib
    import io
    from packages import metrics as ref_metrics
    rng = np.random.RandomState(77)
    per_utt = rng.rand(24, 4)                                   # accuracy, precision, recall, f1 per utterance
    snr = np.array([-5, 0, 5, 10] * 6, dtype=np.float64)
    noise = np.array((["Babble"] * 8) + (["Cafe"] * 8) + (["Car"] * 8))
    keys = ["accuracy", "precision", "recall", "f1score"]
    ci = np.array([ref_metrics.mean_confidence_interval(per_utt[:, j], confidence=c) for j in range(4) for c in (0.95, 0.9)])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref_metrics.compute_stats(metrics_keys=keys, all_metrics=[tuple(r) for r in per_utt], model_data_dir="", confidence=0.95,
                                  all_snr_db=snr)
    text = buf.getvalue()
    save("metrics", per_utt=per_utt, snr=snr, noise=noise, ci=ci, table=np.frombuffer(text.encode(), dtype=np.uint8))


if __name__ == "__main__":
    which = sys.argv[1:] or ["wavenet", "audio", "video", "av", "misc", "metrics", "eval"]
    for w in which:
        globals()["gen_" + w]()
