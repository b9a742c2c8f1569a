"""Pin the CPU oracle against the golden vectors produced by running the reference
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import stategen
from conftest import load_golden, wn_cfg_from
from oracle import frontend, fusion, head, models, resnet18, wavenet

T = torch.from_numpy


@pytest.mark.parametrize("name", ["wn_tiny", "wn_fw3_qc2", "wn_nobias", "wn_w0", "wn_w0_t16"])
def test_wavenet_forward_backward(name):
    g = load_golden(name)
    cfg = wn_cfg_from(g)
    params = {k[2:]: T(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith("p.")}
    x = T(g["x"]).clone().requires_grad_(True)
    y = wavenet.encode(params, x, cfg)
    assert y.shape == g["y"].shape
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=0, atol=2e-6)
    (y * T(g["G"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-4, atol=2e-5)
    for k, p in params.items():
        ref = g["g." + k]
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name", ["wn_tiny", "wn_fw3_qc2"])
def test_wavenet_loops_restatement(name):
    """explicit-index restatement (tap order, left-crop, adaptive pool bins) == reference."""
    g = load_golden(name)
    cfg = wn_cfg_from(g)
    params = {k[2:]: T(v) for k, v in g.items() if k.startswith("p.")}
    y = wavenet.encode_loops(params, T(g["x"]), cfg)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=2e-6)
    assert wavenet.receptive_field(2, [2 ** i for i in range(10)] * 2) == 2048


@pytest.mark.parametrize("name", ["audio_l2_h16", "audio_l1_h32_y3"])
def test_audio_net_packed_lstm(name):
    g = load_golden(name)
    L, H, ydim = [int(v) for v in g["meta"]]
    seed = {"audio_l2_h16": 1, "audio_l1_h32_y3": 2}[name]
    sd = stategen.make_state(stategen.lstm_spec("lstm_audio.", 513, H, L) +
                             stategen.linear_spec("vad_audio", H, ydim), seed)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = T(g["x"]).clone().requires_grad_(True)
    lens = g["lengths"].tolist()
    y = models.audio_net(sd, x, lens, L)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=0, atol=2e-6)
    # padded steps output exactly the Linear bias (packed zeros)
    for b, n in enumerate(lens):
        if n < y.shape[1]:
            np.testing.assert_allclose(y[b, n:].detach().numpy(),
                                       np.broadcast_to(sd["vad_audio.bias"].detach().numpy(), y[b, n:].shape), atol=1e-7)
    loss = head.batch_loss(y, T(g["target"]), lens, 1e-8)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-6)
    loss.backward()
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-4, atol=1e-7)
    for k, p in sd.items():
        np.testing.assert_allclose(p.grad.numpy(), g["g." + k], rtol=1e-4, atol=2e-7)


def _video_state():
    spec = resnet18.trunk_keys("features.") + stategen.lstm_spec("lstm_video.", 512, 16, 2) + \
        stategen.linear_spec("vad_video", 16, 1)
    return stategen.make_state(spec, 7)


def test_video_net_and_trunk():
    g = load_golden("video_h16")
    x = T(g["x"])
    lens = g["lengths"].tolist()
    n_trunk = sum(int(np.prod(s)) for k, s in resnet18.trunk_keys() if not k.endswith(
        ("running_mean", "running_var", "num_batches_tracked")))
    assert n_trunk == 11176512
    sd = _video_state()
    f = models.video_features(sd, x, training=False)
    np.testing.assert_allclose(f.reshape(-1, 512).numpy(), g["feat_eval"], rtol=0, atol=1e-5)
    y = models.video_net(sd, x, lens, 2)
    np.testing.assert_allclose(y.numpy(), g["y_eval"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(models.video_net(sd, x, lens, 2, return_last=True).numpy(), g["y_last_eval"], atol=1e-5)
    np.testing.assert_allclose(models.video_net(sd, x[:1, :1], [1], 2).numpy(), g["y_single_eval"], atol=1e-5)
    sd = _video_state()
    y = models.video_net(sd, x, lens, 2, training=True)
    np.testing.assert_allclose(y.numpy(), g["y_train"], rtol=0, atol=1e-5)
    for k in [k for k in g if k.startswith("rs.")]:
        np.testing.assert_allclose(sd[k[3:]].numpy(), g[k], rtol=1e-5, atol=1e-6)


def test_av_net_concat_and_keys():
    g = load_golden("av_concat_h16")
    keys = [str(k) for k in g["keys"]]
    shapes = [eval(str(s)) for s in g["shapes"]]
    sd = stategen.make_state(list(zip(keys, shapes)), 11)
    lens = g["lengths"].tolist()
    a, v = T(g["audio"]), T(g["video"])
    y = models.av_net(sd, a, v, lens, 2)
    np.testing.assert_allclose(y.numpy(), g["y_eval"], rtol=0, atol=1e-5)
    y = models.av_net(sd, a, v, lens, 2, training=True)
    np.testing.assert_allclose(y.numpy(), g["y_train"], rtol=0, atol=1e-5)
    # ``bn`` is constructed, never used, but is part of the checkpoint (AV_Net.py:33)
    assert "bn.weight" in keys and "lstm_merged.weight_ih_l0" in keys and "vad_merged.bias" in keys
    assert shapes[keys.index("lstm_merged.weight_ih_l0")] == (64, 1025)
    gk = load_golden("av_mcb_keys")
    mk = [str(k) for k in gk["keys"]]
    assert {"mcb.sketch1.h", "mcb.sketch1.s", "mcb.sketch2.h", "mcb.sketch2.s", "mcb_bn.running_var"} <= set(mk)
    assert int(gk["n_params"]) == 27974209
    s = load_golden("sizes")
    assert (int(s["audio"]), int(s["video"]), int(s["av"])) == (14701569, 25873985, 27976257)


def test_misc_losses_sketch_method3():
    g = load_golden("misc")
    np.testing.assert_allclose(head.bce_with_eps(T(g["bce_r"]), T(g["bce_x"]), 1e-8).item(), g["bce"], rtol=1e-6)
    np.testing.assert_allclose(head.bce_with_eps(T(g["bce_big_r"]), T(g["bce_big_x"]), 1e-8).item(), g["bce_big"], rtol=1e-6)
    f = head.f1_scores(T(g["f1_pred"]), T(g["f1_true"]), 1e-8)
    np.testing.assert_allclose([float(v) for v in f], g["f1"], rtol=1e-6)
    z = torch.zeros(5, dtype=torch.int32)
    np.testing.assert_allclose([float(v) for v in head.f1_scores(z, z.long(), 1e-8)], g["f1_zero"], rtol=1e-6)
    y = fusion.count_sketch(T(g["cs_x"]), T(g["cs_h"]), T(g["cs_s"]), 1024)
    np.testing.assert_allclose(y.numpy(), g["cs_y"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(head.last_valid(T(g["m3_seq"]), g["m3_len"]).numpy(), g["m3_out"], atol=0)


def test_mcb_vs_naive_outer_product():
    rng = np.random.RandomState(5)
    h1, h2 = T(rng.randint(0, 1024, 513)), T(rng.randint(0, 1024, 512))
    s1 = T((2 * rng.randint(0, 2, 513) - 1).astype(np.float32))
    s2 = T((2 * rng.randint(0, 2, 512) - 1).astype(np.float32))
    x, y = stategen.rand(6, 2, 3, 513), stategen.rand(7, 2, 3, 512)
    out = fusion.mcb(x, y, h1, s1, h2, s2, 1024)
    ref = fusion.mcb_naive(x, y, h1, s1, h2, s2, 1024)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=0, atol=2e-4)


def test_collates():
    g = load_golden("collate")
    lens = g["lens"].tolist()
    items = [(stategen.rand(50 + i, 513, n), stategen.rand(60 + i, 67, 67, n), stategen.rand(70 + i, 1, n), n)
             for i, n in enumerate(lens)]
    for j, t in enumerate(frontend.collate_many2many(items, 2)):
        np.testing.assert_array_equal(t.numpy(), g["av_%d" % j])
        assert t.is_contiguous()
    for j, t in enumerate(frontend.collate_many2many([(a, y, n) for a, v, y, n in items], 1)):
        np.testing.assert_array_equal(t.numpy(), g["audio_%d" % j])
    for j, t in enumerate(frontend.collate_many2many([(v, y, n) for a, v, y, n in items], 1)):
        np.testing.assert_array_equal(t.numpy(), g["video_%d" % j])
    wl = g["wlens"].tolist()
    items_w = [(stategen.rand(80 + i, wl[i]), v, y, wl[i], n) for i, (a, v, y, n) in enumerate(items)]
    for j, t in enumerate(frontend.collate_many2many_waveform(items_w, True)):
        np.testing.assert_array_equal(t.numpy(), g["avw_%d" % j])
    for j, t in enumerate(frontend.collate_many2many_waveform([(w, y, L, n) for w, v, y, L, n in items_w], False)):
        np.testing.assert_array_equal(t.numpy(), g["aw_%d" % j])


@pytest.mark.parametrize("L", [16000, 16001, 4096 + 768])
def test_stft_vs_naive_dft(L):
    """The reference STFT cannot run on torch 2.x (legacy torch.stft signature): the
    restatement is pinned against an explicit float64 DFT instead ("parity unpinned"
    by reference outputs)."""
    x = stategen.rand(90, L, scale=0.3)
    x = x / x.abs().max()
    S = frontend.stft(x, fs=16e3, wlen_sec=64e-3, hop_percent=0.25, center=False, pad_at_end=True)
    import math
    padded = math.ceil(L / 16e3 / 64e-3 / 0.25) != int(L / 16e3 / 64e-3 / 0.25)
    xin = np.concatenate([x.numpy(), np.zeros(256, np.float32)]) if padded else x.numpy()
    ref = frontend.stft_naive(xin, 1024, 256)
    assert S.shape == (513, ref.shape[1], 2)
    if L == 16000:
        assert S.shape[1] == 60 and padded          # 1 s chunk <-> 60 frames
    np.testing.assert_allclose(S[..., 0].numpy(), ref.real, atol=2e-4)
    np.testing.assert_allclose(S[..., 1].numpy(), ref.imag, atol=2e-4)


def test_metrics_tables_match_reference(capsys):
    """packages.metrics against outputs of the reference's own functions: Student-t confidence
    intervals and the printed METRIC / AVERAGE / CONF. INT. tables (overall + per input SNR)."""
    from packages import metrics
    g = load_golden("metrics")
    per_utt = g["per_utt"]
    got = np.array([metrics.mean_confidence_interval(per_utt[:, j], confidence=c) for j in range(4) for c in (0.95, 0.9)])
    np.testing.assert_array_equal(got, g["ci"])
    capsys.readouterr()
    stats = metrics.compute_stats(metrics_keys=["accuracy", "precision", "recall", "f1score"],
                                  all_metrics=[tuple(r) for r in per_utt], model_data_dir="", confidence=0.95, all_snr_db=g["snr"])
    assert capsys.readouterr().out == bytes(g["table"]).decode()
    assert stats["all"]["f1score"]["avg"] == g["ci"][6][0]


def test_bce_2classes_and_count_sketch_backward_vs_reference():
    """``binary_cross_entropy_2classes`` (models/utils.py:115-116) value + gradients and ``CountSketchFn_backward``
    (compact_bilinear_pooling.py:30-38) as the reference computed them."""
    g = load_golden("misc")
    r1 = T(g["bce2_r1"]).clone().requires_grad_(True)
    r2 = T(g["bce2_r2"]).clone().requires_grad_(True)
    l = head.bce_2classes(r1, r2, T(g["bce2_x"]), 1e-8)
    np.testing.assert_allclose(l.item(), g["bce2"], rtol=1e-6)
    l.backward()
    np.testing.assert_allclose(r1.grad.numpy(), g["bce2_d1"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r2.grad.numpy(), g["bce2_d2"], rtol=1e-5, atol=1e-7)
    x = T(g["cs_x"]).clone().requires_grad_(True)
    y = fusion.count_sketch(x, T(g["cs_h"]), T(g["cs_s"]), 1024)
    (y * T(g["cs_g"])).sum().backward()
    np.testing.assert_array_equal(x.grad.numpy(), g["cs_dx"])


@pytest.mark.parametrize("tag,ydim", [("y1", 1), ("y513", 513)])
def test_reference_checkpoint_and_ibm_head(tag, ydim):
    """N4: the ``.pt`` written by ``torch.save(DeepVAD_audio(2, 32, y_dim).state_dict())`` of the REFERENCE class loads
    (weights_only) and the oracle reproduces the reference's logits on the evaluator's features of a real utterance,
    for the VAD head (y_dim 1) and the IBM head (y_dim 513, ``train_AV_net.py:65-66``)."""
    import os
    from conftest import GOLDEN
    from avvad.train import load_waveform
    g = load_golden("eval_audio")
    sd = torch.load(os.path.join(GOLDEN, "audio_ref_h32_%s.pt" % tag), map_location="cpu", weights_only=True)
    x_t, fs = load_waveform(os.path.join(GOLDEN, "utt_sa1.npz"))
    assert fs == 16000 and x_t.dtype == torch.float32 and float(x_t.abs().max()) <= 1.0
    feats = frontend.audio_features(x_t, T(g["mean"]), T(g["std"]), int(g["n_label"]))
    assert tuple(feats.shape) == tuple(g["feats_shape"]) == (1, 180, 513)
    y = models.audio_net(sd, feats, [feats.shape[1]], 2)
    assert y.shape == (1, 180, ydim)
    np.testing.assert_allclose(y.numpy(), g["logits_" + tag], rtol=0, atol=5e-6)
    soft = torch.sigmoid(y[..., 0])
    np.testing.assert_allclose(soft.numpy(), g["soft_" + tag], atol=2e-6)
    # the drop-in class takes the same file (CPU load only: its forward needs the GPU)
    from packages.models.Audio_Net import DeepVAD_audio
    m = DeepVAD_audio(2, 32, ydim)
    m.load_state_dict(sd)
    assert m.vad_audio.weight.shape == (ydim, 32)
