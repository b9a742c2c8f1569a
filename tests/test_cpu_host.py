"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol (no compute calls
without a GPU), the drop-in classes keep the reference's state_dict keys, host logic (collates, STFT front-end,
metrics), the product path refuses to run without a GPU, and the N>1 data-parallel reducer on gloo."""
import json
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import stategen
from conftest import ROOT, load_golden

T = torch.from_numpy


def test_library_loads_and_exports_every_declared_symbol():
    from avvad import _lib as L
    h = L.lib()
    assert b"gfx950" in h.avvad_version() and h.avvad_abi_version() == L.ABI_VERSION == 3
    header = open(os.path.join(ROOT, "include", "avvad.h")).read()
    declared = set(re.findall(r"\b(avvad_[a-z0-9_]+)\s*\(", header))
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    for name in declared:
        assert hasattr(h, name)


def test_workspace_queries_and_descriptor_validation():
    import ctypes as C
    from avvad import _lib as L
    h = L.lib()
    d = L.TrunkDesc(4, 67, 67, 1, 0.1, 1e-5, 1)
    assert h.avvad_trunk_workspace(C.byref(d)) > 4 * 432832 * 4          # activations alone: 1.73 MB per frame
    assert h.avvad_trunk_workspace(C.byref(L.TrunkDesc(0, 67, 67, 1, 0.1, 1e-5, 1))) == 0
    w = L.WavenetDesc(2, 6143, 1, 32, 32, 256, 2, 16, 20, (C.c_int * 20)(*([2 ** i for i in range(10)] * 2)), 1, 1)
    assert h.avvad_wavenet_workspace(C.byref(w)) > 0
    bad = L.WavenetDesc(2, 100, 1, 32, 32, 256, 2, 16, 20, (C.c_int * 20)(*([2 ** i for i in range(10)] * 2)), 1, 1)
    assert h.avvad_wavenet_workspace(C.byref(bad)) == 0                   # shorter than the receptive field
    assert h.avvad_lstm_workspace(C.byref(L.LstmDesc(4, 6, 768, 1024, None, 1))) > 0
    # NULL pointers are rejected before anything is launched
    assert h.avvad_gemm_f32(None, None, None, None, C.byref(L.GemmDesc(1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0)), None, 0, None) == -1
    assert h.avvad_engine_workspace() == 256 * 576 * 64 * 4    # one weight-gradient partial per CU >= one 128x128 tile per persistent worker
    assert h.avvad_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 1, None) == -1


def test_dropin_state_dict_keys_and_sizes():
    from packages.models.AV_Net import DeepVAD_AV
    from packages.models.Audio_Net import DeepVAD_audio
    from packages.models.Video_Net import DeepVAD_video
    from packages.models.wavenet_autoencoder import wavenet_autoencoder
    from packages.utils import count_parameters
    g = load_golden("av_concat_h16")
    m = DeepVAD_AV(2, 16, 1)
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert [tuple(v.shape) for v in m.state_dict().values()] == [eval(str(s)) for s in g["shapes"]]
    gm = load_golden("av_mcb_keys")
    mm = DeepVAD_AV(2, 16, 1, use_mcb=True)
    assert list(mm.state_dict().keys()) == [str(k) for k in gm["keys"]]
    assert mm.mcb.sketch1.h.dtype == torch.long and set(mm.mcb.sketch2.s.unique().tolist()) <= {-1.0, 1.0}
    mm.float()
    assert mm.mcb.sketch1.h.dtype == torch.long
    s = load_golden("sizes")
    assert count_parameters(DeepVAD_audio(2, 1024, 1)) == int(s["audio"])
    assert count_parameters(DeepVAD_video(2, 1024, 1)) == int(s["video"])
    assert count_parameters(DeepVAD_AV(2, 1024, 1)) == int(s["av"])
    assert count_parameters(DeepVAD_AV(2, 1024, 1, use_mcb=True)) == int(gm["n_params"])
    gw = load_golden("wn_fw3_qc2")
    w = wavenet_autoencoder(3, 2, [1, 2, 4, 1, 2], 8, 6, 5, 7, True)
    assert list(w.state_dict().keys()) == [k[2:] for k in gw if k.startswith("p.")]
    assert w.receptive_field == (3 - 1) * (10 + 1) + 1
    w.load_state_dict({k[2:]: T(v) for k, v in gw.items() if k.startswith("p.")})     # reference checkpoint loads
    # the WaveNet variant re-opens the commented hook: encoder keys + LSTM input = bottleneck + 512
    cfg = dict(filter_width=2, quantization_channel=1, dilations=[1, 2], en_residual_channel=32, en_dilation_channel=32,
               en_bottleneck_width=40, en_pool_kernel_size=4, use_bias=True)
    mw = DeepVAD_AV(2, 16, 1, wavenet_params=cfg)
    assert mw.lstm_merged.input_size == 40 + 512 and "wavenet_en.en_causal_layer.weight" in mw.state_dict()
    # partial load of the video tower by key filter, as scripts/train_AV_net.py:176-187 does
    vid = DeepVAD_video(2, 16, 1)
    sd = m.state_dict()
    sd.update({k: v for k, v in vid.state_dict().items() if "features" in k})
    m.load_state_dict(sd)


def test_names_the_reference_scripts_import():
    """Exactly the ``from packages...`` imports of the six hot-path scripts of the reference must resolve against
    the drop-in package (scripts/train_AV_net.py:16-20, train_audio_net.py:18-21, train_video_net.py:17-19,
    evaluate_{AV,audio,video}_net.py:14-16).  ``packages.data_handling`` / ``packages.visualization`` (HDF5 readers,
    matplotlib figures) are out of scope and not listed."""
    import importlib
    wanted = {
        "packages.models.AV_Net": ["DeepVAD_AV"],
        "packages.models.Video_Net": ["DeepVAD_video"],
        "packages.models.Audio_Net": ["DeepVAD_audio"],
        "packages.models.utils": ["binary_cross_entropy", "binary_cross_entropy_2classes", "f1_loss"],
        "packages.utils": ["count_parameters", "collate_many2many_AV", "collate_many2many_AV_waveform",
                           "collate_many2many_audio", "collate_many2many_audio_waveform", "my_collate",
                           "collate_many2many_video"],
        "packages.processing.stft": ["stft_pytorch"],
        "packages.models.compact_bilinear_pooling": ["CountSketch", "CompactBilinearPooling"],
        "packages.models.wavenet_autoencoder": ["wavenet_autoencoder"],
    }
    for mod, names in wanted.items():
        m = importlib.import_module(mod)
        assert m.__file__.startswith(os.path.join(ROOT, "audio-visual-vad_amd")), m.__file__
        for n in names:
            assert callable(getattr(m, n)), (mod, n)
    # the settings blocks of the entry points parse and name the reference's knobs
    import ast
    for f in ("train_AV_net", "train_audio_net", "train_video_net", "evaluate_AV_net", "evaluate_audio_net", "evaluate_video_net"):
        tree = ast.parse(open(os.path.join(ROOT, "audio-visual-vad_amd", "scripts", f + ".py")).read())
        names = {t.id for n in ast.walk(tree) if isinstance(n, ast.Assign) for t in n.targets if isinstance(t, ast.Name)}
        assert {"lstm_layers", "lstm_hidden_size", "y_dim", "eps", "std_norm"} <= names, (f, names)


def test_schedule_options_are_a_table_not_the_environment(monkeypatch):
    """The library reads AVVAD_* once (first use); later changes go through avvad_set_option only."""
    from avvad import _lib as L
    base = L.get_option("no_streamk")
    monkeypatch.setenv("AVVAD_NO_STREAMK", "all" if base != 1 else "0")
    assert L.get_option("no_streamk") == base                 # the environment is not consulted again
    L.set_option("no_streamk", 1)
    assert L.get_option("no_streamk") == 1
    L.set_option("no_streamk", base)
    with pytest.raises(L.AvvadError):
        L.set_option("no_such_option", 1)
    # every option include/avvad.h names is known to the library, and the integer-valued ones keep their value
    import re
    hdr = open(os.path.join(ROOT, "include", "avvad.h")).read()
    doc = hdr[hdr.index("/* Schedule options"):hdr.index("int avvad_set_option")]
    names = set(re.findall(r'"([a-z0-9_]+)"', doc))
    assert {"no_streamk", "bf16", "max_cus", "wn_flat", "wn_grid", "wn_dx", "no_buf", "wn_bwd_t"} <= names, names
    for n in names:
        v = L.get_option(n)
        L.set_option(n, v)
    for n, v in (("wn_flat", 3), ("wn_grid", 520), ("wn_dx", 2), ("max_cus", 200)):
        old = L.get_option(n)
        L.set_option(n, v)
        assert L.get_option(n) == v
        L.set_option(n, old)


def test_standardisation_stats_and_waveform_loader(tmp_path):
    from avvad.train import Stats, load_waveform
    from conftest import GOLDEN
    np.save(tmp_path / "trainset_audio_mean.npy", np.zeros((513, 1), np.float32))
    np.save(tmp_path / "trainset_audio_std.npy", np.ones((513, 1), np.float32))
    st = Stats.load(str(tmp_path))
    assert st.get("audio_mean", "cpu").shape == (513,) and st.get("video_mean", "cpu") is None
    x = torch.zeros(2, 3, 67, 67)
    assert st.video(x) is x                                    # no video statistics -> untouched
    w, fs = load_waveform(os.path.join(GOLDEN, "utt_sa1.npz"))
    assert fs == 16000 and w.shape == (48100,)
    from scipy.io import wavfile
    wavfile.write(str(tmp_path / "a.wav"), 16000, (w.numpy() * 32768).astype(np.int16))
    w2, _ = load_waveform(str(tmp_path / "a.wav"))
    np.testing.assert_array_equal(w.numpy(), w2.numpy())


def test_no_cpu_fallback():
    from avvad import AvvadError
    from packages.models.Audio_Net import DeepVAD_audio
    from packages.models.utils import binary_cross_entropy
    m = DeepVAD_audio(1, 8, 1)
    with pytest.raises(AvvadError):
        m(torch.zeros(1, 3, 513), [3])
    with pytest.raises(AvvadError):
        binary_cross_entropy(torch.zeros(4, 1), torch.zeros(4, 1), 1e-8)


def test_collates_match_reference_outputs():
    from packages import utils as U
    g = load_golden("collate")
    lens = g["lens"].tolist()
    items = [(stategen.rand(50 + i, 513, n), stategen.rand(60 + i, 67, 67, n), stategen.rand(70 + i, 1, n), n)
             for i, n in enumerate(lens)]
    for j, t in enumerate(U.collate_many2many_AV(items)):
        np.testing.assert_array_equal(t.numpy(), g["av_%d" % j])
        assert t.is_contiguous()
    for j, t in enumerate(U.collate_many2many_audio([(a, y, n) for a, v, y, n in items])):
        np.testing.assert_array_equal(t.numpy(), g["audio_%d" % j])
    for j, t in enumerate(U.collate_many2many_video([(v, y, n) for a, v, y, n in items])):
        np.testing.assert_array_equal(t.numpy(), g["video_%d" % j])
    wl = g["wlens"].tolist()
    items_w = [(stategen.rand(80 + i, wl[i]), v, y, wl[i], n) for i, (a, v, y, n) in enumerate(items)]
    for j, t in enumerate(U.collate_many2many_AV_waveform(items_w)):
        np.testing.assert_array_equal(t.numpy(), g["avw_%d" % j])
    for j, t in enumerate(U.collate_many2many_audio_waveform([(w, y, L, n) for w, v, y, L, n in items_w])):
        np.testing.assert_array_equal(t.numpy(), g["aw_%d" % j])
    out = U.collate_many2many_AV(items)
    assert out[0].dtype == torch.long and out[1].shape == (3, 5, 513) and out[2].shape == (3, 5, 67, 67)
    # many-to-one clip collate: (W,H,C,T_i) -> (B,T,C,H,W)
    clips = [(stategen.rand(1, 6, 5, 3, 4), 1.0, 4), (stategen.rand(2, 6, 5, 3, 2), 0.0, 2)]
    l, d, t = U.my_collate(clips)
    assert d.shape == (2, 4, 3, 5, 6) and float(d[1, 2:].abs().sum()) == 0.0 and t.tolist() == [[1.0], [0.0]]
    np.testing.assert_array_equal(d[0, 1, 2].numpy(), clips[0][0][:, :, 2, 1].t().numpy())


def test_stft_frontend_and_metrics_host_side():
    from oracle import frontend, head
    from packages.models.utils import batch_f1, f1_loss
    from packages.processing.stft import log_power, stft_pytorch
    x = stategen.rand(90, 16000, scale=0.3)
    from avvad import AvvadError
    with pytest.raises(AvvadError):          # host tensors: no CPU / PyTorch fallback, like every other op of the path
        stft_pytorch(x, fs=16e3, wlen_sec=64e-3, win='hann', hop_percent=0.25, center=False, pad_at_end=True)
    ref = frontend.stft(x, fs=16e3, wlen_sec=64e-3, hop_percent=0.25, center=False, pad_at_end=True)
    assert ref.shape == (513, 60, 2)
    np.testing.assert_array_equal(log_power(ref).numpy(), frontend.log_power(ref).numpy())
    with pytest.raises(ValueError):
        stft_pytorch(x, fs=16e3, wlen_sec=50.01e-3)
    g = load_golden("misc")
    f = f1_loss(T(g["f1_pred"]), T(g["f1_true"]), 1e-8)
    np.testing.assert_allclose([float(v) for v in f], g["f1"], rtol=1e-6)
    # batch form == mean over sequences of the per-sequence reference metric on the valid frames
    yh = (stategen.rand(5, 3, 7, 1) > 0).int()
    y = (stategen.rand(6, 3, 7, 1) > 0).long()
    lens = [7, 4, 1]
    per = [head.f1_scores(yh[b, :n].flatten(), y[b, :n].flatten(), 1e-8) for b, n in enumerate(lens)]
    want = [float(sum(p[k] for p in per) / 3) for k in range(4)]
    np.testing.assert_allclose([float(v) for v in batch_f1(yh, y, lens, 1e-8)], want, rtol=1e-6)


# ------------------------------------------------------------------------------------------ N>1 (gloo, CPU)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-visual-vad_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from avvad import dist as avd
    from oracle import head
    avd.init_from_env("gloo")
    torch.manual_seed(0)
    lstm_sd = stategen.make_state(stategen.lstm_spec("l.", 12, 8, 2) + stategen.linear_spec("fc", 8, 1), 3)
    params = {k: torch.nn.Parameter(v.clone()) for k, v in lstm_sd.items()}
    plist = list(params.values())
    flat, offsets = avd.flat_views(plist)
    red = avd.BucketReducer(plist, flat, offsets, bucket_bytes=1 << 10)          # several small buckets
    assert len(red.buckets) > 2
    x = stategen.rand(4, 6, 5, 12)
    tgt = (stategen.rand(5, 6, 5, 1) > 0).float()
    lens = torch.tensor([5, 3, 4, 1, 2, 5])
    xs, ts, ls = avd.shard_batch([x, tgt, lens], rank, world)
    y = head.linear(head.lstm_stack(xs, ls.tolist(), params, "l.", 2), params["fc.weight"], params["fc.bias"])
    head.batch_loss(y, ts, ls.tolist(), 1e-8).backward()
    red.finish()
    if rank == 0:
        torch.save(flat.clone(), out)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_bucket_reducer_gloo_world2(tmp_path):
    """2 ranks, each a shard of the global batch; SUM all-reduce of the flat gradient == single-process
    gradient of the summed loss over the whole batch (scripts/train_AV_net.py:298-302: loss is a SUM)."""
    from avvad import dist as avd
    from oracle import head
    out = str(tmp_path / "flat.pt")
    mp.spawn(_dp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    lstm_sd = stategen.make_state(stategen.lstm_spec("l.", 12, 8, 2) + stategen.linear_spec("fc", 8, 1), 3)
    params = {k: torch.nn.Parameter(v.clone()) for k, v in lstm_sd.items()}
    flat, offsets = avd.flat_views(list(params.values()))
    x = stategen.rand(4, 6, 5, 12)
    tgt = (stategen.rand(5, 6, 5, 1) > 0).float()
    lens = [5, 3, 4, 1, 2, 5]
    y = head.linear(head.lstm_stack(x, lens, params, "l.", 2), params["fc.weight"], params["fc.bias"])
    head.batch_loss(y, tgt, lens, 1e-8).backward()
    np.testing.assert_allclose(got.numpy(), flat.numpy(), rtol=1e-5, atol=1e-7)
    assert float(flat.abs().sum()) > 0
    with pytest.raises(ValueError):
        avd.shard_batch([torch.zeros(5, 2)], 0, 2)


def test_direct_gradient_sinks_notify_the_reducer():
    """Backward kernels accumulate straight into the flat gradient buffer and bypass autograd's AccumulateGrad
    hooks; the bucket reducer must still learn about every parameter (avvad.ops.GRAD_SINKS)."""
    from avvad import dist as avd
    from avvad import ops
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    flat, offsets = avd.flat_views(ps)
    seen = []
    ops.GRAD_SINKS.append(seen.append)
    try:
        tg = [ops._grad_target(p, True) for p in ps]
        assert all(direct for _, direct in tg) and tg[0][0].data_ptr() == flat.data_ptr()
        assert ops._finish_grads(ps, tg) == [None, None] and [id(p) for p in seen] == [id(p) for p in ps]
        fresh = torch.nn.Parameter(torch.randn(2))          # no .grad yet -> a zero temporary goes back to autograd
        g, direct = ops._grad_target(fresh, True)
        assert not direct and float(g.abs().sum()) == 0.0
        assert ops._grad_target(ps[0], False) == (None, False)
    finally:
        ops.GRAD_SINKS.remove(seen.append) if seen.append in ops.GRAD_SINKS else ops.GRAD_SINKS.clear()


def test_bucket_reducer_counts_each_parameter_once_per_step():
    """A parameter whose gradient is written in place can be announced twice (in-place sink + autograd's
    post-accumulate hook): the second announcement must not count, or a bucket is reduced before its last gradients
    exist; finish() re-arms the bookkeeping for the next step."""
    from avvad import dist as avd
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(6)), torch.nn.Parameter(torch.randn(2))]
    flat, offsets = avd.flat_views(ps)
    red = avd.BucketReducer(ps, flat, offsets, bucket_bytes=1 << 30)        # world size 1: one bucket, nothing is sent
    assert len(red.buckets) == 1 and red.buckets[0][2] == 3
    launched = []
    red._launch = lambda b: launched.append(b)
    red._on_grad(ps[0]); red._on_grad(ps[0]); red._on_grad(ps[1]); red._on_grad(ps[1])
    assert red.pending[0] == 2 and launched == []
    red._on_grad(ps[2])
    assert red.pending[0] == 3 and launched == [0]
    red._on_grad(ps[2])
    assert launched == [0]
    red.finish()
    assert red.pending == [0] and not red._seen
    red._on_grad(ps[0])
    assert red.pending[0] == 1


def _dp_absent_worker(rank, world, port, out):
    """Manual 'backward' on CPU tensors: announce gradients in reverse order, as the hooks would."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-visual-vad_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from avvad import dist as avd
    avd.init_from_env("gloo")
    names = ["features.0.weight", "features.1.weight", "bn.weight", "lstm.w", "lstm.b", "vad.weight"]
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (4000, 3000, 10, 5000, 100, 7)]
    flat, offsets = avd.flat_views(ps)
    red = avd.BucketReducer(ps, flat, offsets, bucket_bytes=1 << 30, names=names, min_group_bytes=1 << 10)
    assert [b[2] for b in red.buckets] == [2, 3, 1], red.buckets     # features | bn (too small: rides with lstm) + lstm | vad
    val = lambda r, k, i: float((r + 1) * 100 + k * 10 + i)
    log = []

    def step(k, present_by_rank, early_check=None, double=None):
        flat.zero_()
        for i in (5, 4, 3, 2, 1, 0):                              # backward order
            if i in present_by_rank[rank]:
                ps[i].grad.add_(val(rank, k, i))
                red._on_grad(ps[i])
        if early_check is not None:
            log.append((k, "launched_before_finish", list(red.launched)))
        if double is not None and rank == double:                  # a second backward() before finish(), on one rank only
            red._seen.discard(id(ps[0]))
            red._on_grad(ps[0])
        red.finish()
        for i in range(6):
            want = sum(val(r, k, i) for r in range(world) if i in present_by_rank[r])
            got = flat[offsets[i]:offsets[i] + ps[i].numel()]
            assert float((got - want).abs().max()) == 0.0, (rank, k, i, float(got[0]), want)

    allp = {0, 1, 3, 4, 5}
    step(0, [allp, allp], early_check=True)       # nothing agreed yet: the middle bucket waits for bn.weight until finish()
    step(1, [allp, allp], early_check=True)       # finish(1) reads step 0's bitmap: bn.weight is absent on every rank
    assert red.absent == {2} and red.expected == [2, 2, 1]
    step(2, [allp, allp], early_check=True)       # now the middle bucket goes out from the hooks
    step(3, [allp, allp | {2}])                   # bn.weight comes back on rank 1 ONLY: still summed correctly, nobody hangs
    step(4, [allp | {2}, allp | {2}])             # (finish(4) reads step 3's bitmap: present somewhere -> counted again)
    assert red.absent == set() and red.expected == [2, 3, 1]
    step(5, [allp | {2}, allp | {2}], early_check=True)
    step(6, [allp | {2}, allp | {2}], double=0)   # protocol violation on rank 0 only: no exception inside the hook ...
    raised = False
    try:
        step(7, [allp | {2}, allp | {2}])         # ... every rank reports it at the next finish()
    except RuntimeError as e:
        raised = "protocol violation" in str(e)
    log.append(("raised", raised))
    with open(out + ".%d" % rank, "w") as f:
        json.dump(log, f)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_absent_parameters_are_agreed_between_ranks(tmp_path):
    """Which parameters have no gradient is agreed between the ranks through an all-reduced presence bitmap (never decided
    locally): buckets stop waiting for them, a parameter that comes back on ONE rank only is still reduced by every rank,
    and a double backward() on one rank is reported by all ranks together instead of hanging the others."""
    out = str(tmp_path / "log")
    mp.spawn(_dp_absent_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in range(2):
        with open(out + ".%d" % r) as f:
            log = json.load(f)
        d = {(e[0], e[1]) if len(e) == 3 else e[0]: e[-1] for e in log}
        assert d[(0, "launched_before_finish")] == [True, False, True]
        assert d[(1, "launched_before_finish")] == [True, False, True]
        assert d[(2, "launched_before_finish")] == [True, True, True]
        assert d[(5, "launched_before_finish")] == [True, True, True]
        assert d["raised"] is True


def test_bucket_reducer_close_detaches_the_gradient_sink():
    from avvad import dist as avd
    from avvad import ops
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (40, 30)]
    flat, offsets = avd.flat_views(ps)
    red = avd.BucketReducer(ps, flat, offsets)
    red.world = 1
    red._sink = red._on_grad
    ops.GRAD_SINKS.append(red._sink)
    n0 = len(ops.GRAD_SINKS)
    red.close()
    assert len(ops.GRAD_SINKS) == n0 - 1 and red._sink is None
