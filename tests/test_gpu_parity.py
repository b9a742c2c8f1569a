"""GPU parity tests: the HIP path (through the C ABI of libavvad_hip.so) against the CPU oracle and
the golden vectors produced by the reference.  Tolerance: fp32, |delta| <= 1e-4 on outputs
(BASELINE.json north_star); gradients are compared with a relative bound of the same order."""
import os

import numpy as np
import pytest
import torch

import stategen
from conftest import load_golden, wn_cfg_from

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _report(name, got, ref, atol, rtol=0.0):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got - ref)
    bound = atol + rtol * np.abs(ref)
    worst = float((err - bound).max()) if err.size else 0.0
    msg = "%-44s max|d|=%.3e  max|ref|=%.3e  bound=%.1e+%.1e*|ref|" % (name, err.max() if err.size else 0, np.abs(ref).max() if ref.size else 0, atol, rtol)
    print(msg)
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "parity.log"), "a") as f:
            f.write(msg + "\n")
    except OSError:
        pass
    assert np.isfinite(got).all(), name + ": non-finite values"
    assert worst <= 0, msg


def _grad_tol(ref):
    return 1e-4 * max(1.0, float(np.abs(ref).max()))


def _report_grad(name, got, ref, scale=1.0, rel_bound=2e-3):
    """Gradient parity.  Element-wise bound 1e-4 * max(1, max|ref|); a deep ReLU stack additionally gets a
    relative-L2 escape hatch (<= 2e-3): a pre-activation within ~1e-7 of zero can take the other side of the
    ReLU under a different (equally valid) fp32 summation order -- the MFMA's ordered fmaf chain vs the CPU
    library's vectorised partial sums -- which moves ONE unit's gradient by a finite amount (observed: one
    flip among ~9M ReLU decisions of the 20-layer encoder, |d| 2e-3 on a handful of entries) while every
    forward value stays within 1e-6.  An indexing bug (wrong tap / crop / mask) gives a relative error of
    O(1) and is still caught."""
    got_n = got.detach().cpu().numpy().astype(np.float64)
    ref_n = (ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)).astype(np.float64)
    tol = scale * _grad_tol(ref_n)
    err = np.abs(got_n - ref_n)
    rel = float(np.linalg.norm(got_n - ref_n) / max(np.linalg.norm(ref_n), 1e-30))
    msg = "%-44s max|d|=%.3e  max|ref|=%.3e  relL2=%.2e  tol=%.1e" % (name, err.max(), np.abs(ref_n).max(), rel, tol)
    print(msg)
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "parity.log"), "a") as f:
            f.write(msg + "\n")
    except OSError:
        pass
    assert np.isfinite(got_n).all(), name + ": non-finite values"
    assert err.max() <= tol or rel <= rel_bound, msg


def _relu_flips(out_gpu, ref_inter):
    """ReLU units of the trunk whose on/off decision differs between the GPU forward behind ``out_gpu`` and the oracle's
    intermediates: a pre-activation within ~1e-7 of zero can land on either side under two equally valid fp32
    summation orders (MFMA's ordered fmaf chain vs the CPU library's vectorised partial sums).  Every forward value still
    agrees to 1e-6, but the flipped unit's gradient path is switched on in one run and off in the other, which moves every
    upstream gradient by a finite amount.  Returns (flipped units, total units)."""
    from avvad import ops
    flips = units = 0
    for k, t in ops.trunk_saved_activations(out_gpu).items():
        r = ref_inter[k]
        assert tuple(t.shape) == tuple(r.shape), (k, t.shape, r.shape)
        flips += int(((t.cpu() > 0) != (r > 0)).sum())
        units += r.numel()
    msg = "ReLU decisions that differ from the oracle: %d of %d" % (flips, units)
    print(msg)
    with open(os.path.join(OUT, "parity.log"), "a") as f:
        f.write(msg + "\n")
    # an indexing error flips a sizeable fraction of a layer; rounding flips are single units (observed: ~1 per 9 M)
    assert flips <= 2 + units // 1000000, msg
    return flips, units


def _oracle_trunk_intermediates(state, video, training):
    """post-ReLU activations of the oracle trunk on video (B,T,H,W) with the ``features.*`` entries of ``state``."""
    from oracle import resnet18
    sd = {k: v.detach().clone() for k, v in state.items() if k.startswith("features.")}
    B, Tn, H, W = video.shape
    with torch.no_grad():
        _, inter = resnet18.trunk_forward(sd, video.reshape(B * Tn, 1, H, W).repeat(1, 3, 1, 1), training, return_intermediates=True)
    return inter


def _grad_bound(flips, strict, relaxed=1e-2):
    """Relative-L2 escape of _report_grad: the strict bound unless a ReLU flip has been PROVEN for this run."""
    return strict if flips == 0 else relaxed


# ------------------------------------------------------------------------------------------ GEMM engine
@pytest.mark.parametrize("M,N,K,tA,tB", [(128, 128, 64, 0, 1), (100, 70, 513, 0, 1), (64, 4096, 1024, 0, 1),
                                          (37, 130, 96, 0, 0), (130, 64, 40, 1, 0), (4096, 513, 48, 1, 0),
                                          (48, 1, 1024, 0, 1), (1025, 300, 7, 1, 1), (256, 256, 4096, 0, 0)])
def test_gemm_variants(M, N, K, tA, tB):
    from avvad import ops
    rng = np.random.RandomState(M + N + K)
    A = rng.normal(size=(K, M) if tA else (M, K)).astype(np.float32)
    B = rng.normal(size=(N, K) if tB else (K, N)).astype(np.float32)
    bias = rng.normal(size=(N,)).astype(np.float32)
    ref = (A.T if tA else A).astype(np.float64) @ (B.T if tB else B).astype(np.float64)
    a, b, bs = T(A).to(DEV), T(B).to(DEV), T(bias).to(DEV)
    c = torch.empty(M, N, device=DEV)
    ops.gemm(a, b, c, M, N, K, A.shape[1], B.shape[1], N, bool(tA), bool(tB), bias=bs)
    _report("gemm %dx%dx%d tA%d tB%d +bias" % (M, N, K, tA, tB), c, ref + bias, 1e-5 * np.sqrt(K) * 4)
    c0 = torch.randn(M, N, device=DEV)
    c1 = c0.clone()
    ops.gemm(a, b, c1, M, N, K, A.shape[1], B.shape[1], N, bool(tA), bool(tB), accumulate=True, split_k=4)
    _report("gemm %dx%dx%d split-k accumulate" % (M, N, K), c1, ref + c0.cpu().numpy(), 1e-5 * np.sqrt(K) * 4)


@pytest.mark.parametrize("M,N,K", [(82944, 128, 64), (36992, 64, 576), (25600, 256, 2304), (9216, 512, 300), (1156 * 128 + 5, 128, 96),
                                   (700, 4096, 40), (64, 1024, 4096)])
def test_engine_streamk_fixup_equals_whole_tile(M, N, K):
    """The engine's production schedule (data-parallel rounds + one stream-K round whose split tiles are combined by the
    fix-up kernel) against its whole-tile schedule on the same operands: several full rounds plus a remainder, K so
    short that most workers of the stream-K round get an EMPTY share (1x1 convolutions: that case once read slabs nobody
    had written), fewer tiles than workers, ragged edges.  Also: two runs are bit-identical (no float atomics)."""
    import ctypes as Ct
    from avvad import _lib as L, ops
    rng = np.random.RandomState(M % 1000 + K)
    A = T(rng.normal(size=(M, K)).astype(np.float32)).to(DEV)
    B = T(rng.normal(size=(K, N)).astype(np.float32)).to(DEV)
    bias = T(rng.normal(size=(N,)).astype(np.float32)).to(DEV)
    lib = L.lib()
    st = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.GemmDesc(M, N, K, K, N, N, 0, 0, 0, 1, 0, 0)
    ews = ops.engine_ws(DEV)
    outs = []
    for ws in (None, ews, ews):
        Cc = torch.full((M, N), 7.0, device=DEV)
        L.check(lib.avvad_gemm_f32(L.ptr(A), L.ptr(B), L.ptr(bias), L.ptr(Cc), Ct.byref(d), L.ptr(ws), 0 if ws is None else ws.numel() * 4, st), "gemm")
        outs.append(Cc)
    assert torch.equal(outs[1], outs[2]), "stream-K + fix-up is not reproducible"
    _report("engine %dx%dx%d stream-K vs whole-tile" % (M, N, K), outs[1], outs[0], 2e-5 * np.sqrt(K), 1e-5)
    acc = torch.full((M, N), 0.5, device=DEV)
    d2 = L.GemmDesc(M, N, K, K, N, N, 0, 0, 1, 4, 0, 0)
    L.check(lib.avvad_gemm_f32(L.ptr(A), L.ptr(B), None, L.ptr(acc), Ct.byref(d2), L.ptr(ews), ews.numel() * 4, st), "gemm acc")
    _report("engine %dx%dx%d accumulate" % (M, N, K), acc, outs[0] - bias + 0.5, 2e-5 * np.sqrt(K), 1e-5)


def test_engine_cu_cap_option(lib_options):
    """option max_cus (bench.py --reserve-cus): the persistent grids of the GEMM engine leave CUs free for RCCL's kernels at
    N > 1; a different worker count is a different split of the same sums."""
    from avvad import ops
    rng = np.random.RandomState(3)
    A, B = T(rng.normal(size=(20000, 300)).astype(np.float32)).to(DEV), T(rng.normal(size=(300, 256)).astype(np.float32)).to(DEV)
    c0 = torch.empty(20000, 256, device=DEV)
    ops.gemm(A, B, c0, 20000, 256, 300, 300, 256, 256)
    lib_options("max_cus", 232)
    c1 = torch.empty(20000, 256, device=DEV)
    ops.gemm(A, B, c1, 20000, 256, 300, 300, 256, 256)
    _report("engine with 232 of 256 CUs", c1, c0, 2e-4, 1e-5)


def test_backward_only_cu_reservation(lib_options):
    """option bwd_max_cus (bench.py at N > 1): only the BACKWARD entry points leave CUs to RCCL -- the gradient all-reduce
    overlaps the backward pass, the forward keeps the whole chip.  The cap is in force exactly while a backward entry point
    runs (the forward is bit-identical to the uncapped one, the gradients differ by the summation order of a different worker
    count), and the process-wide max_cus is back to its value afterwards."""
    from avvad import _lib as L, nn as avnn
    from packages.models.Video_Net import DeepVAD_video
    torch.manual_seed(1)
    m = DeepVAD_video(1, 8, 1).to(DEV).train()
    x = torch.randn(160, 67, 67, device=DEV)
    Gd = torch.randn(160, 512, device=DEV) * 1e-3

    def run():
        for p in m.features.parameters():
            p.grad = None
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        f = avnn.trunk_forward(m.features, x, True)
        (f * Gd).sum().backward()
        m.load_state_dict(sd)                     # (running statistics back: both runs start from the same state)
        return f.detach().clone(), [p.grad.clone() for p in m.features.parameters()]
    f0, g0 = run()
    lib_options("bwd_max_cus", 224)
    f1, g1 = run()
    assert L.get_option("max_cus") == 0 and torch.equal(f0, f1)
    worst = max(float((a - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(g1, g0))
    print("backward with 224 of 256 CUs: worst relL2 of a gradient tensor %.2e" % worst)
    assert 0 < worst < 5e-3          # (a flipped ReLU unit under the other summation order moves a gradient by ~1e-3)


# ------------------------------------------------------------------------------------------ WaveNet encoder
@pytest.mark.parametrize("name,alt", [("wn_tiny", 0), ("wn_fw3_qc2", 0), ("wn_nobias", 0), ("wn_w0", 0), ("wn_w0_t16", 0),
                                      ("wn_nobias", 1), ("wn_w0", 1), ("wn_w0_t16", 1), ("wn_w0", 2), ("wn_w0_t16", 2),
                                      ("wn_nobias", 3), ("wn_w0", 3), ("wn_nobias", 4), ("wn_w0", 4), ("wn_w0_t16", 4),
                                      ("wn_w0_t16", 5), ("wn_fw3_qc2", 5), ("wn_w0", 6), ("wn_w0_t16", 6)])
def test_wavenet_golden(name, alt, lib_options):
    """alt=1: the alternate block backward kept in the library (transposed products, no LDS transposes); alt=2: the kernel
    forms picked beside another stream (dx with resident weights, forward with resident weights + cross-tile prefetch);
    alt=3: round 1's flat-addressed kernels; alt=4: the resident-weights form of the fused block backward (the default is the high-occupancy one when the device is ours)."""
    from packages.models.wavenet_autoencoder import wavenet_autoencoder
    if alt == 1:
        lib_options("wn_bwd_t", 1)
    elif alt == 2:
        lib_options("wn_dx", 1)
        lib_options("wn_flat", 2)
    elif alt == 3:
        lib_options("wn_flat", 1)
        lib_options("wn_dx", 3)
    elif alt == 4:
        lib_options("wn_bwd_t", 3)          # the one-wave-per-SIMD form of the fused backward (picked beside another stream)
    elif alt == 5:
        # option bf16 (BASELINE configs[4]): the R = D = 32 block kernels of W0 are fp32-MFMA kernels of their own and do not
        # change; shapes that go through the GEMM engine (fw = 3, qc = 2: the Conv1d weight gradients) multiply bf16-rounded
        # operands there.  Stated tolerance for those gradients: relative L2 <= 1e-2 (2^-8 per operand, random signs).
        lib_options("bf16", 1)
    elif alt == 6:
        lib_options("wn_no_tail_pair", 1)   # the fused tail backward with one wave per time tile (the default pairs two waves)
    g = load_golden(name)
    cfg = wn_cfg_from(g)
    m = wavenet_autoencoder(**cfg)
    m.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("p.")})
    m = m.to(DEV)
    x = T(g["x"]).to(DEV).requires_grad_(True)
    y = m(x)
    _report(name + " forward", y, g["y"], 1e-4)
    (y * T(g["G"]).to(DEV)).sum().backward()
    rb = 1e-2 if alt == 5 else 2e-3
    _report_grad(name + " d/dx", x.grad, g["dx"], rel_bound=rb)
    for k, p in m.named_parameters():
        _report_grad(name + " d/d" + k, p.grad, g["g." + k], rel_bound=rb)


def test_wavenet_backward_is_bit_reproducible():
    """Two runs of the encoder's forward + backward on the same inputs give the SAME BITS in every gradient: the residual
    blocks' and the tail's weight-gradient partial sums meet in fixed order (wave-ordered LDS adds, slab reductions without
    atomics), like the engine's stream-K fix-up."""
    from packages.models.wavenet_autoencoder import wavenet_autoencoder
    cfg = dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2, en_residual_channel=32,
               en_dilation_channel=32, en_bottleneck_width=256, en_pool_kernel_size=16, use_bias=True)
    torch.manual_seed(7)
    m = wavenet_autoencoder(**cfg).to(DEV)
    x = (torch.rand(24, 1, 6143, device=DEV) * 2 - 1).requires_grad_(True)
    Gd = torch.randn(24, 256, 16, device=DEV)
    runs = []
    for _ in range(3):
        for p in m.parameters():
            p.grad = None
        x.grad = None
        (m(x) * Gd).sum().backward()
        torch.cuda.synchronize()
        runs.append([x.grad.clone()] + [p.grad.clone() for p in m.parameters()])
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)


def test_wavenet_batch_and_tails():
    """ragged tile tails (L not a multiple of 32) and B>1 on the MFMA block path vs the oracle."""
    from oracle import wavenet as ow
    from packages.models.wavenet_autoencoder import wavenet_autoencoder
    cfg = dict(filter_width=2, quantization_channel=1, dilations=[1, 2, 4, 8, 16, 3], en_residual_channel=32,
               en_dilation_channel=32, en_bottleneck_width=48, en_pool_kernel_size=7, use_bias=True)
    torch.manual_seed(3)
    m = wavenet_autoencoder(**cfg)
    x = torch.randn(5, 1, 333)
    ref = ow.encode({k: v.detach() for k, v in m.state_dict().items()}, x, cfg)
    _report("wavenet ragged tails B=5 L=333", m.to(DEV)(x.to(DEV)), ref, 1e-4)


# ------------------------------------------------------------------------------------------ heads
@pytest.mark.parametrize("name,seed", [("audio_l2_h16", 1), ("audio_l1_h32_y3", 2)])
def test_audio_net_golden(name, seed):
    from packages.models.Audio_Net import DeepVAD_audio
    from packages.models.utils import batch_binary_cross_entropy
    g = load_golden(name)
    L, H, ydim = [int(v) for v in g["meta"]]
    m = DeepVAD_audio(L, H, ydim)
    m.load_state_dict(stategen.make_state(stategen.lstm_spec("lstm_audio.", 513, H, L) +
                                          stategen.linear_spec("vad_audio", H, ydim), seed))
    m = m.to(DEV)
    x = T(g["x"]).to(DEV).requires_grad_(True)
    lens = g["lengths"].tolist()
    y = m(x, lens)
    _report(name + " logits", y, g["y"], 1e-4)
    loss = batch_binary_cross_entropy(y, T(g["target"]).to(DEV), torch.LongTensor(lens), 1e-8)
    _report(name + " loss", loss, g["loss"], 1e-4)
    loss.backward()
    _report(name + " d/dx", x.grad, g["dx"], _grad_tol(g["dx"]))
    for k, p in m.named_parameters():
        _report(name + " d/d" + k, p.grad, g["g." + k], _grad_tol(g["g." + k]))


def test_lstm_full_size():
    """production head size (H=1024, 2 layers, In=768) against the oracle's explicit time loop."""
    from avvad import ops
    from oracle import head
    import torch.nn as nn
    torch.manual_seed(0)
    lstm = nn.LSTM(768, 1024, 2)
    x = torch.randn(4, 6, 768)
    lens = [6, 3, 5, 1]
    sd = {k: v.detach() for k, v in lstm.state_dict().items()}
    ref = head.lstm_stack(x, lens, sd, "", 2)
    y = ops.lstm_stack(x.to(DEV), lens, lstm.to(DEV))
    _report("lstm 2x1024 In=768", y, ref, 1e-4)


@pytest.mark.parametrize("B,H,per_step", [(16, 256, 0), (16, 256, 1), (32, 128, 0), (64, 64, 0), (128, 128, 0), (192, 64, 0),
                                          (256, 1024, 0), (64, 1024, 0), (64, 1024, 1), (48, 512, 0), (32, 256, 0)])
def test_lstm_fused_step_sequence_groups(B, H, per_step, lib_options):
    """The recurrent forward: ONE persistent launch for all steps where the shape allows (B <= 64, H in {256, 512, 1024}:
    h_t handed between workgroups by write-through stores and a flag barrier), else the fused step kernel (MFMA, one launch
    per time step, the batch in groups of min(B, 64) sequences: BASELINE config 1's 256 sequences are four groups);
    per_step=1 forces the step kernels on a persistent-eligible shape.  Ragged lengths, forward + backward against the
    oracle's time loop."""
    from avvad import ops
    from oracle import head
    import torch.nn as nn
    if per_step:
        lib_options("lstm_no_persistent", 1)
    torch.manual_seed(B + H)
    In, Tn = 40, 5 if H < 1024 else (3 if B > 64 else 7)
    lstm = nn.LSTM(In, H, 1)
    x = torch.randn(B, Tn, In)
    lens = [int(v) for v in torch.randint(1, Tn + 1, (B,))]
    lens[0] = Tn
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in lstm.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    ref = head.lstm_stack(xr, lens, sd, "", 1)
    Gd = torch.randn(B, Tn, H)
    (ref * Gd).sum().backward()
    lstm = lstm.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = ops.lstm_stack(xg, lens, lstm)
    _report("lstm B=%d H=%d forward" % (B, H), y, ref, 1e-4)
    (y * Gd.to(DEV)).sum().backward()
    _report_grad("lstm B=%d d/dx" % B, xg.grad, xr.grad)
    for k, p in lstm.named_parameters():
        _report_grad("lstm B=%d d/d%s" % (B, k), p.grad, sd[k].grad)


def test_bce_and_metrics():
    from packages.models.utils import binary_cross_entropy, f1_loss
    g = load_golden("misc")
    r = T(g["bce_r"]).to(DEV).requires_grad_(True)
    loss = binary_cross_entropy(r, T(g["bce_x"]).to(DEV), 1e-8)
    _report("bce", loss, g["bce"], 1e-6)
    loss.backward()
    rr = T(g["bce_r"]).requires_grad_(True)
    from oracle import head
    head.bce_with_eps(rr, T(g["bce_x"]), 1e-8).backward()
    _report("bce grad", r.grad, rr.grad, 1e-6)
    _report("bce saturated logits", binary_cross_entropy(T(g["bce_big_r"]).to(DEV), T(g["bce_big_x"]).to(DEV), 1e-8),
            g["bce_big"], 1e-5)
    f = f1_loss(T(g["f1_pred"]).to(DEV), T(g["f1_true"]).to(DEV), 1e-8)
    _report("f1", torch.stack(list(f)), g["f1"], 1e-6)


def test_adam_matches_torch():
    from avvad.optim import FlatAdam
    torch.manual_seed(1)
    ps = [torch.randn(1000, 37), torch.randn(513)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
    opt_ref = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.999))
    opt = FlatAdam(mine, lr=1e-3, betas=(0.9, 0.999))
    for step in range(3):
        gs = [torch.randn_like(p) for p in ps]
        for p, gq in zip(ref, gs):
            p.grad = gq.clone()
        for p, gq in zip(mine, gs):
            p.grad.copy_(gq.to(DEV))
        opt_ref.step()
        opt.step()
        opt.zero_grad()
    for i, (p, q) in enumerate(zip(mine, ref)):
        _report("adam param %d after 3 steps" % i, p, q, 1e-6)


# ------------------------------------------------------------------------------------------ single convolutions
@pytest.mark.parametrize("N,H,W,C,Co,KS,stride,pad", [(3, 17, 17, 64, 64, 3, 1, 1), (2, 17, 17, 64, 128, 3, 2, 1),
                                                      (5, 9, 9, 128, 128, 3, 1, 1), (2, 17, 17, 64, 128, 1, 2, 0),
                                                      (3, 5, 5, 256, 512, 3, 2, 1), (4, 3, 3, 512, 512, 3, 1, 1),
                                                      (2, 11, 7, 32, 36, 3, 1, 1), (2, 67, 67, 1, 64, 7, 2, 3),
                                                      (3, 40, 53, 1, 64, 7, 2, 3), (1, 120, 120, 1, 64, 7, 2, 3)])
def test_conv2d_fwd_dgrad_wgrad(N, H, W, C, Co, KS, stride, pad):
    """implicit-GEMM convolution (odd spatial sizes, stride 2, 1x1) and the 7x7 stem (LDS-resident frame kernel at
    67x67 and at a ragged 40x53; the engine fallback at 120x120, whose padded frame exceeds the LDS budget) vs
    F.conv2d + autograd."""
    import ctypes as Ct
    import torch.nn.functional as F
    from avvad import _lib as L
    rng = np.random.RandomState(N * H + C)
    x = T(rng.normal(size=(N, C, H, W)).astype(np.float32)).requires_grad_(True)
    w = T((rng.normal(size=(Co, C, KS, KS)) / np.sqrt(C * KS * KS)).astype(np.float32)).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad)
    gy = T(rng.normal(size=tuple(y.shape)).astype(np.float32))
    y.backward(gy)
    lib = L.lib()
    st = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(N, H, W, C, Co, KS, stride, pad)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    wd_ = w.detach().to(DEV)
    wf = torch.empty(KS * KS * C * Co, device=DEV)
    wdg = torch.empty(KS * KS * C * Co, device=DEV)
    L.check(lib.avvad_conv2d_pack_weights(L.ptr(wd_), L.ptr(wf), L.ptr(wdg), Ct.byref(d), st), "pack")
    yd = torch.empty(N, y.shape[2], y.shape[3], Co, device=DEV)
    from avvad import ops
    ews = ops.engine_ws(DEV)
    wsz = ews.numel() * 4
    L.check(lib.avvad_conv2d_fwd(L.ptr(xd), L.ptr(wf), L.ptr(yd), Ct.byref(d), L.ptr(ews), wsz, st), "fwd")
    tag = "conv %dx%dx%dx%d->%d k%d s%d" % (N, H, W, C, Co, KS, stride)
    _report(tag + " fwd", yd.permute(0, 3, 1, 2), y, 2e-5, 1e-5)
    gyd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    if C > 1 and Co % 32 == 0:      # dgrad contracts over Co in 32-deep K tiles
        dx = torch.empty_like(xd)
        L.check(lib.avvad_conv2d_dgrad(L.ptr(gyd), L.ptr(wdg), L.ptr(dx), Ct.byref(d), 0, L.ptr(ews), wsz, st), "dgrad")
        _report(tag + " dgrad", dx.permute(0, 3, 1, 2), x.grad, 2e-5, 1e-5)
    dw = torch.empty(KS * KS * C, Co, device=DEV)
    L.check(lib.avvad_conv2d_wgrad(L.ptr(xd), L.ptr(gyd), L.ptr(dw), Ct.byref(d), L.ptr(ews), wsz, st), "wgrad")
    # bit-reproducible: tiles cut along K are combined by the fix-up kernel in a fixed order (no float atomics), and the
    # whole-tile schedule (no workspace) agrees with it up to the summation order
    dw2 = torch.empty_like(dw)
    L.check(lib.avvad_conv2d_wgrad(L.ptr(xd), L.ptr(gyd), L.ptr(dw2), Ct.byref(d), L.ptr(ews), wsz, st), "wgrad again")
    assert torch.equal(dw, dw2), "wgrad is not reproducible run to run"
    y2 = torch.empty_like(yd)
    L.check(lib.avvad_conv2d_fwd(L.ptr(xd), L.ptr(wf), L.ptr(y2), Ct.byref(d), None, 0, st), "fwd, whole-tile schedule")
    _report(tag + " fwd whole-tile vs stream-K", y2, yd, 2e-5, 1e-5)
    ref_dw = w.grad.permute(2, 3, 1, 0).reshape(KS * KS * C, Co)
    # (stem: the contraction runs over every output pixel of every frame; as ONE fp32 chain -- the whole-tile debug
    #  schedule -- it drifts to 2.5e-4 on sums of magnitude 170, split-K partial sums stay under 1e-4)
    _report(tag + " wgrad", dw, ref_dw, 3e-4 if C == 1 else 1e-4, 1e-5)


@pytest.mark.parametrize("N,H,W,C,Co,KS,stride,pad", [(64, 17, 17, 64, 64, 3, 1, 1), (48, 17, 17, 64, 128, 3, 2, 1), (96, 9, 9, 128, 128, 3, 1, 1)])
def test_conv2d_fallback_paths_agree(N, H, W, C, Co, KS, stride, pad, lib_options):
    """The forms that only other operand sizes / tile counts select give the same BITS as the defaults on the same problem:
    flat-addressed gathers with validity selects (operands of 2 GiB and more; option no_buf) against the buffer-addressed
    ones, and the four-wave fix-up kernel (deep splits; option no_fixup1) against the one-wave-per-strip kernel."""
    import ctypes as Ct
    from avvad import _lib as L, ops
    lib = L.lib()
    st = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(N, H, W, C, Co, KS, stride, pad)
    Ho = (H + 2 * pad - KS) // stride + 1
    torch.manual_seed(N + C)
    x = torch.randn(N, H, W, C, device=DEV)
    gy = torch.randn(N, Ho, Ho, Co, device=DEV)
    w = torch.randn(Co, C, KS, KS, device=DEV) / (C * KS * KS) ** 0.5
    wf = torch.empty(KS * KS * C * Co, device=DEV)
    wdg = torch.empty(KS * KS * C * Co, device=DEV)
    L.check(lib.avvad_conv2d_pack_weights(L.ptr(w), L.ptr(wf), L.ptr(wdg), Ct.byref(d), st), "pack")
    ews = ops.engine_ws(DEV)
    wsz = ews.numel() * 4

    def run():
        y = torch.empty(N, Ho, Ho, Co, device=DEV)
        dx = torch.empty_like(x)
        dw = torch.empty(KS * KS * C, Co, device=DEV)
        L.check(lib.avvad_conv2d_fwd(L.ptr(x), L.ptr(wf), L.ptr(y), Ct.byref(d), L.ptr(ews), wsz, st), "fwd")
        L.check(lib.avvad_conv2d_dgrad(L.ptr(gy), L.ptr(wdg), L.ptr(dx), Ct.byref(d), 0, L.ptr(ews), wsz, st), "dgrad")
        L.check(lib.avvad_conv2d_wgrad(L.ptr(x), L.ptr(gy), L.ptr(dw), Ct.byref(d), L.ptr(ews), wsz, st), "wgrad")
        torch.cuda.synchronize()
        return y, dx, dw
    base = run()
    lib_options("no_buf", 1)
    flat = run()
    lib_options("no_buf", 0)
    for a, b in zip(base, flat):
        assert torch.equal(a, b)                       # same arithmetic, different addressing
    lib_options("no_fixup1", 1)
    four = run()
    for a, b, name in zip(base, four, ("fwd", "dgrad", "wgrad")):
        _report("conv %s: four-wave fix-up vs one wave per strip" % name, b, a, 1e-5, 1e-6)   # (summation order of the slabs differs)


@pytest.mark.parametrize("N,H,W", [(3, 17, 17), (70, 9, 13), (1, 3, 3), (1, 1, 1), (1024, 17, 17)])
def test_conv64_weights_stationary_kernel(N, H, W, lib_options):
    """The 64 -> 64 channel 3x3 / 1 / 1 convolutions (ResNet layer1) run on their own weights-stationary kernel (conv64.h):
    forward, data gradient, accumulating data gradient and weight gradient (output-stationary kernel + ordered reduction of the
    per-CU partials) against the engine's implicit GEMM on the same operands (option
    no_conv64; different summation order: 1e-5 relative to the result's scale), at ragged pixel counts (a last tile of 3, 9 and
    1 pixels), a single pixel, and the benchmark's full size; the small cases also against F.conv2d + autograd."""
    import ctypes as Ct
    import torch.nn.functional as F
    from avvad import _lib as L, ops
    lib = L.lib()
    st = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(N, H, W, 64, 64, 3, 1, 1)
    torch.manual_seed(N + H)
    x = torch.randn(N, H, W, 64, device=DEV)
    gy = torch.randn(N, H, W, 64, device=DEV)
    w = torch.randn(64, 64, 3, 3, device=DEV) / 24.0
    wf = torch.empty(9 * 64 * 64, device=DEV)
    wdg = torch.empty(9 * 64 * 64, device=DEV)
    L.check(lib.avvad_conv2d_pack_weights(L.ptr(w), L.ptr(wf), L.ptr(wdg), Ct.byref(d), st), "pack")
    ews = ops.engine_ws(DEV)
    wsz = ews.numel() * 4
    dx0 = torch.randn(N, H, W, 64, device=DEV)

    def run():
        y = torch.empty(N, H, W, 64, device=DEV)
        dx = torch.empty_like(x)
        dxa = dx0.clone()
        L.check(lib.avvad_conv2d_fwd(L.ptr(x), L.ptr(wf), L.ptr(y), Ct.byref(d), L.ptr(ews), wsz, st), "fwd")
        L.check(lib.avvad_conv2d_dgrad(L.ptr(gy), L.ptr(wdg), L.ptr(dx), Ct.byref(d), 0, L.ptr(ews), wsz, st), "dgrad")
        L.check(lib.avvad_conv2d_dgrad(L.ptr(gy), L.ptr(wdg), L.ptr(dxa), Ct.byref(d), 1, L.ptr(ews), wsz, st), "dgrad +=")
        dw = torch.full((9 * 64, 64), float("nan"), device=DEV)
        L.check(lib.avvad_conv2d_wgrad(L.ptr(x), L.ptr(gy), L.ptr(dw), Ct.byref(d), L.ptr(ews), wsz, st), "wgrad")
        torch.cuda.synchronize()
        return y, dx, dxa, dw
    direct = run()
    again = run()
    for a, b in zip(direct, again):
        assert torch.equal(a, b)                       # run to run: the same bits
    # a capped grid (option max_cus: what a data-parallel rank's backward runs under) deals the same tiles to fewer workgroups:
    # forward / data gradient bit for bit, the weight gradient up to the grouping of its per-workgroup partials
    lib_options("max_cus", 200)
    capped = run()
    lib_options("max_cus", 0)
    for a, b in zip(direct[:3], capped[:3]):
        assert torch.equal(a, b)
    assert _max_rel(direct[3], capped[3]) < 1e-5
    lib_options("no_conv64", 1)
    engine = run()
    for a, b, name in zip(direct, engine, ("fwd", "dgrad", "dgrad +=", "wgrad")):
        _report("conv64 %dx%dx%d %s: weights-stationary kernel vs engine" % (N, H, W, name), a, b, 1e-5 * float(b.abs().max()), 0.0)
    assert torch.equal(direct[2] - dx0, direct[2] - dx0) and float((direct[2] - (dx0 + direct[1])).abs().max()) <= 2e-6 * float(direct[2].abs().max())
    if N <= 70:
        xr = x.permute(0, 3, 1, 2).cpu().requires_grad_(True)
        yr = F.conv2d(xr, w.cpu(), None, 1, 1)
        yr.backward(gy.permute(0, 3, 1, 2).cpu())
        _report("conv64 %dx%dx%d fwd vs F.conv2d" % (N, H, W), direct[0].permute(0, 3, 1, 2), yr.detach(), 2e-5, 1e-5)
        _report("conv64 %dx%dx%d dgrad vs autograd" % (N, H, W), direct[1].permute(0, 3, 1, 2), xr.grad, 2e-5, 1e-5)
        wr = w.cpu().requires_grad_(True)
        F.conv2d(x.permute(0, 3, 1, 2).cpu(), wr, None, 1, 1).backward(gy.permute(0, 3, 1, 2).cpu())
        _report("conv64 %dx%dx%d wgrad vs autograd" % (N, H, W), direct[3], wr.grad.permute(2, 3, 1, 0).reshape(9 * 64, 64), 1e-4, 1e-5)


# ------------------------------------------------------------------------------------------ trunk
def _video_state():
    from oracle import resnet18
    spec = resnet18.trunk_keys("features.") + stategen.lstm_spec("lstm_video.", 512, 16, 2) + \
        stategen.linear_spec("vad_video", 16, 1)
    return stategen.make_state(spec, 7)


def test_video_net_golden_eval_and_train():
    from packages.models.Video_Net import DeepVAD_video
    g = load_golden("video_h16")
    x = T(g["x"]).to(DEV)
    lens = g["lengths"].tolist()
    m = DeepVAD_video(2, 16, 1)
    m.load_state_dict(_video_state())
    m = m.to(DEV).eval()
    from avvad import nn as avnn
    f = avnn.video_features(m.features, x, False)
    # features reach |x| ~ 200 with the seeded BN statistics: 1e-4 absolute is below fp32 resolution there,
    # so intermediate features get atol 1e-4 + rtol 1e-5; the logits keep the plain 1e-4 of the north star
    _report("trunk features eval", f.reshape(-1, 512), g["feat_eval"], 1e-4, 1e-5)
    _report("video net eval", m(x, lens), g["y_eval"], 1e-4)
    _report("video net eval return_last", m(x, lens, return_last=True), g["y_last_eval"], 1e-4)
    _report("video net single frame", m(x[:1, :1], [1]), g["y_single_eval"], 1e-4)
    m.load_state_dict(_video_state())
    m.train()
    _report("video net train-mode BN", m(x, torch.LongTensor(lens)), g["y_train"], 1e-4)
    sd = m.state_dict()
    for k in [k for k in g if k.startswith("rs.")]:
        _report("running stat " + k[3:], sd[k[3:]], g[k], 1e-5, 1e-5)
    assert int(sd["features.1.num_batches_tracked"]) == 1


@pytest.mark.parametrize("training,streamk", [(True, False), (False, False), (True, True), (False, True)])
def test_trunk_backward_vs_oracle(training, streamk, lib_options):
    """gradients of a random projection of the features w.r.t. every trunk parameter.

    streamk=False runs the engine with whole-tile scheduling (option no_streamk), streamk=True the production schedule
    (data-parallel rounds + a stream-K round whose split tiles are combined by the fix-up kernel in a fixed order).  Both
    are bit-reproducible -- round 1's float-atomic stream-K was not -- and both are held to the strict bounds.  The one
    legitimate way to miss them is a ReLU unit whose pre-activation sits within rounding of zero and lands on the other
    side than in the oracle (different but equally valid fp32 summation orders): that moves every upstream gradient by
    ~3e-3 relative L2 while the forward agrees to 1e-6.  The test therefore COUNTS such units from the saved activations
    (_relu_flips) and relaxes the bound only when one is found; an indexing bug flips a whole layer's worth and fails."""
    from oracle import resnet18
    from avvad import nn as avnn
    from packages.models.Video_Net import DeepVAD_video
    lib_options("no_streamk", 0 if streamk else 1)
    sd0 = _video_state()
    N = 6
    x = stategen.rand(21, N, 67, 67)
    G = stategen.rand(22, N, 512)
    sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone())
          for k, v in sd0.items() if k.startswith("features.")}
    ref, inter = resnet18.trunk_forward(sd, x[:, None].repeat(1, 3, 1, 1), training, return_intermediates=True)
    (ref * G).sum().backward()
    m = DeepVAD_video(2, 16, 1)
    m.load_state_dict(sd0)
    m = m.to(DEV).train(training)
    f = avnn.trunk_forward(m.features, x.to(DEV), training)
    _report("trunk fwd (training=%s, streamk=%s)" % (training, streamk), f, ref, 1e-4, 1e-5)
    flips, _ = _relu_flips(f, inter)
    (f * G.to(DEV)).sum().backward()
    # both schedules are deterministic now (no float atomics): both are held to the strict bounds, which open up only when
    # a flipped ReLU unit has been found in THIS run's activations
    rel = _grad_bound(flips, 2e-3 if not training else 5e-3)
    for k, p in m.features.named_parameters():
        _report_grad("trunk d/d%s" % k, p.grad, sd["features." + k].grad, 2.0, rel)


def test_av_net_golden_concat():
    from packages.models.AV_Net import DeepVAD_AV
    g = load_golden("av_concat_h16")
    keys = [str(k) for k in g["keys"]]
    shapes = [eval(str(s)) for s in g["shapes"]]
    m = DeepVAD_AV(2, 16, 1)
    m.load_state_dict(stategen.make_state(list(zip(keys, shapes)), 11))
    m = m.to(DEV).eval()
    a, v = T(g["audio"]).to(DEV), T(g["video"]).to(DEV)
    lens = g["lengths"].tolist()
    _report("AV net eval", m(a, v, lens), g["y_eval"], 1e-4)
    m.train()
    _report("AV net train", m(a, v, torch.LongTensor(lens).to(DEV)), g["y_train"], 1e-4)


def test_av_net_mcb_fusion_vs_oracle():
    """use_mcb=True (the reference's default fusion): count sketch + circular convolution + signed sqrt + whole-tensor
    L2 norm + BatchNorm1d(eps=1e-8), forward and all gradients, train and eval mode.  The reference forward cannot
    run on torch 2.x (torch.rfft): the oracle restates it with torch.fft and is pinned against the naive
    outer-product sketch (tests/test_oracle_golden.py) -- "parity unpinned" by reference outputs."""
    from oracle import head, models
    from packages.models.AV_Net import DeepVAD_AV
    from packages.models.utils import batch_binary_cross_entropy
    torch.manual_seed(2)
    m = DeepVAD_AV(1, 32, 1, use_mcb=True, eps=1e-8)
    B, Tn = 3, 4
    a = torch.randn(B, Tn, 513)
    v = torch.randn(B, Tn, 67, 67)
    tgt = (torch.rand(B, Tn, 1) > 0.5).float()
    lens = [4, 2, 3]
    for training in (False, True):
        sd = {k: (t.detach().clone().requires_grad_(True) if t.dtype == torch.float32 and "running" not in k and ".s" not in k[-2:]
                  else t.detach().clone()) for k, t in m.state_dict().items()}
        ar = a.clone().requires_grad_(True)
        ref = models.av_net(sd, ar, v, lens, 1, use_mcb=True, eps=1e-8, training=training)
        ref_loss = head.batch_loss(ref, tgt, lens, 1e-8)
        ref_loss.backward()
        mg = DeepVAD_AV(1, 32, 1, use_mcb=True, eps=1e-8)
        mg.load_state_dict(m.state_dict())
        mg = mg.to(DEV).train(training)
        ag = a.clone().to(DEV).requires_grad_(True)
        y = mg(ag, v.to(DEV), lens)
        tag = "AV+MCB %s" % ("train" if training else "eval")
        _report(tag + " logits", y, ref, 1e-4)
        flips, _ = _relu_flips(y, _oracle_trunk_intermediates(m.state_dict(), v, training))
        loss = batch_binary_cross_entropy(y, tgt.to(DEV), lens, 1e-8)
        loss.backward()
        rel = _grad_bound(flips, 5e-3 if training else 2e-3)
        _report_grad(tag + " d/d audio", ag.grad, ar.grad, 2.0, rel)
        for k in ("mcb_bn.weight", "mcb_bn.bias", "lstm_merged.weight_ih_l0", "features.7.1.conv2.weight", "features.0.weight"):
            _report_grad(tag + " d/d" + k, dict(mg.named_parameters())[k].grad, sd[k].grad, 2.0, rel)
        if training:
            _report(tag + " running_var", mg.mcb_bn.running_var, sd["mcb_bn.running_var"], 1e-6, 1e-4)
            assert int(mg.mcb_bn.num_batches_tracked) == 1
    from avvad import AvvadError
    with pytest.raises(AvvadError):
        mg.mcb(ag.cpu(), ag.cpu())            # the bare modules run on the GPU only: no PyTorch fallback


def test_av_wavenet_end_to_end_vs_oracle():
    """The north-star model: WaveNet encoder + ResNet-18 tower + LSTM/FC head, loss and all gradients."""
    from oracle import head, models
    from packages.models.AV_Net import DeepVAD_AV
    from packages.models.utils import batch_binary_cross_entropy
    wcfg = dict(filter_width=2, quantization_channel=1, dilations=[1, 2, 4, 8, 16, 32], en_residual_channel=32,
                en_dilation_channel=32, en_bottleneck_width=64, en_pool_kernel_size=4, use_bias=True)
    torch.manual_seed(5)
    m = DeepVAD_AV(2, 32, 1, wavenet_params=wcfg)
    B, Tn = 3, 4
    rf = 64
    wave = torch.randn(B, 1, Tn * 256 + rf - 1) * 0.3
    video = torch.randn(B, Tn, 67, 67)
    tgt = (torch.rand(B, Tn, 1) > 0.5).float()
    lens = [4, 2, 3]
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sd = {k: (v.detach().clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.detach().clone())
          for k, v in m.state_dict().items()}
    ref = models.av_net(sd, wave, video, lens, 2, training=True, wavenet_cfg=wcfg)
    ref_loss = head.batch_loss(ref, tgt, lens, 1e-8)
    ref_loss.backward()
    m = m.to(DEV).train()
    y = m(wave.to(DEV), video.to(DEV), torch.LongTensor(lens))
    _report("AV+WaveNet logits (train)", y, ref, 1e-4)
    flips, _ = _relu_flips(y, _oracle_trunk_intermediates(sd0, video, True))
    loss = batch_binary_cross_entropy(y, tgt.to(DEV), lens, 1e-8)
    _report("AV+WaveNet loss", loss, ref_loss, 1e-4)
    loss.backward()
    for k, p in m.named_parameters():
        if p.grad is None:
            assert k.startswith("bn."), k          # the unused BatchNorm1d of the reference
            continue
        # train-mode BatchNorm over 12 frames amplifies rounding differences; the bound relaxes only on a proven ReLU flip
        _report_grad("AV+WaveNet d/d" + k, p.grad, sd[k].grad, 2.0, _grad_bound(flips, 5e-3))


# ------------------------------------------------------------------------------------------ entry points
def test_train_and_evaluate_entry_points(tmp_path, monkeypatch):
    """scripts/train_AV_net.py + evaluate_AV_net.py bodies on synthetic ragged batches (WaveNet variant):
    the loss must go down and the evaluator must write the reference's *_y_hat_{soft,hard}.pt files."""
    import os
    from avvad import train as T
    from packages.models.AV_Net import DeepVAD_AV
    monkeypatch.chdir(tmp_path)
    wcfg = dict(filter_width=2, quantization_channel=1, dilations=[1, 2, 4, 8], en_residual_channel=32,
                en_dilation_channel=32, en_bottleneck_width=32, en_pool_kernel_size=16, use_bias=True)

    class Fixed(T.SyntheticAV):     # fixed T so the pool size matches every item
        def __init__(self, *a, **k):
            k.update(t_min=16, t_max=16, rf=16)
            super().__init__(*a, **k)
    monkeypatch.setattr(T, "SyntheticAV", Fixed)
    make = lambda: DeepVAD_AV(1, 32, 1, wavenet_params=wcfg)
    monkeypatch.setenv("AVVAD_EPOCHS", "1")
    model = T.train_main("av", make, "unit", waveform=True, epochs=1, batch_size=8, n_items=32, lr=1e-3,
                         out_dir=str(tmp_path / "m"))
    ck = [f for f in os.listdir(tmp_path / "m") if f.endswith(".pt")]
    assert len(ck) == 1 and ck[0].startswith("Video_Net_epoch_001_vloss_")
    log = open(tmp_path / "m" / "output_batch.log").read()
    assert "Number of learnable parameters" in log and "====> Epoch:  1" in log
    T.evaluate_main("av", make, checkpoint=str(tmp_path / "m" / ck[0]), waveform=True, n_items=3, out_dir=str(tmp_path / "e"))
    outs = sorted(os.listdir(tmp_path / "e"))
    assert outs == sorted(["utt%04d_%s.pt" % (i, k) for i in range(3) for k in ("y_hat_hard", "y_hat_soft", "label")])
    soft = torch.load(tmp_path / "e" / "utt0000_y_hat_soft.pt", weights_only=True)
    assert soft.shape == (1, 16) and float(soft.min()) >= 0 and float(soft.max()) <= 1
    # run_metrics: per-utterance accuracy / precision / recall / F1 -> mean +- Student-t half-width
    stats = T.metrics_main(str(tmp_path / "e"))
    assert set(stats["all"]) == {"accuracy", "precision", "recall", "f1score"}
    assert 0.0 <= stats["all"]["accuracy"]["avg"] <= 1.0


def test_training_steps_match_cpu_adam():
    """3 optimisation steps (FlatAdam over the flat buffer, gradients accumulated in place by the HIP backward)
    against the oracle model trained with torch.optim.Adam on the CPU: same loss trajectory."""
    from avvad.optim import FlatAdam
    from oracle import head, models
    from packages.models.AV_Net import DeepVAD_AV
    from packages.models.utils import batch_binary_cross_entropy
    wcfg = dict(filter_width=2, quantization_channel=1, dilations=[1, 2, 4, 8], en_residual_channel=32,
                en_dilation_channel=32, en_bottleneck_width=32, en_pool_kernel_size=4, use_bias=True)
    torch.manual_seed(11)
    m = DeepVAD_AV(1, 32, 1, wavenet_params=wcfg)
    B, Tn = 4, 4
    wave = torch.randn(B, 1, Tn * 256 + 15) * 0.3
    video = torch.randn(B, Tn, 67, 67)
    tgt = (torch.rand(B, Tn, 1) > 0.5).float()
    lens = [4, 3, 4, 2]
    sd = {k: (v.detach().clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.detach().clone())
          for k, v in m.state_dict().items()}
    cpu_params = [v for k, v in sd.items() if v.requires_grad and not k.startswith("bn.")]
    opt_ref = torch.optim.Adam(cpu_params, lr=1e-3, betas=(0.9, 0.999))
    m = m.to(DEV).train()
    opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
    ref_losses, losses = [], []
    for step in range(3):
        y = models.av_net(sd, wave, video, lens, 1, training=True, wavenet_cfg=wcfg)
        l = head.batch_loss(y, tgt, lens, 1e-8)
        opt_ref.zero_grad()
        l.backward()
        opt_ref.step()
        ref_losses.append(float(l))
        yg = m(wave.to(DEV), video.to(DEV), torch.LongTensor(lens))
        lg = batch_binary_cross_entropy(yg, tgt.to(DEV), lens, 1e-8)
        lg.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(lg))
    _report("loss trajectory over 3 Adam steps", np.array(losses), np.array(ref_losses), 2e-3)
    assert losses[2] < losses[0]
    # the in-place path really was used: .grad tensors are views of the flat buffer
    p0 = next(m.parameters())
    assert p0.grad.data_ptr() >= opt.flat_grad.data_ptr() and p0.grad.data_ptr() < opt.flat_grad.data_ptr() + 4 * opt.flat_grad.numel()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_av_training_is_bit_reproducible(dtype, lib_options):
    """(bf16: the same under BASELINE configs[4]'s arithmetic.)  Two trainings from the same initial state (W0 encoder, ResNet-18 trunk, 2 x LSTM(1024), FC; 16 sequences x 16
    frame-pairs; 2 Adam steps, the encoder on its side stream) end in the SAME BITS: no kernel of the step adds in arrival
    order (stream-K fix-up, BatchNorm's two-stage sums, the encoder's slab reductions, one-block loss)."""
    import copy
    from avvad.optim import FlatAdam
    from packages.models.AV_Net import DeepVAD_AV
    from packages.models.utils import batch_binary_cross_entropy
    wcfg = dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2, en_residual_channel=32,
                en_dilation_channel=32, en_bottleneck_width=256, en_pool_kernel_size=16, use_bias=True)
    if dtype == "bf16":
        lib_options("bf16", 1)
    torch.manual_seed(3)
    m0 = DeepVAD_AV(2, 1024, 1, wavenet_params=wcfg)
    B, Tn = 16, 16
    wave = (torch.rand(B, 1, Tn * 256 + 2047) * 2 - 1).to(DEV)
    video = torch.randn(B, Tn, 67, 67).to(DEV)
    tgt = (torch.rand(B, Tn, 1) > 0.5).float().to(DEV)
    lens = torch.randint(3, Tn + 1, (B,))
    lens[0] = Tn
    finals = []
    for run in range(2):
        m = copy.deepcopy(m0).to(DEV).train()
        opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
        for step in range(2):
            loss = batch_binary_cross_entropy(m(wave, video, lens), tgt, lens.tolist(), 1e-8)
            loss.backward()
            opt.step()
            opt.zero_grad()
        torch.cuda.synchronize()
        finals.append(([p.detach().clone() for p in m.parameters()] + [b.detach().clone() for b in m.buffers()], float(loss.detach())))
    assert finals[0][1] == finals[1][1]
    for a, b in zip(finals[0][0], finals[1][0]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("L", [16000, 16001, 4096 + 768])
def test_stft_frontend_gpu(L):
    """GPU STFT (framing + Hann + DFT as one MFMA GEMM) vs the oracle restatement of stft_pytorch (torch.stft on the
    CPU) incl. the pad / no-pad branch, and the fused log-power features of a ragged-free batch."""
    from oracle import frontend
    from packages.processing.stft import log_power_spectrogram, stft_pytorch
    x = stategen.rand(90, L, scale=0.3)
    x = x / x.abs().max()
    ref = frontend.stft(x, fs=16e3, wlen_sec=64e-3, hop_percent=0.25, center=False, pad_at_end=True)
    S = stft_pytorch(x.to(DEV), fs=16e3, wlen_sec=64e-3, win='hann', hop_percent=0.25, center=False, pad_at_end=True)
    assert tuple(S.shape) == tuple(ref.shape)
    _report("stft re/im L=%d" % L, S, ref, 5e-4)
    xb = torch.stack([x, x.flip(0) * 0.5])
    lp = log_power_spectrogram(xb.to(DEV))
    pw_ref = torch.stack([(frontend.stft(r, fs=16e3, wlen_sec=64e-3, hop_percent=0.25, center=False) ** 2).sum(-1).t() for r in xb])
    _report("stft power L=%d" % L, torch.exp(lp) - 1e-8, pw_ref, 2e-3, 1e-4)
    # the reference's default center=True (reflect padding), same kernel behind it
    ref_c = frontend.stft(x, fs=16e3, wlen_sec=64e-3, hop_percent=0.25, center=True, pad_at_end=True)
    S_c = stft_pytorch(x.to(DEV), fs=16e3, wlen_sec=64e-3, win='hann', hop_percent=0.25, center=True, pad_at_end=True)
    assert tuple(S_c.shape) == tuple(ref_c.shape)
    _report("stft centred re/im L=%d" % L, S_c, ref_c, 5e-4)
    from avvad import AvvadError
    with pytest.raises(AvvadError):      # no silent library fallback for what the kernel does not implement
        stft_pytorch(x.to(DEV), fs=16e3, wlen_sec=64e-3, win=torch.ones(1024, device=DEV), center=False)


def test_training_is_reproducible_under_allocator_churn_and_stream_timing():
    """Three Adam steps of the small AV model (encoder on the side HIP stream), repeated 8 times in one process with the
    caching allocator perturbed in between (different block addresses, stale contents, hipMalloc stalls that shift the two
    streams against each other): logits and every gradient must come out in the SAME BITS.  This is what caught the
    optimiser overtaking the encoder's backward on the side stream (gradients written in place bypass autograd's stream
    join) and a bias gradient summed with float atomics; a fixed allocation pattern hides both."""
    import dp_gpu_case as case
    from avvad.optim import FlatAdam
    from packages.models.utils import batch_binary_cross_entropy

    def run(churn):
        if churn:
            torch.cuda.empty_cache()
            junk = [torch.full((1 << (10 + i % 14),), float("nan"), device=DEV) for i in range(churn)]
            del junk
        m = case.make_model().to(DEV).train()
        wave, video, target, lengths = [t.to(DEV) for t in case.make_batch()]
        opt = FlatAdam(m.parameters(), lr=1e-3)
        for step in range(3):
            y = m(wave, video, lengths)
            loss = batch_binary_cross_entropy(y, target, lengths, 1e-8)
            loss.backward()
            if step < 2:
                opt.step()
                opt.zero_grad()
        g = opt.flat_grad.detach().clone()        # (no synchronize: reading on the main stream must be safe by itself)
        return y.detach().clone(), g
    y0, g0 = run(0)
    assert torch.isfinite(g0).all() and float(g0.abs().sum()) > 0
    for i in range(8):
        y, g = run(5 + 7 * i)
        assert torch.equal(y, y0), "logits differ in run %d" % i
        assert torch.equal(g, g0), "gradients differ in run %d: max %.3e" % (i, float((g - g0).abs().max()))


# ------------------------------------------------------------------------------------------ bf16 arithmetic (BASELINE configs[4])
# relative L2 of a gradient tensor vs the fp32 oracle (measured on this 12-frame model: head 3.8-7.1e-2, trunk 1.3e-1 at the
# last block rising to 3.0e-1 at the stem, whose gradient has passed 20 bf16 convolutions).  The figures move by +-2e-2 with
# the fp32 summation order INSIDE the convolutions: an activation that lands on the other side of a bf16 rounding boundary
# changes by 2^-8 relative, and a 12-frame train-mode BatchNorm passes that on (the head's weight_ih_l1: 4.2e-2 with the engine's
# layer-1 kernels, 7.1e-2 with conv64::kernel16 -- the same exact products, another order).  The bounds are sanity bounds for a
# toy; the benched model's logits (3e-2 of max|ref|) and the exact-product tests carry the real claim.
BF16_GRAD_REL = {"head": 1e-1, "encoder": 1e-1, "trunk": 3.5e-1}


@pytest.mark.parametrize("N,H,W,C,Co,KS,stride,pad", [(3, 17, 17, 64, 64, 3, 1, 1), (2, 17, 17, 64, 128, 3, 2, 1), (5, 9, 9, 128, 128, 3, 1, 1),
                                                      (2, 17, 17, 64, 128, 1, 2, 0), (4, 3, 3, 512, 512, 3, 1, 1), (700, 9, 9, 128, 128, 3, 1, 1)])
def test_bf16_convolutions_vs_rounded_operands(N, H, W, C, Co, KS, stride, pad, lib_options):
    """Option "bf16": operands are rounded to bf16 (round to nearest even) on their way into LDS, multiplied by
    v_mfma_f32_32x32x16_bf16, accumulated in fp32.  Exact check of that arithmetic: F.conv2d in fp32 on operands that
    were rounded to bf16 BEFOREHAND has the very same products, so only the fp32 summation order differs (1e-5-level) --
    a wrong k-slot, lane or row in the bf16 LDS image would be an O(1) error.  Forward, data gradient, weight gradient."""
    import ctypes as Ct
    import torch.nn.functional as F
    from avvad import _lib as L, ops
    lib_options("bf16", 1)
    rng = np.random.RandomState(N * H + C + 1)
    r16 = lambda t: t.bfloat16().float()
    x = r16(T(rng.normal(size=(N, C, H, W)).astype(np.float32))).requires_grad_(True)
    w = r16(T((rng.normal(size=(Co, C, KS, KS)) / np.sqrt(C * KS * KS)).astype(np.float32))).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad)
    gy = r16(T(rng.normal(size=tuple(y.shape)).astype(np.float32)))
    y.backward(gy)
    lib = L.lib()
    st = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(N, H, W, C, Co, KS, stride, pad)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    wf = torch.empty(KS * KS * C * Co, device=DEV)
    wdg = torch.empty(KS * KS * C * Co, device=DEV)
    L.check(lib.avvad_conv2d_pack_weights(L.ptr(w.detach().to(DEV)), L.ptr(wf), L.ptr(wdg), Ct.byref(d), st), "pack")
    ews = ops.engine_ws(DEV)
    wsz = ews.numel() * 4
    yd = torch.empty(N, y.shape[2], y.shape[3], Co, device=DEV)
    L.check(lib.avvad_conv2d_fwd(L.ptr(xd), L.ptr(wf), L.ptr(yd), Ct.byref(d), L.ptr(ews), wsz, st), "fwd")
    tag = "bf16 conv %dx%dx%dx%d->%d k%d s%d" % (N, H, W, C, Co, KS, stride)
    _report(tag + " fwd", yd.permute(0, 3, 1, 2), y, 2e-5, 2e-5)
    gyd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dx = torch.empty_like(xd)
    L.check(lib.avvad_conv2d_dgrad(L.ptr(gyd), L.ptr(wdg), L.ptr(dx), Ct.byref(d), 0, L.ptr(ews), wsz, st), "dgrad")
    _report(tag + " dgrad", dx.permute(0, 3, 1, 2), x.grad, 2e-5, 2e-5)
    dw = torch.empty(KS * KS * C, Co, device=DEV)
    L.check(lib.avvad_conv2d_wgrad(L.ptr(xd), L.ptr(gyd), L.ptr(dw), Ct.byref(d), L.ptr(ews), wsz, st), "wgrad")
    _report(tag + " wgrad", dw, w.grad.permute(2, 3, 1, 0).reshape(KS * KS * C, Co), 1e-4 * np.sqrt(N), 1e-4)
    # and the distance to the UNROUNDED fp32 product is what bf16 operands cost: ~3e-3 relative
    y32 = F.conv2d(x.detach(), w.detach(), None, stride, pad)
    assert float((yd.permute(0, 3, 1, 2).cpu() - y32).norm() / y32.norm()) < 1e-5      # (operands already representable)


@pytest.mark.parametrize("N,H,W,C,Co,KS,stride,pad", [(3, 17, 17, 64, 64, 3, 1, 1), (2, 17, 17, 64, 128, 3, 2, 1), (5, 9, 9, 128, 128, 3, 1, 1),
                                                      (2, 17, 17, 64, 128, 1, 2, 0), (4, 3, 3, 512, 512, 3, 1, 1), (700, 9, 9, 128, 128, 3, 1, 1),
                                                      (300, 5, 5, 256, 256, 3, 1, 1), (96, 9, 9, 128, 256, 3, 2, 1), (40, 17, 17, 64, 64, 3, 1, 1),
                                                      (130, 3, 3, 512, 512, 3, 1, 1), (256, 9, 9, 128, 256, 3, 2, 1), (200, 5, 5, 256, 512, 3, 2, 1),
                                                      (333, 9, 9, 128, 256, 3, 1, 1), (1024, 17, 17, 64, 64, 3, 1, 1)])
def test_bf16_data_path_convolutions(N, H, W, C, Co, KS, stride, pad):
    """The bf16 DATA PATH's convolutions (csrc/bgemm.h; what the trunk runs under option bf16 = 1): operands are bf16 in
    memory -- NHWC bf16 activations / output gradients, K-contiguous bf16 weight packs -- staged with 16-byte loads, the
    weight gradient's operands transposed by ds_read_b64_tr_b16, fp32 accumulation and fp32 results.  Exact check as above:
    F.conv2d in fp32 on the bf16 values has the very same products; only the fp32 summation order differs.  A wrong lane /
    k-slot / chunk in the LDS images, the transposed reads or the weight packs is an O(1) error.  Covers whole-tile rounds,
    stream-K rounds with the ordered fix-up (N = 700, 300), the four stride-2 parity classes, the 1x1 downsample, and -- from
    128 images up, where the tile count fits the stream-K pool -- the position-class schedule that skips the zero padding.
    The 64 -> 64 channel shapes run the bf16 form of the weights-stationary kernel (conv64::kernel16), at 1024 frames with four
    to five tiles per wave (the cross-tile prefetch)."""
    import ctypes as Ct
    import torch.nn.functional as F
    from avvad import _lib as L, ops
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    rng = np.random.RandomState(N * H + C + 7)
    r16 = lambda t: t.bfloat16().float()
    x = r16(T(rng.normal(size=(N, C, H, W)).astype(np.float32))).requires_grad_(True)
    w = r16(T((rng.normal(size=(Co, C, KS, KS)) / np.sqrt(C * KS * KS)).astype(np.float32))).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad)
    gy = r16(T(rng.normal(size=tuple(y.shape)).astype(np.float32)))
    y.backward(gy)
    lib = L.lib()
    st = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(N, H, W, C, Co, KS, stride, pad)
    x16 = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).bfloat16()
    wf16 = torch.empty(KS * KS * C * Co, device=DEV, dtype=torch.bfloat16)
    wd16 = torch.empty(KS * KS * C * Co, device=DEV, dtype=torch.bfloat16)
    L.check(lib.avvad_conv2d_pack_weights_bf16(L.ptr(w.detach().to(DEV)), L.ptr(wf16), L.ptr(wd16), Ct.byref(d), st), "pack bf16")
    ews = ops.engine_ws(DEV)
    wsz = ews.numel() * 4
    tag = "bf16 path conv %dx%dx%dx%d->%d k%d s%d" % (N, H, W, C, Co, KS, stride)
    yd = torch.full((N, y.shape[2], y.shape[3], Co), float("nan"), device=DEV)
    L.check(lib.avvad_conv2d_fwd_bf16(L.ptr(x16), L.ptr(wf16), L.ptr(yd), Ct.byref(d), L.ptr(ews), wsz, st), "fwd bf16")
    _report(tag + " fwd", yd.permute(0, 3, 1, 2), y, 2e-5, 2e-5)
    yd2 = torch.full_like(yd, float("nan"))
    L.check(lib.avvad_conv2d_fwd_bf16(L.ptr(x16), L.ptr(wf16), L.ptr(yd2), Ct.byref(d), None, 0, st), "fwd bf16, whole tiles")
    _report(tag + " fwd (no scratch: whole-tile schedule)", yd2.permute(0, 3, 1, 2), y, 2e-5, 2e-5)
    gy16 = gy.permute(0, 2, 3, 1).contiguous().to(DEV).bfloat16()
    dx = torch.full((N, H, W, C), float("nan"), device=DEV)
    L.check(lib.avvad_conv2d_dgrad_bf16(L.ptr(gy16), L.ptr(wd16), L.ptr(dx), Ct.byref(d), 0, L.ptr(ews), wsz, st), "dgrad bf16")
    _report(tag + " dgrad", dx.permute(0, 3, 1, 2), x.grad, 2e-5, 2e-5)
    L.check(lib.avvad_conv2d_dgrad_bf16(L.ptr(gy16), L.ptr(wd16), L.ptr(dx), Ct.byref(d), 1, L.ptr(ews), wsz, st), "dgrad bf16 +=")
    _report(tag + " dgrad accumulate", dx.permute(0, 3, 1, 2), 2 * x.grad, 4e-5, 4e-5)
    dw = torch.full((KS * KS * C, Co), float("nan"), device=DEV)
    L.check(lib.avvad_conv2d_wgrad_bf16(L.ptr(x16), L.ptr(gy16), L.ptr(dw), Ct.byref(d), L.ptr(ews), wsz, st), "wgrad bf16")
    wref = w.grad.permute(2, 3, 1, 0).reshape(KS * KS * C, Co)
    # (fp32 sums of N*Ho*Wo exact products in two different orders: 1e-4 * sqrt(N), and never less than 4e-6 of the largest sum --
    #  at 1024 frames an element is a sum of 296 k products of magnitude up to 2e3)
    _report(tag + " wgrad", dw, wref, max(1e-4 * np.sqrt(N), 4e-6 * float(wref.abs().max())), 1e-4)
    dw2 = torch.empty_like(dw)
    L.check(lib.avvad_conv2d_wgrad_bf16(L.ptr(x16), L.ptr(gy16), L.ptr(dw2), Ct.byref(d), L.ptr(ews), wsz, st), "wgrad bf16")
    assert torch.equal(dw, dw2)                               # ordered fix-up: run to run the same bits
    bad = L.ConvDesc(N, H, W, 32, Co, KS, stride, pad)          # 32 channels: not a whole 64-channel chunk
    assert lib.avvad_conv2d_fwd_bf16(L.ptr(x16), L.ptr(wf16), L.ptr(yd), Ct.byref(bad), None, 0, st) == -1


def test_bf16_gemm_variants(lib_options):
    from avvad import ops
    lib_options("bf16", 1)
    rng = np.random.RandomState(5)
    for (M, N, K, tA, tB) in [(128, 128, 64, 0, 1), (100, 70, 513, 0, 1), (64, 4096, 1024, 0, 1), (37, 130, 96, 0, 0), (130, 64, 40, 1, 0),
                              (4096, 513, 48, 1, 0), (1025, 300, 7, 1, 1), (1024, 768, 4096, 0, 0)]:
        A = T(rng.normal(size=(K, M) if tA else (M, K)).astype(np.float32)).bfloat16().float()
        B = T(rng.normal(size=(N, K) if tB else (K, N)).astype(np.float32)).bfloat16().float()
        ref = (A.t() if tA else A).double() @ (B.t() if tB else B).double()
        c = torch.empty(M, N, device=DEV)
        ops.gemm(A.to(DEV), B.to(DEV), c, M, N, K, A.shape[1], B.shape[1], N, bool(tA), bool(tB))
        _report("bf16 gemm %dx%dx%d tA%d tB%d" % (M, N, K, tA, tB), c, ref, 1e-5 * np.sqrt(K) * 4)
    # unrounded operands: the bf16 rounding itself, relative L2 ~ 2^-9 * sqrt(2)
    A = T(rng.normal(size=(512, 768)).astype(np.float32)); B = T(rng.normal(size=(768, 640)).astype(np.float32))
    c = torch.empty(512, 640, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), c, 512, 640, 768, 768, 640, 640)
    rel = float((c.cpu().double() - A.double() @ B.double()).norm() / (A.double() @ B.double()).norm())
    print("bf16 gemm on fp32 operands: relL2 %.2e" % rel)
    assert 5e-4 < rel < 6e-3


def test_bf16_av_training_step_vs_fp32_oracle(lib_options):
    """BASELINE configs[4] arithmetic end to end on the AV model (WaveNet encoder + trunk + LSTM head): bf16 operands in
    every convolution / dense GEMM, fp32 BatchNorm statistics, LSTM cell, loss.  SURVEY 7's tolerance for this mode: about
    2e-2 relative on the logits vs the fp32 oracle; every big gradient tensor within a relative-L2 bound of the oracle's
    (BF16_GRAD_REL: by where the tensor sits -- the stem's gradient has passed 20 bf16 convolutions and train-mode BatchNorm
    over only 12 frames)."""
    from oracle import head, models
    from packages.models.AV_Net import DeepVAD_AV
    from packages.models.utils import batch_binary_cross_entropy
    wcfg = dict(filter_width=2, quantization_channel=1, dilations=[1, 2, 4, 8, 16, 32], en_residual_channel=32,
                en_dilation_channel=32, en_bottleneck_width=64, en_pool_kernel_size=4, use_bias=True)
    torch.manual_seed(5)
    m = DeepVAD_AV(2, 32, 1, wavenet_params=wcfg)
    B, Tn = 3, 4
    wave = torch.randn(B, 1, Tn * 256 + 63) * 0.3
    video = torch.randn(B, Tn, 67, 67)
    tgt = (torch.rand(B, Tn, 1) > 0.5).float()
    lens = [4, 2, 3]
    sd = {k: (v.detach().clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.detach().clone())
          for k, v in m.state_dict().items()}
    ref = models.av_net(sd, wave, video, lens, 2, training=True, wavenet_cfg=wcfg)
    ref_loss = head.batch_loss(ref, tgt, lens, 1e-8)
    ref_loss.backward()
    lib_options("bf16", 1)
    m = m.to(DEV).train()
    y = m(wave.to(DEV), video.to(DEV), torch.LongTensor(lens))
    scale = float(ref.abs().max())
    _report("bf16 AV logits vs fp32 oracle", y, ref, 2e-2 * scale)
    loss = batch_binary_cross_entropy(y, tgt.to(DEV), lens, 1e-8)
    _report("bf16 AV loss", loss, ref_loss, 2e-2 * float(ref_loss))
    loss.backward()
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        g, r = p.grad.cpu().flatten().double(), sd[k].grad.flatten().double()
        assert torch.isfinite(g).all(), k
        if r.numel() >= 4096:
            rel = float((g - r).norm() / r.norm().clamp_min(1e-30))
            # per-tensor relative-L2 bounds: the head sees one bf16 product per operand; a trunk gradient has passed up to 20
            # bf16 convolutions and train-mode BatchNorm over 12 frames (bound by depth: the stem's is the loosest)
            bound = BF16_GRAD_REL["head"] if k.startswith(("lstm", "vad")) else (BF16_GRAD_REL["encoder"] if k.startswith("wavenet") else BF16_GRAD_REL["trunk"])
            msg = "bf16 grad relL2 %-40s %.3e (bound %.1e)" % (k, rel, bound)
            print(msg)
            with open(os.path.join(OUT, "parity.log"), "a") as f:
                f.write(msg + "\n")
            assert rel < bound, msg


# ------------------------------------------------------------------------------------------ boundary: bare modules, losses
def test_count_sketch_and_compact_bilinear_pooling_modules():
    """``CountSketch.forward`` / ``CompactBilinearPooling.forward`` as stand-alone modules (compact_bilinear_pooling.py
    :59-114,222-263): the sketch against the reference's own output (pinned fixture ``cs_y`` / ``cs_dx``), the pooled
    vector + both input gradients against the oracle's FFT form (itself pinned to the naive outer-product definition)."""
    from oracle import fusion
    from packages.models.compact_bilinear_pooling import CompactBilinearPooling, CountSketch
    g = load_golden("misc")
    cs = CountSketch(513, 1024, T(g["cs_h"]), T(g["cs_s"])).to(DEV)
    x = T(g["cs_x"]).to(DEV).requires_grad_(True)
    y = cs(x)
    _report("CountSketch.forward vs reference", y, g["cs_y"], 1e-6)
    (y * T(g["cs_g"]).to(DEV)).sum().backward()
    _report("CountSketch backward vs reference", x.grad, g["cs_dx"], 1e-6)
    rng = np.random.RandomState(5)
    h1, h2 = T(rng.randint(0, 1024, 513)), T(rng.randint(0, 1024, 512))
    s1 = T((2 * rng.randint(0, 2, 513) - 1).astype(np.float32))
    s2 = T((2 * rng.randint(0, 2, 512) - 1).astype(np.float32))
    a, v, G = stategen.rand(61, 2, 3, 513), stategen.rand(62, 2, 3, 512), stategen.rand(63, 2, 3, 1024)
    ar, vr = a.clone().requires_grad_(True), v.clone().requires_grad_(True)
    ref = fusion.mcb(ar, vr, h1, s1, h2, s2, 1024)
    (ref * G).sum().backward()
    m = CompactBilinearPooling(513, 512, 1024, h1, s1, h2, s2).to(DEV)
    ag, vg = a.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    out = m(ag, vg)
    _report("CompactBilinearPooling.forward", out, ref, 2e-4, 1e-5)
    (out * G.to(DEV)).sum().backward()
    _report_grad("CompactBilinearPooling d/dx", ag.grad, ar.grad)
    _report_grad("CompactBilinearPooling d/dy", vg.grad, vr.grad)
    sq = CompactBilinearPooling(513, 513, 1024, h1, s1, T(rng.randint(0, 1024, 513)), s1.clone()).to(DEV)
    _report("CompactBilinearPooling(x) == (x, x)", sq(ag.detach()), sq(ag.detach(), ag.detach()), 1e-5, 1e-6)   # LDS-atomic bucket order


def test_bce_2classes_vs_reference():
    from packages.models.utils import binary_cross_entropy_2classes
    g = load_golden("misc")
    r1 = T(g["bce2_r1"]).to(DEV).requires_grad_(True)
    r2 = T(g["bce2_r2"]).to(DEV).requires_grad_(True)
    loss = binary_cross_entropy_2classes(r1, r2, T(g["bce2_x"]).to(DEV), 1e-8)
    _report("bce_2classes", loss, g["bce2"], 1e-6)
    (loss * 3.0).backward()
    _report("bce_2classes d/dr1", r1.grad, 3.0 * g["bce2_d1"], 1e-6, 1e-5)
    _report("bce_2classes d/dr2", r2.grad, 3.0 * g["bce2_d2"], 1e-6, 1e-5)


@pytest.mark.parametrize("frozen", [("dil_w",), ("dense_w",), ("dil_w", "dense_w"), ("dil_b", "dense_b"), ("bott_w",)])
def test_wavenet_partially_frozen_blocks(frozen):
    """Gradients of a residual block are wanted independently: freezing one weight (or both weights but not the biases)
    must leave the others exact (the fused MFMA backward once skipped them all and still returned OK)."""
    from oracle import wavenet as ow
    from packages.models.wavenet_autoencoder import wavenet_autoencoder
    cfg = dict(filter_width=2, quantization_channel=1, dilations=[1, 2, 4], en_residual_channel=32,
               en_dilation_channel=32, en_bottleneck_width=256, en_pool_kernel_size=5, use_bias=True)
    torch.manual_seed(4)
    m = wavenet_autoencoder(**cfg)
    x = torch.randn(3, 1, 200)
    G = torch.randn(3, 256, 5)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    (ow.encode(sd, x, cfg) * G).sum().backward()
    pick = {"dil_w": "en_dilation_layer_stack.1.weight", "dense_w": "en_dense_layer_stack.1.weight",
            "dil_b": "en_dilation_layer_stack.1.bias", "dense_b": "en_dense_layer_stack.1.bias", "bott_w": "bottleneck_layer.weight"}
    frozen_keys = {pick[f] for f in frozen}
    m = m.to(DEV)
    for k, p in m.named_parameters():
        p.requires_grad = k not in frozen_keys
    (m(x.to(DEV)) * G.to(DEV)).sum().backward()
    for k, p in m.named_parameters():
        if k in frozen_keys:
            assert p.grad is None, k
        else:
            _report_grad("frozen %s: d/d%s" % ("+".join(frozen), k), p.grad, sd[k].grad)


def test_stem_maxpool_ties_follow_torch():
    """Flat image regions give identical stem outputs, i.e. exact positive ties inside a 3x3 pooling window: torch
    routes the gradient to the FIRST maximum in scan order only (its saved argmax); counting every tied element once
    per window inflated the stem gradients."""
    from oracle import resnet18
    from avvad import nn as avnn
    from packages.models.Video_Net import DeepVAD_video
    sd0 = _video_state()
    N = 4
    x = stategen.rand(31, N, 67, 67)
    x[0] = 0.7                                  # constant frame: every interior stem output of a channel is identical
    x[1, :40] = -0.3                            # constant band (black border)
    x[2, :, 20:] = 1.5
    G = stategen.rand(32, N, 512)
    sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone())
          for k, v in sd0.items() if k.startswith("features.")}
    ref, inter = resnet18.trunk_forward(sd, x[:, None].repeat(1, 3, 1, 1), False, return_intermediates=True)
    (ref * G).sum().backward()
    m = DeepVAD_video(2, 16, 1)
    m.load_state_dict(sd0)
    m = m.to(DEV).eval()
    f = avnn.trunk_forward(m.features, x.to(DEV), False)
    _report("trunk fwd with flat regions", f, ref, 1e-4, 1e-5)
    flips, _ = _relu_flips(f, inter)
    (f * G.to(DEV)).sum().backward()
    for k in ("0.weight", "1.weight", "1.bias"):
        _report_grad("tie-breaking: trunk d/d%s" % k, dict(m.features.named_parameters())[k].grad, sd["features." + k].grad, 2.0,
                     _grad_bound(flips, 2e-3))


# ------------------------------------------------------------------------------------------ C1: evaluate_audio_net plumbing, K19
def test_audio_evaluator_plumbing_on_a_real_utterance(tmp_path):
    """BASELINE configs[0] plumbing (scripts/evaluate_audio_net.py:107-180) on one utterance of the reference's
    data/subset: x/max|x| -> STFT -> power -> log -> crop -> standardise -> classifier -> sigmoid -> threshold, every
    arithmetic step on the GPU, against (a) the oracle's features and (b) the logits / soft / hard outputs the
    REFERENCE's own DeepVAD_audio produced from a checkpoint it wrote (tests/golden/audio_ref_h32_*.pt), for the VAD
    head and the IBM head (y_dim 513).  The STFT itself is pinned only to the oracle restatement (torch 2.x cannot run
    stft_pytorch): "parity unpinned" for that stage."""
    from conftest import GOLDEN
    from avvad import train as TR
    from oracle import frontend
    from packages.models.Audio_Net import DeepVAD_audio
    g = load_golden("eval_audio")
    wav = os.path.join(GOLDEN, "utt_sa1.npz")
    x_t, fs = TR.load_waveform(wav)
    stats = TR.Stats(audio_mean=g["mean"], audio_std=g["std"])
    n_label = int(g["n_label"])
    feats = TR.audio_features(x_t.to(DEV), stats, n_label)
    ref_feats = frontend.audio_features(x_t, T(g["mean"]), T(g["std"]), n_label)
    _report("evaluator features (1,180,513)", feats, ref_feats, 2e-3, 1e-4)     # log of a 1024-term fp32 DFT power
    for tag, ydim in (("y1", 1), ("y513", 513)):
        ck = os.path.join(GOLDEN, "audio_ref_h32_%s.pt" % tag)
        out = tmp_path / tag
        lab = torch.zeros(ydim, n_label)
        TR.evaluate_main("audio", lambda: DeepVAD_audio(2, 32, ydim), checkpoint=ck, out_dir=str(out), wav_list=[wav],
                         stats=stats, labels={wav: lab})
        soft = torch.load(out / "utt_sa1_y_hat_soft.pt", weights_only=True)
        hard = torch.load(out / "utt_sa1_y_hat_hard.pt", weights_only=True)
        assert soft.shape == (1, n_label) and hard.dtype == torch.int32
        _report("evaluator soft output (%s)" % tag, soft, g["soft_" + tag], 1e-4)
        flips = int((hard.numpy() != g["hard_" + tag]).sum())
        margin = np.abs(g["soft_" + tag] - 0.5)
        assert flips == 0 or float(margin[hard.numpy() != g["hard_" + tag]].max()) < 1e-4, flips
        m = DeepVAD_audio(2, 32, ydim)
        m.load_state_dict(torch.load(ck, map_location="cpu", weights_only=True))
        y = m.to(DEV).eval()(feats, [n_label])
        _report("DeepVAD_audio(y_dim=%d) logits vs reference" % ydim, y, g["logits_" + tag], 1e-4)


def test_input_standardisation_in_the_train_loop():
    """K19: ``(x - mean.T) / (std + eps).T`` on audio features (513 per-bin statistics) and video (scalar statistics)
    as applied by ``forward_batch`` (scripts/train_AV_net.py:286-291) vs the oracle."""
    from avvad import train as TR
    from oracle import frontend
    a, v = stategen.rand(71, 3, 5, 513), stategen.rand(72, 3, 5, 67, 67)
    am, as_ = stategen.rand(73, 513, 1), stategen.rand(74, 513, 1).abs() + 0.5
    vm, vs = torch.tensor([[0.4]]), torch.tensor([[2.5]])
    st = TR.Stats(am.numpy(), as_.numpy(), vm.numpy(), vs.numpy())
    _report("standardise audio", st.audio(a.to(DEV)), frontend.standardize(a, am, as_), 1e-6, 1e-6)
    _report("standardise video", st.video(v.to(DEV)), (v - vm.T) / (vs + 1e-8).T, 1e-6, 1e-6)
    seen = {}

    class Probe(torch.nn.Module):
        def forward(self, x, vid, lengths):
            seen["a"], seen["v"] = x, vid
            return x[..., :1]
    batch = (torch.LongTensor([5, 5, 5]), a, v, torch.zeros(3, 5, 1))
    TR.forward_batch(Probe(), "av", batch, torch.device(DEV), False, st)
    _report("forward_batch standardises audio", seen["a"], frontend.standardize(a, am, as_), 1e-6, 1e-6)
    _report("forward_batch standardises video", seen["v"], (v - vm.T) / (vs + 1e-8).T, 1e-6, 1e-6)


# ------------------------------------------------------------------------------------------ BASELINE-size properties
@pytest.mark.gpu
@pytest.mark.parametrize("B,Lin,dil,grid", [(3, 700, 1, 0), (2, 1200, 64, 8), (5, 2300, 512, 8), (4, 1027, 16, 16), (16, 16000, 64, 0),
                                             (256, 16000, 512, 0), (256, 15999, 1, 0), (64, 14977, 256, 0)])
def test_wavenet_block_kernel_forms_agree(B, Lin, dil, grid, lib_options):
    """One residual block through avvad_wavenet_block_fwd in its three forms (option wn_flat: 1 flat dword addressing,
    2 buffer dword, 3 wide dwordx4, 4 high occupancy, 5 LDS-DMA) against the plain torch fp32 statement of the block (wavenet_autoencoder.py:78-86);
    a small workgroup cap (wn_grid) makes every wave walk several tiles, tails included."""
    import ctypes as C
    import torch.nn.functional as F
    from avvad import _lib as L_
    lib = L_.lib()
    torch.manual_seed(B * 1000 + dil)
    s_in = torch.randn(B, 32, Lin, device=DEV)
    wd, bd = torch.randn(32, 32, 2, device=DEV) * 0.2, torch.randn(32, device=DEV) * 0.1
    we, be = torch.randn(32, 32, 1, device=DEV) * 0.2, torch.randn(32, device=DEV) * 0.1
    big = B * Lin > 1 << 20          # BASELINE configs[1] planes: the flat-addressed kernel (form 1) is the reference
    if not big:
        ref = F.conv1d(F.relu(F.conv1d(F.relu(s_in.double()), wd.double(), bd.double(), dilation=dil)), we.double(), be.double())
        ref = (ref + s_in.double()[:, :, dil:]).float()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = {}
    for form in (1, 2, 3, 4, 5):
        lib_options("wn_flat", form)
        lib_options("wn_grid", grid)
        out = torch.full((B, 32, Lin - dil), float("nan"), device=DEV)
        L_.check(lib.avvad_wavenet_block_fwd(L_.ptr(s_in), L_.ptr(wd), L_.ptr(bd), L_.ptr(we), L_.ptr(be), L_.ptr(out), B, Lin,
                                             dil, st), "avvad_wavenet_block_fwd")
        torch.cuda.synchronize()
        assert torch.isfinite(out).all(), "form %d left samples unwritten" % form
        if big and form == 1:
            ref = out
        assert (out - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), "form %d" % form
        outs[form] = out
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[1], outs[4])     # same arithmetic, different addressing
    assert torch.equal(outs[1], outs[5])                                       # ... and the LDS-DMA form (operands through LDS)


def _max_rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_full_size_c2_wavenet_batch_split():
    """BASELINE configs[1] shape: 256 one-second chunks (256,1,16000) through the W0 encoder.  No oracle run at this
    size (94 MB of activations per chunk on the CPU): the size-independent property is that sequences are
    independent -- the batch of 256 equals 4 runs of 64 (catches 32-bit index overflow, tile tails, grid caps)."""
    from packages.models.wavenet_autoencoder import wavenet_autoencoder
    cfg = dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2, en_residual_channel=32,
               en_dilation_channel=32, en_bottleneck_width=256, en_pool_kernel_size=60, use_bias=True)
    torch.manual_seed(0)
    m = wavenet_autoencoder(**cfg).to(DEV)
    x = torch.rand(256, 1, 16000, device=DEV) * 2 - 1
    with torch.no_grad():
        full = m(x)
        parts = torch.cat([m(x[i:i + 64]) for i in range(0, 256, 64)])
    assert full.shape == (256, 256, 60) and torch.isfinite(full).all()
    assert _max_rel(full, parts) == 0.0          # the forward encoder is deterministic and batch-independent


def test_full_size_c2_wavenet_training_backward():
    """BASELINE configs[1]: audio_net TRAINING at batch 256 of one-second chunks -- forward AND backward at full size
    (21 activation planes of 524 MB each in the workspace).  Checks: (a) sequence 0 is the reference's own ``wn_w0``
    sample, so its output row must equal the fixture; (b) shard-sum: the gradient of every parameter over the batch of
    256 equals the sum over four shards of 64 (sequences are independent, the loss is a plain sum); (c) the input
    gradient of sequence 0 equals the fixture's ``dx``."""
    from packages.models.wavenet_autoencoder import wavenet_autoencoder
    g = load_golden("wn_w0")
    cfg = wn_cfg_from(g)
    m = wavenet_autoencoder(**cfg)
    m.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("p.")})
    m = m.to(DEV)
    B = 256
    torch.manual_seed(0)
    x = torch.rand(B, 1, 16000, device=DEV) - 0.5
    x[0] = T(g["x"]).to(DEV)[0]
    G = torch.randn(B, 256, 60, device=DEV) * 0.1
    G[0] = T(g["G"]).to(DEV)[0]
    params = list(m.parameters())

    def run(sl):
        for p in params:
            p.grad = None
        xi = x[sl].clone().requires_grad_(True)
        y = m(xi)
        (y * G[sl]).sum().backward()
        torch.cuda.synchronize()
        return y.detach(), xi.grad.detach(), [p.grad.clone() for p in params]

    y, dx, gf = run(slice(0, B))
    assert y.shape == (B, 256, 60) and torch.isfinite(y).all() and torch.isfinite(dx).all()
    _report("C2 B=256: sequence 0 forward == wn_w0 fixture", y[:1], g["y"], 1e-4)
    _report_grad("C2 B=256: sequence 0 d/dx == wn_w0 fixture", dx[:1], g["dx"])
    acc = [torch.zeros_like(t, dtype=torch.float64) for t in gf]
    for i in range(0, B, 64):
        yi, dxi, gi = run(slice(i, i + 64))
        assert _max_rel(yi, y[i:i + 64]) == 0.0 and _max_rel(dxi, dx[i:i + 64]) < 1e-6
        for a_, t in zip(acc, gi):
            a_ += t.double()
    for (k, _), f, a_ in zip(m.named_parameters(), gf, acc):
        rel = float((f.double() - a_).norm() / a_.norm().clamp_min(1e-30))
        print("C2 shard-sum d/d%-40s relL2 %.2e" % (k, rel))
        assert rel < 1e-4, (k, rel)             # only the fp32 summation order over (sequence, time) differs


def test_full_size_c3_trunk_train_mode_backward(lib_options):
    """BASELINE configs[2]: 512 lip crops through the trunk in TRAINING mode (batch statistics), forward + backward at
    full size.  The oracle can afford this size once (about 0.7 TFLOP on the host cores): features, updated running
    statistics and a spread of weight / BN gradients are compared on the full batch under the production schedule; then
    a 6-frame slice of the same frames is re-run alone (eval-mode BatchNorm, whole-tile schedule: bit-reproducible)
    and every gradient is held to the strict bounds."""
    from oracle import resnet18
    from avvad import nn as avnn
    from packages.models.Video_Net import DeepVAD_video
    sd0 = _video_state()
    m = DeepVAD_video(2, 16, 1)
    m.load_state_dict(sd0)
    m = m.to(DEV).train()
    N = 512
    x = stategen.rand(41, N, 67, 67)
    G = stategen.rand(42, N, 512)
    f = avnn.trunk_forward(m.features, x.to(DEV), True)
    assert f.shape == (N, 512) and torch.isfinite(f).all()
    sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone())
          for k, v in sd0.items() if k.startswith("features.")}
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ref, inter = resnet18.trunk_forward(sd, x[:, None].repeat(1, 3, 1, 1), True, return_intermediates=True)
    (ref * G).sum().backward()
    _report("C3 train-mode trunk fwd N=512", f, ref, 1e-4, 1e-5)
    flips, _ = _relu_flips(f, inter)              # (before backward: it frees the saved activations)
    (f * G.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    for k, p in m.features.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    new = m.state_dict()
    for k in ("features.1.running_mean", "features.1.running_var", "features.7.1.bn2.running_mean", "features.7.1.bn2.running_var"):
        _report("C3 running stat " + k, new[k], sd[k], 1e-5, 1e-5)
    for k in ("0.weight", "1.weight", "4.0.conv1.weight", "5.0.downsample.0.weight", "6.1.bn2.bias", "7.1.conv2.weight"):
        _report_grad("C3 N=512 trunk d/d%s" % k, dict(m.features.named_parameters())[k].grad, sd["features." + k].grad, 2.0,
                     _grad_bound(flips, 2e-3))
    # (c) 6-frame slice alone, eval-mode BN, whole-tile schedule, strict bounds
    lib_options("no_streamk", 1)
    m2 = DeepVAD_video(2, 16, 1)
    m2.load_state_dict(sd0)
    m2 = m2.to(DEV).eval()
    sl = x[100:106]
    sd2 = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone())
           for k, v in sd0.items() if k.startswith("features.")}
    r2, inter2 = resnet18.trunk_forward(sd2, sl[:, None].repeat(1, 3, 1, 1), False, return_intermediates=True)
    (r2 * G[100:106]).sum().backward()
    f2 = avnn.trunk_forward(m2.features, sl.to(DEV), False)
    _report("C3 6-frame slice eval fwd", f2, r2, 1e-4, 1e-5)
    flips2, _ = _relu_flips(f2, inter2)
    (f2 * G[100:106].to(DEV)).sum().backward()
    for k, p in m2.features.named_parameters():
        _report_grad("C3 slice d/d%s" % k, p.grad, sd2["features." + k].grad, 2.0, _grad_bound(flips2, 2e-3))


def test_full_size_c3_trunk_batch_split_and_scale(lib_options):
    """BASELINE configs[2] shape: 512 lip crops.  Eval-mode trunk: frames are independent, so 512 == 2 x 256.  Every
    schedule is deterministic (split tiles are combined in a fixed order): under the whole-tile schedule a frame's sums do not
    depend on the batch it sits in and the two runs agree BIT FOR BIT; under the production stream-K schedule the split
    points move with the batch size, i.e. only the fp32 summation order inside a tile differs (1e-6 of the largest value).
    Positive homogeneity of the ReLU network in eval mode with the BN shifts zero: f(a*x) == a*f(x), a a power of two: exact."""
    from packages.models.Video_Net import DeepVAD_video
    from avvad import nn as avnn
    torch.manual_seed(0)
    m = DeepVAD_video(1, 8, 1).to(DEV).eval()
    x = torch.randn(512, 67, 67, device=DEV)
    with torch.no_grad():
        full = avnn.trunk_forward(m.features, x, False)
        parts = torch.cat([avnn.trunk_forward(m.features, x[:256], False), avnn.trunk_forward(m.features, x[256:], False)])
        assert full.shape == (512, 512) and torch.isfinite(full).all()
        print("C3 eval 512 vs 2x256, stream-K: max rel %.2e" % _max_rel(full, parts))
        assert _max_rel(full, parts) < 1e-6
        # running_mean = 0, beta = 0 (fresh BatchNorm) -> the eval network is conv/scale/ReLU/max/mean only
        half = avnn.trunk_forward(m.features, 0.5 * x, False)
        assert torch.equal(half, 0.5 * full)
        lib_options("no_streamk", 1)
        full_w = avnn.trunk_forward(m.features, x, False)
        parts_w = torch.cat([avnn.trunk_forward(m.features, x[:256], False), avnn.trunk_forward(m.features, x[256:], False)])
        assert torch.equal(full_w, parts_w)


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


@pytest.mark.parametrize("dtype,whole_tiles", [("f32", False), ("f32", True), ("bf16", True)])
def test_full_size_c4_dp_shard_gradient_sum(dtype, whole_tiles, lib_options):
    """BASELINE configs[3]/[4] per-GPU shard (64 sequences x 16 frames, W0 encoder, 2xLSTM1024): the data-parallel
    property on ONE GPU.  With BatchNorm in eval mode the summed-over-sequences loss makes
    grad(full batch) == grad(shard 0) + grad(shard 1) -- exactly what the SUM all-reduce relies on.  Every schedule is
    deterministic; what differs between the three runs is the fp32 summation order over samples (split points move with
    the batch size) and, rarely, a ReLU unit whose pre-activation sits within rounding of zero.  f32: relative L2 <= 2e-4 per
    tensor.  bf16 (BASELINE configs[4]'s arithmetic at its real shape): the same property holds -- the rounding of a
    sample's activations does not depend on its batch -- checked under the whole-tile schedule (see below), bound 2e-3."""
    from packages.models.AV_Net import DeepVAD_AV
    from packages.models.utils import batch_binary_cross_entropy
    bench = _load_bench()
    # f32, production schedule: the stream-K split points (and, from 128 frames up, the position-class tiling) move with the
    # batch size, so a sample's fp32 sums differ in ORDER between the full batch and its shards; a ReLU unit whose
    # pre-activation sits within rounding of zero then lands on the other side and moves every gradient upstream of it by
    # ~1e-3 (measured worst tensor 4e-4 .. 1.3e-3, always the stem / early trunk weights): bound 5e-3, a wrong shard sum is O(1).
    # Whole-tile schedule: a sample's sums do not depend on its batch at all -- only the final sums over samples differ in order.
    if dtype == "bf16":
        lib_options("bf16", 1)
        # Rounding to bf16 is discontinuous: a 1e-7 difference in an fp32 sum (stream-K split points move with the batch size)
        # flips the rounding of a few activations by a whole bf16 step, those flip more downstream, and after 20 layers
        # two runs differ by the bf16 noise level itself (measured, tools/lab/shard_probe.py: trunk features 3.5e-3, deep
        # gradients 5-9 %).  Under the whole-tile schedule a sample's sums do not depend on the batch it sits in, the
        # bf16 roundings agree bit for bit and the shard-sum property is exact up to the fp32 order of the final sums.
    if whole_tiles:
        lib_options("no_streamk", 1)
    torch.manual_seed(0)
    m = DeepVAD_AV(2, 1024, 1, wavenet_params=bench.W0).to(DEV).eval()
    wave, video, target, lengths = bench.make_inputs(torch, 64, 1234, torch.device(DEV))
    lengths = lengths.clone()
    lengths[::3] = 11                                   # ragged
    params = [p for n, p in m.named_parameters() if not n.startswith("bn.")]

    def grads(sl):
        for p in params:
            p.grad = None
        y = m(wave[sl], video[sl], lengths[sl])
        loss = batch_binary_cross_entropy(y, target[sl], lengths[sl], 1e-8)
        loss.backward()
        return float(loss.detach()), [p.grad.clone() for p in params]

    lf, gf = grads(slice(0, 64))
    l0, g0 = grads(slice(0, 32))
    l1, g1 = grads(slice(32, 64))
    lf2, gf2 = grads(slice(0, 64))
    assert lf == lf2 and all(torch.equal(a, b) for a, b in zip(gf, gf2))      # run to run: the same bits
    assert abs(lf - (l0 + l1)) < (1e-5 if dtype == "f32" else 1e-3) * abs(lf)
    rels = sorted(((float(((a + b) - f).norm() / f.norm().clamp_min(1e-30)), n) for (n, _), f, a, b in
                   zip([q for q in m.named_parameters() if not q[0].startswith("bn.")], gf, g0, g1)), reverse=True)
    print("worst tensors:", rels[:6])
    worst = rels[0][0]
    print("DP shard-sum property (%s, %s schedule): loss %.4f = %.4f + %.4f, worst relL2 over %d tensors %.2e"
          % (dtype, "whole-tile" if whole_tiles else "production", lf, l0, l1, len(gf), worst))
    assert worst < (2e-3 if dtype == "bf16" else (1e-4 if whole_tiles else 5e-3))


def test_bf16_benched_model_ragged_logits_vs_fp32_oracle(lib_options):
    """BASELINE configs[4] at the BENCHED model (W0 encoder, ResNet-18, 2 x LSTM(1024), FC; sequences of the benchmark's
    shape): logits of a ragged 4-sequence slice in train mode under option bf16 against the fp32 CPU oracle, within
    3e-2 * max|ref| (see below); the same slice in fp32 within 1e-4 (the north-star bound) for contrast.
    This is bench.py's parity probe (`cpu_ref_max_abs_delta`) as a test."""
    from oracle import models
    from packages.models.AV_Net import DeepVAD_AV
    bench = _load_bench()
    torch.manual_seed(0)
    m = DeepVAD_AV(2, 1024, 1, wavenet_params=bench.W0)
    wave, video, target, lengths = bench.make_inputs(torch, 4, 4321, None)
    lens = [16, 9, 12, 5]
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ref = models.av_net(sd, wave, video, lens, 2, training=True, wavenet_cfg=bench.W0)
    m = m.to(DEV).train()
    with torch.no_grad():
        y32 = m(wave.to(DEV), video.to(DEV), torch.LongTensor(lens))
    _report("benched model, fp32 ragged logits", y32, ref, 1e-4)
    lib_options("bf16", 1)
    m.load_state_dict(sd)                                    # (the fp32 pass moved the running statistics)
    with torch.no_grad():
        y16 = m(wave.to(DEV), video.to(DEV), torch.LongTensor(lens))
    # bf16 tolerance: SURVEY 7 gives "about 2e-2 relative on the logits"; measured on this slice 1.9e-2 with fp32 storage and
    # operands rounded while staging (option bf16 = 2) and 2.6e-2 on the bf16 DATA PATH, whose residual stream (each block's
    # output, added back as the next block's identity) is itself stored in bf16: 3e-2 * max|ref|.
    _report("benched model, bf16 ragged logits", y16, ref, 3e-2 * float(ref.abs().max()))
    lib_options("bf16", 2)
    m.load_state_dict(sd)
    with torch.no_grad():
        y16c = m(wave.to(DEV), video.to(DEV), torch.LongTensor(lens))
    _report("benched model, bf16 (rounded while staging) ragged logits", y16c, ref, 3e-2 * float(ref.abs().max()))
    assert float((y16.cpu() - ref).abs().max()) > 1e-6      # the option really changed the arithmetic
    for b, n in enumerate(lens):                             # padded steps: the Linear bias, exactly, in either arithmetic
        assert torch.equal(y16[b, n:], y32[b, n:])


@pytest.mark.parametrize("N,H,W,C,Co,stride", [(200, 5, 5, 256, 256, 1), (130, 3, 3, 512, 512, 1), (256, 9, 9, 128, 256, 2),
                                               (128, 5, 5, 256, 512, 2), (300, 3, 3, 256, 128, 1), (1024, 3, 3, 512, 512, 1),
                                               (130, 6, 8, 128, 128, 2), (200, 9, 5, 256, 128, 2)])
def test_conv2d_position_classes_skip_the_zero_padding(N, H, W, C, Co, stride, lib_options):
    """3x3 / pad 1 convolutions whose tile count fits the stream-K pool run position-major (igemm.h "position classes"): a
    tile's rows share one grid position, its K loop walks only the taps that fall inside the image -- on a 3x3 grid 40 % of
    the products are multiplications by the zero padding; the weight gradient runs tap-major and contracts, per tap, only the
    grid positions at which that tap is inside the image.  Forward, data and weight gradient against torch's conv2d (same
    bounds as the dense schedule), with image counts that are not whole tiles (padded class rows), both strides; the skipped products
    are exact zeros, so the dense schedule (option no_cls) must agree to the last bits of the fp32 summation order.
    Stride 2: the data gradient's four parity classes (1 / 2 / 2 / 4 live taps) are position classes of ONE product
    (ClassSched::s2) -- odd and even grids, the even ones ending in a row that only tap 2 reaches."""
    import ctypes as Ct
    import torch.nn.functional as F
    from avvad import _lib as L, ops
    rng = np.random.RandomState(N + H * 7 + C)
    x = T(rng.normal(size=(N, C, H, W)).astype(np.float32)).requires_grad_(True)
    w = T((rng.normal(size=(Co, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32))
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, stride, 1)
    gy = T(rng.normal(size=tuple(y.shape)).astype(np.float32))
    y.backward(gy)
    wgrad_ref = w.grad.permute(2, 3, 1, 0).reshape(9 * C, Co)
    w = w.detach()
    lib = L.lib()
    st = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(N, H, W, C, Co, 3, stride, 1)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    wf = torch.empty(9 * C * Co, device=DEV)
    wdg = torch.empty(9 * C * Co, device=DEV)
    L.check(lib.avvad_conv2d_pack_weights(L.ptr(w.to(DEV)), L.ptr(wf), L.ptr(wdg), Ct.byref(d), st), "pack")
    ews = ops.engine_ws(DEV)
    gyd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    outs = {}
    for no_cls in (0, 1):
        lib_options("no_cls", no_cls)
        yd = torch.full((N, y.shape[2], y.shape[3], Co), float("nan"), device=DEV)
        L.check(lib.avvad_conv2d_fwd(L.ptr(xd), L.ptr(wf), L.ptr(yd), Ct.byref(d), L.ptr(ews), ews.numel() * 4, st), "fwd")
        tag = "conv %dx%dx%dx%d->%d s%d %s" % (N, H, W, C, Co, stride, "dense" if no_cls else "classes")
        _report(tag + " fwd", yd.permute(0, 3, 1, 2), y, 2e-5, 2e-5)
        dx = torch.full((N, H, W, C), float("nan"), device=DEV)
        L.check(lib.avvad_conv2d_dgrad(L.ptr(gyd), L.ptr(wdg), L.ptr(dx), Ct.byref(d), 0, L.ptr(ews), ews.numel() * 4, st), "dgrad")
        _report(tag + " dgrad", dx.permute(0, 3, 1, 2), x.grad, 2e-5, 2e-5)
        L.check(lib.avvad_conv2d_dgrad(L.ptr(gyd), L.ptr(wdg), L.ptr(dx), Ct.byref(d), 1, L.ptr(ews), ews.numel() * 4, st), "dgrad +=")
        _report(tag + " dgrad accumulate", dx.permute(0, 3, 1, 2), 2 * x.grad, 4e-5, 4e-5)
        dw = torch.full((9 * C, Co), float("nan"), device=DEV)
        L.check(lib.avvad_conv2d_wgrad(L.ptr(xd), L.ptr(gyd), L.ptr(dw), Ct.byref(d), L.ptr(ews), ews.numel() * 4, st), "wgrad")
        _report(tag + " wgrad", dw, wgrad_ref, 1e-4 * np.sqrt(N), 1e-4)
        outs[no_cls] = (yd, dx, dw)
        yd2 = torch.empty_like(yd)
        L.check(lib.avvad_conv2d_fwd(L.ptr(xd), L.ptr(wf), L.ptr(yd2), Ct.byref(d), L.ptr(ews), ews.numel() * 4, st), "fwd")
        assert torch.equal(yd, yd2)                              # run to run: the same bits
    assert _max_rel(outs[0][0], outs[1][0]) < 2e-6 and _max_rel(outs[0][1], outs[1][1]) < 4e-6
    assert _max_rel(outs[0][2], outs[1][2]) < 1e-5


def test_conv2d_refuses_kernels_beyond_the_tap_mask():
    """The im2col gathers keep one validity bit per tap in a 32-bit word: a 7x7 kernel on >= 32 channels must be refused
    (AVVAD_EINVAL), not computed with aliased taps."""
    import ctypes as Ct
    from avvad import _lib as L
    lib = L.lib()
    st = Ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(2, 17, 17, 32, 64, 7, 1, 3)
    x = torch.zeros(2, 17, 17, 32, device=DEV); w = torch.zeros(49 * 32, 64, device=DEV); y = torch.zeros(2, 17, 17, 64, device=DEV)
    assert lib.avvad_conv2d_fwd(L.ptr(x), L.ptr(w), L.ptr(y), Ct.byref(d), None, 0, st) == -1
    assert lib.avvad_conv2d_dgrad(L.ptr(y), L.ptr(w), L.ptr(x), Ct.byref(d), 0, None, 0, st) == -1
    d5 = L.ConvDesc(2, 17, 17, 32, 64, 5, 1, 2)               # 25 taps fit
    assert lib.avvad_conv2d_fwd(L.ptr(x), L.ptr(w), L.ptr(y), Ct.byref(d5), None, 0, st) == 0


def test_rccl_path_on_one_gpu(tmp_path):
    """The RCCL code path executed on the one GPU there is: a fresh child process (torch.distributed.run, WORLD_SIZE=1,
    backend nccl) trains the AV model with BucketReducer(force_hooks=True) -- ProcessGroupNCCL, hooks + in-place sinks,
    async all_reduce on slices of the flat CUDA buffer, the side-stream ordering, the presence-bitmap collective -- and
    without a reducer; at world size 1 the SUM is the identity, so the gradients must agree BIT FOR BIT."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "rccl1.pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("AVVAD_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_one_gpu_worker.py"), out]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired as e:
        raise AssertionError("RCCL one-rank worker did not finish in 300 s:\n" + str(e.stderr or "")[-3000:])
    assert r.returncode == 0 and os.path.exists(out), "RCCL one-rank worker failed (exit %d):\n%s" % (r.returncode, r.stderr[-3000:])
    got = torch.load(out, weights_only=True)
    info = got["info"]
    print("RCCL world-1: buckets %s, launched from hooks %s, absent %s, not launched %s, |grad| %.4e\n  diff %s" % (
        info["buckets"], info["launched_from_hooks_last_step"], [info["names"][i] for i in info["absent"]],
        [[info["names"][j] for j in range(len(info["names"])) if info["bucket_ranges"][b][0] <= info["offsets"][j] < info["bucket_ranges"][b][1]]
         for b in info["not_launched"]], float(got["with"].norm()), info["diff"]))
    assert got["loss"][0] == got["loss"][1] and float(got["with"].norm()) > 0
    assert info["diff"]["without_vs_without"] == [] and info["diff"]["with_vs_without"] == [], info["diff"]
    assert torch.equal(got["with"], got["without"])
    assert info["buckets"] > 2 and [info["names"][i] for i in info["absent"]] == ["bn.weight", "bn.bias"]   # agreed absent ...
    for b in info["not_launched"]:           # ... so every bucket that holds anything else left from the hooks
        lo, hi = info["bucket_ranges"][b][0], info["bucket_ranges"][b][1]
        assert all(j in info["absent"] for j in range(len(info["names"])) if lo <= info["offsets"][j] < hi), (b, info["not_launched"])
    assert info["launched_from_hooks_last_step"] >= info["buckets"] - 1


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_two_rank_gpu_data_parallel_step(tmp_path, lib_options, monkeypatch, overlap):
    """The N > 1 path on the real kernels: two processes (both on cuda:0, gradients exchanged through gloo -- RCCL
    refuses two ranks per device), each a shard of the batch; the SUM all-reduce of the flat gradient must equal the
    single-process gradient of the whole batch (loss is a sum over sequences; eval-mode BatchNorm so that shards are
    independent).  Exercises hooks + in-place gradient sinks + the side HIP stream + bucket launches together.
    Both sides run the whole-tile schedule (AVVAD_NO_STREAMK=all): a sample's forward is then bit-identical whatever
    batch it sits in, so no ReLU decision can differ between the shard runs and the whole-batch run (with stream-K
    the split points move with the batch size, and one flipped unit moves this small model's gradient by 1e-2).
    Both stream layouts are run: they change the order in which gradients become ready, and with it which bucket
    launches when (overlap=0 exposed a double-counted readiness signal that reduced a bucket too early)."""
    import socket
    import subprocess
    import sys
    import dp_gpu_case as case
    from avvad.optim import FlatAdam
    from packages.models.utils import batch_binary_cross_entropy
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "flat_grad.pt")
    from avvad import ops
    lib_options("no_streamk", 1)                 # this process (the whole-batch reference run below) ...
    monkeypatch.setattr(ops, "_OVERLAP", overlap == "1")
    env = dict(os.environ, AVVAD_DIST_BACKEND="gloo", AVVAD_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               AVVAD_NO_STREAMK="all", AVVAD_OVERLAP=overlap)      # ... and the workers (read once when they load the library)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_gpu_worker.py"), out]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired as e:
        # a hang in the two-rank path (collective order, a stream dependency) must read as a failure, not a skip
        raise AssertionError("two-rank workers did not finish in 300 s:\n" + str(e.stderr or "")[-3000:])
    if r.returncode != 0 or not os.path.exists(out):
        env_msgs = ("address already in use", "EADDRINUSE", "Gloo is not available", "gloo backend is not available")
        if any(m.lower() in r.stderr.lower() for m in env_msgs):      # recognised environment problems only
            pytest.skip("two-rank launch not possible on this box: " + r.stderr[-400:].replace("\n", " | "))
        raise AssertionError("two-rank worker failed (exit %d):\n%s" % (r.returncode, r.stderr[-3000:]))
    got = torch.load(out, weights_only=True)
    model = case.make_model().to(DEV).eval()
    wave, video, target, lengths = [t.to(DEV) for t in case.make_batch()]
    opt = FlatAdam(model.parameters(), lr=1e-3)
    refs = []
    for step in range(2):                                # the same two steps (one SGD update in between) on the whole batch
        loss = batch_binary_cross_entropy(model(wave, video, lengths), target, lengths, 1e-8)
        loss.backward()
        refs.append(opt.flat_grad.detach().cpu().clone())
        if step == 0:
            opt.flat.add_(opt.flat_grad, alpha=-case.SGD_LR)
            opt.zero_grad()
    rel0 = float((got["step0"] - refs[0]).norm() / refs[0].norm())
    rel1 = float((got["step1"] - refs[1]).norm() / refs[1].norm())
    print("two-rank DP: |flat grad| %.4e, relL2(all-reduced shards vs whole batch) step 1: %.2e, step 2: %.2e" % (float(refs[1].norm()), rel0, rel1))
    # First step: the forward is bit-identical on both sides, only the order of the sum over samples differs: 1e-5.
    # Second step: its weights carry the first step's 1e-7 differences, and ONE ReLU unit of this small model that sits on a
    # zero crossing may then decide differently -- measured once (under a CU cap, tools/lab/dp_two_rank_probe.py): 5e-4, every
    # tensor upstream of one layer-2 unit off by ~1e-3.  A reducer fault (a bucket reduced early, twice or not at all) is O(0.1+).
    assert float(refs[0].norm()) > 0 and rel0 < 1e-5 and rel1 < 5e-3
