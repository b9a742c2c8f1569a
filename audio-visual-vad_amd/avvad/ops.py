"""torch.autograd wrappers around the C ABI of libavvad_hip.so.

PyTorch is plumbing here: it owns device memory, streams and the autograd tape;
every arithmetic step of the hot path runs in the HIP kernels.  All tensors
must be fp32, contiguous and live on the GPU -- anything else raises.
"""
import ctypes as C
import os

import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_SIDE = {}


_SIDE_PRIO = int(os.environ.get("AVVAD_SIDE_PRIO", "-1"))      # read once at import
_OVERLAP = os.environ.get("AVVAD_OVERLAP", "1") != "0"


def side_stream():
    """The per-device side HIP stream on which independent sub-graphs (the audio encoder) run next to the main one."""
    dev = torch.cuda.current_device()
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream(device=dev, priority=_SIDE_PRIO)
    return _SIDE[dev]


def side_streams():
    return list(_SIDE.values())


_FORK = {}        # device -> the stream that forked work onto the side stream (DeepVAD_AV.forward's caller stream)


def note_fork(main):
    _FORK[torch.cuda.current_device()] = main


def _join_side_after_backward():
    """Called from a Function.backward that ran on the side stream and wrote parameter gradients IN PLACE.  autograd
    joins the streams of a backward pass through its AccumulateGrad nodes; gradients written in place bypass those, so
    the engine never learns that the side stream took part and the caller's stream would NOT wait for it when
    backward() returns: an optimiser step (or a .cpu() of a gradient) could overtake the encoder's backward kernels.
    (Observed: bimodal 1e-7 differences between identical trainings of a tiny model -- Adam read the encoder's gradients
    before or after they were complete.)  The join is queued as an end-of-backward callback, so the trunk's backward on
    the main stream is not made to wait in the middle of the pass."""
    if not _SIDE:                 # no side stream was ever created (audio-only / video-only models, CPU-side unit tests)
        return
    dev = torch.cuda.current_device()
    side = _SIDE.get(dev)
    if side is None or torch.cuda.current_stream() != side:
        return
    main = _FORK.get(dev) or torch.cuda.default_stream()
    torch.autograd.Variable._execution_engine.queue_callback(lambda: main.wait_stream(side))


def overlap_enabled():
    return _OVERLAP


def set_overlap(flag):
    """Run the audio encoder on the side stream (True, default) or in line on the current stream."""
    global _OVERLAP
    _OVERLAP = bool(flag)


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise L.AvvadError("%s must be a GPU tensor: the AV-VAD hot path has no CPU fallback "
                           "(use the oracle/ package for CPU checks)" % name)
    if t.dtype != torch.float32:
        raise L.AvvadError("%s must be float32, got %s" % (name, t.dtype))
    return t.contiguous()


def _ws(nbytes, device):
    if nbytes == 0:
        raise L.AvvadError("workspace query failed (bad descriptor)")
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)


# ---------------------------------------------------------------------------
# Direct gradient accumulation.  Every backward kernel ACCUMULATES (+=) into the gradient pointer it is
# given, so when a parameter already owns a gradient buffer (FlatAdam re-homes all of them into one flat
# buffer) the kernel writes there directly and the Function returns None for that input: no zero-fill, no
# temporary, no autograd add kernel per parameter (~300 tiny launches per step saved).  Because autograd's
# AccumulateGrad node is bypassed, its post-accumulate hooks do not fire: listeners (the DP bucket reducer)
# register in GRAD_SINKS and are told about every parameter written this way.
GRAD_SINKS = []


def _grad_target(p, needed):
    """-> (tensor to accumulate into or None, direct?)"""
    if not needed or p is None:
        return None, False
    g = getattr(p, "grad", None)
    if isinstance(p, torch.nn.Parameter) and g is not None and g.is_contiguous() and g.dtype == torch.float32:
        return g, True
    return torch.zeros_like(p), False


def _finish_grads(params, targets):
    """returned gradients for autograd (None where written in place) + sink notification"""
    out = []
    if any(direct for _, direct in targets):
        _join_side_after_backward()
    for p, (g, direct) in zip(params, targets):
        if direct:
            for sink in GRAD_SINKS:
                sink(p)
            out.append(None)
        else:
            out.append(g)
    return out


def lengths_i32(lengths, device):
    """`lengths` arrives as a LongTensor (train loop) or a Python list (eval)."""
    if isinstance(lengths, torch.Tensor):
        return lengths.to(device=device, dtype=torch.int32).contiguous()
    return torch.tensor([int(v) for v in lengths], dtype=torch.int32, device=device)


# --------------------------------------------------------------------------- GEMM / Linear
def engine_ws(device):
    """Scratch of the GEMM engine (partial tiles of its stream-K round): caller-allocated per call, like every
    workspace of the C ABI; torch's caching allocator makes that a pointer bump."""
    return torch.empty(L.lib().avvad_engine_workspace() // 4, dtype=torch.float32, device=device)


def gemm(A, B, C_out, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, accumulate=False, split_k=1):
    d = L.GemmDesc(M, N, K, lda, ldb, ldc, int(transA), int(transB), int(accumulate), split_k, 0, 0)
    ws = engine_ws(C_out.device)
    L.check(L.lib().avvad_gemm_f32(L.ptr(A), L.ptr(B), L.ptr(bias), L.ptr(C_out), C.byref(d), L.ptr(ws), ws.numel() * 4,
                                   _stream()), "avvad_gemm_f32")


class LinearFn(torch.autograd.Function):
    """y = x W^T + b   (nn.Linear: Audio_Net.py:59, Video_Net.py:116, AV_Net.py:140)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = _dev(x, "x").view(-1, x.shape[-1])
        w = _dev(weight, "weight")
        rows, K = x2.shape
        N = w.shape[0]
        y = torch.empty(rows, N, dtype=torch.float32, device=x.device)
        gemm(x2, w, y, rows, N, K, K, K, N, transB=True, bias=_dev(bias, "bias") if bias is not None else None)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.prm = (weight, bias)
        return y.view(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        rows, K = x2.shape
        N = w.shape[0]
        dy2 = _dev(dy, "dy").view(rows, N)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            gemm(dy2, w, dx, rows, K, N, N, K, K)                       # [rows,N] x [N,K]
            dx = dx.view(dy.shape[:-1] + (K,))
        tw = _grad_target(ctx.prm[0], ctx.needs_input_grad[1])
        tb = _grad_target(ctx.prm[1], ctx.has_bias and ctx.needs_input_grad[2])
        if tw[0] is not None:
            gemm(dy2, x2, tw[0], N, K, rows, N, K, K, transA=True, accumulate=True)   # dy^T x
        if tb[0] is not None:
            L.check(L.lib().avvad_colsum_acc(L.ptr(dy2), rows, N, L.ptr(tb[0]), _stream()), "avvad_colsum_acc")
        dw, db = _finish_grads(ctx.prm, (tw, tb))
        return dx, dw, db


# --------------------------------------------------------------------------- LSTM layer
class LstmLayerFn(torch.autograd.Function):
    """One unidirectional LSTM layer over a padded batch with packed-sequence semantics."""

    @staticmethod
    def forward(ctx, x, lens32, w_ih, w_hh, b_ih, b_hh):
        x = _dev(x, "x")
        B, T, In = x.shape
        H = w_hh.shape[1]
        w_ih, w_hh, b_ih, b_hh = (_dev(t, n) for t, n in ((w_ih, "w_ih"), (w_hh, "w_hh"), (b_ih, "b_ih"), (b_hh, "b_hh")))
        d = L.LstmDesc(B, T, In, H, lens32.data_ptr(), 1)
        ws = _ws(L.lib().avvad_lstm_workspace(C.byref(d)), x.device)
        y = torch.empty(B, T, H, dtype=torch.float32, device=x.device)
        L.check(L.lib().avvad_lstm_layer_fwd(L.ptr(x), L.ptr(w_ih), L.ptr(w_hh), L.ptr(b_ih), L.ptr(b_hh), L.ptr(y),
                                             C.byref(d), L.ptr(ws), ws.numel() * 4, _stream()), "avvad_lstm_layer_fwd")
        ctx.save_for_backward(x, lens32, w_ih, w_hh, y, ws)
        ctx.prm = (w_ih, w_hh, b_ih, b_hh)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, lens32, w_ih, w_hh, y, ws = ctx.saved_tensors
        B, T, In = x.shape
        H = w_hh.shape[1]
        dy = _dev(dy, "dy")
        d = L.LstmDesc(B, T, In, H, lens32.data_ptr(), 1)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        tg = [_grad_target(p, ctx.needs_input_grad[2 + i]) for i, p in enumerate(ctx.prm)]
        L.check(L.lib().avvad_lstm_layer_bwd(L.ptr(x), L.ptr(w_ih), L.ptr(w_hh), L.ptr(y), L.ptr(dy), L.ptr(dx),
                                             L.ptr(tg[0][0]), L.ptr(tg[1][0]), L.ptr(tg[2][0]), L.ptr(tg[3][0]), C.byref(d),
                                             L.ptr(ws), ws.numel() * 4, _stream()), "avvad_lstm_layer_bwd")
        dw_ih, dw_hh, db_ih, db_hh = _finish_grads(ctx.prm, tg)
        return dx, None, dw_ih, dw_hh, db_ih, db_hh


def lstm_stack(x, lengths, lstm_module):
    """Runs the parameters held by an ``nn.LSTM`` container through the HIP layers."""
    if lstm_module.bidirectional or not lstm_module.bias or lstm_module.proj_size:
        raise L.AvvadError("only the unidirectional, biased LSTM of the reference is supported")
    lens32 = lengths_i32(lengths, x.device)
    y = x
    for l in range(lstm_module.num_layers):
        y = LstmLayerFn.apply(y, lens32, getattr(lstm_module, "weight_ih_l%d" % l), getattr(lstm_module, "weight_hh_l%d" % l),
                              getattr(lstm_module, "bias_ih_l%d" % l), getattr(lstm_module, "bias_hh_l%d" % l))
    return y


# --------------------------------------------------------------------------- WaveNet encoder
class WavenetFn(torch.autograd.Function):
    """wave (B,qc,L) -> (B,Bn,P).  params: causal_w, causal_b, bott_w, bott_b, then per layer
    dil_w, dil_b, dense_w, dense_b (biases None when use_bias is False)."""

    @staticmethod
    def _desc(cfg, B, Lin, dil_arr, save):
        # shared_device: this call runs on the side stream next to the trunk's kernels (DeepVAD_AV with overlap)
        dev = torch.cuda.current_device()
        shared = _OVERLAP and dev in _SIDE and torch.cuda.current_stream() == _SIDE[dev]
        return L.WavenetDesc(B, Lin, cfg["quantization_channel"], cfg["en_residual_channel"], cfg["en_dilation_channel"],
                             cfg["en_bottleneck_width"], cfg["filter_width"], cfg["en_pool_kernel_size"],
                             len(cfg["dilations"]), dil_arr, int(cfg["use_bias"]), int(save), int(shared))

    @staticmethod
    def _ptrs(ts, n):
        cw, cb, bw, bb = ts[:4]
        rest = ts[4:]
        arrs = [L.ptr_array(rest[k::4][:n]) for k in range(4)]
        p = L.WavenetPtrs(L.ptr(cw), L.ptr(cb), arrs[0], arrs[1], arrs[2], arrs[3], L.ptr(bw), L.ptr(bb))
        return p, arrs  # keep arrs alive

    @staticmethod
    def forward(ctx, wave, cfg, *params):
        wave = _dev(wave, "wave")
        B, qc, Lin = wave.shape
        n = len(cfg["dilations"])
        if qc != cfg["quantization_channel"] or len(params) != 4 + 4 * n:
            raise L.AvvadError("wavenet: bad input channels / parameter list")
        ctx.owners = params          # the Parameter objects themselves (their .grad may be written directly)
        params = tuple(None if t is None else _dev(t, "param") for t in params)
        dil_arr = (C.c_int * max(1, n))(*cfg["dilations"])
        save = any(ctx.needs_input_grad)
        d = WavenetFn._desc(cfg, B, Lin, dil_arr, save)
        ws = _ws(L.lib().avvad_wavenet_workspace(C.byref(d)), wave.device)
        out = torch.empty(B, cfg["en_bottleneck_width"], cfg["en_pool_kernel_size"], dtype=torch.float32, device=wave.device)
        p, keep = WavenetFn._ptrs(params, n)
        L.check(L.lib().avvad_wavenet_fwd(L.ptr(wave), C.byref(p), L.ptr(out), C.byref(d), L.ptr(ws), ws.numel() * 4,
                                          _stream()), "avvad_wavenet_fwd")
        ctx.cfg = cfg
        ctx.save_for_backward(wave, ws, *[t for t in params if t is not None])
        ctx.mask = [t is not None for t in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        cfg = ctx.cfg
        saved = ctx.saved_tensors
        wave, ws = saved[0], saved[1]
        it = iter(saved[2:])
        params = tuple(next(it) if m else None for m in ctx.mask)
        n = len(cfg["dilations"])
        B, qc, Lin = wave.shape
        dil_arr = (C.c_int * max(1, n))(*cfg["dilations"])
        d = WavenetFn._desc(cfg, B, Lin, dil_arr, True)
        owners = ctx.owners
        tg = [_grad_target(o if o is not None else t, t is not None and ctx.needs_input_grad[2 + i])
              for i, (o, t) in enumerate(zip(owners, params))]
        grads = tuple(g for g, _ in tg)
        dwave = torch.empty_like(wave) if ctx.needs_input_grad[0] else None
        p, keep1 = WavenetFn._ptrs(params, n)
        g, keep2 = WavenetFn._ptrs(grads, n)
        L.check(L.lib().avvad_wavenet_bwd(L.ptr(wave), C.byref(p), L.ptr(_dev(dout, "dout")), C.byref(g), L.ptr(dwave),
                                          C.byref(d), L.ptr(ws), ws.numel() * 4, _stream()), "avvad_wavenet_bwd")
        grads = tuple(_finish_grads([o if o is not None else t for o, t in zip(owners, params)], tg))
        return (dwave, None) + grads


# --------------------------------------------------------------------------- ResNet-18 trunk
class TrunkFn(torch.autograd.Function):
    """frames (N,H,W) -> (N,512).  tensors: 20 conv_w, 20 bn_w, 20 bn_b, 20 running_mean, 20 running_var
    in the conv index order of include/avvad.h.  Running stats are updated in place when training."""

    @staticmethod
    def _params(ts):
        n = L.TRUNK_NCONV
        p = L.TrunkParams()
        for i in range(n):
            p.conv_w[i] = ts[i].data_ptr()
            p.bn_w[i] = ts[n + i].data_ptr()
            p.bn_b[i] = ts[2 * n + i].data_ptr()
            p.bn_rm[i] = ts[3 * n + i].data_ptr()
            p.bn_rv[i] = ts[4 * n + i].data_ptr()
        return p

    @staticmethod
    def forward(ctx, frames, training, momentum, eps, *ts):
        frames = _dev(frames, "frames")
        N, H, W = frames.shape
        n = L.TRUNK_NCONV
        if len(ts) != 5 * n:
            raise L.AvvadError("trunk: expected %d tensors" % (5 * n))
        owners = ts
        ts = tuple(_dev(t, "trunk tensor") for t in ts)
        save = any(ctx.needs_input_grad[4:4 + 3 * n])
        d = L.TrunkDesc(N, H, W, int(training), float(momentum), float(eps), int(save))
        ws = _ws(L.lib().avvad_trunk_workspace(C.byref(d)), frames.device)
        feat = torch.empty(N, 512, dtype=torch.float32, device=frames.device)
        p = TrunkFn._params(ts)
        L.check(L.lib().avvad_trunk_fwd(L.ptr(frames), C.byref(p), L.ptr(feat), C.byref(d), L.ptr(ws), ws.numel() * 4,
                                        _stream()), "avvad_trunk_fwd")
        if save:
            ctx.save_for_backward(frames, ws, *ts)
            ctx.cfg = (int(training), float(momentum), float(eps))
            ctx.owners = owners
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        saved = ctx.saved_tensors
        frames, ws, ts = saved[0], saved[1], saved[2:]
        N, H, W = frames.shape
        n = L.TRUNK_NCONV
        training, momentum, eps = ctx.cfg
        d = L.TrunkDesc(N, H, W, training, momentum, eps, 1)
        p = TrunkFn._params(ts)
        g = L.TrunkGrads()
        tg = [_grad_target(ctx.owners[i], ctx.needs_input_grad[4 + i]) for i in range(3 * n)]
        for i, (gt, _) in enumerate(tg):
            if gt is not None:
                (g.conv_w, g.bn_w, g.bn_b)[i // n][i % n] = gt.data_ptr()
        L.check(L.lib().avvad_trunk_bwd(L.ptr(frames), C.byref(p), L.ptr(_dev(dfeat, "dfeat")), C.byref(g), C.byref(d),
                                        L.ptr(ws), ws.numel() * 4, _stream()), "avvad_trunk_bwd")
        grads = _finish_grads(ctx.owners[:3 * n], tg)
        return (None, None, None, None) + tuple(grads) + (None,) * (2 * n)


def trunk_saved_activations(feat):
    """Test support: the post-ReLU activations kept for backward by the TrunkFn node behind ``feat``, as a dict
    name -> (N,C,H,W) view: ``pool``, ``<stage>.<block>.a1``, ``<stage>.<block>`` (the oracle's names)."""
    todo, seen, node = [feat.grad_fn], set(), None          # breadth-first over the autograd graph behind `feat`
    while todo:
        f = todo.pop(0)
        if f is None or f in seen:
            continue
        seen.add(f)
        if type(f).__name__ == "TrunkFnBackward":
            node = f
            break
        todo.extend(nf for nf, _ in f.next_functions)
    if node is None:
        raise L.AvvadError("no TrunkFn node behind this tensor (was the forward run with gradients enabled?)")
    frames, ws = node.saved_tensors[0], node.saved_tensors[1]
    N, H, W = frames.shape
    training, momentum, eps = node.cfg
    d = L.TrunkDesc(N, H, W, training, momentum, eps, 1)
    out = {}
    for idx in range(17):
        off, c, h, w = C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
        L.check(L.lib().avvad_trunk_activation(C.byref(d), idx, C.byref(off), C.byref(c), C.byref(h), C.byref(w)), "avvad_trunk_activation")
        t = ws[off.value: off.value + N * h.value * w.value * c.value].view(N, h.value, w.value, c.value).permute(0, 3, 1, 2)
        k = (idx - 1) // 2
        name = "pool" if idx == 0 else "%d.%d%s" % (4 + k // 2, k % 2, ".a1" if (idx - 1) % 2 == 0 else "")
        out[name] = t
    return out


# --------------------------------------------------------------------------- MCB fusion
class McbFusionFn(torch.autograd.Function):
    """audio (B,T,A), video (B,T,V) -> BatchNorm1d(L2norm(ssqrt(MCB(audio, video))))  (B,T,D)  (AV_Net.py:109-121)."""

    @staticmethod
    def forward(ctx, audio, video, h1, s1, h2, s2, bn_w, bn_b, rm, rv, eps, training, momentum):
        audio, video = _dev(audio, "audio"), _dev(video, "video")
        B, T, A = audio.shape
        V = video.shape[-1]
        D = bn_w.numel()
        for t, n in ((h1, "h1"), (h2, "h2")):
            if not t.is_cuda or t.dtype != torch.long:
                raise L.AvvadError("%s must be an int64 GPU tensor" % n)
        d = L.McbDesc(B * T, A, V, D, float(eps), int(training), float(momentum), 1)
        ws = _ws(L.lib().avvad_mcb_workspace(C.byref(d)), audio.device)
        out = torch.empty(B, T, D, dtype=torch.float32, device=audio.device)
        L.check(L.lib().avvad_mcb_fusion_fwd(L.ptr(audio), L.ptr(video), L.ptr(h1), L.ptr(_dev(s1, "s1")), L.ptr(h2),
                                             L.ptr(_dev(s2, "s2")), L.ptr(_dev(bn_w, "bn_w")), L.ptr(_dev(bn_b, "bn_b")),
                                             L.ptr(rm), L.ptr(rv), L.ptr(out), C.byref(d), L.ptr(ws), ws.numel() * 4,
                                             _stream()), "avvad_mcb_fusion_fwd")
        ctx.save_for_backward(audio, video, h1, s1, h2, s2, bn_w, ws)
        ctx.cfg = (float(eps), int(training), float(momentum))
        ctx.prm = (bn_w, bn_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        audio, video, h1, s1, h2, s2, bn_w, ws = ctx.saved_tensors
        B, T, A = audio.shape
        V, D = video.shape[-1], bn_w.numel()
        eps, training, momentum = ctx.cfg
        d = L.McbDesc(B * T, A, V, D, eps, training, momentum, 1)
        da = torch.empty_like(audio) if ctx.needs_input_grad[0] else None
        dv = torch.empty_like(video) if ctx.needs_input_grad[1] else None
        tg = [_grad_target(ctx.prm[0], ctx.needs_input_grad[6]), _grad_target(ctx.prm[1], ctx.needs_input_grad[7])]
        L.check(L.lib().avvad_mcb_fusion_bwd(L.ptr(audio), L.ptr(video), L.ptr(h1), L.ptr(s1), L.ptr(h2), L.ptr(s2),
                                             L.ptr(bn_w), L.ptr(_dev(dout, "dout")), L.ptr(da), L.ptr(dv), L.ptr(tg[0][0]),
                                             L.ptr(tg[1][0]), C.byref(d), L.ptr(ws), ws.numel() * 4, _stream()),
                "avvad_mcb_fusion_bwd")
        dw, db = _finish_grads(ctx.prm, tg)
        return da, dv, None, None, None, None, dw, db, None, None, None, None, None


class CountSketchFn(torch.autograd.Function):
    """psi(x, h, s): out[..., h[i]] += s[i] x[..., i]  (compact_bilinear_pooling.py:7-27,41-57)."""

    @staticmethod
    def forward(ctx, h, s, output_size, x):
        x = _dev(x, "x")
        if not h.is_cuda or h.dtype != torch.long:
            raise L.AvvadError("h must be an int64 GPU tensor")
        In = x.shape[-1]
        rows = x.numel() // In
        out = torch.empty(x.shape[:-1] + (output_size,), dtype=torch.float32, device=x.device)
        L.check(L.lib().avvad_count_sketch_fwd(L.ptr(x), L.ptr(h), L.ptr(_dev(s, "s")), L.ptr(out), rows, In, output_size,
                                               _stream()), "avvad_count_sketch_fwd")
        ctx.save_for_backward(h, s)
        ctx.dims = (tuple(x.shape), rows, In, output_size)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, s = ctx.saved_tensors
        shape, rows, In, D = ctx.dims
        dx = torch.empty(shape, dtype=torch.float32, device=dout.device)
        L.check(L.lib().avvad_count_sketch_bwd(L.ptr(_dev(dout, "dout")), L.ptr(h), L.ptr(s), L.ptr(dx), rows, In, D, _stream()),
                "avvad_count_sketch_bwd")
        return None, None, None, dx


class CompactBilinearPoolingFn(torch.autograd.Function):
    """psi(x,h1,s1) (*) psi(y,h2,s2): the raw pooled vector (compact_bilinear_pooling.py:140-220)."""

    @staticmethod
    def forward(ctx, h1, s1, h2, s2, output_size, x, y):
        x, y = _dev(x, "x"), _dev(y, "y")
        if x.shape[:-1] != y.shape[:-1]:
            raise L.AvvadError("compact bilinear pooling: leading dimensions differ")
        for t, n in ((h1, "h1"), (h2, "h2")):
            if not t.is_cuda or t.dtype != torch.long:
                raise L.AvvadError("%s must be an int64 GPU tensor" % n)
        A, V = x.shape[-1], y.shape[-1]
        rows = x.numel() // A
        out = torch.empty(x.shape[:-1] + (output_size,), dtype=torch.float32, device=x.device)
        L.check(L.lib().avvad_mcb_fwd(L.ptr(x), L.ptr(y), L.ptr(h1), L.ptr(_dev(s1, "s1")), L.ptr(h2), L.ptr(_dev(s2, "s2")),
                                      L.ptr(out), rows, A, V, output_size, _stream()), "avvad_mcb_fwd")
        ctx.save_for_backward(h1, s1, h2, s2, x, y)
        ctx.dims = (rows, A, V, output_size)
        return out

    @staticmethod
    def backward(ctx, dout):
        h1, s1, h2, s2, x, y = ctx.saved_tensors
        rows, A, V, D = ctx.dims
        dx = torch.empty_like(x) if ctx.needs_input_grad[5] else None
        dy = torch.empty_like(y) if ctx.needs_input_grad[6] else None
        L.check(L.lib().avvad_mcb_bwd(L.ptr(x), L.ptr(y), L.ptr(h1), L.ptr(s1), L.ptr(h2), L.ptr(s2), L.ptr(_dev(dout, "dout")),
                                      L.ptr(dx), L.ptr(dy), rows, A, V, D, _stream()), "avvad_mcb_bwd")
        return None, None, None, None, None, dx, dy


# --------------------------------------------------------------------------- loss
class Bce2ClassesFn(torch.autograd.Function):
    """-mean(sum(x log(r1+eps) + (1-x) log(r2+eps), dim=-1))  (models/utils.py:115-116), r1/r2 probabilities."""

    @staticmethod
    def forward(ctx, r1, r2, x, eps):
        r1, r2 = _dev(r1, "r1"), _dev(r2, "r2")
        x = _dev(x.to(torch.float32), "x")
        if r1.shape != r2.shape or r1.shape != x.shape:
            raise L.AvvadError("binary_cross_entropy_2classes: r1, r2, x must have the same shape")
        Y = r1.shape[-1]
        rows = r1.numel() // Y
        loss = torch.empty(1, dtype=torch.float32, device=r1.device)
        d1, d2 = torch.empty_like(r1), torch.empty_like(r2)
        L.check(L.lib().avvad_bce_2classes(L.ptr(r1), L.ptr(r2), L.ptr(x), L.ptr(loss), L.ptr(d1), L.ptr(d2), rows, Y, float(eps),
                                           _stream()), "avvad_bce_2classes")
        ctx.save_for_backward(d1, d2)
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        out = []
        for g in ctx.saved_tensors:
            g = g.clone()
            L.check(L.lib().avvad_scale_by_device_scalar(L.ptr(g), L.ptr(_dev(dloss, "dloss").view(1)), g.numel(), _stream()),
                    "avvad_scale_by_device_scalar")
            out.append(g)
        return out[0], out[1], None, None



class MaskedBceFn(torch.autograd.Function):
    """sum_b mean_{t<len_b} BCE-with-eps(logits, targets)  (models/utils.py:108-113 + train_AV_net.py:298-301)."""

    @staticmethod
    def forward(ctx, logits, targets, lens32, eps):
        logits = _dev(logits, "logits")
        targets = _dev(targets.to(torch.float32), "targets")
        B, T, Y = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits)
        L.check(L.lib().avvad_bce_masked(L.ptr(logits), L.ptr(targets), L.ptr(lens32), L.ptr(loss), L.ptr(dl), B, T, Y,
                                         float(eps), _stream()), "avvad_bce_masked")
        ctx.save_for_backward(dl)
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        (dl,) = ctx.saved_tensors
        g = dl.clone()
        L.check(L.lib().avvad_scale_by_device_scalar(L.ptr(g), L.ptr(_dev(dloss, "dloss").view(1)), g.numel(), _stream()),
                "avvad_scale_by_device_scalar")
        return g, None, None, None


def masked_bce(logits, targets, lengths, eps=1e-8):
    return MaskedBceFn.apply(logits, targets, lengths_i32(lengths, logits.device), eps)


# --------------------------------------------------------------------------- layout helpers
class ConcatColsFn(torch.autograd.Function):
    """torch.cat([a, b], dim=2) (AV_Net.py:124), each branch written straight into the buffer."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _dev(a, "a"), _dev(b, "b")
        rows = a.numel() // a.shape[-1]
        ca, cb = a.shape[-1], b.shape[-1]
        y = torch.empty(a.shape[:-1] + (ca + cb,), dtype=torch.float32, device=a.device)
        lib = L.lib()
        L.check(lib.avvad_copy_cols(L.ptr(a), L.ptr(y), rows, ca, ca, 0, ca + cb, 0, _stream()), "avvad_copy_cols")
        L.check(lib.avvad_copy_cols(L.ptr(b), L.ptr(y), rows, cb, cb, 0, ca + cb, ca, _stream()), "avvad_copy_cols")
        ctx.dims = (rows, ca, cb, a.shape, b.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, ca, cb, sa, sb = ctx.dims
        dy = _dev(dy, "dy")
        lib = L.lib()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty(sa, dtype=torch.float32, device=dy.device)
            L.check(lib.avvad_copy_cols(L.ptr(dy), L.ptr(da), rows, ca, ca + cb, 0, ca, 0, _stream()), "avvad_copy_cols")
        if ctx.needs_input_grad[1]:
            db = torch.empty(sb, dtype=torch.float32, device=dy.device)
            L.check(lib.avvad_copy_cols(L.ptr(dy), L.ptr(db), rows, cb, ca + cb, ca, cb, 0, _stream()), "avvad_copy_cols")
        return da, db


class TransposeLast2Fn(torch.autograd.Function):
    """(B,C,T) -> (B,T,C): encoder output to the LSTM's batch-first layout."""

    @staticmethod
    def forward(ctx, x):
        x = _dev(x, "x")
        B, Cc, T = x.shape
        y = torch.empty(B, T, Cc, dtype=torch.float32, device=x.device)
        L.check(L.lib().avvad_transpose_last2(L.ptr(x), L.ptr(y), B, Cc, T, _stream()), "avvad_transpose_last2")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _dev(dy, "dy")
        B, T, Cc = dy.shape
        dx = torch.empty(B, Cc, T, dtype=torch.float32, device=dy.device)
        L.check(L.lib().avvad_transpose_last2(L.ptr(dy), L.ptr(dx), B, T, Cc, _stream()), "avvad_transpose_last2")
        return dx


# --------------------------------------------------------------------------- STFT front-end (no gradient)
def n_frames(L, n_fft, hop, pad_at_end=True, fs=16e3):
    """frame count of stft_pytorch(center=False): one hop of zeros is appended when the utterance is not a whole
    number of hops (packages/processing/stft.py:134-139)."""
    import math
    wlen_sec, hop_percent = n_fft / fs, hop / n_fft
    if pad_at_end:
        v = L / fs / wlen_sec / hop_percent
        if math.ceil(v) != int(v):
            L = L + hop
    return (L - n_fft) // hop + 1


def stft(wave, n_fft=1024, hop=256, mode=0, eps=1e-8, pad_at_end=True, fs=16e3, mean=None, std=None, norm_eps=1e-8):
    """wave (B,L) or (L,) on the GPU.  mode 0: log-power (B,T,F); 1: power (B,T,F); 2: legacy real view (F,T,2).
    With ``mean`` / ``std`` (F values each, mode 0) the train-set standardisation (x - mean) / (std + norm_eps) of the
    evaluate scripts is applied in the same pass (avvad_stft_features)."""
    w = _dev(wave, "wave")
    w2 = w.view(1, -1) if w.dim() == 1 else w
    B, Ls = w2.shape
    T = n_frames(Ls, n_fft, hop, pad_at_end, fs)
    F = n_fft // 2 + 1
    d = L.StftDesc(B, Ls, n_fft, hop, T, float(eps))
    ws = _ws(L.lib().avvad_stft_workspace(C.byref(d)), w.device)
    out = torch.empty((F, T, 2) if mode == 2 else (B, T, F), dtype=torch.float32, device=w.device)
    if mean is not None:
        if mode != 0 or std is None:
            raise L.AvvadError("standardisation is fused into the log-power mode only and needs both mean and std")
        mean, std = _dev(mean, "mean").reshape(-1), _dev(std, "std").reshape(-1)
        if mean.numel() != F or std.numel() != F:
            raise L.AvvadError("mean / std must hold %d values" % F)
        L.check(L.lib().avvad_stft_features(L.ptr(w2), L.ptr(mean), L.ptr(std), L.ptr(out), C.byref(d), float(norm_eps), L.ptr(ws),
                                            ws.numel() * 4, _stream()), "avvad_stft_features")
        return out
    L.check(L.lib().avvad_stft(L.ptr(w2), L.ptr(out), C.byref(d), mode, L.ptr(ws), ws.numel() * 4, _stream()), "avvad_stft")
    return out


def peak_normalize(wave):
    """x / max|x| per utterance (evaluate_audio_net.py:125-127); wave (B,L) or (L,)."""
    w = _dev(wave, "wave")
    w2 = w.view(1, -1) if w.dim() == 1 else w
    out = torch.empty_like(w2)
    L.check(L.lib().avvad_peak_normalize(L.ptr(w2), L.ptr(out), w2.shape[0], w2.shape[1], _stream()), "avvad_peak_normalize")
    return out.view(w.shape)


def standardize(x, mean, std, eps=1e-8):
    """(x - mean.T) / (std + eps).T of the train / evaluate loops (train_AV_net.py:286-291).  ``mean`` / ``std`` hold
    either one value per feature of the last axis (audio: (513,1)) or a single value (video: (1,1))."""
    x = _dev(x, "x")
    mean, std = _dev(mean, "mean").reshape(-1), _dev(std, "std").reshape(-1)
    F = x.shape[-1]
    nstat = mean.numel()
    if std.numel() != nstat or nstat not in (1, F):
        raise L.AvvadError("standardize: statistics must hold 1 or %d values, got %d / %d" % (F, nstat, std.numel()))
    out = torch.empty_like(x)
    L.check(L.lib().avvad_standardize(L.ptr(x), L.ptr(mean), L.ptr(std), L.ptr(out), x.numel() // F, F, nstat, float(eps),
                                      _stream()), "avvad_standardize")
    return out
