#!/usr/bin/env python3
"""bench.py -- AV frame-pairs/s of one data-parallel training step on N MI355X of one node.

  python bench.py                                   # N=1, defaults finish in ~2 minutes
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W     # one rank per GPU over RCCL

Default workload "c4" (BASELINE.json configs[3]/[4], SURVEY.md 8d "C4/C5"; weak scaling): per GPU 64 sequences x
T=16 = 1024 AV frame-pairs: raw 16 kHz waveform (64,1,16*256+2047) -> WaveNet encoder "W0" (20 dilated layers,
R=D=32, Bn=256, P=16), 67x67 gray lip crops (64,16,67,67) -> ResNet-18 trunk, concat -> 2xLSTM(1024) -> FC(1),
masked summed BCE, backward, bucketed RCCL all-reduce of the flat gradient, fused Adam.  fp32 end to end
(fp32-input MFMA), synthetic inputs resident in HBM, random-init weights.  At N=8 this is exactly C5's shape
(8192 global frame-pairs; C5's bf16 arithmetic is `--dtype bf16`, never the headline line).

Other workloads (one JSON line each, same contract):
  --config c2   audio_net training: wave (256,1,16000) -> W0 (P=60) -> 2xLSTM(1024) -> FC; 15 360 frames / step
  --config c3   video_net training: 512 lip crops as (32,16,67,67) -> ResNet-18 -> 2xLSTM(1024) -> FC; 512 frames / step
  --forward-only        time the inference forward (eval mode, no autograd tape) instead of the training step

One "step" = forward + loss + backward + gradient exchange + Adam + zero_grad.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "audio-visual-vad_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HOP, H_IMG = 256, 67
RF = 2048


def w0(pool):
    return dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2, en_residual_channel=32,
                en_dilation_channel=32, en_bottleneck_width=256, en_pool_kernel_size=pool, use_bias=True)


N_SEQ, T_FRAMES = 64, 16
W0 = w0(T_FRAMES)
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0     # HBM3E spec (6.29 TB/s measured copy)
RESERVE_CUS_DP = 16       # N > 1: CUs the persistent conv grids leave to RCCL's channel workgroups
# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_summary.py -> profiles/r03_pmc_*_per_kernel.csv);
# FETCH_SIZE counts 128-B requests at 64 B on gfx950 -> doubled.  Bytes per launch of the two roofline kernels.
TRAFFIC = {"conv_fwd": None, "wn_layer": None}
try:
    with open(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")) as _f:
        TRAFFIC.update(json.load(_f))
except OSError:
    pass

CONFIGS = {
    # name: (kind, sequences per GPU, frames per sequence, samples per sequence, frames counted per step)
    "c4": dict(kind="av", n_seq=64, T=16, L=16 * HOP + RF - 1,
               what="AV_net fused training step (WaveNet-W0 encoder + ResNet-18 trunk + concat + 2xLSTM1024 + FC, masked BCE, "
                    "backward, RCCL all-reduce, fused Adam): BASELINE configs[3]/[4] per-GPU shard"),
    "c2": dict(kind="audio", n_seq=256, T=60, L=16000,
               what="audio_net training step (WaveNet-W0 encoder on 256 one-second 16 kHz chunks + 2xLSTM1024 + FC, masked BCE, "
                    "backward, fused Adam): BASELINE configs[1]"),
    "c3": dict(kind="video", n_seq=32, T=16, L=0,
               what="video_net training step (ResNet-18 trunk on 512 67x67 lip crops + 2xLSTM1024 + FC, masked BCE, backward, "
                    "fused Adam): BASELINE configs[2]"),
}


def make_inputs(torch, n_seq, seed, device, T=T_FRAMES, L=None, kind="av"):
    g = torch.Generator().manual_seed(seed)
    L = T * HOP + RF - 1 if L is None else L
    wave = torch.rand(n_seq, 1, max(L, 1), generator=g) * 2 - 1
    wave = wave / wave.abs().amax(dim=2, keepdim=True)            # peak-normalised like data_handling.py:441
    video = torch.randn(n_seq, T, H_IMG, H_IMG, generator=g) if kind != "audio" else torch.zeros(1)
    target = (torch.rand(n_seq, T, 1, generator=g) > 0.5).float()
    lengths = torch.full((n_seq,), T, dtype=torch.long)
    return [t.to(device) if device else t for t in (wave, video, target, lengths)]


def trunk_conv_shapes(n):
    """(C, Co, H, W, KS, stride, pad) of the 19 NHWC trunk convolutions behind the stem, for n frames."""
    out, cin, h = [], 64, 17
    for s, c in enumerate((64, 128, 256, 512)):
        for b in range(2):
            st = 2 if (b == 0 and s > 0) else 1
            out.append((cin, c, h, h, 3, st, 1))
            ho = (h + 2 - 3) // st + 1
            out.append((c, c, ho, ho, 3, 1, 1))
            if st == 2:
                out.append((cin, c, h, h, 1, 2, 0))
            cin, h = c, ho
    return out


# ---------------------------------------------------------------------------------------------- algorithmic work
def encoder_work(n_seq, L, cfg):
    """forward FLOPs and layer-at-a-time activation bytes (one read + one write of every layer's input/output plane)
    of the W0 encoder on n_seq sequences of L samples."""
    R, D, fw, Bn = cfg["en_residual_channel"], cfg["en_dilation_channel"], cfg["filter_width"], cfg["en_bottleneck_width"]
    Li = L - (fw - 1)
    flops = 2.0 * n_seq * Li * R * cfg["quantization_channel"] * fw
    byts = 4.0 * n_seq * (L + R * Li)
    for d in cfg["dilations"]:
        Lo = Li - d * (fw - 1)
        flops += 2.0 * n_seq * Lo * (D * R * fw + R * D)
        byts += 4.0 * n_seq * R * (Li + Lo)
        Li = Lo
    flops += 2.0 * n_seq * Li * Bn * R
    byts += 4.0 * n_seq * (R * Li + Bn * cfg["en_pool_kernel_size"])
    return flops, byts


def trunk_work(n):
    """forward FLOPs (3 identical input channels folded: conv1 reads 1 channel) and layer-at-a-time bytes per SURVEY 8d
    (2.5 MB / frame: every activation written once and read once, residual re-reads)."""
    flops = 2.0 * n * 34 * 34 * 64 * 49
    for (c, co, h, w, ks, st, pad) in trunk_conv_shapes(n):
        ho = (h + 2 * pad - ks) // st + 1
        flops += 2.0 * n * ho * ho * co * ks * ks * c
    return flops, 2.5e6 * n


def head_work(n_seq, T, in_dim, H=1024):
    flops = 2.0 * n_seq * T * (4 * H * (in_dim + H) + 4 * H * 2 * H + H)
    byts = 4.0 * (4 * H * (in_dim + H) + 4 * H * 2 * H) * (1 + T) + 4.0 * n_seq * T * (in_dim + 10 * H)   # W_hh re-streamed per step
    return flops, byts


def step_work(cfg_name, forward_only):
    c = CONFIGS[cfg_name]
    fl = by = 0.0
    in_dim = 0
    if c["kind"] in ("av", "audio"):
        f, b = encoder_work(c["n_seq"], c["L"], w0(c["T"]))
        fl, by, in_dim = fl + f, by + b, in_dim + 256
    if c["kind"] in ("av", "video"):
        f, b = trunk_work(c["n_seq"] * c["T"])
        fl, by, in_dim = fl + f, by + b, in_dim + 512
    f, b = head_work(c["n_seq"], c["T"], in_dim)
    fl, by = fl + f, by + b
    k = 1.0 if forward_only else 3.0       # backward = data + weight gradients: 2x the forward products
    return fl * k, by * k


# ---------------------------------------------------------------------------------------------- roofline probes
def _events(torch, fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps      # ms


def roofline_probe(torch, n_frames, reps=5, dtype="f32"):
    """Per-launch duration (HIP events on the launch stream = torch's current stream) of the dominant kernel: the
    fp32-MFMA implicit-GEMM convolution igemm::kernel<128,128,true,512,Im2colFwd<true>,ColTapRows<true>,...>, i.e. the forward of
    every trunk conv with Cout >= 128 (15 launches per step).  achieved = algorithmic FLOPs of those launches
    (2*N*Ho*Wo*Co*KS^2*C each) / their summed duration, everything the launch needs included (the GEMM kernel and,
    where a shape's tile count leaves a stream-K round, its fix-up kernel).  `traffic` is the
    per-launch HBM-side byte count from the rocprofv3 PMC passes committed under profiles/."""
    from avvad import _lib as L
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot_flop, exe_flop, tot_ms, n_launch = 0.0, 0.0, 0.0, 0
    per = []
    ews = torch.empty(lib.avvad_engine_workspace() // 4, device="cuda")     # the engine's stream-K scratch (caller-allocated)
    layer1 = None
    for (c, co, h, w, ks, stride, pad) in trunk_conv_shapes(n_frames):
        if co < 128 and not (dtype == "f32" and c == 64 and co == 64 and ks == 3 and layer1 is None):
            continue
        ho = (h + 2 * pad - ks) // stride + 1
        x = torch.randn(n_frames, h, w, c, device="cuda")
        wf = torch.randn(ks * ks * c, co, device="cuda") * 0.05
        y = torch.empty(n_frames, ho, ho, co, device="cuda")
        d = L.ConvDesc(n_frames, h, w, c, co, ks, stride, pad)
        if dtype == "bf16":          # the bf16 data path's kernel: operands bf16 in HBM (csrc/bgemm.h)
            x, wf = x.bfloat16(), wf.bfloat16()
            fn = lib.avvad_conv2d_fwd_bf16
        else:
            fn = lib.avvad_conv2d_fwd
        L.check(fn(L.ptr(x), L.ptr(wf), L.ptr(y), C.byref(d), L.ptr(ews), ews.numel() * 4, st), "conv fwd")
        ms = _events(torch, lambda: fn(L.ptr(x), L.ptr(wf), L.ptr(y), C.byref(d), L.ptr(ews), ews.numel() * 4, st), reps)
        flop = 2.0 * n_frames * ho * ho * co * ks * ks * c
        if co < 128:                 # the 64 -> 64 channel convolutions of layer1 (4 per step): their own kernel, reported beside the
            layer1 = {"kernel": "conv64::kernel<false> (weights-stationary: the 576x64 weight image in LDS, A fetched straight into "
                                "MFMA operand registers; csrc/conv64.h)", "launches_per_step": 4, "us": round(1e3 * ms, 1),
                      "TFLOPs": round(flop / ms / 1e9, 1), "frac": round(flop / ms / 1e9 / PEAK_F32_TFLOPS, 4)}
            continue                 # dominant kernel, not inside its average
        per.append((c, co, h, ks, stride, ms, flop / ms / 1e9))
        tot_flop += flop
        # products actually executed: 3x3 / pad 1 convolutions that run position-major skip the taps that fall into the
        # padding (csrc/igemm.h "position classes"; same eligibility rule as csrc/trunk.hip conv_fwd_cls_ok)
        live = 1.0
        if ks == 3 and pad == 1 and n_frames >= 128 and ho >= 2 and ho * ho * -(-n_frames // 128) * -(-co // 128) <= 1024:
            last = 1 if (ho - 1) * stride - pad + 2 > h - 1 else 0
            live = ((3 * ho - 1 - last) / (3.0 * ho)) ** 2
        exe_flop += flop * live
        tot_ms += ms
        n_launch += 1
    ach = tot_flop / tot_ms / 1e9
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
    kname = ("bgemm::kernel<128,128,false,Im2colFwd[Cls],RowPairs[Cls],Epi{Store,Cls}> (trunk conv forward on the bf16 data path, Cout>=128: "
             "bf16 operands in HBM, v_mfma_f32_32x32x16_bf16; position classes on the small grids) + its stream-K fix-up" if dtype == "bf16" else
             "igemm::kernel<128,128,true,512,Im2colFwd[Cls],ColTapRows[Cls],Epi{Store,Cls}> (trunk conv forward, Cout>=128: fp32 MFMA implicit "
             "GEMM, buffer-addressed gathers; the 3x3 convolutions of the 9x9 / 5x5 / 3x3 grids run position-major and skip their zero "
             "padding) + its stream-K fix-up")
    return {"bound": "mfma", "kernel": kname,
            "launches_per_step": n_launch, "avg_launch_us": round(1e3 * tot_ms / n_launch, 2),
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            # `achieved` counts ALGORITHMIC FLOPs (SURVEY 8d's dense 2*N*Ho*Wo*Co*KS^2*C); the products really multiplied:
            "executed_tflops": round(exe_flop / tot_ms / 1e9, 2), "executed_frac": round(exe_flop / tot_ms / 1e9 / peak, 4),
            "traffic": TRAFFIC.get("conv_fwd") if dtype == "f32" else None,
            "traffic_unit": "bytes/launch (HBM side, rocprofv3 PMC, profiles/r03_pmc_*_per_kernel.csv)",
            "per_shape": [{"C": a, "Co": b, "HW": c_, "k": d_, "s": e, "us": round(1e3 * f, 1), "TFLOPs": round(g, 1)}
                          for (a, b, c_, d_, e, f, g) in per],
            "layer1_conv64": layer1}


def roofline_probe_hbm(torch, n_seq, L, reps=10):
    """The dominant HBM-bound kernel: one layer-at-a-time residual block of the encoder (wn_block_fwd_occ: R = D = 32,
    filter width 2; planes of >= 8192 samples take the dwordx4 form wn_block_fwd_w4), timed per launch with HIP events through the
    single-layer entry point avvad_wavenet_block_fwd.  Algorithmic bytes per launch = one read of s_in + one write of
    s_out (the second tap and the residual are re-reads of the same plane: L2 hits by design) = 256 B per output sample."""
    from avvad import _lib as L_
    lib = L_.lib()
    if not hasattr(lib, "avvad_wavenet_block_fwd"):
        return None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for dil in (64, 512):
        Lin = L - 1 - 63                       # the plane a d=64 layer of the first stack sees (after causal + d=1..32)
        Lo = Lin - dil
        s_in = torch.randn(n_seq, 32, Lin, device="cuda")
        s_out = torch.empty(n_seq, 32, Lo, device="cuda")
        wd, bd = torch.randn(32, 32, 2, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.1
        we, be = torch.randn(32, 32, 1, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.1
        call = lambda: lib.avvad_wavenet_block_fwd(L_.ptr(s_in), L_.ptr(wd), L_.ptr(bd), L_.ptr(we), L_.ptr(be), L_.ptr(s_out),
                                                   n_seq, Lin, dil, st)
        L_.check(call(), "avvad_wavenet_block_fwd")
        ms = _events(torch, call, reps)
        byts = 4.0 * n_seq * 32 * (Lin + Lo)
        out.append((dil, ms, byts))
    ms = sum(m for _, m, _ in out) / len(out)
    byts = sum(b for _, _, b in out) / len(out)
    ach = byts / ms / 1e6
    return {"bound": "hbm", "kernel": "wn_block_fwd_occ<0> (encoder residual block, layer-at-a-time, R=D=32 fw=2, buffer addressing, 4 waves/SIMD, XCD-aware walk)",
            "avg_launch_us": round(1e3 * ms, 2), "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(ach / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_launch": int(byts), "traffic": TRAFFIC.get("wn_layer"),
            "per_dilation": [{"d": d, "us": round(1e3 * m, 1), "GBs": round(b / m / 1e6, 1)} for d, m, b in out]}


def cpu_baseline(torch, n_seq=32):
    """The CPU oracle (a restatement of the reference's PyTorch-CPU arithmetic: kind "port") timed on this
    host: forward + loss + backward of the same AV model on n_seq x 16 frame-pairs, all host threads."""
    from oracle import head, models
    from packages.models.AV_Net import DeepVAD_AV
    # the GPU box gives one-GPU jobs a 16-CPU share of a much larger host: os.cpu_count() would oversubscribe
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    torch.manual_seed(0)
    m = DeepVAD_AV(2, 1024, 1, wavenet_params=W0)
    sd = {k: (v.detach().clone().requires_grad_(v.dtype == torch.float32 and "running" not in k)) for k, v in m.state_dict().items()}
    wave, video, target, lengths = make_inputs(torch, n_seq, 99, None)
    lens = lengths.tolist()
    times = []
    for it in range(11):
        t0 = time.perf_counter()
        y = models.av_net(sd, wave, video, lens, 2, training=True, wavenet_cfg=W0)
        loss = head.batch_loss(y, target, lens, 1e-8)
        loss.backward()
        times.append(time.perf_counter() - t0)
        for v in sd.values():
            if v.requires_grad:
                v.grad = None
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(n_seq * T_FRAMES / t, 1), "unit": "frame-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle AV model (WaveNet W0 + ResNet-18 + 2xLSTM1024 + FC), forward+loss+backward, %d seq x %d frames, "
                      "median of 10 after 1 warm-up, %.2f s each" % (n_seq, T_FRAMES, t)}


def parity_probe(torch, model):
    """CPU-reference max|delta| on a RAGGED slice of the benched model itself: the benchmark's own weights
    (WaveNet W0, ResNet-18, 2xLSTM(1024), FC), 4 sequences of the benchmark's shape with lengths 16/9/12/5, train-mode
    BatchNorm, logits vs the oracle on the host."""
    from oracle import models
    wave, video, target, lengths = make_inputs(torch, 4, 4321, None)
    lens = [16, 9, 12, 5]
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref = models.av_net(sd, wave, video, lens, 2, training=model.training, wavenet_cfg=W0)
    rs = {k: model.state_dict()[k].clone() for k in sd if "running" in k or "num_batches" in k}
    with torch.no_grad():
        y = model(wave.cuda(), video.cuda(), torch.LongTensor(lens))
    model.load_state_dict(rs, strict=False)                      # the probe must not move the running statistics
    return float((y.detach().cpu() - ref).abs().max()), float(ref.abs().max())


def bench_c1(args):
    """BASELINE configs[0]: the audio_net WaveNet-encoder forward on the CPU, one 16000-sample chunk, batch 1, through the
    evaluate_audio_net.process_utt-shaped plumbing (peak-normalise -> encoder -> (1,60,Bn) -> 2xLSTM(1024) -> FC -> sigmoid ->
    threshold).  The reference's own CPU-runnable case: timed on the oracle (a restatement of the reference's PyTorch-CPU
    arithmetic), cores stated; where a GPU is present the same chunk also runs through the HIP path and the max |delta| of
    the logits is reported beside it."""
    import torch
    from oracle import models
    from packages.models.Audio_Net import DeepVAD_audio
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    cfg = w0(60)
    torch.manual_seed(0)
    m = DeepVAD_audio(2, 1024, 1, wavenet_params=cfg)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    wave = torch.rand(1, 1, 16000, generator=g) * 2 - 1

    def cpu_fwd():
        with torch.no_grad():
            x = wave / wave.abs().amax(dim=2, keepdim=True)                    # evaluate_audio_net.py:125-127
            y = models.audio_net(sd, x, [60], 2, wavenet_cfg=cfg)               # encoder -> (1,60,Bn) -> LSTM -> FC
            return y, (torch.sigmoid(y) > 0.5)                                 # :236-250 soft / hard decisions
    times = []
    for _ in range(args.warmup + max(5, args.steps)):
        t0 = time.perf_counter()
        y_ref, _ = cpu_fwd()
        times.append(time.perf_counter() - t0)
    t = sorted(times[args.warmup:])[len(times[args.warmup:]) // 2]
    out = {"metric": "audio_net WaveNet-encoder forward on CPU (frames/s)", "value": round(60 / t, 1), "unit": "frames/s", "n_gpus": 0,
           "steps": len(times) - args.warmup, "warmup": args.warmup, "ms_per_step": round(1e3 * t, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE configs[0]: audio_net (WaveNet-W0 encoder P=60 + 2xLSTM1024 + FC) forward on one 16000-sample "
                                  "chunk, batch 1, evaluate_audio_net.process_utt plumbing, CPU oracle (kind: port)", "name": "c1",
                      "cores": cores, "samples": 16000, "frames": 60}}
    if torch.cuda.is_available():
        mg = m.to("cuda:0").eval()
        wg = wave.to("cuda:0")
        from avvad import ops

        def gpu_fwd():
            with torch.no_grad():
                return mg(ops.peak_normalize(wg.view(1, -1)).view(1, 1, -1), [60])
        y = gpu_fwd()
        torch.cuda.synchronize()
        ms = _events(torch, gpu_fwd, 20)
        out["gpu"] = {"ms": round(ms, 3), "frames_per_s": round(60 / ms * 1e3, 1),
                      "cpu_ref_max_abs_delta": float((y.cpu() - y_ref).abs().max())}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS) + ["c1"], default="c4")
    ap.add_argument("--blocks", type=int, default=3, help="timed blocks of --steps steps each; the MEDIAN block is reported")
    ap.add_argument("--global-frame-pairs", type=int, default=0, help="c4 only: fix the GLOBAL batch (BASELINE configs[3]: 1024 over 2 and 4 "
                    "GPUs) and shard it over the ranks -- strong scaling; default: 1024 frame-pairs PER GPU (weak scaling)")
    ap.add_argument("--forward-only", action="store_true", help="inference forward instead of the training step")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="bf16: BASELINE configs[4] arithmetic (own line, never the headline)")
    ap.add_argument("--reserve-cus", type=int, default=-1, help="CUs left free by the persistent conv grids (room for RCCL kernels); "
                    "default: 0 on one GPU, %d at N > 1 (DESIGN.md 5)" % RESERVE_CUS_DP)
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline / parity probes")
    ap.add_argument("--ab", default="", help="tuning aid: after the timed region, time the step under library option NAME=VALUE\n"
                    "against its current value, A/B/A/B in this process (stderr; boxes differ by several %)")
    args = ap.parse_args()
    if args.config == "c1":
        return bench_c1(args)

    import torch
    import torch.distributed as dist
    from avvad import _lib as L
    from avvad import dist as avd
    from avvad.optim import FlatAdam
    from packages.models.AV_Net import DeepVAD_AV
    from packages.models.Audio_Net import DeepVAD_audio
    from packages.models.Video_Net import DeepVAD_video
    from packages.models.utils import batch_binary_cross_entropy

    rank, world, local = avd.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    reserve = args.reserve_cus if args.reserve_cus >= 0 else (RESERVE_CUS_DP if world > 1 else 0)
    if reserve:
        # only the BACKWARD entry points leave CUs free: that is when the bucketed all-reduce runs (the forward has no exchange)
        L.set_option("bwd_max_cus", 256 - reserve)
    if args.dtype == "bf16":
        if L.lib().avvad_set_option(b"bf16", 1) != 0:
            raise SystemExit("this build of libavvad_hip.so has no bf16 path")

    cfg = dict(CONFIGS[args.config])
    if args.global_frame_pairs:
        if args.config != "c4" or args.global_frame_pairs % (cfg["T"] * world):
            raise SystemExit("--global-frame-pairs: c4 only, and a multiple of %d x the number of GPUs" % cfg["T"])
        cfg["n_seq"] = args.global_frame_pairs // (cfg["T"] * world)
    kind, n_seq, T = cfg["kind"], cfg["n_seq"], cfg["T"]
    torch.manual_seed(0)                       # identical initial weights on every rank
    if kind == "av":
        model = DeepVAD_AV(2, 1024, 1, use_mcb=False, eps=1e-8, wavenet_params=w0(T))
    elif kind == "audio":
        model = DeepVAD_audio(2, 1024, 1, wavenet_params=w0(T))
    else:
        model = DeepVAD_video(2, 1024, 1)
    model = model.to(dev).train(not args.forward_only)
    wave, video, target, lengths = make_inputs(torch, n_seq, 1234 + rank, dev, T=T, L=cfg["L"], kind=kind)

    def fwd():
        if kind == "av":
            return model(wave, video, lengths)
        if kind == "audio":
            return model(wave, lengths)
        return model(video, lengths)

    if args.forward_only:
        for p_ in model.parameters():
            p_.requires_grad = False

        def step():
            with torch.no_grad():
                return fwd().sum()
    else:
        opt = FlatAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.999))
        reducer = avd.BucketReducer(opt.params, opt.flat_grad, opt.offsets,
                                    names=[n for n, q in model.named_parameters() if q.requires_grad])

        def step():
            loss = batch_binary_cross_entropy(fwd(), target, lengths, 1e-8)
            loss.backward()
            reducer.finish()
            opt.step()
            opt.zero_grad()
            return loss

    def log(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log("warm-up done")
    # --blocks timed blocks of EXACTLY --steps steps each, every one bracketed by a barrier + synchronize on both sides and
    # taken as the MAX over ranks; the MEDIAN block is the reported one (a single 0.4 s region let box-to-box and
    # clock-ramp noise of a few % decide the headline).
    block_dt = []
    for blk in range(max(1, args.blocks)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        block_dt.append(dt)
        log("timed block %d: %.3f s for %d steps" % (blk, dt, args.steps))
    dt = sorted(block_dt)[len(block_dt) // 2]
    final_loss = float(loss.item())
    log("timed region (median block): %.3f s for %d steps" % (dt, args.steps))
    if args.ab and world == 1:
        from avvad import _lib as L_
        name, val = args.ab.split("=")
        base = L_.get_option(name)
        for rep in range(3):
            for v_ in (base, int(val)):
                L_.set_option(name, v_)
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                log("A/B %s=%d: %.3f ms/step" % (name, v_, 1e3 * (time.perf_counter() - ta) / args.steps))
        L_.set_option(name, base)

    if rank == 0:
        fp_per_step = world * n_seq * T
        flops, byts = step_work(args.config, args.forward_only)
        t_step = dt / args.steps
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        unit = {"av": "frame-pairs/s", "audio": "frames/s", "video": "frames/s"}[kind]
        mode = "fwd" if args.forward_only else "fwd+bwd"
        out = {"metric": "AV frame-pairs/sec (%s)" % mode if kind == "av" else "%s_net frames/sec (%s)" % (kind, mode),
               "value": round(fp_per_step * args.steps / dt, 1),
               "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * t_step, 3), "higher_is_better": True,
               "scaling": "strong" if args.global_frame_pairs else "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": cfg["what"] if not args.forward_only else
                          cfg["what"].replace("training step", "inference forward").replace(", masked BCE, backward, RCCL all-reduce, fused Adam", "")
                          .replace(", masked BCE, backward, fused Adam", ""),
                          "name": args.config, "per_gpu_frame_pairs": n_seq * T, "global_frame_pairs": fp_per_step,
                          "sequences_per_gpu": n_seq, "frames_per_sequence": T, "samples_per_sequence": cfg["L"],
                          "parallelism": "dp%d" % world, "final_loss": round(final_loss, 4), "reserved_cus": reserve,
                          "timed_blocks_ms_per_step": [round(1e3 * b / args.steps, 3) for b in block_dt], "reported_block": "median",
                          # whole-step fractions of SURVEY 8d (per GPU): algorithmic FLOPs / layer-at-a-time bytes over the step time
                          "flop_frac": round(flops / t_step / 1e12 / peak, 4), "hbm_frac": round(byts / t_step / 1e9 / PEAK_HBM_GBS, 4),
                          "algorithmic_tflop_per_step": round(flops / 1e12, 4), "layerwise_gb_per_step": round(byts / 1e9, 3)}}
        if not args.no_extras:
            if kind != "audio":
                out["roofline"] = roofline_probe(torch, n_seq * T, dtype=args.dtype)
            hb = roofline_probe_hbm(torch, n_seq if kind != "video" else 64, cfg["L"] or (16 * HOP + RF - 1))
            if hb is not None:
                out["roofline_hbm" if "roofline" in out else "roofline"] = hb
            log("roofline probes done")
            if world == 1 and kind == "av":     # CPU baseline and CPU-reference delta: rank 0 at N=1 only
                # (bf16: the delta of the bf16 arithmetic to the fp32 CPU reference, BASELINE configs[4]'s own tolerance applies)
                out["cpu_ref_max_abs_delta"], out["cpu_ref_max_abs"] = parity_probe(torch, model)
                # the same probe at the INITIAL weights (what the parity tests bound: 1e-4 fp32, 3e-2 of max|ref| bf16); the
                # line above is taken at the weights the timed steps left behind -- saturated logits of a model that has
                # memorised its one batch, where a rounding difference is amplified the most
                torch.manual_seed(0)
                fresh = DeepVAD_AV(2, 1024, 1, use_mcb=False, eps=1e-8, wavenet_params=w0(T)).to(dev).train()
                out["cpu_ref_init_max_abs_delta"], out["cpu_ref_init_max_abs"] = parity_probe(torch, fresh)
                del fresh
                log("parity probe done")
                out["cpu_baseline"] = cpu_baseline(torch)
                log("cpu baseline done")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
