#!/usr/bin/env python3
"""bench.py -- AV frame-pairs/s of one data-parallel training step on N MI355X of one node.

  python bench.py                                   # N=1, defaults finish in ~2 minutes
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W     # one rank per GPU over RCCL

Workload (BASELINE.json configs[3]/[4], SURVEY.md 8d "C4/C5"; weak scaling): per GPU 64 sequences x T=16 =
1024 AV frame-pairs: raw 16 kHz waveform (64,1,16*256+2047) -> WaveNet encoder "W0" (20 dilated layers,
R=D=32, Bn=256, P=16), 67x67 gray lip crops (64,16,67,67) -> ResNet-18 trunk, concat -> 2xLSTM(1024) -> FC(1),
masked summed BCE, backward, bucketed RCCL all-reduce of the flat gradient, fused Adam.  fp32 end to end
(fp32-input MFMA), synthetic inputs resident in HBM, random-init weights.  At N=8 this is exactly C5
(8192 global frame-pairs).

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

N_SEQ, T_FRAMES, HOP, H_IMG = 64, 16, 256, 67
W0 = dict(filter_width=2, quantization_channel=1, dilations=[2 ** i for i in range(10)] * 2, en_residual_channel=32,
          en_dilation_channel=32, en_bottleneck_width=256, en_pool_kernel_size=T_FRAMES, use_bias=True)
RF = 2048
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak
# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --no-extras`, dominant kernel, per
# launch: FETCH_SIZE 99,916 KB x 2 (gfx950 counts 128-B requests at 64 B) + WRITE_SIZE 56,047 KB
# (tools/pmc_summary.py -> profiles/r01_pmc_{fetch,write}_size_per_kernel.csv)
TRAFFIC_PER_LAUNCH_BYTES = int((2 * 99916 + 56047) * 1024)


def make_inputs(torch, n_seq, seed, device):
    g = torch.Generator().manual_seed(seed)
    wave = torch.rand(n_seq, 1, T_FRAMES * HOP + RF - 1, generator=g) * 2 - 1
    wave = wave / wave.abs().amax(dim=2, keepdim=True)            # peak-normalised like data_handling.py:441
    video = torch.randn(n_seq, T_FRAMES, H_IMG, H_IMG, generator=g)
    target = (torch.rand(n_seq, T_FRAMES, 1, generator=g) > 0.5).float()
    lengths = torch.full((n_seq,), T_FRAMES, dtype=torch.long)
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


def roofline_probe(torch, n_frames, reps=5):
    """Per-launch duration (HIP events on the launch stream) of the dominant kernel: the fp32-MFMA
    implicit-GEMM convolution igemm::kernel<128,128,true,512,Im2colFwd,ColTapRows,EpiStore>, i.e. the forward
    of every trunk conv with Cout >= 128 (15 launches per step).  achieved = algorithmic FLOPs of those
    launches (2*N*Ho*Wo*Co*KS^2*C each) / their summed duration.  `traffic` is NOT measured here: it is the
    per-launch HBM-side byte count from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE in separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B/lane reads)."""
    from avvad import _lib as L
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot_flop, tot_ms, n_launch = 0.0, 0.0, 0
    per = []
    for (c, co, h, w, ks, stride, pad) in trunk_conv_shapes(n_frames):
        if co < 128:
            continue
        ho = (h + 2 * pad - ks) // stride + 1
        x = torch.randn(n_frames, h, w, c, device="cuda")
        wf = torch.randn(ks * ks * c, co, device="cuda") * 0.05
        y = torch.empty(n_frames, ho, ho, co, device="cuda")
        d = L.ConvDesc(n_frames, h, w, c, co, ks, stride, pad)
        for _ in range(2):
            L.check(lib.avvad_conv2d_fwd(L.ptr(x), L.ptr(wf), L.ptr(y), C.byref(d), st), "conv fwd")
        # time the GEMM kernel alone: the stream-K zero-fill of the output is a separate 13 us kernel with its own
        # line in the rocprof summary (zero_strided); with it skipped the events bracket only igemm::kernel launches
        os.environ["AVVAD_SKIP_ZERO"] = "1"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.avvad_conv2d_fwd(L.ptr(x), L.ptr(wf), L.ptr(y), C.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        del os.environ["AVVAD_SKIP_ZERO"]
        ms = e0.elapsed_time(e1) / reps
        flop = 2.0 * n_frames * ho * ho * co * ks * ks * c
        per.append((c, co, h, ks, stride, ms, flop / ms / 1e9))
        tot_flop += flop
        tot_ms += ms
        n_launch += 1
    ach = tot_flop / tot_ms / 1e9
    return {"bound": "mfma", "kernel": "igemm::kernel<128,128,true,512,Im2colFwd,ColTapRows,EpiStore> (trunk conv forward, Cout>=128)",
            "launches_per_step": n_launch, "avg_launch_us": round(1e3 * tot_ms / n_launch, 2),
            "achieved": round(ach, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_TFLOPS, 4),
            "traffic": TRAFFIC_PER_LAUNCH_BYTES, "traffic_unit": "bytes/launch (HBM side, rocprofv3 PMC, profiles/r01_pmc_*_size_per_kernel.csv)",
            "per_shape": [{"C": a, "Co": b, "HW": c_, "k": d_, "s": e, "us": round(1e3 * f, 1), "TFLOPs": round(g, 1)}
                          for (a, b, c_, d_, e, f, g) in per]}


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


def parity_probe(torch):
    """CPU-reference max|delta| of the logits on a small ragged AV batch (the second half of the metric)."""
    from oracle import models
    from packages.models.AV_Net import DeepVAD_AV
    cfg = dict(W0, en_pool_kernel_size=4)
    torch.manual_seed(1)
    m = DeepVAD_AV(2, 64, 1, wavenet_params=cfg)
    wave = torch.randn(3, 1, 4 * HOP + RF - 1) * 0.3
    video = torch.randn(3, 4, H_IMG, H_IMG)
    lens = [4, 2, 3]
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ref = models.av_net(sd, wave, video, lens, 2, training=True, wavenet_cfg=cfg)
    y = m.to("cuda").train()(wave.cuda(), video.cuda(), lens)
    return float((y.detach().cpu() - ref).abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline / parity probes")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from avvad import dist as avd
    from avvad.optim import FlatAdam
    from packages.models.AV_Net import DeepVAD_AV
    from packages.models.utils import batch_binary_cross_entropy

    rank, world, local = avd.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(0)                       # identical initial weights on every rank
    model = DeepVAD_AV(2, 1024, 1, use_mcb=False, eps=1e-8, wavenet_params=W0).to(dev).train()
    wave, video, target, lengths = make_inputs(torch, N_SEQ, 1234 + rank, dev)
    opt = FlatAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.999))
    reducer = avd.BucketReducer(opt.params, opt.flat_grad, opt.offsets)

    def step():
        y = model(wave, video, lengths)
        loss = batch_binary_cross_entropy(y, target, lengths, 1e-8)
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
    final_loss = float(loss.item())
    log("timed region: %.3f s for %d steps" % (dt, args.steps))

    if rank == 0:
        fp_per_step = world * N_SEQ * T_FRAMES
        out = {"metric": "AV frame-pairs/sec (fwd+bwd)", "value": round(fp_per_step * args.steps / dt, 1),
               "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "AV_net fused training step (WaveNet-W0 encoder + ResNet-18 trunk + concat + 2xLSTM1024 + FC, "
                                      "masked BCE, backward, RCCL all-reduce, fused Adam): BASELINE configs[3]/[4] per-GPU shard",
                          "per_gpu_frame_pairs": N_SEQ * T_FRAMES, "global_frame_pairs": fp_per_step, "sequences_per_gpu": N_SEQ,
                          "frames_per_sequence": T_FRAMES, "samples_per_sequence": T_FRAMES * HOP + RF - 1,
                          "parallelism": "dp%d" % world, "final_loss": round(final_loss, 4)}}
        if not args.no_extras:
            out["roofline"] = roofline_probe(torch, N_SEQ * T_FRAMES)
            log("roofline probe done")
            if world == 1:     # CPU baseline and CPU-reference delta: rank 0 at N=1 only
                out["cpu_ref_max_abs_delta"] = parity_probe(torch)
                log("parity probe done")
                out["cpu_baseline"] = cpu_baseline(torch)
                log("cpu baseline done")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
