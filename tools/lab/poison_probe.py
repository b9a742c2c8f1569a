"""Lab: does any kernel's result depend on what the caching allocator hands out (stale workspace contents, block
addresses)?  Runs one AV forward+backward per memory state and reports the first intermediate that differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "audio-visual-vad_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from avvad import ops, nn as avnn
from avvad.optim import FlatAdam
from packages.models.utils import batch_binary_cross_entropy
import dp_gpu_case as case
dev = torch.device("cuda", 0)
OVERLAP = os.environ.get("PROBE_OVERLAP", "1") == "1"

def run(poison, steps=1):
    if poison is not None:
        torch.cuda.empty_cache()
        big = torch.full((3 << 28,), poison, device=dev)
        del big
    m = case.make_model().to(dev).train()
    wave, video, target, lengths = [t.to(dev) for t in case.make_batch()]
    opt = FlatAdam(m.parameters(), lr=1e-3)
    cap = {}
    for step in range(steps):
        main, side = torch.cuda.current_stream(), ops.side_stream()
        if OVERLAP:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                audio = ops.TransposeLast2Fn.apply(m.wavenet_en(wave))
            feats = avnn.video_features(m.features, video, True)
            main.wait_stream(side)
            audio.record_stream(main)
        else:
            feats = avnn.video_features(m.features, video, True)
            audio = ops.TransposeLast2Fn.apply(m.wavenet_en(wave))
        cat = ops.ConcatColsFn.apply(audio, feats)
        h = ops.lstm_stack(cat, lengths, m.lstm_merged)
        y = ops.LinearFn.apply(h, m.vad_merged.weight, m.vad_merged.bias)
        for name, t in (("d_audio", audio), ("d_feats", feats), ("d_cat", cat), ("d_h", h), ("d_y", y)):
            t.register_hook(lambda g, name=name: cap.__setitem__(name, g.detach().clone()))
        cap["audio"], cap["feats"], cap["h"], cap["y"] = audio.detach().clone(), feats.detach().clone(), h.detach().clone(), y.detach().clone()
        loss = batch_binary_cross_entropy(y, target, lengths, 1e-8)
        loss.backward()
        if step + 1 < steps:
            opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        if p.grad is not None:
            cap["g:" + n] = p.grad.detach().clone()
    return cap

steps = int(os.environ.get("PROBE_STEPS", "1"))
c0 = run(None, steps)
order = ["audio", "feats", "h", "y", "d_y", "d_h", "d_cat", "d_feats", "d_audio"]
for tag, poison in (("zeros", 0.0), ("nan", float("nan")), ("big", 1e30), ("none", None), ("nan2", float("nan"))):
    c = run(poison, steps)
    bad = [k for k in order + sorted(k for k in c0 if k.startswith("g:")) if not torch.equal(c[k], c0[k])]
    print(tag, "differing:", len(bad), bad[:12], flush=True)
