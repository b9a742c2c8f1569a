"""Lab: run the same AV training step(s) N times in one process and count distinct outcomes (forward logits, gradients)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "audio-visual-vad_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from avvad.optim import FlatAdam
from packages.models.utils import batch_binary_cross_entropy
import dp_gpu_case as case
dev = torch.device("cuda", 0)

def run(steps, churn):
    if churn:   # perturb the allocator state / timing between runs
        torch.cuda.empty_cache()
        junk = [torch.empty(1 << (10 + i % 14), device=dev) for i in range(churn)]
        del junk
    m = case.make_model().to(dev).train()
    wave, video, target, lengths = [t.to(dev) for t in case.make_batch()]
    opt = FlatAdam(m.parameters(), lr=1e-3)
    for step in range(steps):
        y = m(wave, video, lengths)
        loss = batch_binary_cross_entropy(y, target, lengths, 1e-8)
        loss.backward()
        if step + 1 < steps:
            opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    return y.detach().clone(), opt.flat_grad.detach().clone(), list(opt.offsets), names

steps = int(os.environ.get("PROBE_STEPS", "3"))
y0, g0, off, names = run(steps, 0)
for i in range(int(os.environ.get("PROBE_RUNS", "10"))):
    y, g, _, _ = run(steps, i * 7)
    bad = [n for j, n in enumerate(names) if not torch.equal(g[off[j]:off[j+1]], g0[off[j]:off[j+1]])]
    ok = [n for j, n in enumerate(names) if torch.equal(g[off[j]:off[j+1]], g0[off[j]:off[j+1]])]
    print("run", i, "logits equal", torch.equal(y, y0), "max|dy|", float((y - y0).abs().max()), "differing grads", len(bad), "equal:", ok if len(ok) < 6 else len(ok), bad[:3], flush=True)
