"""The two-rank test's model in ONE process: gradient of the whole batch vs the sum over its two shards, first and second step
(SGD update in between), per parameter.  Run with AVVAD_MAX_CUS=240 AVVAD_NO_STREAMK=all to see what the cap changes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-vad_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dp_gpu_case as case
from packages.models.utils import batch_binary_cross_entropy
DEV = torch.device("cuda", 0)
wave, video, target, lengths = [t.to(DEV) for t in case.make_batch()]

def grads(model, sl):
    for p in model.parameters():
        p.grad = None
    loss = batch_binary_cross_entropy(model(wave[sl], video[sl], lengths[sl].cpu()), target[sl], lengths[sl].cpu(), 1e-8)
    loss.backward()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

model = case.make_model().to(DEV).eval()
for step in range(2):
    gw = grads(model, slice(0, 4)); g0 = grads(model, slice(0, 2)); g1 = grads(model, slice(2, 4))
    rels = sorted(((float(((g0[n] + g1[n]) - gw[n]).norm() / gw[n].norm().clamp_min(1e-30)), n) for n in gw), reverse=True)
    tot = float(torch.cat([((g0[n] + g1[n]) - gw[n]).flatten() for n in gw]).norm() / torch.cat([gw[n].flatten() for n in gw]).norm())
    print("step %d: total relL2 %.2e; worst %s" % (step, tot, [(("%.1e" % r), n) for r, n in rels[:5]]), flush=True)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n in gw:
                p.add_(gw[n], alpha=-case.SGD_LR)
