"""Diagnostic: 2-rank DP gradient vs single process, under combinations of AVVAD_OVERLAP / AVVAD_DP_LATE."""
import os, subprocess, sys, socket, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "audio-visual-vad_amd"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import torch
os.environ["AVVAD_NO_STREAMK"] = "all"
import dp_gpu_case as case
from avvad.optim import FlatAdam
from packages.models.utils import batch_binary_cross_entropy

def single(overlap):
    os.environ["AVVAD_OVERLAP"] = overlap
    model = case.make_model().to("cuda").eval()
    wave, video, target, lengths = [t.to("cuda") for t in case.make_batch()]
    opt = FlatAdam(model.parameters(), lr=1e-3)
    batch_binary_cross_entropy(model(wave, video, lengths), target, lengths, 1e-8).backward()
    torch.cuda.synchronize()
    return opt.flat_grad.detach().cpu().clone(), opt.offsets, [n for n, _ in model.named_parameters()]

def two_rank(overlap, late):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    out = tempfile.mktemp(suffix=".pt")
    env = dict(os.environ, AVVAD_DIST_BACKEND="gloo", AVVAD_FORCE_DEVICE="0", AVVAD_OVERLAP=overlap)
    if late: env["AVVAD_DP_LATE"] = "1"
    else: env.pop("AVVAD_DP_LATE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(R, "tests", "dp_gpu_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(out, weights_only=True)

s0, offs, names = single("0")
s1, _, _ = single("1")
print("single overlap0 vs overlap1 relL2 %.2e" % float((s0 - s1).norm() / s1.norm()))
for ov in ("0", "1"):
    for late in (False, True):
        g = two_rank(ov, late)
        ref = s0
        rel = float((g - ref).norm() / ref.norm())
        print("two-rank overlap=%s late=%d : relL2 vs single %.2e" % (ov, late, rel))
        if rel > 1e-4:
            worst = []
            for i, n in enumerate(names):
                a, b = g[offs[i]:offs[i + 1]], ref[offs[i]:offs[i + 1]]
                d = float((a - b).norm() / max(float(b.norm()), 1e-20))
                if d > 1e-4: worst.append((d, n))
            worst.sort(reverse=True)
            print("   differing params:", [(round(d, 4), n) for d, n in worst[:12]], "count", len(worst))
