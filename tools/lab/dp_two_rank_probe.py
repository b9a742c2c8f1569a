"""tests/test_gpu_parity.py::test_two_rank_gpu_data_parallel_step with a per-parameter breakdown of (all-reduced shards - whole
batch): which tensors differ when the pair disagrees.  Environment (AVVAD_MAX_CUS, AVVAD_NO_CONV64, ...) is inherited by the
workers.   python tools/lab/dp_two_rank_probe.py [overlap 0|1]"""
import os, sys, socket, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-vad_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dp_gpu_case as case
from avvad import _lib as L, ops
from avvad.optim import FlatAdam
from packages.models.utils import batch_binary_cross_entropy
overlap = sys.argv[1] if len(sys.argv) > 1 else "1"
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
out = os.path.join(tempfile.mkdtemp(), "flat.pt")
env = dict(os.environ, AVVAD_DIST_BACKEND="gloo", AVVAD_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", AVVAD_NO_STREAMK="all", AVVAD_OVERLAP=overlap)
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
       os.path.join(ROOT, "tests", "dp_gpu_worker.py"), out]
r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
assert r.returncode == 0, r.stderr[-2000:]
got = torch.load(out, weights_only=True)["step1"]
DEV = torch.device("cuda", 0)
L.set_option("no_streamk", 1)
ops._OVERLAP = overlap == "1"
model = case.make_model().to(DEV).eval()
wave, video, target, lengths = [t.to(DEV) for t in case.make_batch()]
opt = FlatAdam(model.parameters(), lr=1e-3)
for step in range(2):
    loss = batch_binary_cross_entropy(model(wave, video, lengths), target, lengths, 1e-8)
    loss.backward()
    if step == 0:
        opt.flat.add_(opt.flat_grad, alpha=-case.SGD_LR); opt.zero_grad()
ref = opt.flat_grad.detach().cpu()
print("total relL2 %.2e" % float((got - ref).norm() / ref.norm()))
names = [n for n, p in model.named_parameters() if p.requires_grad]
rows = []
for n, p, off in zip(names, opt.params, opt.offsets):
    a, b = got[off:off + p.numel()], ref[off:off + p.numel()]
    rows.append((float((a - b).norm() / b.norm().clamp_min(1e-30)), float((a - b).abs().max()), float(b.abs().max()), n))
for r_ in sorted(rows, reverse=True)[:12]:
    print("%.2e  max|d| %.2e  max|ref| %.2e  %s" % r_)
