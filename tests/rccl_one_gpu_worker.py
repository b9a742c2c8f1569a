"""Worker of tests/test_gpu_parity.py::test_rccl_path_on_one_gpu (launched by torch.distributed.run as a fresh child, one
rank, backend nccl = RCCL).  Runs two training steps of the AV model twice from the same state: once with a
BucketReducer(force_hooks=True) -- hooks, in-place gradient sinks, async all_reduce on bucket slices of the flat CUDA
buffer, the side-stream ordering, the presence-bitmap collective all execute on RCCL -- and once without a reducer.  At
world size 1 the SUM all-reduce is the identity, so the two flat gradients must agree bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "audio-visual-vad_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from avvad import dist as avd  # noqa: E402
from avvad.optim import FlatAdam  # noqa: E402
from packages.models.utils import batch_binary_cross_entropy  # noqa: E402
import dp_gpu_case as case  # noqa: E402


def run(use_reducer, dev):
    model = case.make_model().to(dev).train()
    wave, video, target, lengths = [t.to(dev) for t in case.make_batch()]
    opt = FlatAdam(model.parameters(), lr=1e-3)
    red = None
    if use_reducer:
        red = avd.BucketReducer(opt.params, opt.flat_grad, opt.offsets, bucket_bytes=1 << 20, force_hooks=True,
                                names=[n_ for n_, p_ in model.named_parameters() if p_.requires_grad], min_group_bytes=1 << 12)
        assert red.active and len(red.buckets) > 2
    launched_from_hooks = 0
    for step in range(3):                                # step 2 runs with the agreed-absent set (the unused `bn`)
        loss = batch_binary_cross_entropy(model(wave, video, lengths), target, lengths, 1e-8)
        loss.backward()
        if red is not None:
            launched_from_hooks = sum(red.launched)
            launched_list = list(red.launched)
            red.finish()
        if step < 2:
            opt.step()
            opt.zero_grad()
    torch.cuda.synchronize()
    names = [n_ for n_, p_ in model.named_parameters() if p_.requires_grad]
    info = {"names": names, "offsets": list(opt.offsets)}
    if red is not None:
        info.update({"buckets": len(red.buckets), "launched_from_hooks_last_step": launched_from_hooks,
                     "absent": sorted(red.absent), "bucket_ranges": [list(b) for b in red.buckets],
                     "not_launched": [i for i, l in enumerate(launched_list) if not l]})
        red.close()
    return opt.flat_grad.detach().clone(), float(loss.detach()), info


def main():
    out = sys.argv[1]
    # (the process group is created before anything touches the GPU: this worker is a fresh child of the test)
    dist.init_process_group(backend="nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g0, l0, _ = run(False, dev)
    g1, l1, info = run(True, dev)
    g2, l2, _ = run(False, dev)
    # diagnostics: which parameters' gradients differ between the runs (none expected)
    diff = {}
    for tag, a, b in (("with_vs_without", g1, g0), ("without_vs_without", g2, g0)):
        bad = []
        for i, n_ in enumerate(info["names"]):
            sl = slice(info["offsets"][i], info["offsets"][i + 1])
            if not torch.equal(a[sl], b[sl]):
                bad.append((n_, float((a[sl] - b[sl]).abs().max()), float(b[sl].abs().max())))
        diff[tag] = bad
    info["diff"] = diff
    torch.save({"with": g1.cpu(), "without": g0.cpu(), "loss": (l1, l0), "info": info}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
