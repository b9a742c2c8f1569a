"""Worker of tests/test_gpu_parity.py::test_two_rank_gpu_data_parallel_step (launched by torch.distributed.run).
Both ranks sit on cuda:0 and exchange gradients through gloo (RCCL refuses two ranks per device): everything else
-- the HIP kernels, in-place gradient sinks, the two HIP streams, bucket launches from hooks -- is the production path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "audio-visual-vad_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from avvad import dist as avd  # noqa: E402
from avvad.optim import FlatAdam  # noqa: E402
from packages.models.AV_Net import DeepVAD_AV  # noqa: E402
from packages.models.utils import batch_binary_cross_entropy  # noqa: E402
import dp_gpu_case as case  # noqa: E402


def main():
    out = sys.argv[1]
    rank, world, local = avd.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    model = case.make_model().to(dev).eval()            # eval: BatchNorm uses running statistics, so shards are independent
    wave, video, target, lengths = [t.to(dev) for t in case.make_batch()]
    opt = FlatAdam(model.parameters(), lr=1e-3)
    red = avd.BucketReducer(opt.params, opt.flat_grad, opt.offsets, bucket_bytes=1 << 20,   # several buckets, cut at sub-modules
                            names=[n_ for n_, p_ in model.named_parameters() if p_.requires_grad], min_group_bytes=1 << 12)
    lengths_s, wave_s, video_s, target_s = avd.shard_batch([lengths, wave, video, target], rank, world)
    first = None
    for step in range(2):                                # two steps: the reducer's per-step bookkeeping is reset in between
        y = model(wave_s, video_s, lengths_s)
        loss = batch_binary_cross_entropy(y, target_s, lengths_s, 1e-8)
        loss.backward()
        red.finish()
        if step == 0:
            torch.cuda.synchronize()
            first = opt.flat_grad.detach().cpu().clone()
            opt.flat.add_(opt.flat_grad, alpha=-case.SGD_LR)      # plain SGD (Adam would amplify 1e-7 differences to +-lr)
            opt.zero_grad()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"step0": first, "step1": opt.flat_grad.detach().cpu()}, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
