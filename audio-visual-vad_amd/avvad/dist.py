"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce (SUM) of the flat gradient
buffer over RCCL/xGMI, launched as soon as a bucket's last gradient has been accumulated so the exchange
overlaps the rest of backward.

Replaces ``nn.parallel.DataParallel(model, device_ids=[0,1,2,3])`` (``scripts/train_AV_net.py:193``), which
broadcasts all parameters and reduces all gradients through GPU 0 every step.  The loss is a SUM over
sequences (``train_AV_net.py:298-302``), so a SUM all-reduce (no division by world size) reproduces the
single-process gradient of the global batch; BatchNorm statistics stay per replica, as under DataParallel.

The reducer only needs ``torch.distributed`` and flat views, so it runs unchanged on the gloo backend with
CPU tensors -- that is how the N>1 path is tested without GPUs."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the launcher (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal aids for a one-GPU box (RCCL refuses two ranks on one device): AVVAD_DIST_BACKEND=gloo moves the
    # all-reduce through the host, AVVAD_FORCE_DEVICE=0 puts every rank on that device.  Never set in production.
    backend = os.environ.get("AVVAD_DIST_BACKEND", backend)
    if "AVVAD_FORCE_DEVICE" in os.environ:
        local = int(os.environ["AVVAD_FORCE_DEVICE"])
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(tensors, rank, world):
    """Even split of the global batch on dim 0 (what DataParallel.scatter does)."""
    out = []
    for t in tensors:
        n = t.shape[0]
        if n % world:
            raise ValueError("global batch %d not divisible by world size %d" % (n, world))
        k = n // world
        out.append(t[rank * k:(rank + 1) * k])
    return out


def flat_views(params):
    """Re-home ``.grad`` of every trainable parameter into one flat buffer (CPU or GPU);
    returns (flat_grad, offsets).  FlatAdam does the same for GPU training."""
    params = [p for p in params if p.requires_grad]
    offsets = [0]
    for p in params:
        offsets.append(offsets[-1] + (p.numel() + 63) // 64 * 64)
    flat = torch.zeros(offsets[-1], dtype=params[0].dtype, device=params[0].device)
    for p, o in zip(params, offsets):
        p.grad = flat[o:o + p.numel()].view(p.shape)
    return flat, offsets


class BucketReducer:
    """params: trainable parameters whose ``.grad`` are views into ``flat_grad`` at ``offsets`` (elements).
    Buckets are contiguous ranges of the flat buffer of about ``bucket_bytes`` (xGMI is point-to-point:
    few large messages beat many small ones; 25 MB keeps a ring step well above the latency floor).

    The flat buffer is laid out in forward (registration) order and backward produces gradients in reverse, so buckets
    complete from the END of the buffer towards its start.  With ``names`` (``model.named_parameters()`` order) a bucket
    is also closed where the top-level sub-module changes (``features`` | ``wavenet_en`` | ``lstm_*`` | ``vad_*``) once
    it holds ``min_group_bytes``: the head's buckets then never wait for trunk gradients and overlap the whole trunk
    backward.  Parameters that received no gradient in the previous step (the reference's unused ``bn`` of
    ``AV_Net.py:33``, frozen sub-modules) are dropped from the readiness count, so their bucket still launches from the
    hooks instead of from ``finish()``."""

    def __init__(self, params, flat_grad, offsets, bucket_bytes=25 << 20, group=None, names=None, min_group_bytes=4 << 20):
        self.flat_grad = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = list(params)             # keep the objects alive: the maps below are keyed by id()
        self.buckets = []                      # (start, end, n_params)
        self.param_bucket = {}
        tops = [n.split(".")[0] for n in names] if names is not None else None
        start, count = 0, 0
        for i, p in enumerate(self.params):
            end = offsets[i + 1]
            count += 1
            self.param_bucket[id(p)] = len(self.buckets)
            size = (end - start) * 4
            last = i == len(self.params) - 1
            group_edge = tops is not None and not last and tops[i + 1] != tops[i] and size >= min_group_bytes
            if size >= bucket_bytes or group_edge or last:
                self.buckets.append((start, end, count))
                start, count = end, 0
        self.expected = [b[2] for b in self.buckets]    # parameters a bucket waits for (absent ones removed)
        self.absent = set()                    # id(p) of parameters that got no gradient last step
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.handles = []
        self._hooks = []
        self._names = dict(zip(map(id, self.params), names)) if names is not None else {}
        self._seen = set()
        self._sink = None
        self._trace = bool(os.environ.get("AVVAD_DP_TRACE"))      # debug aids, read once
        self._late = bool(os.environ.get("AVVAD_DP_LATE"))
        if self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            try:                      # gradients written in place by the HIP backward bypass autograd's hooks
                from . import ops
                self._sink = self._on_grad
                ops.GRAD_SINKS.append(self._sink)
            except ImportError:       # CPU-only use of the reducer (gloo tests)
                pass

    def close(self):
        """Detach from autograd and from the HIP backward's gradient sinks (a process may build several reducers)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._sink is not None:
            from . import ops
            if self._sink in ops.GRAD_SINKS:
                ops.GRAD_SINKS.remove(self._sink)
            self._sink = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _on_grad(self, p):
        """A parameter's gradient for this step is complete (all kernels that write it are enqueued).  Called by
        autograd's post-accumulate hook and/or by the HIP backward's in-place sinks: a parameter written in place can be
        announced by BOTH (observed on torch 2.10: the hook also fires for the ``None`` the Function returns), so
        repeats within a step are dropped -- counting them launched a bucket's all-reduce before its last gradients
        existed (caught by tests/test_gpu_parity.py::test_two_rank_gpu_data_parallel_step).  Assumes what holds for these
        models: every parameter is written by exactly one backward Function per step (no gradient accumulation over
        several backward() calls between two finish() calls: that raises below instead of reducing too early)."""
        if id(p) not in self.param_bucket or id(p) in self._seen:
            return
        self._seen.add(id(p))
        b = self.param_bucket[id(p)]
        if id(p) in self.absent:               # came back (e.g. unfrozen): wait for it again from now on
            self.absent.discard(id(p))
            self.expected[b] += 1
        if self.launched[b]:
            raise RuntimeError("BucketReducer: gradient of %s announced after its bucket was all-reduced (a parameter that "
                               "had no gradient in the previous step got one now, or backward() ran twice before finish())"
                               % self._names.get(id(p), tuple(p.shape)))
        self.pending[b] += 1
        if self._trace:
            print("[dp] grad ready: param %s shape %s -> bucket %d pending %d/%d" % (self._names.get(id(p), "?"), tuple(p.shape), b, self.pending[b], self.expected[b]), flush=True)
        if self.pending[b] == self.expected[b] and not self._late:
            self._launch(b)

    def _launch(self, b):
        if self.launched[b]:
            return
        s, e, _ = self.buckets[b]
        self.launched[b] = True
        if self.flat_grad.is_cuda:
            # A bucket can hold gradients written on different HIP streams (the audio encoder's backward runs on the
            # side stream, ops.side_stream()); the collective is ordered after the CURRENT stream only, so make that
            # stream wait for everything the others have been given so far (all of this bucket's writers are enqueued).
            from . import ops
            cur = torch.cuda.current_stream()
            for st in ops.side_streams() + [torch.cuda.default_stream()]:
                if st != cur:
                    cur.wait_stream(st)
        self.handles.append(dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Call after backward: launches whatever did not fire (parameters without a gradient this step)
        and waits for every bucket."""
        if self.world > 1:
            for b in range(len(self.buckets)):
                self._launch(b)
            for h in self.handles:
                h.wait()
            # parameters nobody announced this step stop counting towards their bucket's readiness
            for p in self.params:
                if id(p) not in self._seen and id(p) not in self.absent:
                    self.absent.add(id(p))
                    self.expected[self.param_bucket[id(p)]] -= 1
        self.handles = []
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self._seen = set()
