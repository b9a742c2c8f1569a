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
    backward.

    Parameters without a gradient (the reference's unused ``bn`` of ``AV_Net.py:33``, frozen sub-modules) must not hold
    their bucket back until ``finish()``.  Which parameters those are is AGREED between the ranks, never decided
    locally: every ``finish()`` all-reduces a presence bitmap (one float per parameter, asynchronously; it is read at
    the next ``finish()``, a whole step later, so no step waits for it), and a parameter that NO rank announced stops
    counting towards its bucket's readiness from the step after.  Such "absent" parameters' slices are cut out of their
    bucket's collective and reduced by ``finish()`` in collectives of their own -- identical on every rank, after every
    possible writer -- so one that gets a gradient again on some rank
    only (a data-dependent branch, an unfrozen module) is still summed correctly and nobody waits in a collective the
    others never enter.  Rules the caller keeps: ONE ``backward()`` per ``finish()`` (no gradient accumulation over
    several backward passes: a gradient announced after its bucket went out is reported by every rank at the next
    ``finish()``), and the flat gradient is zeroed between steps (``FlatAdam.zero_grad``).

    ``force_hooks``: register the hooks / sinks and run the collectives even at world size 1 (a one-GPU box can then
    execute the RCCL path end to end: ``tests/test_gpu_parity.py::test_rccl_path_on_one_gpu``)."""

    def __init__(self, params, flat_grad, offsets, bucket_bytes=25 << 20, group=None, names=None, min_group_bytes=4 << 20,
                 force_hooks=False):
        self.flat_grad = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force_hooks and dist.is_initialized())
        self.params = list(params)             # keep the objects alive: the maps below are keyed by id()
        self.offsets = list(offsets)
        self.index = {id(p): i for i, p in enumerate(self.params)}
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
        self.expected = [b[2] for b in self.buckets]    # parameters a bucket waits for (agreed-absent ones removed)
        self.absent = set()                    # parameter indices no rank announced in the last agreed step
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.handles = []
        self._hooks = []
        self._names = dict(zip(map(id, self.params), names)) if names is not None else {}
        self._seen = set()
        self._error = ""                       # a protocol violation seen on THIS rank in the current step
        self._meta = None                      # (handle, device tensor, host tensor, event) of the previous finish()
        self._sink = None
        self._trace = bool(os.environ.get("AVVAD_DP_TRACE"))      # debug aids, read once
        self._late = bool(os.environ.get("AVVAD_DP_LATE"))
        if self.active:
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
        existed (caught by tests/test_gpu_parity.py::test_two_rank_gpu_data_parallel_step).  Never raises (an exception
        inside an autograd hook on ONE rank would leave its peers waiting in a collective): violations are recorded and
        reported by every rank together at the next finish()."""
        if id(p) not in self.param_bucket or id(p) in self._seen:
            return
        self._seen.add(id(p))
        if self.index[id(p)] in self.absent:   # agreed absent: reduced on its own by finish(), whoever has a gradient
            return
        b = self.param_bucket[id(p)]
        if self.launched[b]:
            self._error = ("gradient of %s announced after its bucket was all-reduced (backward() ran twice before finish()?)"
                           % (self._names.get(id(p), tuple(p.shape)),))
            return
        self.pending[b] += 1
        if self._trace:
            print("[dp] grad ready: param %s shape %s -> bucket %d pending %d/%d" % (self._names.get(id(p), "?"), tuple(p.shape), b, self.pending[b], self.expected[b]), flush=True)
        if self.pending[b] == self.expected[b] and not self._late:
            self._launch(b)

    def _join_streams(self):
        """A bucket can hold gradients written on different HIP streams (the audio encoder's backward runs on the side
        stream, ops.side_stream()); a collective is ordered after the CURRENT stream only, so make that stream wait for
        everything the others have been given so far (all of the slice's writers are enqueued by now)."""
        if self.flat_grad.is_cuda:
            from . import ops
            cur = torch.cuda.current_stream()
            for st in ops.side_streams() + [torch.cuda.default_stream()]:
                if st != cur:
                    cur.wait_stream(st)

    def _reduce(self, s, e):
        self.handles.append(dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        self._join_streams()
        for s, e in self._segments(b):
            self._reduce(s, e)

    def _segments(self, b):
        """Bucket b's range of the flat buffer WITHOUT the slices of agreed-absent parameters (finish() reduces those on
        their own, after every possible writer; reduced here as well they would be summed twice)."""
        s, e, _ = self.buckets[b]
        if not self.absent:
            return [(s, e)]
        segs, cur = [], s
        for i in sorted(self.absent):
            if self.param_bucket[id(self.params[i])] != b:
                continue
            if self.offsets[i] > cur:
                segs.append((cur, self.offsets[i]))
            cur = self.offsets[i + 1]
        if e > cur:
            segs.append((cur, e))
        return segs

    def _agree(self):
        """Read the presence bitmap the PREVIOUS finish() all-reduced: the same numbers on every rank."""
        if self._meta is None:
            return
        handle, dev_t, host_t, evt = self._meta
        self._meta = None
        handle.wait()
        if evt is not None:
            evt.synchronize()
        counts = host_t if host_t is not None else dev_t
        if float(counts[-1]) > 0:
            raise RuntimeError("BucketReducer: a rank reported a protocol violation in the previous step (%s)"
                               % (getattr(self, "_last_error", "") or "on another rank: see its log"))
        absent = {i for i in range(len(self.params)) if float(counts[i]) == 0.0}
        if absent != self.absent:
            self.absent = absent
            self.expected = [b[2] for b in self.buckets]
            for i in absent:
                self.expected[self.param_bucket[id(self.params[i])]] -= 1

    def finish(self):
        """Call after backward: launches whatever did not fire, reduces the agreed-absent parameters' slices, waits for
        every collective and sends this step's presence bitmap on its way."""
        if self.active:
            stale_error = self._error
            for b in range(len(self.buckets)):
                self._launch(b)
            # agreed-absent parameters: their own collectives (contiguous runs coalesced), the same on every rank
            run = None
            for i in sorted(self.absent) + [None]:
                if run is not None and (i is None or i != run[1] + 1):
                    self._join_streams()
                    self._reduce(self.offsets[run[0]], self.offsets[run[1] + 1])
                    run = None
                if i is not None:
                    run = (i, i) if run is None else (run[0], i)
            for h in self.handles:
                h.wait()
            self._agree()                      # (the previous step's bitmap: long complete)
            meta = torch.zeros(len(self.params) + 1, dtype=torch.float32)
            for p in self.params:
                if id(p) in self._seen:
                    meta[self.index[id(p)]] = 1.0
            meta[-1] = 1.0 if stale_error else 0.0
            self._last_error, self._error = stale_error, ""
            dev_t = meta.to(self.flat_grad.device)
            handle = dist.all_reduce(dev_t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            host_t = evt = None
            if dev_t.is_cuda:
                handle.wait()                  # stream-orders the copy behind the collective; the host does not block
                host_t = torch.empty(meta.shape, dtype=torch.float32, pin_memory=True)
                host_t.copy_(dev_t, non_blocking=True)
                evt = torch.cuda.Event()
                evt.record()
            self._meta = (handle, dev_t, host_t, evt)
        elif self._error:
            msg, self._error = self._error, ""
            raise RuntimeError("BucketReducer: " + msg)
        self.handles = []
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self._seen = set()
