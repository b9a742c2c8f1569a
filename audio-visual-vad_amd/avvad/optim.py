"""Fused Adam over one flat fp32 buffer (csrc/misc.hip::adam_kernel).

Replaces ``torch.optim.Adam(model.parameters(), lr, betas=(0.9, 0.999))`` + ``step()`` + ``zero_grad()`` of
``scripts/train_AV_net.py:238,305-307``.  Layout for 288 GB of HBM: every trainable parameter becomes a view
into ONE contiguous buffer, every ``.grad`` a view into a second one, so the optimiser is a single kernel
launch and data-parallel training all-reduces a handful of large contiguous buckets (avvad.dist) instead of
one message per tensor."""
import ctypes as C

import torch

from . import _lib as L


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise L.AvvadError("FlatAdam needs GPU parameters (no CPU fallback)")
        sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for n in sizes:
            self.offsets.append(self.offsets[-1] + (n + 63) // 64 * 64)      # 256-byte aligned views
        total = self.offsets[-1]
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                self.flat[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                g = self.flat_grad[o:o + n].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g                       # autograd accumulates in place into the flat buffer
        self.lr, self.betas, self.eps = lr, betas, eps
        self.t = 0

    def _join(self):
        """Gradients may have been written on the side HIP stream (ops.side_stream()): order this stream behind it."""
        from . import ops
        cur = torch.cuda.current_stream()
        for st in ops.side_streams():
            if st != cur:
                cur.wait_stream(st)

    def step(self):
        self._join()
        self.t += 1
        L.check(L.lib().avvad_adam_step(L.ptr(self.flat), L.ptr(self.flat_grad), L.ptr(self.exp_avg),
                                        L.ptr(self.exp_avg_sq), self.flat.numel(), self.lr, self.betas[0],
                                        self.betas[1], self.eps, self.t,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), "avvad_adam_step")

    def zero_grad(self):
        self._join()
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):       # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
