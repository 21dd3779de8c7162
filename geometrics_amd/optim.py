"""Adam for the replicated 0N-GCN parameters on top of geom_adam_step_f32: every parameter tensor
in one launch, step counter on the device (HIP-graph replayable).  Same update as
torch.optim.Adam(lr, betas, eps) without weight decay / amsgrad (what GEOMetrics.py:73 uses)."""
import ctypes

import torch

from . import _lib


class FusedAdam:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if len(self.params) > 16:
            raise RuntimeError("FusedAdam handles at most 16 parameter tensors per group")
        for p in self.params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("FusedAdam needs contiguous fp32 parameters on a HIP device")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.state = torch.zeros(3, dtype=torch.float32, device=self.params[0].device)
        n = len(self.params)
        self._sizes = (ctypes.c_int64 * n)(*[p.numel() for p in self.params])
        self._ptr_array = ctypes.c_void_p * n

    def _ptrs(self, tensors):
        return self._ptr_array(*[t.data_ptr() for t in tensors])

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self, grads=None, grad_scale=1.0):
        """grads: tensors to read instead of p.grad (e.g. views of an all-reduced flat bucket)."""
        if grads is None:
            grads = [p.grad for p in self.params]
        grads = [g.contiguous() for g in grads]
        with torch.cuda.device(self.params[0].device):
            _lib.call("geom_adam_step_f32", len(self.params), self._ptrs([p.data for p in self.params]),
                      self._ptrs(grads), self._ptrs(self.exp_avg), self._ptrs(self.exp_avg_sq), self._sizes,
                      float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(grad_scale),
                      self.state.data_ptr())
