"""Adam for the replicated 0N-GCN parameters on top of geom_adam_step_f32: up to 64 parameter tensors per
launch, step counter on the device and advanced inside the kernel (HIP-graph replayable, no tick launch).  Same
update as torch.optim.Adam(lr, betas, eps) without weight decay / amsgrad (what GEOMetrics.py:73 uses).  Any number
of tensors: they are issued in chunks of 64 that all use the bias corrections of the same step -- only the last
chunk advances the state."""
import ctypes

import torch

from . import _lib


class _InBackward:
    def __init__(self, opt):
        self.opt = opt

    def __enter__(self):
        from . import layers
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("FusedAdam.in_backward() would step on the LOCAL gradients, before the all-reduce across the "
                               "%d ranks; data-parallel steps call step(bucket.views, grad_scale=1/world) after the exchange"
                               % dist.get_world_size())
        self.prev = layers._backward_optimizer
        layers._backward_optimizer = self.opt
        self.opt._stepped_in_backward = False
        return self.opt

    def __exit__(self, *exc):
        from . import layers
        layers._backward_optimizer = self.prev
        return False


class FusedAdam:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise RuntimeError("FusedAdam got no trainable parameters")
        for p in self.params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("FusedAdam needs contiguous fp32 parameters on a HIP device")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.state = torch.zeros(_lib.ADAM_STATE_WORDS, dtype=torch.float32, device=self.params[0].device)

    @staticmethod
    def _ptrs(tensors):
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    @property
    def step_count(self):
        """Steps taken so far (one host read)."""
        return int(self.state[0].item())

    def in_backward(self):
        """Context manager around forward + backward of ONE iteration: the step is applied INSIDE the backward pass, in the
        launch that finishes the gradients (layers' end-of-pass reduction, geom_dense_reduce_adam_f32) -- no optimiser
        launch of its own, no second read of the gradients.  It happens only if (a) parameter-gradient deferral is on
        (`layers.deferred_parameter_gradients()`) and (b) that one launch finishes the gradient of EVERY parameter of this
        optimiser, each exactly once; the following `step()` (no arguments, grad_scale 1) is then a no-op.  Otherwise
        nothing changes and `step()` does the work.  For `zero_grad(); backward(); step()` loops -- not for gradient
        accumulation over several passes, and not for data-parallel steps (they reduce the gradients across ranks between
        backward and step)."""
        return _InBackward(self)

    def zero_grad(self):
        self._stepped_in_backward = False       # a new iteration: an in-backward step nobody consumed must not swallow a later step()
        for p in self.params:
            p.grad = None

    def step(self, grads=None, grad_scale=1.0):
        """grads: tensors to read instead of p.grad (e.g. views of an all-reduced flat bucket).  A parameter whose
        gradient is None is left untouched, as torch.optim.Adam does (the reference block's bn14 is never used)."""
        if grads is None:
            if getattr(self, "_stepped_in_backward", False) and grad_scale == 1.0:
                self._stepped_in_backward = False       # the backward pass's reduction launch has applied this step
                return
        if getattr(self, "_stepped_in_backward", False):
            # the pass already applied a step with the parameters' own gradients and scale 1: a second, different step on top
            # of it is never what the caller meant
            self._stepped_in_backward = False
            raise RuntimeError("FusedAdam.step(grads=..., grad_scale=...) after a backward pass that already applied the step "
                               "(in_backward()): this iteration would be stepped twice")
        if grads is None:
            grads = [p.grad for p in self.params]
        live = [i for i, g in enumerate(grads) if g is not None]
        if not live:
            return
        grads = [None if g is None else g.contiguous() for g in grads]
        m = _lib.ADAM_MAX_TENSORS
        with torch.cuda.device(self.params[0].device):
            for c0 in range(0, len(live), m):
                chunk = live[c0:c0 + m]
                pick = lambda seq: self._ptrs([seq[i] for i in chunk])
                sizes = (ctypes.c_int64 * len(chunk))(*[self.params[i].numel() for i in chunk])
                _lib.call("geom_adam_step_f32", len(chunk), pick([p.data for p in self.params]), pick(grads),
                          pick(self.exp_avg), pick(self.exp_avg_sq), sizes, float(self.lr), float(self.betas[0]),
                          float(self.betas[1]), float(self.eps), float(grad_scale), self.state.data_ptr(),
                          int(c0 + m >= len(live)))       # only the last chunk advances the device-side step state
