"""Data-parallel sharding of independent meshes: one process per GPU, RCCL over xGMI
(torch.distributed backend "nccl" on ROCm; "gloo" for the CPU tests).

The hot path has no data-path exchange: sampling, both arg-min scans and the losses are
per-mesh.  The only collectives are (a) one all-reduce of a <=32-byte loss/metric vector per
step and (b) one bucketed all-reduce of the (replicated) 0N-GCN parameter gradients --
1.04 MB for the 963-192-192-192 stack, latency-bound on xGMI (7 links x ~153 GB/s per GPU).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, world, local_rank) from the torchrun environment; single process when unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # GEOM_DIST_BACKEND is a test hook (e.g. two ranks sharing one GPU over gloo); production = RCCL
            backend = os.environ.get("GEOM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total, rank, world):
    """Contiguous [first, first+count) block of `total` meshes owned by `rank` (config 5: 64 -> 8x8).
    Remainders go to the lowest ranks."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def all_reduce_sum_(t, force=False):
    """In-place SUM over the ranks.  force: issue the collective even in a 1-rank group (how the test suite executes RCCL
    itself -- backend "nccl" -- on a single GPU: tests/test_dist_step_gpu.py)."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def global_mean_loss(local_loss_sum, local_count):
    """Mean over ALL meshes of per-mesh losses: all-reduce [sum, count] (8 bytes)."""
    total = local_loss_sum.reshape(()).float()
    vec = torch.stack([total, total.new_full((), float(local_count))])   # no host->device copy: graph-capturable
    all_reduce_sum_(vec)
    return vec[0] / vec[1]


class GradBucket:
    """One flat fp32 buffer for the DP exchange: every parameter gradient plus `extra` trailing scalars
    (e.g. the shard's loss), so a step needs exactly ONE all-reduce.  Autograd leaves a fresh
    .grad on every parameter (no accumulate kernels); `pack()` gathers them with one launch (bind=True: the layers' launches
    write them in place and pack() has nothing to do),
    `all_reduce()` sums the buffer across ranks, `views` are per-parameter views of the reduced buffer
    (the 1/world scale is applied inside the optimiser kernel)."""

    def __init__(self, params, extra=0, force_collective=False, bind=False):
        self.force = force_collective
        self.params = [p for p in params if p.requires_grad]
        ref = self.params[0]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel + extra, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.extra = self.flat[self.numel:]
        # bind=True: the layers' gradient launches write straight into the views (layers.bind_gradient_targets), so that
        # pack() only has to place the trailing scalars and whatever gradient did not come from those launches
        self.bound = bool(bind)
        if bind:
            from . import layers
            layers.bind_gradient_targets(self.params, self.views)

    def pack(self, *scalars):
        """Gather into the flat buffer whatever is not there yet.  Returns True when that took a launch (an unbound bucket
        always; a bound one only for gradients that did not land in their views, missing gradients, or scalars) -- a
        data-parallel step that orders its collective behind an event recorded EARLIER must then wait for these copies too."""
        if not self.bound:
            parts = [p.grad.reshape(-1) for p in self.params] + [s.reshape(1).to(self.flat.dtype) for s in scalars]
            torch.cat(parts, out=self.flat)
            return True
        launched = False
        for p, view in zip(self.params, self.views):
            if p.grad is None:
                view.zero_()
                launched = True
            elif p.grad.data_ptr() != view.data_ptr():       # produced elsewhere (a library fallback): gather it
                view.copy_(p.grad)
                launched = True
        if scalars:
            torch.stack([s.reshape(()).to(self.flat.dtype) for s in scalars], out=self.extra[:len(scalars)])
            launched = True
        return launched

    def all_reduce(self):
        all_reduce_sum_(self.flat, self.force)
        return self.views

    def all_reduce_async(self):
        """Start the all-reduce behind everything queued on the CURRENT stream so far and return at once: work queued on
        this stream afterwards runs BESIDE the collective (RCCL has its own stream); `wait()` on the returned handle orders
        the current stream behind the collective (None: nothing was issued -- a single process)."""
        if dist.is_initialized() and (dist.get_world_size() > 1 or self.force):
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
        return None

    def pack_all_reduce(self, *scalars):
        self.pack(*scalars)
        return self.all_reduce()


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
