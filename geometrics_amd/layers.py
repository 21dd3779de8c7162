"""0N-GCN layers with the reference's class names, constructor signatures, parameter names
and initialisers (reference layers.py:14-189), so `models.py` and pretrained state_dicts
(`gcN.weight1`, `gcN.weight`, `gcN.bias`, `weight_Ws.0`, `weight_Bs.0`) work unchanged.

`forward(input, adj, activation)` still receives the DENSE [V,V] adjacency the callers pass
(models.py:241, GEOMetrics.py:120).  The layer derives CSR and CSR^T from it once (cached on
the tensor's identity + version) and replaces the reference's dense `adj @ support[..., :k]`,
`torch.cat` and bias add by one fused HIP kernel (csrc/zn_gcn.hip); the dense feature GEMM
`input @ W` stays a library GEMM (rocBLAS/hipBLASLt through torch.matmul).
"""
import ctypes
import math
import os
import threading
import weakref

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import Module
from torch.nn.parameter import Parameter

from . import _lib
from . import dense as _dense_kernels
from . import fused as _fused

_ACT_NONE, _ACT_RELU, _ACT_ELU = 0, 1, 2


# --------------------------------------------------------------- CSR cache ----
class _Csr:
    __slots__ = ("rowptr", "col", "val", "rowptr_t", "col_t", "val_t", "nv", "nnz",
                 "ell_w", "ell_col", "ell_val", "ell_col_t", "ell_val_t", "inv_deg", "over", "over_t",
                 "_deform_tail")      # (geometrics_amd.deform: the rows' entries beyond the table as a second fixed-width table)


_csr_cache = {}   # id(adjacency tensor) -> (weakref, version, _Csr)


def _to_csr(dense):
    nz = dense.nonzero()                       # row-major order == CSR order
    rows, cols = nz[:, 0], nz[:, 1]
    counts = torch.bincount(rows, minlength=dense.shape[0])
    rowptr = torch.zeros(dense.shape[0] + 1, dtype=torch.int64, device=dense.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), cols.to(torch.int32).contiguous(), dense[rows, cols].to(torch.float32).contiguous()


def _to_ell(rowptr, col, val, width):
    """[V][width] neighbour table holding the first `width` entries of every CSR row (unused slots col = -1 /
    val = 0) + the CSR tail (over_ptr, over_col, over_val) of the rows that are longer, or None when none is."""
    nv = rowptr.numel() - 1
    lens = (rowptr[1:] - rowptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(nv, device=col.device), lens)
    slot = torch.arange(col.numel(), device=col.device) - rowptr[:-1].long()[rows]
    ell_col = torch.full((nv, width), -1, dtype=torch.int32, device=col.device)
    ell_val = torch.zeros((nv, width), dtype=torch.float32, device=col.device)
    head = slot < width
    ell_col[rows[head], slot[head]] = col[head]
    ell_val[rows[head], slot[head]] = val[head]
    over = None
    if not bool(head.all()):
        over_ptr = torch.zeros(nv + 1, dtype=torch.int64, device=col.device)
        over_ptr[1:] = torch.cumsum((lens - width).clamp_min(0), 0)
        over = (over_ptr.to(torch.int32).contiguous(), col[~head].contiguous(), val[~head].contiguous())
    return ell_col.contiguous(), ell_val.contiguous(), over


def _ell_width(lens_a, lens_b):
    """Table width of the fast kernel: the smallest of 8 / 16 that leaves at most 5 % of the rows (of either
    orientation) with a CSR tail -- an icosphere (rows of 6-7) gets 8 and no tail, the reference's 482.obj (rows of
    5-9 and two poles of 33) gets 8 with a 10-row tail; 0 = irregular degrees, generic CSR kernel."""
    n = max(int(lens_a.numel()), 1)
    for w in (8, 16):
        if max(int((lens_a > w).sum()), int((lens_b > w).sum())) <= 0.05 * n:
            return w
    return 0


def _finish_csr(c):
    """Derived tables shared by every construction route: 1/(degree without the self loop) and, for bounded
    degrees, the fixed-stride ELL neighbour tables of the fast aggregation kernel."""
    c.nv = int(c.rowptr.numel()) - 1
    c.nnz = int(c.col.numel())
    # 1 / (neighbours without the self loop): what batch_get_lap_info divides by on the binary adjacency
    c.inv_deg = (1.0 / ((c.rowptr[1:] - c.rowptr[:-1]).float() - 1.0)).contiguous()
    c.ell_w = _ell_width(c.rowptr[1:] - c.rowptr[:-1], c.rowptr_t[1:] - c.rowptr_t[:-1]) if c.nv else 0
    c.ell_col = c.ell_val = c.ell_col_t = c.ell_val_t = c.over = c.over_t = None
    if c.ell_w:
        c.ell_col, c.ell_val, c.over = _to_ell(c.rowptr, c.col, c.val, c.ell_w)
        c.ell_col_t, c.ell_val_t, c.over_t = _to_ell(c.rowptr_t, c.col_t, c.val_t, c.ell_w)
    return c


def csr_from_parts(rowptr, col, val, rowptr_t, col_t, val_t):
    """An adjacency handed over already in CSR (+ CSR^T): int32 rowptr/col, fp32 val, on the device.  The result
    can be passed wherever a layer takes `adj` (geometrics_amd.ragged builds block-diagonal batches this way)."""
    c = _Csr()
    c.rowptr, c.col, c.val, c.rowptr_t, c.col_t, c.val_t = rowptr, col, val, rowptr_t, col_t, val_t
    with torch.no_grad():
        return _finish_csr(c)


def adjacency_csr(adj):
    """CSR + CSR^T (+ ELL tables) of a dense [V,V] adjacency, cached per TENSOR OBJECT and in-place
    version.  The key is the object's identity guarded by a weak reference -- never the data pointer,
    which the caching allocator hands to the next adjacency of the same size (auto_encoder.py builds a
    fresh one per mesh).  Building reads the dense matrix once (one host sync, at first use only)."""
    if isinstance(adj, _Csr):
        return adj
    if hasattr(adj, "csr") and isinstance(adj.csr, _Csr):     # a RaggedMeshBatch
        return adj.csr
    if adj.dim() != 2 or adj.shape[0] != adj.shape[1]:
        raise RuntimeError("adjacency must be a square [V,V] tensor, got %s" % (tuple(adj.shape),))
    key = id(adj)
    hit = _csr_cache.get(key)
    if hit is not None and hit[0]() is adj and hit[1] == adj._version:
        return hit[2]
    if not adj.is_cuda:
        raise RuntimeError("adjacency must live on a HIP device; geometrics_amd has no CPU path")
    with torch.no_grad():
        c = _Csr()
        dense = adj.detach()
        c.rowptr, c.col, c.val = _to_csr(dense)
        c.rowptr_t, c.col_t, c.val_t = _to_csr(dense.t())
        _finish_csr(c)
    _csr_cache[key] = (weakref.ref(adj, lambda _ref, k=key: _csr_cache.pop(k, None)), adj._version, c)
    return c


# ------------------------------------------------------- fused aggregation ----
def _activation_code(activation):
    """ReLU / ELU(alpha=1) are folded into the kernel epilogue (and their derivative into the
    backward read); any other callable is applied by the caller after an un-activated kernel."""
    if activation is F.relu or activation is torch.relu:
        return _ACT_RELU
    if activation is F.elu:
        return _ACT_ELU
    return _ACT_NONE


def aggregate_forward(s, bias_c, csr, k, act, out, want_mask=False):
    """Launch out = act([A . s[..., :k] | s[..., k:]] + bias) on contiguous fp32 [B,V,C] tensors: the fixed-stride table
    kernel when the adjacency has one (bounded degrees; long rows continue in its CSR tail), the generic CSR kernel
    otherwise.  Returns the ReLU sign mask when one was asked for and written."""
    b, nv, c = s.shape
    mask = None
    with torch.cuda.device(s.device):
        code = _lib.EUNSUPPORTED
        if csr.ell_w:   # bounded-degree mesh: fixed-stride neighbour table, no rowptr round trip
            if want_mask and act == _ACT_RELU:
                # one sign bit per output element: the backward takes relu' from it instead of re-reading `out`
                words = _lib.lib().geom_zn_gcn_relu_mask_words(b, nv, c, k)
                if words:
                    mask = torch.empty(words, dtype=torch.int16, device=s.device)
            over = csr.over or (None, None, None)
            code = _lib.lib().geom_zn_gcn_aggregate_ell_fwd_f32(
                b, nv, c, k, csr.ell_w, csr.ell_col.data_ptr(), csr.ell_val.data_ptr(), _lib.ptr(over[0]),
                _lib.ptr(over[1]), _lib.ptr(over[2]), s.data_ptr(),
                _lib.ptr(bias_c), act, out.data_ptr(), _lib.ptr(mask), _lib.stream_ptr())
            if code == _lib.EUNSUPPORTED:
                mask = None
        if code == _lib.EUNSUPPORTED:
            _lib.call("geom_zn_gcn_aggregate_fwd_f32", b, nv, c, k, csr.rowptr.data_ptr(), csr.col.data_ptr(),
                      csr.val.data_ptr(), s.data_ptr(), _lib.ptr(bias_c), act, out.data_ptr())
        else:
            _lib.check(code, "geom_zn_gcn_aggregate_ell_fwd_f32")
    return mask


# ---- bias gradients of a whole backward pass finished in ONE launch -----------------------------------------------
# The aggregation backward leaves per-workgroup partial column sums; reducing them is a launch-floor kernel (4.7 us) per
# layer -- 14 of them in a deformation block.  Inside an autograd backward pass the reduction is postponed instead: the
# partials are queued, and a callback at the END of the pass (the engine's queue_callback, what DDP uses for its own
# finalisation) reduces all of them with one geom_colsum_batch_f32 launch on the stream they were produced on.  The
# bias gradient handed to autograd is therefore complete when backward() returns, but not while the pass is running;
# a pass in which something could read it earlier -- an existing .grad to accumulate into, a hook on the bias, a bias that
# receives gradients from more than one node (the engine adds them on arrival) -- takes the immediate reduction, and so
# does every call outside an engine-run pass.
# Deferral is OPT-IN (round-2 advice): what cannot be seen from here -- C++ hooks on the AccumulateGrad node (torch DDP's
# Reducer copies the gradient into its bucket on arrival), a second consumer of the parameter that is a plain torch op --
# would read the placeholder before the flush.  bench.py and the deformation block's own training step switch it on
# (`with layers.deferred_parameter_gradients():`), nothing else does; a placeholder is zero-filled when
# `_zero_fill_deferred` is set (debugging aid: detect_anomaly trips over uninitialised memory otherwise).
defer_parameter_gradients = False     # False: every bias / weight gradient is reduced where it is produced
_zero_fill_deferred = False

# Gradient targets (data-parallel steps: geometrics_amd.dist.GradBucket(bind=True)): a parameter bound to a tensor of its
# shape gets its gradient WRITTEN THERE by the launches of this module -- the reduction launch at the end of the pass writes
# straight into the flat all-reduce bucket, autograd adopts the tensor as `.grad` (a fresh, contiguous tensor object of
# the parameter's layout is taken as is, not copied), and the bucket's pack launch has nothing left to gather.
_gradient_targets = {}


def bind_gradient_targets(params, tensors):
    """`tensors[i]` (same shape / dtype / device as `params[i]`, contiguous) receives the gradient of `params[i]` from now on;
    None unbinds.  Only gradients this module produces itself land there (a library-product fallback returns its own tensor)."""
    for p, t in zip(params, tensors):
        key = id(p)
        if t is None:
            _gradient_targets.pop(key, None)
            continue
        if t.shape != p.shape or t.dtype != p.dtype or t.device != p.device or not t.is_contiguous():
            raise ValueError("a gradient target must match its parameter's shape, dtype and device and be contiguous")
        _gradient_targets[key] = (weakref.ref(p, lambda _r, k=key: _gradient_targets.pop(k, None)), t)


def _gradient_buffer(param, like):
    """Where the gradient of `param` (may be None: unknown) goes: its bound target, else a new tensor like `like`."""
    hit = _gradient_targets.get(id(param)) if param is not None else None
    # (a parameter that already holds a gradient is being ACCUMULATED into: the new gradient must not overwrite the old one's
    # memory, which is what the target is by then)
    # (a parameter fed by more than one live autograd node -- a shared weight or bias, a layer applied twice -- has its
    # gradients ADDED by the engine: each node needs memory of its own, or the second would overwrite the first's before the
    # sum is formed; GradBucket.pack() gathers a gradient that did not land in its view by copy)
    if (hit is not None and hit[0]() is param and hit[1].shape == like.shape and param.grad is None
            and _bias_user_count(param) <= 1):
        return hit[1].detach()           # a fresh tensor object over the target's memory (autograd adopts it as .grad)
    return torch.zeros_like(like) if _zero_fill_deferred else torch.empty_like(like)
use_matrix_core_products = True       # the layers' dense gradients on csrc/dense_gemm.hip where dense.plan says so
use_any_shape_products = True         # every other width: csrc/dense_any.hip (False: the library's products)


class deferred_parameter_gradients:
    """Context manager: inside it, bias / weight gradients of a backward pass are finished by batched launches at the end
    of the pass (the caller guarantees that nothing reads a parameter gradient before backward() returns)."""

    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        global defer_parameter_gradients
        self.prev = defer_parameter_gradients
        defer_parameter_gradients = self.enabled
        return self

    def __exit__(self, *exc):
        global defer_parameter_gradients
        defer_parameter_gradients = self.prev
        return False
# ---- the input gradient of a layer whose input is a LEAF, issued AFTER the parameter gradients of the pass are final ------
# Data-parallel steps (bench.py, N > 1): the only thing the gradient all-reduce has to wait for is the end-of-pass reduction
# launch; the first layer's input gradient dX = G . W^T (65 us at the BASELINE shard, a third of the all-reduce window an
# 8-GPU ring needs) is needed by nobody before backward() returns -- its consumer is the leaf's .grad.  Inside
# `late_input_gradients()` such a product is therefore postponed like the parameter-gradient reductions: the node hands
# autograd the (not yet written) gradient buffer, and the end-of-pass callback launches the product AFTER the reduction
# launch and after `parameter_gradients_ready` callbacks have run -- which is where a data-parallel step records the event
# its collective waits for, so that the all-reduce travels while the product runs.  Same safeguards as the other deferrals
# (leaf without hooks or an existing .grad, a single consumer, an engine-run pass); opt-in, nothing else uses it.
late_input_gradient_products = False
_pending_late = {}        # autograd graph-task id -> [(g2, w2, product buffer, stream, input ref)]
parameter_gradients_ready = []   # callables run by the end-of-pass callback right after the reduction launch(es)


class late_input_gradients:
    """Context manager around forward + backward (see above); `on_parameter_gradients` = a callable run inside the
    end-of-pass callback once every parameter gradient of the pass has been launched, before the postponed products."""

    def __init__(self, on_parameter_gradients=None, enabled=True, collect=None):
        """collect: a list -- the postponed products are NOT launched by the callback but appended to it as callables
        (each launches one product into the gradient buffer autograd already holds); the caller launches them, e.g. after
        issuing a collective.  They stay valid for as long as their tensors do (a captured step replays them every step)."""
        self.enabled, self.hook, self.collect = enabled, on_parameter_gradients, collect

    def __enter__(self):
        global late_input_gradient_products
        self.prev = late_input_gradient_products
        late_input_gradient_products = self.enabled
        if self.hook is not None:
            parameter_gradients_ready.append(self.hook)
        self.prev_collect = _late_collect[0]
        _late_collect[0] = self.collect
        return self

    def __exit__(self, *exc):
        global late_input_gradient_products
        late_input_gradient_products = self.prev
        if self.hook is not None:
            parameter_gradients_ready.remove(self.hook)
        _late_collect[0] = self.prev_collect
        return False


def _may_postpone_input_gradient(x):
    """A leaf whose gradient nobody observes before backward() returns: no tensor hooks (they would not see the postponed
    part), an engine-run first-order pass, deferral of the parameter gradients active (the flush this rides on).  The
    postponing node returns NO gradient for the leaf to the engine; the end-of-pass callback launches the product into a
    buffer of its own and then SETS the leaf's .grad to it -- or adds it to what is there: whatever other consumers of the
    leaf contributed through the engine during the pass (or an earlier pass left behind) is kept, so any number of
    consumers is correct.  (Round 4 handed the engine the not-yet-written buffer instead; a second consumer's gradient was
    added to uninitialised memory and then overwritten: round-4 advice.)  Only under .backward(): torch.autograd.grad(...,
    inputs=[leaf]) collects what the ENGINE carries and fails loudly ("not used in the graph") for such a leaf."""
    if not (late_input_gradient_products and defer_parameter_gradients) or not hasattr(torch._C, "_current_graph_task_id"):
        return False
    if torch._C._current_graph_task_id() < 0 or torch.is_grad_enabled():
        return False
    return (x.is_leaf and not x._backward_hooks and not getattr(x, "_post_accumulate_grad_hooks", None))


_late_collect = [None]


def _late_product(g2, w2, out, stream):
    def launch():
        with torch.cuda.stream(stream), torch.no_grad():
            if (_own_products_preferred() and use_matrix_core_products and w2.is_contiguous() and g2.is_contiguous() and g2.shape[0] >= 512
                    and w2.shape[1] % 16 == 0 and _dense_kernels.supported(w2.shape[0], w2.shape[1], g2.shape[0])):
                _dense_kernels.backward_input(g2, w2, out=out)
            else:
                torch.mm(g2, w2.t(), out=out)
    return launch


def _flush_late(task):
    for hook in list(parameter_gradients_ready):
        hook()
    for g2, w2, out, stream, x_ref in _pending_late.pop(task, []):
        jobs = [_late_product(g2, w2, out, stream)]
        x = x_ref()
        if x is not None:
            if x.grad is None:
                x.grad = out.view(x.shape)       # the product's own buffer becomes the leaf's gradient: no copy
            else:                                # other consumers of the leaf (or an earlier pass) were there first: add
                def add(x=x, out=out, stream=stream):
                    with torch.cuda.stream(stream), torch.no_grad():
                        x.grad.add_(out.view_as(x.grad))
                jobs.append(add)
        if _late_collect[0] is None:
            for job in jobs:
                job()
        else:
            _late_collect[0].extend(jobs)


_pending_colsums = {}     # autograd graph-task id -> [(partials, rows, cols, out alias, stream)]
_pending_reduce = {}      # autograd graph-task id -> [(rows, cin, c, workspace, dW alias, stream, param ref)]: weight-gradient partials
# bias parameter -> the live autograd nodes that produce a gradient for it.  A bias shared by two layers (or a layer applied
# twice) gets its gradients ADDED by the engine inside the pass, which reads them on arrival: more than one live node means
# immediate reduction for all of them.  Nodes leave the set when their graph is freed.
_bias_users = {}          # id(bias) -> (weak reference to the bias, WeakSet of nodes); keyed by identity, tensors do not compare


def _register_bias_user(bias, node):
    key = id(bias)
    entry = _bias_users.get(key)
    if entry is None or entry[0]() is not bias:
        entry = _bias_users[key] = (weakref.ref(bias, lambda _ref, k=key: _bias_users.pop(k, None)), weakref.WeakSet())
    entry[1].add(node)


def _bias_user_count(bias):
    entry = _bias_users.get(id(bias))
    return len(entry[1]) if entry is not None and entry[0]() is bias else 0


def _alias(t):
    """A second tensor object over t's memory (no view relation): keeps the storage alive without being a reference to
    the tensor itself, so that autograd still finds the gradient unshared and stores it instead of cloning it."""
    return torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), t.storage_offset(), t.size(), t.stride())


def _may_defer(bias, opted_in=False):
    """opted_in: the forward ran inside weight_gradient_batching() -- that context is the caller's explicit request."""
    if not (defer_parameter_gradients or opted_in) or bias is None or not hasattr(torch._C, "_current_graph_task_id"):
        return False
    if torch._C._current_graph_task_id() < 0 or torch.is_grad_enabled():      # not an engine pass / double backward
        return False
    acc = getattr(bias, "grad_fn", None)
    if acc is not None:      # not a leaf
        return False
    if _bias_user_count(bias) > 1:      # another node of a live graph feeds the same parameter
        return False
    return (bias.grad is None and not bias._backward_hooks and not getattr(bias, "_post_accumulate_grad_hooks", None)
            and bias.is_leaf)


def _check_landed(param_ref, out):
    """After a deferred reduction: the gradient autograd stored for the parameter must BE the buffer the flush wrote.  If
    the engine kept a copy instead (it clones a gradient it does not hold the only reference to), the finished values are
    copied into it -- never leave a placeholder behind silently."""
    param = param_ref() if param_ref is not None else None
    if param is None or param.grad is None or param.grad.data_ptr() == out.data_ptr():
        return
    if param.grad.shape == out.shape or param.grad.numel() == out.numel():
        param.grad.copy_(out.view_as(param.grad))
    else:
        raise RuntimeError("geometrics_amd: a deferred parameter gradient did not reach its parameter (shape %s vs %s)"
                           % (tuple(param.grad.shape), tuple(out.shape)))


_flush_registered = set()     # graph tasks whose end-of-pass callback is queued
_backward_optimizer = None     # optim.FusedAdam.fuse_into_backward(): its step rides in the end-of-pass launch when that launch covers it


def _register_flush(task):
    """ONE end-of-pass callback per backward pass: it finishes the pending bias-gradient column sums AND the pending
    weight-gradient partial sums, in one launch where their shapes allow."""
    if task in _flush_registered:
        return
    for stale in [t for t in _flush_registered if t < task - 64]:      # passes that died of an exception
        _flush_registered.discard(stale)
        _pending_colsums.pop(stale, None)
        _pending_reduce.pop(stale, None)
        _pending_late.pop(stale, None)
    _flush_registered.add(task)
    torch.autograd.Variable._execution_engine.queue_callback(lambda: _flush_pass(task))


def _flush_pass(task):
    try:
        _flush_parameter_gradients(task)
    finally:
        _flush_late(task)      # the postponed input-gradient products: behind the reduction launch(es) and the ready-callbacks


def _flush_parameter_gradients(task):
    _flush_registered.discard(task)
    _flush_pending_gradients(task)


def _flush_pending_gradients(task):
    red = _pending_reduce.get(task, [])
    cs = list(_pending_colsums.get(task, []))
    streams = {j[5] for j in red} | {j[4] for j in cs}
    joint = (red and cs and len(streams) == 1 and all(j[2] % 4 == 0 for j in cs)
             and 2 * len(red) + len(cs) <= _lib.DENSE_MAX_REDUCE_JOBS)
    if not joint:
        _flush_colsums(task)
        _flush_reduce(task)
        return
    _pending_reduce.pop(task, None)
    _pending_colsums.pop(task, None)
    stream = next(iter(streams))
    n, m = len(red), len(cs)
    ints = lambda seq: (ctypes.c_int * len(seq))(*seq)
    ptrs = lambda seq: (ctypes.c_void_p * len(seq))(*[t.data_ptr() for t in seq])
    opt = _backward_optimizer
    slots = None
    if opt is not None:     # the optimiser's step rides along when this launch finishes the gradient of every one of its parameters
        owners = [j[6]() if j[6] is not None else None for j in red] + [j[5]() if j[5] is not None else None for j in cs]
        index = {id(p): k for k, p in enumerate(opt.params)}
        slots = [index.get(id(p)) if p is not None else None for p in owners]
        if None in slots or sorted(slots) != list(range(len(opt.params))) or opt.params[0].device != stream.device:
            slots = None
    with torch.cuda.device(stream.device):
        if slots is None:
            _lib.check(_lib.lib().geom_dense_reduce2_f32(
                n, ints([j[0] for j in red]), ints([j[1] for j in red]), ints([j[2] for j in red]), ptrs([j[3] for j in red]),
                ptrs([j[4] for j in red]), None, m, ptrs([j[0] for j in cs]), ints([j[1] for j in cs]), ints([j[2] for j in cs]),
                ptrs([j[3] for j in cs]), stream.cuda_stream), "geom_dense_reduce2_f32")
        else:
            pick = lambda seq, ks: ptrs([seq[k] for k in ks])
            wk, bk = slots[:n], slots[n:]
            params = [p.data for p in opt.params]
            _lib.check(_lib.lib().geom_dense_reduce_adam_f32(
                n, ints([j[0] for j in red]), ints([j[1] for j in red]), ints([j[2] for j in red]), ptrs([j[3] for j in red]),
                ptrs([j[4] for j in red]), pick(params, wk), pick(opt.exp_avg, wk), pick(opt.exp_avg_sq, wk),
                m, ptrs([j[0] for j in cs]), ints([j[1] for j in cs]), ints([j[2] for j in cs]), ptrs([j[3] for j in cs]),
                pick(params, bk), pick(opt.exp_avg, bk), pick(opt.exp_avg_sq, bk), float(opt.lr), float(opt.betas[0]),
                float(opt.betas[1]), float(opt.eps), opt.state.data_ptr(), stream.cuda_stream), "geom_dense_reduce_adam_f32")
            opt._stepped_in_backward = True
    for job in red:
        _check_landed(job[6], job[4])
    for job in cs:
        _check_landed(job[5], job[3])


def _flush_colsums(task):
    jobs = _pending_colsums.pop(task, [])
    by_stream = {}
    for job in jobs:
        by_stream.setdefault(job[4], []).append(job)
    for stream, group in by_stream.items():
        with torch.cuda.device(stream.device):
            for c0 in range(0, len(group), _lib.COLSUM_MAX_JOBS):
                chunk = group[c0:c0 + _lib.COLSUM_MAX_JOBS]
                n = len(chunk)
                _lib.check(_lib.lib().geom_colsum_batch_f32(
                    n, (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in chunk]), (ctypes.c_int * n)(*[j[1] for j in chunk]),
                    (ctypes.c_int * n)(*[j[2] for j in chunk]), (ctypes.c_void_p * n)(*[j[3].data_ptr() for j in chunk]),
                    stream.cuda_stream), "geom_colsum_batch_f32")
    for job in jobs:
        _check_landed(job[5], job[3])


def _queue_colsum(partials, rows, cols, out, param=None):
    task = torch._C._current_graph_task_id()
    jobs = _pending_colsums.get(task)
    if jobs is None:
        jobs = _pending_colsums[task] = []
    _register_flush(task)
    jobs.append((partials, rows, cols, _alias(out), torch.cuda.current_stream(out.device),
                 None if param is None else weakref.ref(param)))


def _finish_colsum(partials, rows, cols, grad_bias, bias, defer):
    """Per-workgroup column sums -> the bias gradient: queued for the end-of-pass launch, or one launch now."""
    if defer:
        _queue_colsum(partials, rows, cols, grad_bias, bias)
        return
    with torch.cuda.device(partials.device):
        _lib.check(_lib.lib().geom_colsum_batch_f32(
            1, (ctypes.c_void_p * 1)(partials.data_ptr()), (ctypes.c_int * 1)(rows), (ctypes.c_int * 1)(cols),
            (ctypes.c_void_p * 1)(grad_bias.data_ptr()), _lib.stream_ptr()), "geom_colsum_batch_f32")


def aggregate_backward(g, csr, k, act, out, mask, want_bias, bias=None, arena=None):
    """grad_support = [A^T . g'[..., :k] | g'[..., k:]] with g' = g * act'(out) (relu' from the sign mask when there is
    one), and the bias gradient = column sums of g' out of the same launch (+ a fixed-order reduction: at once, or -- given
    the bias parameter, inside a backward pass -- batched at the end of the pass, see above)."""
    b, nv, c = g.shape
    grad_support = _new_like(g, "grad_support", arena, descending=True)
    grad_bias = scratch = None
    defer = False
    if want_bias:   # column sums of g come out of the same kernel (per-block partials + fixed-order reduce)
        grad_bias = _gradient_buffer(bias, bias) if bias is not None and bias.shape == (c,) and bias.dtype == torch.float32 \
            else torch.empty(c, dtype=torch.float32, device=g.device)
        scratch = torch.empty(_lib.lib().geom_zn_gcn_bwd_scratch_floats(b, nv, c), dtype=torch.float32,
                              device=g.device)
        defer = _may_defer(bias, opted_in=arena is not None)
    now = None if defer else grad_bias
    with torch.cuda.device(g.device):
        code = _lib.EUNSUPPORTED
        ell_w = csr.ell_w
        if ell_w:
            over = csr.over_t or (None, None, None)
            code = _lib.lib().geom_zn_gcn_aggregate_ell_bwd_f32(
                b, nv, c, k, ell_w, csr.ell_col_t.data_ptr(), csr.ell_val_t.data_ptr(), _lib.ptr(over[0]),
                _lib.ptr(over[1]), _lib.ptr(over[2]), g.data_ptr(),
                _lib.ptr(out), _lib.ptr(mask), act, grad_support.data_ptr(), _lib.ptr(now), _lib.ptr(scratch),
                _lib.stream_ptr())
        if code == _lib.EUNSUPPORTED:
            ell_w = 0
            _lib.call("geom_zn_gcn_aggregate_bwd_f32", b, nv, c, k, csr.rowptr_t.data_ptr(),
                      csr.col_t.data_ptr(), csr.val_t.data_ptr(), g.data_ptr(), _lib.ptr(out), act,
                      grad_support.data_ptr(), _lib.ptr(now), _lib.ptr(scratch))
        else:
            _lib.check(code, "geom_zn_gcn_aggregate_ell_bwd_f32")
    if defer:
        _queue_colsum(scratch, int(_lib.lib().geom_zn_gcn_bwd_partial_rows(b, nv, c, k, ell_w)), c, grad_bias, bias)
    return grad_support, grad_bias


class _ZeroNAggregate(torch.autograd.Function):
    """out = act([A . S[..., :k] | S[..., k:]] + bias)  for S [B,V,C]: one kernel, S read once.
    Backward: grad_S = [A^T . g[..., :k] | g[..., k:]] with g = grad_out * act'(out), again one kernel."""

    @staticmethod
    def forward(ctx, support, bias, csr, k, act):
        s = _lib.require(support, "support", torch.float32, 3)
        if s.shape[1] != csr.nv:
            raise RuntimeError("support has %d vertices but the adjacency has %d" % (s.shape[1], csr.nv))
        bias_c = None if bias is None else _lib.require(bias, "bias", torch.float32, 1)
        ctx.arena = current_slabs()
        out = _new_like(s, "aggregated", ctx.arena)
        mask = aggregate_forward(s, bias_c, csr, k, act, out, want_mask=support.requires_grad)
        ctx.csr, ctx.k, ctx.act, ctx.has_bias = csr, k, act, bias is not None
        ctx.bias_ref = None if bias is None else weakref.ref(bias)
        if bias is not None and ctx.needs_input_grad[1]:
            _register_bias_user(bias, ctx)
        ctx.masked = mask is not None
        if mask is not None:
            ctx.save_for_backward(mask)
        elif act != _ACT_NONE:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        g = grad_out.contiguous()
        act = ctx.act
        mask = ctx.saved_tensors[0] if ctx.masked else None
        out = ctx.saved_tensors[0] if (act != _ACT_NONE and not ctx.masked) else None
        grad_support, grad_bias = aggregate_backward(g, ctx.csr, ctx.k, act, out, mask,
                                                     ctx.has_bias and ctx.needs_input_grad[1],
                                                     ctx.bias_ref() if ctx.bias_ref is not None else None, ctx.arena)
        return (grad_support if ctx.needs_input_grad[0] else None), grad_bias, None, None, None


class _ZeroNAggregateHead(torch.autograd.Function):
    """positions = base + scale * act([A . S[..., :k] | S[..., k:]] + bias)[..., :3]: the aggregation of the layer whose
    three leading channels are a stage's coordinate update (GEOMetrics.py:121,126,131), with the update in the same two
    launches.  Forward: the aggregation kernel also writes the new positions.  Backward: the gradient of the layer output
    is [scale * grad_pos | 0 ...] by construction -- it is never materialised (15.7 MB of zeros at the BASELINE shard) nor
    read back; the aggregation backward synthesises it from grad_pos."""

    @staticmethod
    def forward(ctx, support, bias, base, csr, k, act, scale, down=None):
        s = _lib.require(support, "support", torch.float32, 3)
        base_c = _lib.require(base, "base", torch.float32, 3, 3)
        ctx.down = down        # a _StackLink: the boundary launch that produced `support` (zero_n_stack_positions)
        b, nv, c = s.shape
        bias_c = None if bias is None else _lib.require(bias, "bias", torch.float32, 1)
        out = torch.empty_like(s)
        pos = torch.empty_like(base_c)
        mask = None
        if act == _ACT_RELU:
            mask = torch.empty(_lib.lib().geom_zn_gcn_relu_mask_words(b, nv, c, k), dtype=torch.int16, device=s.device)
        over = csr.over or (None, None, None)
        with torch.cuda.device(s.device):
            _lib.call("geom_zn_gcn_aggregate_ell_head_fwd_f32", b, nv, c, k, csr.ell_w, csr.ell_col.data_ptr(),
                      csr.ell_val.data_ptr(), _lib.ptr(over[0]), _lib.ptr(over[1]), _lib.ptr(over[2]), s.data_ptr(),
                      _lib.ptr(bias_c), act, out.data_ptr(), _lib.ptr(mask), base_c.data_ptr(), float(scale), pos.data_ptr())
        ctx.csr, ctx.k, ctx.act, ctx.scale, ctx.shape = csr, k, act, float(scale), (b, nv, c)
        ctx.bias_ref = None if bias is None else weakref.ref(bias)
        if bias is not None and ctx.needs_input_grad[1]:
            _register_bias_user(bias, ctx)
        if mask is not None:
            ctx.save_for_backward(mask)
        return pos

    @staticmethod
    def backward(ctx, grad_pos):
        gp = grad_pos.contiguous()
        b, nv, c = ctx.shape
        csr, k = ctx.csr, ctx.k
        mask = ctx.saved_tensors[0] if ctx.saved_tensors else None
        grad_support = torch.empty(b, nv, c, dtype=torch.float32, device=gp.device)
        grad_bias = scratch = None
        defer = False
        bias = ctx.bias_ref() if ctx.bias_ref is not None else None
        if ctx.needs_input_grad[1] and ctx.bias_ref is not None:
            grad_bias = _gradient_buffer(bias, bias) if bias is not None and bias.shape == (c,) and bias.dtype == torch.float32 \
                else torch.empty(c, dtype=torch.float32, device=gp.device)
            scratch = torch.empty(_lib.lib().geom_zn_gcn_bwd_scratch_floats(b, nv, c), dtype=torch.float32, device=gp.device)
            defer = _may_defer(bias)
        down = ctx.down
        # (the launch below takes THIS layer's aggregation backward: it must be a shape the boundary kernel serves -- 192 wide,
        # k = 64, table width 8 without long rows -- and `wt` the transposed weight of a 192-row product; else: the separate operators)
        if (down is not None and down.wt is not None and down.wanted and ctx.needs_input_grad[0]
                and _fused.plan(b * nv)["bwd"] and down.wt.dim() == 2 and down.wt.shape[0] == c
                and _fused.supported(csr, c, k, down.wt.shape[1])):
            # this aggregation backward AND the input gradient of the product below it in one launch (csrc/zn_stack.hip);
            # the boundary below picks its input gradient up from the link instead of computing it
            rows = _fused.partial_rows(b, nv)
            partial = torch.empty(rows, c, dtype=torch.float32, device=gp.device) if grad_bias is not None else None
            _fused.layer_backward(None, None, mask, csr, k, ctx.act, down.wt, g_out=grad_support, grad_in=down.take_dx(b, nv),
                                  colsum_partial=partial, grad_pos=gp, head_scale=ctx.scale, shape=(b, nv, c))
            down.stamp(grad_support)
            if grad_bias is not None:
                _finish_colsum(partial, rows, c, grad_bias, bias, defer)
            return grad_support, grad_bias, (gp if ctx.needs_input_grad[2] else None), None, None, None, None, None
        over = csr.over_t or (None, None, None)
        with torch.cuda.device(gp.device):
            _lib.call("geom_zn_gcn_aggregate_ell_head_bwd_f32", b, nv, c, k, csr.ell_w, csr.ell_col_t.data_ptr(),
                      csr.ell_val_t.data_ptr(), _lib.ptr(over[0]), _lib.ptr(over[1]), _lib.ptr(over[2]), gp.data_ptr(),
                      ctx.scale, _lib.ptr(mask), ctx.act, grad_support.data_ptr(), _lib.ptr(None if defer else grad_bias),
                      _lib.ptr(scratch))
        if defer:
            _queue_colsum(scratch, int(_lib.lib().geom_zn_gcn_bwd_partial_rows(b, nv, c, k, csr.ell_w)), c, grad_bias, bias)
        return (grad_support if ctx.needs_input_grad[0] else None), grad_bias, (gp if ctx.needs_input_grad[2] else None), \
            None, None, None, None, None


def zero_n_aggregate_head(support, adj, bias, k, activation, base, scale, down=None):
    """base + scale * zero_n_aggregate(...)[..., :3] for a [B,V,C] support: fused (see _ZeroNAggregateHead) for split-3
    layers on a bounded-degree mesh with ReLU or no activation, the two separate operators otherwise."""
    act = _ACT_NONE if activation is None else _activation_code(activation)
    fused = (torch.is_tensor(support) and support.dim() == 3 and support.is_cuda and support.dtype == torch.float32
             and not (torch.is_tensor(adj) and adj.dim() == 3) and (activation is None or act == _ACT_RELU))
    if fused:
        csr = adjacency_csr(adj)
        c = support.shape[-1]
        fused = bool(csr.ell_w) and k > 0 and k % 4 == 0 and c == 3 * k
    if not fused:
        from .ops import VertexHead
        return VertexHead.apply(base, zero_n_aggregate(support, adj, bias, k, activation), scale)
    return _ZeroNAggregateHead.apply(support, bias, base, csr, k, act, scale, down)


def zero_n_aggregate(support, adj, bias, k, activation=None):
    """Shared tail of every 0N-GCN layer; accepts [V,C] or [B,V,C] support.  Returns the
    ACTIVATED output when `activation` is given (fused for relu / elu)."""
    if torch.is_tensor(adj) and adj.dim() == 3:
        # one dense adjacency PER MESH ([B,V,V]): what the reference's torch.matmul(adj, support[..., :k]) also accepts
        # (layers.py:111, 146).  Not a shape the reference drivers produce; served by the same dense product.
        out = torch.cat((torch.matmul(adj, support[..., :k]), support[..., k:]), dim=-1)
        if bias is not None:
            out = out + bias
        return out if activation is None else activation(out)
    csr = adjacency_csr(adj)
    act = _ACT_NONE if activation is None else _activation_code(activation)
    s3 = support.unsqueeze(0) if support.dim() == 2 else support
    out = _ZeroNAggregate.apply(s3, bias, csr, k, act)
    if support.dim() == 2:
        out = out.squeeze(0)
    if activation is not None and act == _ACT_NONE:
        out = activation(out)
    return out


# ---- weight gradients of a stack of equal layers as ONE batched product --------------------------------------------------
# dW = X^T . G of a hidden layer (K = b*V rows against a 192 x 192 output) is the library's least efficient product: 22 us at
# the reference's training shape for 0.57 GFLOP, and a deformation block issues twelve of them, one per layer.  The
# gradients are independent of each other, so inside `weight_gradient_batching()` they are postponed to the end of the
# backward pass (same mechanism and same safeguards as the bias gradients above) and issued as one strided-batched
# product per run of equal layers: 242 -> 72 us for the twelve.  A strided-batched product wants its operands at a
# regular pitch, so while the context is active the layers' activations and gradients are carved out of stacked buffers
# (`_Slabs`): consecutive equal-shape allocations sit one pitch apart, forward ones ascending, backward ones descending
# (the backward pass meets the layers in reverse), which makes X_l, G_l and dW_l all ascending in l.
class _Slabs:
    """Stacked buffers for the tensors of one forward/backward pass: take(kind, shape) returns the next [shape] slot of a
    [slots, *shape] buffer of that kind and shape (a fresh buffer when the current one is used up: the run of equal
    layers is then split there).  slots = the depth of the stack (untaken slots cost address space only)."""
    def __init__(self, slots):
        self.open = {}
        self.slots = max(2, int(slots))

    def take(self, kind, shape, device, descending=False):
        shape = tuple(shape)
        key = (kind, shape, device, descending)
        slots = self.slots
        cur = self.open.get(key)
        if cur is None or cur[1] == slots:
            cur = self.open[key] = [torch.empty((slots,) + shape, dtype=torch.float32, device=device), 0]
        index = slots - 1 - cur[1] if descending else cur[1]
        cur[1] += 1
        return cur[0][index]


_active = threading.local()    # .slabs = the arena of the forward pass running on this thread (weight_gradient_batching)


def current_slabs():
    return getattr(_active, "slabs", None)


_pending_dense = {}       # autograd graph-task id -> [(x2d, g2d, dW slot alias, stream)] in backward order


class weight_gradient_batching:
    """Context manager for the FORWARD pass of a stack of layers: their weight gradients are computed at the end of the
    backward pass, batched over runs of equal layers (see above).  Results differ from the per-layer products only by the
    library kernel's summation order (1e-6 relative)."""

    def __init__(self, depth=4):
        """depth: how many equal layers follow each other at most (the slots of one stacked buffer)."""
        self.depth = depth

    def __enter__(self):
        self.outer = current_slabs()
        # nothing to batch without a backward pass: an inference forward keeps its plain, progressively freed allocations
        _active.slabs = _Slabs(self.depth) if torch.is_grad_enabled() else None
        return self

    def __exit__(self, *exc):
        _active.slabs = self.outer
        return False


def _new_like(t, kind, arena, descending=False):
    """Allocation of an activation / gradient: a slot of the pass's stacked buffers when batching is on."""
    if arena is None:
        return torch.empty_like(t)
    return arena.take(kind, t.shape, t.device, descending)


def _regular_run(tensors):
    """(first tensor, pitch in elements) when the tensors sit at one constant positive pitch inside one storage."""
    first = tensors[0]
    if len(tensors) == 1:
        return first, 0
    base = first.untyped_storage().data_ptr()
    if any(t.untyped_storage().data_ptr() != base or t.stride() != first.stride() for t in tensors):
        return None
    step = tensors[1].storage_offset() - first.storage_offset()
    if step <= 0 or any(tensors[i + 1].storage_offset() - tensors[i].storage_offset() != step for i in range(len(tensors) - 1)):
        return None
    return first, step


def _flush_dense(task):
    jobs = _pending_dense.pop(task, [])
    jobs.reverse()                                   # layer order: every operand ascending
    _flush_dense_products([job[:4] for job in jobs])
    for job in jobs:
        _check_landed(job[4], job[2])


def _flush_dense_products(jobs):
    i = 0
    while i < len(jobs):
        x, g, out, stream = jobs[i]
        j = i + 1
        while j < len(jobs) and jobs[j][0].shape == x.shape and jobs[j][1].shape == g.shape and jobs[j][3] == stream:
            j += 1
        group = jobs[i:j]
        with torch.cuda.stream(stream), torch.no_grad():
            runs = [_regular_run([job[k] for job in group]) for k in range(3)] if len(group) > 1 else None
            if runs and all(runs):
                n = len(group)
                xb = torch.as_strided(runs[0][0], (n,) + tuple(x.shape), (runs[0][1],) + tuple(x.stride()))
                gb = torch.as_strided(runs[1][0], (n,) + tuple(g.shape), (runs[1][1],) + tuple(g.stride()))
                ob = torch.as_strided(runs[2][0], (n,) + tuple(out.shape), (runs[2][1],) + tuple(out.stride()))
                torch.bmm(xb.transpose(1, 2), gb, out=ob)
            else:
                for xx, gg, oo, _ in group:
                    # (a layer of its own width -- the block's 192 -> 3 coordinate head: 7712 summed rows against a 192 x 3 output
                    # is 75 us in the library, which runs it without a split; the any-shape kernel splits the sum)
                    _weight_gradient_product(xx, gg, out=oo)
        i = j


def _takes_any_shape_kernel(rows, cin, c):
    """Shapes whose products run on csrc/dense_any.hip: whatever the 192-column kernels (dense.plan) do not take."""
    return (use_any_shape_products and use_matrix_core_products and _dense_kernels.plan(rows, cin, c)["dw"] != "mfma"
            and _dense_kernels.any_supported(rows, cin, c))


# The library's products of this path are fast only with the recorded selections of geometrics_amd/tuning (TunableOp: validated
# against the PyTorch / hipBLASLt build, so a library update REJECTS the file and the default heuristic runs the 963-wide
# products at 85 us instead of 62).  When gemm_tuning.enable() was called and the file was rejected, the forward products and
# the wide input gradient of the 192-column layers take this package's own matrix-core kernels instead (csrc/dense_gemm.hip:
# within a few per cent of the tuned library, measured by tools/time_dense.py and bench.py --own-products): the headline does
# not hang on a version-locked file.  None = that rule; True / False force it (tests, A/B).
own_dense_products = None


def _own_products_preferred():
    if own_dense_products is not None:
        return bool(own_dense_products)
    env = os.environ.get("GEOM_OWN_PRODUCTS")
    if env is not None:
        return env not in ("", "0")
    from . import gemm_tuning
    return gemm_tuning.status == "library default (tuning file rejected)"      # (not: "tuned at start-up", gemm_tuning.tune_products)


def _own_kernel_takes(x, w2):
    return (use_matrix_core_products and x.is_cuda and x.dtype == torch.float32 and w2.dtype == torch.float32 and w2.dim() == 2
            and w2.is_contiguous() and x.shape[-1] == w2.shape[0] and w2.shape[1] % 16 == 0
            and _dense_kernels.supported(w2.shape[0], w2.shape[1], x.numel() // max(x.shape[-1], 1))
            and x.numel() // max(x.shape[-1], 1) >= 512)


def _library_or_own_forward(x, w2):
    """x @ w2 by the library, or by geom_dense_fwd_f32 when the library runs untuned (see above)."""
    if _own_products_preferred() and _own_kernel_takes(x, w2):
        x2 = x.reshape(-1, x.shape[-1])
        return _dense_kernels.forward(x2 if x2.is_contiguous() else x2.contiguous(), w2).view(x.shape[:-1] + (w2.shape[1],))
    return torch.matmul(x, w2)


def _library_or_own_input_gradient(g2, w2):
    """g2 @ w2^T ([rows, c] x [cin, c]^T) by the library, or by geom_dense_bwd_input_f32 when the library runs untuned."""
    if (_own_products_preferred() and use_matrix_core_products and g2.is_cuda and g2.dtype == torch.float32 and g2.dim() == 2
            and g2.is_contiguous() and w2.is_contiguous() and w2.dim() == 2 and g2.shape[1] == w2.shape[1] and g2.shape[0] >= 512
            and w2.shape[1] % 16 == 0 and _dense_kernels.supported(w2.shape[0], w2.shape[1], g2.shape[0])):
        return _dense_kernels.backward_input(g2, w2)
    return torch.matmul(g2, w2.t())


def _forward_product(x, w2):
    """x [..., cin] @ w2 [cin, c]: ONE rule for which kernel computes it, whichever autograd node wraps it (the routes of a
    layer must agree bit for bit in the forward: tests compare them)."""
    rows = x.numel() // x.shape[-1] if x.shape[-1] else 0
    if (x.is_cuda and x.dtype == torch.float32 and w2.dtype == torch.float32 and w2.dim() == 2 and w2.is_contiguous()
            and x.shape[-1] == w2.shape[0] and _takes_any_shape_kernel(rows, x.shape[-1], w2.shape[-1])):
        x2 = x.reshape(-1, x.shape[-1])
        return _dense_kernels.gemm(x2 if x2.is_contiguous() else x2.contiguous(), w2).view(x.shape[:-1] + (w2.shape[1],))
    return _library_or_own_forward(x, w2)


def _weight_gradient_product(x2, g2, out=None):
    """x2^T @ g2 ([rows, cin]^T x [rows, c]): the any-shape kernel's split product where _takes_any_shape_kernel says so (the
    library runs a long sum against a small output without a split: 75 us for the 192 -> 3 head at 7712 rows), else the library."""
    if (x2.is_cuda and x2.dtype == torch.float32 and g2.dtype == torch.float32 and x2.dim() == 2 and g2.dim() == 2
            and x2.stride(1) == 1 and g2.stride(1) == 1 and (out is None or (out.dim() == 2 and out.stride(1) == 1))
            and _takes_any_shape_kernel(x2.shape[0], x2.shape[1], g2.shape[1])):
        return _dense_kernels.gemm(x2, g2, trans_a=True, out=out)
    return torch.mm(x2.t(), g2) if out is None else torch.mm(x2.t(), g2, out=out)


class _Dense(torch.autograd.Function):
    """support = input @ W for [.., Cin] x [Cin, Cout] (W may carry the reference's leading 1: [1, Cin, Cout]) with the
    weight gradient postponed to the end of the backward pass (used only inside weight_gradient_batching();
    torch.matmul otherwise).  Takes the PARAMETER itself, so that its gradient goes straight to the leaf."""

    @staticmethod
    def forward(ctx, x, w, arena):
        ctx.save_for_backward(x, w)
        ctx.arena = arena
        ctx.w_ref = weakref.ref(w)
        if ctx.needs_input_grad[1]:
            _register_bias_user(w, ctx)
        return _forward_product(x, w.reshape(w.shape[-2:]))

    @staticmethod
    def backward(ctx, grad):
        x, w = ctx.saved_tensors
        g = grad.contiguous()
        w2 = w.reshape(w.shape[-2:])
        grad_x = torch.matmul(g, w2.t()) if ctx.needs_input_grad[0] else None
        grad_w = None
        if ctx.needs_input_grad[1]:
            x2, g2 = x.reshape(-1, x.shape[-1]), g.reshape(-1, g.shape[-1])
            param = ctx.w_ref()
            if param is not None and x2.is_contiguous() and _may_defer(param, opted_in=True):
                grad_w = ctx.arena.take("dW", w.shape, w.device, descending=True)
                task = torch._C._current_graph_task_id()
                jobs = _pending_dense.get(task)
                if jobs is None:
                    for stale in [t for t in _pending_dense if t < task - 64]:
                        del _pending_dense[stale]
                    jobs = _pending_dense[task] = []
                    torch.autograd.Variable._execution_engine.queue_callback(lambda: _flush_dense(task))
                jobs.append((x2, g2, _alias(grad_w).view(w2.shape), torch.cuda.current_stream(w.device), ctx.w_ref))
            else:
                grad_w = _weight_gradient_product(x2, g2).view(w.shape)
        return grad_x, grad_w, None


# ---- the layer's dense products on the fp32 matrix cores (csrc/dense_gemm.hip) -----------------------------------------


def _flush_reduce(task):
    jobs = _pending_reduce.pop(task, [])
    by_stream = {}
    for job in jobs:
        by_stream.setdefault(job[5], []).append(job[:5] + (None,))
    for stream, group in by_stream.items():
        _dense_kernels.reduce(group, stream.cuda_stream)
    for job in jobs:
        _check_landed(job[6], job[4])


def _finish_weight_gradient(w_ref, w, rows, cin, c, ws):
    """The weight gradient whose split partial sums are in `ws`: queued for the reduction launch at the end of the pass where
    deferral is allowed (its buffer handed to autograd now), reduced at once otherwise."""
    param = w_ref()
    grad_w = _gradient_buffer(param, w)
    if param is not None and _may_defer(param):
        task = torch._C._current_graph_task_id()
        jobs = _pending_reduce.get(task)
        if jobs is None:
            jobs = _pending_reduce[task] = []
        _register_flush(task)
        jobs.append((rows, cin, c, ws, _alias(grad_w).view(cin, c), torch.cuda.current_stream(w.device), w_ref))
    else:
        _dense_kernels.reduce([(rows, cin, c, ws, grad_w.view(cin, c), None)])
    return grad_w


class _DenseMM(torch.autograd.Function):
    """support = input @ W with both gradients on the matrix-core kernels where `dense.plan` puts them: the input
    gradient and the split partial sums of the weight gradient in ONE launch (two workgroups per CU) for the 192-wide
    layers, the partial sums alone for the 963-wide one; the partials of all layers of a backward pass are added up by one
    reduction launch at its end when parameter-gradient deferral is on (see `defer_parameter_gradients`), at once
    otherwise.  Same values either way: the reduction order is fixed by the shape."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.w_ref = weakref.ref(w)
        if ctx.needs_input_grad[1]:
            _register_bias_user(w, ctx)
        return _library_or_own_forward(x, w.reshape(w.shape[-2:]))

    @staticmethod
    def backward(ctx, grad):
        x, w = ctx.saved_tensors
        g2 = grad.reshape(-1, grad.shape[-1]).contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        w2 = w.reshape(w.shape[-2:])
        rows, cin = x2.shape
        c = g2.shape[1]
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        plan = _dense_kernels.plan(rows, cin, c)
        if not x2.is_contiguous() or not w2.is_contiguous() or not (need_w and plan["dw"] == "mfma"):
            grad_x = _library_or_own_input_gradient(g2, w2).view(x.shape) if need_x else None
            grad_w = _weight_gradient_product(x2, g2).view(w.shape) if need_w else None
            return grad_x, grad_w
        ws = _dense_kernels.weight_workspace(rows, cin, c, x.device)
        grad_x = None
        late = None
        if need_x and plan["pair"]:
            grad_x = torch.empty_like(x)
            _dense_kernels.backward_pair(x2, g2, w2, grad_x.view(rows, cin), ws)
        else:
            if need_x and _may_postpone_input_gradient(x):
                # no gradient for x through the engine: the end-of-pass callback launches the product behind the reduction
                # launch and makes its buffer the leaf's .grad (_flush_late)
                late = (g2, w2, torch.empty(rows, cin, dtype=x.dtype, device=x.device), torch.cuda.current_stream(x.device),
                        weakref.ref(x))
            elif need_x:
                grad_x = _library_or_own_input_gradient(g2, w2).view(x.shape)
            _dense_kernels.backward_weight_partials(x2, g2, ws)
        if late is not None:
            task = torch._C._current_graph_task_id()
            _pending_late.setdefault(task, []).append(late)
            _register_flush(task)
        return grad_x, _finish_weight_gradient(ctx.w_ref, w, rows, cin, c, ws)


class _DenseAny(torch.autograd.Function):
    """support = input @ W and both gradients on the any-shape matrix-core kernel (csrc/dense_any.hip): the layers whose
    widths the 192-column kernels do not take -- the mesh encoder's 3 / 60 / ... / 300-wide ZERON_GCN layers, whose weight
    gradients (18 432 summed rows against a 300 x 300 output) the library runs without a split, 73-97 us each."""

    @staticmethod
    def forward(ctx, x, w):
        x2 = x.reshape(-1, x.shape[-1])
        w2 = w.reshape(w.shape[-2:])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        ctx.save_for_backward(x2, w)
        ctx.x_shape = x.shape
        return _forward_product(x2, w2).view(x.shape[:-1] + (w2.shape[1],))

    @staticmethod
    def backward(ctx, grad):
        x2, w = ctx.saved_tensors
        w2 = w.reshape(w.shape[-2:])
        g2 = grad.reshape(-1, grad.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        grad_x = grad_w = None
        if ctx.needs_input_grad[0]:
            grad_x = _dense_kernels.gemm(g2, w2, trans_b=True).view(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            grad_w = _dense_kernels.gemm(x2, g2, trans_a=True).view(w.shape)
        return grad_x, grad_w


def _dense(x, w):
    """input @ weight of a 0N-GCN layer; w = the layer's weight parameter ([Cin, Cout] or [1, Cin, Cout])."""
    arena = current_slabs()
    plain = (not x.is_cuda or x.dtype != torch.float32 or w.dtype != torch.float32
             or not (w.dim() == 2 or (w.dim() == 3 and w.shape[0] == 1))
             or not (x.requires_grad or w.requires_grad) or not torch.is_grad_enabled())
    if not plain and arena is None and use_matrix_core_products:
        rows = x.numel() // x.shape[-1]
        if w.requires_grad and _dense_kernels.plan(rows, x.shape[-1], w.shape[-1])["dw"] == "mfma":
            return _DenseMM.apply(x, w)
        if w.is_contiguous() and _takes_any_shape_kernel(rows, x.shape[-1], w.shape[-1]):
            return _DenseAny.apply(x, w)
    if plain or arena is None:
        # ([1,Cin,Cout]: one GEMM, not B broadcast bmm's; the same kernel as the differentiable routes pick for the shape)
        return _forward_product(x, w.squeeze(0) if w.dim() == 3 else w)
    return _Dense.apply(x, w, arena)


# ---- a stack of layers with its layer BOUNDARIES as single launches (csrc/zn_stack.hip) -----------------------------------
# Between two consecutive 192-wide layers the aggregation of the first is the operand load of the second's product, and in the
# backward pass the aggregation backward of a layer is the operand load of its own input-gradient product.  The boundary
# launch does both; which boundaries take it is `fused.plan`'s decision (measured: it pays from ~6 meshes of 2562 vertices
# forward, ~10 backward).  The values are those of the separate operators: the aggregation's bit for bit, the products within
# fp32 summation order.
class _StackLink:
    """What two neighbouring launches of a stack hand each other outside autograd's edges: the boundary's forward launch
    leaves its weight transposed (`wt`); the launch ABOVE it in the backward pass (the next boundary's or the head's fused
    backward) computes this boundary's input gradient with it and leaves it in `dx`, stamped with the address of the support
    gradient it was computed from -- the boundary uses it only when that very tensor arrives as its incoming gradient (an
    engine that summed several consumers' gradients hands over another tensor, and the product is computed here as usual)."""
    __slots__ = ("wt", "dx", "g_ref", "g_version", "wanted")

    def __init__(self):
        self.wt = self.dx = self.g_ref = None
        self.g_version = -1
        self.wanted = False

    def stamp(self, g):
        """`dx` was computed from the support gradient `g`: remember the tensor itself (held, so that its address cannot be
        recycled) and its version counter (an engine that accumulates another consumer's gradient IN PLACE keeps the address
        and bumps the version)."""
        self.g_ref, self.g_version = g, g._version

    def claim_dx(self, g):
        """The precomputed input gradient if `g` is the very tensor (same memory, same version) it was computed from."""
        dx, ref, ver = self.dx, self.g_ref, self.g_version
        self.dx = self.g_ref = None
        if dx is None or ref is None:
            return None
        same = (g.data_ptr() == ref.data_ptr() and g.shape.numel() == ref.shape.numel() and g._version == ver
                and ref._version == ver)
        return dx if same else None

    def take_dx(self, b, nv):
        self.dx = torch.empty(b, nv, self.wt.shape[1], dtype=torch.float32, device=self.wt.device)
        return self.dx


class _FusedBoundary(torch.autograd.Function):
    """support_next = act([A . S[..., :k] | S[..., k:]] + bias) @ W_next -- layer L's aggregation and layer L+1's product."""

    @staticmethod
    def forward(ctx, support, bias, w_next, csr, k, act, up, down):
        s = _lib.require(support, "support", torch.float32, 3)
        b, nv, c = s.shape
        bias_c = None if bias is None else _lib.require(bias, "bias", torch.float32, 1)
        w2 = w_next.reshape(w_next.shape[-2:])
        need = any(ctx.needs_input_grad[:3])
        mask = None
        if need and act == _ACT_RELU:
            mask = torch.empty(_lib.lib().geom_zn_gcn_relu_mask_words(b, nv, c, k), dtype=torch.int16, device=s.device)
        if need:
            up.wt = torch.empty(w2.shape[1], c, dtype=torch.float32, device=s.device)
            up.wanted = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        x, s_next = _fused.layer_forward(s, bias_c, csr, k, act, w2, mask=mask, wt_out=up.wt)
        ctx.csr, ctx.k, ctx.act, ctx.up, ctx.down = csr, k, act, up, down
        ctx.bias_ref = None if bias is None else weakref.ref(bias)
        ctx.w_ref = weakref.ref(w_next)
        if bias is not None and ctx.needs_input_grad[1]:
            _register_bias_user(bias, ctx)
        if ctx.needs_input_grad[2]:
            _register_bias_user(w_next, ctx)
        ctx.masked = mask is not None
        ctx.save_for_backward(x, w_next, *([mask] if mask is not None else []))
        return s_next

    @staticmethod
    def backward(ctx, grad_next):
        x, w = ctx.saved_tensors[:2]
        mask = ctx.saved_tensors[2] if ctx.masked else None
        b, nv, c = x.shape
        rows = b * nv
        w2 = w.reshape(w.shape[-2:])
        n_out = w2.shape[1]
        g2 = grad_next.reshape(rows, n_out).contiguous()
        x2 = x.view(rows, c)
        need_s, need_b, need_w = ctx.needs_input_grad[:3]
        need_b = need_b and ctx.bias_ref is not None
        up, down, csr, k, act = ctx.up, ctx.down, ctx.csr, ctx.k, ctx.act
        # ---- layer L+1's product: dW partials, and dX unless the launch above left it in the link
        dx = up.claim_dx(g2)
        plan = _dense_kernels.plan(rows, c, n_out)
        grad_w = ws = None
        split = need_w and plan["dw"] == "mfma" and w2.is_contiguous()
        if split:
            ws = _dense_kernels.weight_workspace(rows, c, n_out, x.device)
        if dx is None and (need_s or need_b):
            if split and plan["pair"]:
                dx = torch.empty_like(x)
                _dense_kernels.backward_pair(x2, g2, w2, dx.view(rows, c), ws)
            else:
                dx = torch.matmul(g2, w2.t()).view(x.shape)
                if split:
                    _dense_kernels.backward_weight_partials(x2, g2, ws)
        elif split:
            _dense_kernels.backward_weight_partials(x2, g2, ws)
        if split:
            grad_w = _finish_weight_gradient(ctx.w_ref, w, rows, c, n_out, ws)
        elif need_w:
            grad_w = _weight_gradient_product(x2, g2).view(w.shape)
        if not (need_s or need_b):
            return None, None, grad_w, None, None, None, None, None
        # ---- layer L's aggregation: with the input gradient of ITS product in the same launch when the boundary below wants it
        bias = ctx.bias_ref() if ctx.bias_ref is not None else None
        out = x if (act != _ACT_NONE and mask is None) else None
        if down is not None and down.wt is not None and down.wanted and _fused.plan(rows)["bwd"]:
            grad_bias = partial = None
            prows = _fused.partial_rows(b, nv)
            if need_b:
                grad_bias = _gradient_buffer(bias, bias) if bias is not None and bias.shape == (c,) \
                    else torch.empty(c, dtype=torch.float32, device=x.device)
                partial = torch.empty(prows, c, dtype=torch.float32, device=x.device)
            grad_support, _ = _fused.layer_backward(dx, out, mask, csr, k, act, down.wt, grad_in=down.take_dx(b, nv),
                                                    colsum_partial=partial)
            down.stamp(grad_support)
            if need_b:
                _finish_colsum(partial, prows, c, grad_bias, bias, _may_defer(bias))
        else:
            grad_support, grad_bias = aggregate_backward(dx, csr, k, act, out, mask, need_b, bias)
        return (grad_support if need_s else None), grad_bias, grad_w, None, None, None, None, None


def _boundary_fuses(x, csr, prev, layer, act, activation):
    """Whether the boundary between `prev` and `layer` takes the single launch."""
    if not (use_matrix_core_products and current_slabs() is None and isinstance(csr, _Csr)):
        return False
    if activation is not None and act == _ACT_NONE:          # a foreign callable: applied by the caller between the operators
        return False
    w = layer._weight()
    c = prev._weight().shape[-1]
    if w.dtype != torch.float32 or w.shape[-2] != c or not w.is_contiguous() or prev.split != 3 or c % 3:
        return False
    rows = x.numel() // x.shape[-1]
    return _fused.supported(csr, c, c // 3, w.shape[-1]) and _fused.plan(rows)["fwd"]


def _stack_supports(x, adj, stack, head):
    """The front of a stack up to the support of its last layer: (support, link of the last boundary, csr)."""
    act = _ACT_NONE if head["activation"] is None else _activation_code(head["activation"])
    activation = head["activation"]
    batched = torch.is_tensor(x) and x.dim() == 3 and x.is_cuda and x.dtype == torch.float32 \
        and not (torch.is_tensor(adj) and adj.dim() == 3)
    csr = adjacency_csr(adj) if batched else None
    s = _dense(x, stack[0]._weight())
    link = None
    for prev, layer in zip(stack[:-1], stack[1:]):
        if batched and _boundary_fuses(x, csr, prev, layer, act, activation):
            up = _StackLink()
            s = _FusedBoundary.apply(s, prev.bias, layer._weight(), csr, s.shape[-1] // prev.split, act, up, link)
            link = up
        else:
            h = zero_n_aggregate(s, adj, prev.bias, s.shape[-1] // prev.split, activation)
            s = _dense(h, layer._weight())
            link = None
    return s, link


def zero_n_stack(x, adj, stack, activation):
    """stack[-1](... stack[1](stack[0](x, adj, activation), adj, activation) ...): consecutive 0N-GCN layers applied to one
    adjacency (GEOMetrics.py:117-131 runs such runs per deformation stage), their boundaries as single launches where
    `fused.plan` says so.  Same values as calling the layers one by one."""
    s, _ = _stack_supports(x, adj, stack, {"activation": activation})
    return zero_n_aggregate(s, adj, stack[-1].bias, s.shape[-1] // stack[-1].split, activation)


def zero_n_stack_positions(x, adj, stack, activation, base, scale):
    """base + scale * zero_n_stack(x, adj, stack, activation)[..., :3] -- the coordinate update of a deformation stage."""
    s, link = _stack_supports(x, adj, stack, {"activation": activation})
    return zero_n_aggregate_head(s, adj, stack[-1].bias, s.shape[-1] // stack[-1].split, activation, base, scale, down=link)


def _uniform(t, bound):
    t.data.uniform_(-bound, bound)


# ------------------------------------------------------------------ layers ----
class _ZeroNBase(Module):
    """Common body: `split` = denominator of the aggregated column fraction (10 or 3)."""
    split = 10

    def _weight(self):
        raise NotImplementedError

    def forward(self, input, adj, activation):
        support = _dense(input, self._weight())
        return zero_n_aggregate(support, adj, self.bias, support.shape[-1] // self.split, activation)

    def forward_positions(self, input, adj, activation, base, scale):
        """base + scale * self(input, adj, activation)[..., :3] -- the coordinate update of a deformation stage
        (GEOMetrics.py:121,126,131) without materialising the layer output's gradient (see zero_n_aggregate_head)."""
        support = _dense(input, self._weight())
        return zero_n_aggregate_head(support, adj, self.bias, support.shape[-1] // self.split, activation, base, scale)


class ZERON_GCN(_ZeroNBase):
    """Unbatched layer, first C//10 output channels neighbour-aggregated (reference layers.py:14-41)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        _uniform(self.weight, 6.0 / math.sqrt(self.weight.size(0) + self.weight.size(1)))
        if self.bias is not None:
            self.bias.data.zero_()

    def _weight(self):
        return self.weight


class BatchZERON_GCN(ZERON_GCN):
    """Batched [B,V,Cin] input, same parameters and split (reference layers.py:121-152)."""


class Batch_Image_ZERON_GCNGCN(_ZeroNBase):
    """Batched layer with a [1,Cin,Cout] `weight1` and the first C//3 channels aggregated
    (reference layers.py:84-116; the initialiser's quirk of using size(0)==1 is kept)."""
    split = 3

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight1 = Parameter(torch.empty(1, in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        _uniform(self.weight1, 0.3 * 6.0 / math.sqrt(self.weight1.size(1) + self.weight1.size(0)))
        if self.bias is not None:
            _uniform(self.bias, 0.1)

    def _weight(self):
        return self.weight1


class _MaxPoolBase(Module):
    """Aggregate like a 0N-GCN layer (split 10), then max over the vertex axis."""

    def __init__(self, in_features, print_length):
        super().__init__()
        self.in_features, self.print_length = in_features, print_length
        self.weight_Ws = nn.ParameterList([Parameter(torch.empty(in_features, print_length))])
        self.weight_Bs = nn.ParameterList([Parameter(torch.empty(print_length))])
        self.reset_parameters()

    def reset_parameters(self):
        _uniform(self.weight_Bs[0], 6.0 / math.sqrt(self.weight_Bs[0].size(0)))
        _uniform(self.weight_Ws[0], 6.0 / math.sqrt(self.weight_Ws[0].size(0) + self.weight_Ws[0].size(1)))

    def _pre_activation(self, r_s, adj):
        support = _dense(r_s, self.weight_Ws[0])
        return zero_n_aggregate(support, adj, self.weight_Bs[0], support.shape[-1] // 10)


class GCNMax(_MaxPoolBase):
    """Unbatched: max over vertices of activation(v) (reference layers.py:43-79).  Given a RaggedMeshBatch as
    `adj` (r_s = the concatenated [sum(V), Cin] features) it returns one row per mesh, [B, print_length]."""

    def forward(self, r_s, adj, activation):
        support = _dense(r_s, self.weight_Ws[0])
        acted = zero_n_aggregate(support, adj, self.weight_Bs[0], support.shape[-1] // 10, activation)
        if hasattr(adj, "offsets"):
            from .ops import SegmentMax
            return SegmentMax.apply(acted, adj.offsets, adj.max_len)
        return torch.max(acted, dim=0)[0]


class BatchGCNMax(_MaxPoolBase):
    """Batched: max over vertices of the PRE-activation values -- the reference computes the
    activation and then discards it (layers.py:186-187); preserved."""

    def forward(self, r_s, adj, activation):
        return torch.max(self._pre_activation(r_s, adj), dim=1)[0]
