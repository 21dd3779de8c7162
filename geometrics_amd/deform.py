"""The hidden layers of the mesh deformation block as ONE launch per layer and direction (csrc/deform_block.hip).

reference: models.py:237-297 -- `x = F.relu(self.bnK(self.gcK(features, adj, F.relu)))` thirteen times (gcK: layers.py:107-116,
bnK = nn.BatchNorm1d(verts)), `features = features + x; features /= 2` after every second layer.  As separate operators a
hidden layer is product -> aggregation -> per-vertex BatchNorm (three launches each way, the activation written and re-read
between them); here layer i's launch computes

    Z_i = aggregate(S_i) + bias_i ;  X_{i+1} = ReLU(BN_i(Z_i)) (+ residual, / 2) ;  S_{i+1} = X_{i+1} . W_{i+1}

and its backward launch the aggregation backward + input-gradient product of layer i + 1 and the BatchNorm backward of layer i.
The chain is one autograd node: its input is the FIRST layer's raw support S_1 = cat(features, pooled) . W_1 (that product and
its gradients stay with layers._dense), its output the block's final features X_14; the coordinate head gc15 stays the
existing layer.  Weight gradients of the twelve equal layers: one strided-batched product over the stacked activations and
support gradients; bias gradients: per-vertex column sums out of the backward launches, added up by one reduction.

Where every workgroup of a launch is resident at once (482 vertices on an MI355X: yes) the thirteen launches of a direction
are ONE (`chain`): a vertex's workgroup runs layer after layer and waits for its neighbours' rows inside the launch.

`serves()` says when the launches apply (192-wide block, k = 64, b <= 16, bounded-degree table of width 8, training mode,
local BatchNorm statistics, fp32 on a HIP device); everything else takes the separate operators (models.py).
"""
import ctypes
import os
import threading

import torch

from . import _lib
from . import layers as _layers

# models.py:252-292: which earlier activation a layer's output is averaged with.  Key: layer; value: "lead" (the leading 192
# columns of the block input) or j = X_j, the INPUT of layer j (X_{j} = output of layer j - 1).
RESIDUALS = {2: "lead", 4: 3, 6: 5, 8: 7, 10: 9, 12: 11, 13: 13}
LAYERS = 13

enabled = True      # tests / A-B timing: False keeps the separate operators
relu = True         # tests only: False drops the ReLU of every hidden layer (a smooth chain: every launch of the backward can then
#                     be held to a tight float64 bound -- under ReLU a pre-activation within rounding of zero may fall on either
#                     side in any two fp32 evaluations and switch a whole unit's term)


def serves(block, features, pooled, csr):
    """Whether the fused launches serve this call of `block` (a models.BatchMeshDeformationBlock)."""
    if not enabled or block.hidden != 192 or not block.training or not torch.is_grad_enabled():
        return False
    if not (features.is_cuda and features.dtype == torch.float32 and pooled.dtype == torch.float32 and features.dim() == 3):
        return False
    if features.shape[0] > 16 or features.shape[0] * features.shape[1] * 192 >= 2 ** 29:
        return False
    if csr.ell_w != 8 or _tail_tables(csr) is False:
        return False
    for i in range(1, LAYERS + 1):
        gc, bn = getattr(block, "gc%d" % i), getattr(block, "bn%d" % i)
        if gc.bias is None or gc.weight1.shape[-1] != 192 or (i > 1 and gc.weight1.shape[-2] != 192) or bn._synchronised():
            return False
    return True


def _p(t):
    return None if t is None else t.data_ptr()


TAIL = 32      # GEOM_DEFORM_TAIL: entries of an adjacency row beyond the 8-wide neighbour table that the launches take


def _tail_tables(csr):
    """(tail_col, tail_val, tail_col_t, tail_val_t): the entries of every row beyond the neighbour table as a second,
    TAIL-wide table [nv, TAIL] (col -1 = padding) for A and A^T, or Nones where no row is longer; False when a row has more
    than 8 + TAIL entries (the launches then do not serve the adjacency).  Cached on the csr object."""
    cached = getattr(csr, "_deform_tail", None)
    if cached is not None:
        return cached
    out = []
    for over in (csr.over, csr.over_t):
        if over is None:
            out += [None, None]
            continue
        ptr, col, val = over
        lens = (ptr[1:] - ptr[:-1]).long()
        if int(lens.max()) > TAIL:
            csr._deform_tail = False
            return False
        nv = lens.numel()
        rows = torch.repeat_interleave(torch.arange(nv, device=col.device), lens)
        slot = torch.arange(col.numel(), device=col.device) - ptr[:-1].long()[rows]
        tcol = torch.full((nv, TAIL), -1, dtype=torch.int32, device=col.device)
        tval = torch.zeros((nv, TAIL), dtype=torch.float32, device=col.device)
        tcol[rows, slot] = col
        tval[rows, slot] = val
        out += [tcol.contiguous(), tval.contiguous()]
    csr._deform_tail = tuple(out)
    return csr._deform_tail


def pack_weights(weights, zero=None):
    """(fwd, bwd) [len(weights), 36864] each: every [192,192] weight (and its transpose) in the order a wave of the layer
    launches keeps its slice in registers -- ONE launch for all layers of a block (geom_deform_pack_weights_zero_f32).
    zero (optional int32 tensor): cleared by the same launch (the chain launches' counters)."""
    n = len(weights)
    w2 = [w.reshape(192, 192) if w.is_contiguous() else w.reshape(192, 192).contiguous() for w in weights]
    dev = w2[0].device
    fwd = torch.empty(n, 192 * 192, dtype=torch.float32, device=dev)
    bwd = torch.empty(n, 192 * 192, dtype=torch.float32, device=dev)
    ptrs = (ctypes.c_void_p * n)(*[w.data_ptr() for w in w2])
    with torch.cuda.device(dev):
        _lib.call("geom_deform_pack_weights_zero_f32", n, ptrs, fwd.data_ptr(), bwd.data_ptr(), _p(zero),
                  0 if zero is None else zero.numel())
    return fwd, bwd


# The hidden layers of a block as ONE launch per direction (geom_deform_chain_fwd_f32: a vertex's workgroup waits for its
# neighbours' rows inside the launch) where every workgroup of the launch is resident at once; False: one launch per layer.
#
# ONE chain launch at a time per device: a chain launch needs all its workgroups resident, and two of them that start together
# on two streams can each hold half of the chip and wait for the other half (the waits give up after seconds and the outputs
# are NaN -- loud, but a lost step).  A process therefore gives the chain launches of a device to ONE stream at a time: the
# stream that asks first owns them until it is idle, other streams issue the layers one by one meanwhile.  A stream that is being
# captured always records chain launches.  (Not covered: two PROCESSES sharing a GPU, and captured graphs replayed concurrently
# with other chain work -- set deform.chain = False (or GEOM_DEFORM_CHAIN=0 in the environment) there.)
chain = os.environ.get("GEOM_DEFORM_CHAIN", "1") != "0"
CTR_STRIDE = 32      # ints per vertex counter (one 128-byte line each)
_chain_owner = {}    # device index -> the torch stream that issues chain launches
_chain_lock = threading.Lock()


def chain_fits(nv, device):
    if not chain:
        return False
    with torch.cuda.device(device):
        if not _lib.lib().geom_deform_chain_fits(int(nv)):
            return False
        mine = torch.cuda.current_stream(device)
        if torch.cuda.is_current_stream_capturing():
            return True      # (nothing runs now; whoever replays the graph keeps other chain work off the device meanwhile)
    with _chain_lock:
        owner = _chain_owner.get(device.index)
        if owner is None or owner == mine:
            _chain_owner[device.index] = mine
            return True
        try:                          # the owner has nothing in flight any more: the launches move to this stream
            idle = owner.query()
        except RuntimeError:          # (it is being captured)
            idle = False
        if idle:
            _chain_owner[device.index] = mine
        return idle


def _forward_args(s_in, bias, csr, bn_w, bn_b, run_mean, run_var, training, momentum, eps, relu, res, scale, z_out, x_out,
                  save_mean, save_invstd, w_next=None, s_out=None, w_head=None, s_head=None):
    b, nv, c = s_in.shape
    tail = _tail_tables(csr)
    return _lib.DeformFwd(b, nv, c, 64, csr.ell_w, _p(s_in), _p(bias), _p(csr.ell_col), _p(csr.ell_val), _p(tail[0]), _p(tail[1]),
                          _p(bn_w), _p(bn_b), _p(run_mean), _p(run_var), int(training), float(momentum), float(eps),
                          int(relu), _p(res), res.stride(1) if res is not None else 0, float(scale), _p(z_out), _p(x_out),
                          _p(save_mean), _p(save_invstd), _p(w_next), _p(s_out), _p(w_head), _p(s_head), 0)


def layer_forward(s_in, *args, **kwargs):
    """One forward launch (geom_deform_layer_fwd_f32); w_next = the next layer's weight PACKED (pack_weights()[0][l]); see
    include/geom_hip.h for the operands."""
    a = _forward_args(s_in, *args, **kwargs)
    with torch.cuda.device(s_in.device):
        _lib.call("geom_deform_layer_fwd_f32", ctypes.addressof(a))


def chain_forward(layers, done, device):
    """`layers` (lists of layer_forward's arguments) as ONE launch (geom_deform_chain_fwd_f32); done: int32 [nv * CTR_STRIDE],
    zero."""
    structs = (_lib.DeformFwd * len(layers))(*[_forward_args(*a, **k) for a, k in layers])
    with torch.cuda.device(device):
        _lib.call("geom_deform_chain_fwd_f32", len(layers), ctypes.addressof(structs), done.data_ptr())


def _rows192(t, shape):
    """(tensor, row pitch in floats) of a [B,V,192] gradient as the backward launch reads it: in place when it is row-major
    with contiguous rows at any pitch (a column slice of a wider buffer), a contiguous copy otherwise."""
    if t is None:
        return None, 0
    b, nv, c = shape
    ok = (t.dim() == 3 and tuple(t.shape) == (b, nv, c) and t.is_cuda and t.dtype == torch.float32 and t.stride(2) == 1
          and t.stride(1) >= c and t.stride(0) == nv * t.stride(1) and t.data_ptr() % 4 == 0)
    if not ok:
        t = t.contiguous()
    return t, t.stride(1)


def _backward_args(shape, csr, z, bn_w, bn_b, save_mean, save_invstd, relu, has_res, scale, dz, grad_bn_w, grad_bn_b,
                   dz_up=None, ds_up=None, wt_up=None, g=None, g2=None, grad_res=None, colsum=None, ds_head=None, w_head=None,
                   x_top=None, dw_head=None):
    """(struct, tensors it points into that were made here: the caller keeps them alive until the launch is issued)"""
    b, nv, c = shape
    tail = _tail_tables(csr)
    g, g_ld = _rows192(g, shape)
    g2, g2_ld = _rows192(g2, shape)
    a = _lib.DeformBwd(b, nv, c, 64, csr.ell_w, _p(dz_up), _p(csr.ell_col_t), _p(csr.ell_val_t), _p(tail[2]), _p(tail[3]),
                       _p(ds_up), _p(wt_up), _p(g), _p(g2), g_ld, g2_ld, _p(z), _p(bn_w), _p(bn_b), _p(save_mean),
                       _p(save_invstd), int(relu), int(has_res), float(scale), _p(grad_res), _p(dz), _p(grad_bn_w),
                       _p(grad_bn_b), _p(colsum), _p(ds_head), _p(w_head), _p(x_top), _p(dw_head), 0)
    return a, (g, g2)


def layer_backward(shape, csr, z, *args, **kwargs):
    """One backward launch (geom_deform_layer_bwd_f32); wt_up = the layer above's weight, TRANSPOSED and packed
    (pack_weights()[1][l])."""
    a, _keep = _backward_args(shape, csr, z, *args, **kwargs)
    with torch.cuda.device(z.device):
        _lib.call("geom_deform_layer_bwd_f32", ctypes.addressof(a))


def chain_backward(layers, done, device, ds_first=None):
    """`layers` (lists of layer_backward's arguments, the top layer first) as ONE launch (geom_deform_chain_bwd_f32); ds_first:
    receives the aggregation backward of the last layer's dZ (the chain's first layer has no activation behind its support)."""
    built = [_backward_args(*a, **k) for a, k in layers]
    structs = (_lib.DeformBwd * len(built))(*[b[0] for b in built])
    with torch.cuda.device(device):
        _lib.call("geom_deform_chain_bwd_f32", len(built), ctypes.addressof(structs), done.data_ptr(), _p(ds_first))


class _HiddenChain(torch.autograd.Function):
    """X_14 = the thirteen hidden layers applied to S_1 (see the module docstring).  apply(s1, lead, csr, stats, momentum,
    eps, *biases[13], *weights[12] (gc2..gc13), *bn_weights[13], *bn_biases[13]); stats = [(running_mean, running_var)] * 13.
    Returns the final features TWICE (two tensor objects over one memory: one for the coordinate head, one for the caller;
    their gradients meet inside the first backward launch instead of in an add pass)."""

    @staticmethod
    def forward(ctx, s1, lead, csr, stats, momentum, eps, w_head, *params):
        """w_head: the coordinate head's weight ([1,192,3] / [192,3]: models.py:219 gc15) or None.  Given, the head's product
        rides in the last layer's launch and the SECOND output is its raw support [B,V,3] (the caller aggregates it and adds
        the bias) instead of a second handle on the features."""
        L = LAYERS
        biases, weights = params[:L], params[L:2 * L - 1]
        bn_w, bn_b = params[2 * L - 1:3 * L - 1], params[3 * L - 1:4 * L - 1]
        s1 = _lib.require(s1, "s1", torch.float32, 3, 192)
        lead, _ = _rows192(lead, tuple(s1.shape))       # in place when it is the leading columns of the block's wide input
        b, nv, c = s1.shape
        dev = s1.device
        f32 = dict(dtype=torch.float32, device=dev)
        xs = torch.empty(L, b, nv, c, **f32)          # xs[i - 1] = X_{i+1}, the output of layer i
        zs = torch.empty(L, b, nv, c, **f32)          # zs[i - 1] = Z_i, what BN_i normalised
        means, invstds = torch.empty(L, nv, **f32), torch.empty(L, nv, **f32)
        s_buf = (torch.empty(b, nv, c, **f32), torch.empty(b, nv, c, **f32))
        s_cur = s1
        as_chain = chain_fits(nv, dev)
        # (counters of the forward and of the backward chain launch, cleared by the packing launch)
        counters = torch.empty(2, nv * CTR_STRIDE, dtype=torch.int32, device=dev) if as_chain else None
        w2, wts = pack_weights(weights, counters)     # w2[i - 1] / wts[i - 1] = W_{i+1} / its transpose, in register-slice order
        calls = []
        for i in range(1, L + 1):
            src = RESIDUALS.get(i)
            res = None if src is None else (lead if src == "lead" else xs[src - 2])
            nxt = i < L
            head = w_head is not None and not nxt
            if head:
                wh = w_head.reshape(c, 3)
                wh = wh if wh.is_contiguous() else wh.contiguous()
                s_head = torch.empty(b, nv, 3, **f32)
            call = ((s_cur, biases[i - 1], csr, bn_w[i - 1], bn_b[i - 1], stats[i - 1][0], stats[i - 1][1], True, momentum, eps,
                     relu, res, 0.5, zs[i - 1], xs[i - 1], means[i - 1], invstds[i - 1]),
                    dict(w_next=w2[i - 1] if nxt else None, s_out=s_buf[i & 1] if nxt else None,
                         w_head=wh if head else None, s_head=s_head if head else None))
            if as_chain:
                calls.append(call)
            else:
                layer_forward(*call[0], **call[1])
            s_cur = s_buf[i & 1]
        if as_chain:
            chain_forward(calls, counters[0], dev)
        ctx.counters = counters[1] if as_chain else None
        ctx.csr, ctx.relu, ctx.head = csr, relu, w_head is not None
        ctx.head_shape = None if w_head is None else tuple(w_head.shape)
        ctx.set_materialize_grads(False)      # a handle nobody uses (the last block's features) arrives as None, not as 5.9 MB of zeros
        ctx.save_for_backward(xs, zs, means, invstds, wts, *bn_w, *bn_b, *([wh] if w_head is not None else []))
        out = xs[L - 1]
        if w_head is not None:
            return out, s_head
        return out, _layers._alias(out)

    @staticmethod
    def backward(ctx, g_a, g_b):
        L = LAYERS
        saved = ctx.saved_tensors
        xs, zs, means, invstds, wts = saved[:5]
        bn_w, bn_b = saved[5:5 + L], saved[5 + L:5 + 2 * L]
        csr = ctx.csr
        _, b, nv, c = xs.shape
        dev = xs.device
        f32 = dict(dtype=torch.float32, device=dev)
        n_in = 7 + 4 * L - 1
        if g_a is None and g_b is None:
            return (None,) * n_in
        ds_head = w_head = dw_head = None
        if ctx.head:                       # g_b is the gradient of the head's raw support [B,V,3]
            w_head = saved[5 + 2 * L]
            if g_b is not None:
                ds_head = g_b.contiguous()
                dw_head = torch.empty(nv, c * 3, **f32) if ctx.needs_input_grad[6] else None
            g_top, g_top2 = g_a, None
        else:
            g_top, g_top2 = (g_a, g_b) if g_a is not None else (g_b, None)      # (read in place at any row pitch: layer_backward)
        dzs = torch.empty(L, b, nv, c, **f32)         # dzs[i - 1] = dZ_i
        dss = torch.empty(L - 1, b, nv, c, **f32)     # dss[i - 2] = dS_i = gradient of layer i's raw support, i = 2..L
        g_bnw, g_bnb = torch.empty(L, nv, **f32), torch.empty(L, nv, **f32)
        colsum = torch.empty(L, nv, c, **f32)
        pending = {}                                   # j -> gradient that reaches X_j through a residual average
        g_lead = None
        # (the forward ran as one launch: so does the backward, if this stream still owns the device's chain launches)
        counters = ctx.counters
        # (its counters are good for ONE pass: a second backward over a retained graph takes the launches per layer)
        calls = [] if ctx.counters is not None and chain_fits(nv, dev) else None
        if calls is not None:
            ctx.counters = None
        for i in range(L, 0, -1):
            src = RESIDUALS.get(i)
            grad_res = torch.empty(b, nv, c, **f32) if src is not None else None
            common = dict(relu=ctx.relu, has_res=src is not None, scale=0.5, dz=dzs[i - 1], grad_bn_w=g_bnw[i - 1], grad_bn_b=g_bnb[i - 1],
                          grad_res=grad_res, colsum=colsum[i - 1])
            where = ((b, nv, c), csr, zs[i - 1], bn_w[i - 1], bn_b[i - 1], means[i - 1], invstds[i - 1])
            if i == L:
                what = dict(g=g_top, g2=g_top2, ds_head=ds_head, w_head=w_head if ds_head is not None else None,
                            x_top=xs[L - 1] if dw_head is not None else None, dw_head=dw_head, **common)
            else:
                what = dict(dz_up=dzs[i], ds_up=dss[i - 1], wt_up=wts[i - 1], g2=pending.pop(i + 1, None), **common)
            if calls is None:
                layer_backward(*where, **what)
            else:
                calls.append((where, what))
            if src == "lead":
                g_lead = grad_res
            elif src is not None:
                if src in pending:
                    if calls is not None:
                        raise RuntimeError("two residual gradients for one layer: not expressible inside one launch")
                    pending[src] = pending[src] + grad_res
                else:
                    pending[src] = grad_res
        assert not pending, "a residual gradient was left without its layer"
        # dS_1 = aggregation backward of the first layer (no activation between S_1 and Z_1): a last step of the chain launch, or
        # the existing operator
        if calls is not None:
            g_s1 = torch.empty(b, nv, c, **f32)
            chain_backward(calls, counters, dev, ds_first=g_s1)
        else:
            g_s1, _ = _layers.aggregate_backward(dzs[0], csr, 64, _layers._ACT_NONE, None, None, False)
        rows = b * nv
        g_w = torch.bmm(xs[:L - 1].view(L - 1, rows, c).transpose(1, 2), dss.view(L - 1, rows, c))     # dW_i = X_i^T . dS_i, i = 2..L
        # bias gradients: the vertices' column sums added up in vertex order, all 13 layers in ONE launch (torch's reduction over
        # the middle axis of [13, 482, 192] took 38 us; geom_colsum_batch_f32 is the reduction the aggregation backward's
        # partials go through: fixed order, ~5 us)
        g_bias = torch.empty(L, c, **f32)
        jobs = [(colsum[i], c, g_bias[i]) for i in range(L)]
        g_head = None
        if dw_head is not None:            # the head's weight gradient: the vertices' [192, 3] partials added up by the same launch
            g_head = torch.empty(c * 3, **f32)
            jobs.append((dw_head, c * 3, g_head))
        n = len(jobs)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().geom_colsum_batch_f32(
                n, (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs]), (ctypes.c_int * n)(*([nv] * n)),
                (ctypes.c_int * n)(*[j[1] for j in jobs]), (ctypes.c_void_p * n)(*[j[2].data_ptr() for j in jobs]),
                _lib.stream_ptr()), "geom_colsum_batch_f32")
        grads = [g_bias[i] for i in range(L)] + [g_w[i].view(1, c, c) for i in range(L - 1)] \
            + [g_bnw[i] for i in range(L)] + [g_bnb[i] for i in range(L)]
        g_wh = None
        if g_head is not None:
            g_wh = g_head.view(ctx.head_shape)
        return (g_s1, g_lead, None, None, None, None, g_wh, *grads)


def hidden_chain(block, s1, lead, csr, head=None):
    """The thirteen hidden layers of `block` applied to the first layer's raw support; returns (features, features) -- see
    _HiddenChain -- or, with head = the block's coordinate layer (192 -> 3), (features, raw support of the head)."""
    L = LAYERS
    gcs = [getattr(block, "gc%d" % i) for i in range(1, L + 1)]
    bns = [getattr(block, "bn%d" % i) for i in range(1, L + 1)]
    for bn in bns:
        bn._pending_batches += 1
    stats = [(bn.running_mean, bn.running_var) for bn in bns]
    params = [g.bias for g in gcs] + [g.weight1 for g in gcs[1:]] + [bn.weight for bn in bns] + [bn.bias for bn in bns]
    w_head = None
    if head is not None and tuple(head.weight1.shape[-2:]) == (192, 3):
        w_head = head.weight1
    return _HiddenChain.apply(s1, lead, csr, stats, bns[0].momentum, bns[0].eps, w_head, *params)
