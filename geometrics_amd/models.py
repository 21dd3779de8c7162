"""The mesh deformation block of the reference (models.py:203-297) on the fused HIP kernels:
same attribute names, constructor and `forward(features, pooled, adj) -> (features, coords)`, and
the same state_dict keys (`gcN.weight1`, `gcN.bias`, `bnN.weight/bias/running_mean/running_var/
num_batches_tracked`), so the reference's checkpoints load unchanged.

Per layer pair the reference runs GEMM, dense adjacency product, cat, bias add, a two-pass
BatchNorm1d(verts), ReLU, add and divide as separate eager ops.  At the block's own width (192, batch <= 16: the
reference's training shape) every hidden layer is ONE launch per direction -- aggregation + BatchNorm1d(verts) + ReLU +
residual average + the next layer's product (geometrics_amd/deform.py, csrc/deform_block.hip); every other shape takes
GEMM -> one aggregation kernel (csrc/zn_gcn.hip) -> one per-vertex BN + ReLU (+ residual average) kernel (csrc/vertex_bn.hip).
A BatchNorm output that feeds the next layer AND a later residual average is handed out as two tensor objects over
the same memory (`tap`), so that its two upstream gradients meet inside the BN backward kernel instead of in a separate
accumulation pass; the block input's leading columns (the first residual) are tapped the same way (`_InputTap`).
(Measured and rejected: aggregation + BatchNorm in ONE launch -- LAB_NOTES.md section 4.)
Data-parallel note: by default every rank normalises with the statistics of its own shard (fused kernel, no collective,
graph-capturable: DDP semantics).  `VertexBatchNorm.sync_across_ranks = True` switches to the statistics of the GLOBAL
batch (`_SyncVertexBN`: per-rank mean and centred second moment combined by ONE all-reduce forward, one backward), i.e. N
shards normalise exactly as the single-GPU reference does over its whole batch -- capturable into the step's HIP graph with RCCL.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from . import layers as _layers
from .layers import Batch_Image_ZERON_GCNGCN, GCNMax, ZERON_GCN, _alias


def _identity(x):
    return x


class _VertexBN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, residual, training, momentum, eps, relu, scale, tap=False):
        xc = _lib.require(x, "x", torch.float32, 3)
        b, nv, c = xc.shape
        dev = xc.device
        res, res_ld = _residual_operand(residual, xc)   # a column slice of a wider row-major tensor is read in place
        out = _layers._new_like(xc, "normalised", _layers.current_slabs())   # a slot of the pass's stacked buffers when batching is on
        mean = torch.empty(nv, dtype=torch.float32, device=dev)
        invstd = torch.empty(nv, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("geom_vertex_bn_fwd_f32", b, nv, c, xc.data_ptr(), _lib.ptr(weight), _lib.ptr(bias),
                      _lib.ptr(running_mean), _lib.ptr(running_var), int(training), float(momentum), float(eps),
                      int(relu), _lib.ptr(res), res_ld, float(scale), out.data_ptr(), mean.data_ptr(),
                      invstd.data_ptr())
        if training:
            ctx.save_for_backward(xc, weight, bias, mean, invstd)
        else:
            inv = torch.rsqrt(running_var + eps)
            ctx.save_for_backward(xc, weight, bias, running_mean.clone(), inv)
        ctx.relu, ctx.scale, ctx.has_res, ctx.training = relu, scale, residual is not None, training
        # tap: the output twice (two tensor objects over the same memory) -- the caller gives one to the next layer and one
        # to the residual average two layers on; their gradients are summed inside the backward kernel
        return (out, _alias(out)) if tap else out

    @staticmethod
    def backward(ctx, *grads):
        x, weight, bias, mean, invstd = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("backward through the fused vertex BatchNorm is implemented for training mode only")
        given = [t.contiguous() for t in grads if t is not None]
        g, g2 = given[0], (given[1] if len(given) > 1 else None)
        b, nv, c = x.shape
        grad_x = torch.empty_like(x)
        grad_res = torch.empty_like(x) if ctx.has_res else None
        gw = torch.empty(nv, dtype=torch.float32, device=x.device) if weight is not None else None
        gb = torch.empty(nv, dtype=torch.float32, device=x.device) if bias is not None else None
        with torch.cuda.device(x.device):
            _lib.call("geom_vertex_bn_bwd_f32", b, nv, c, x.data_ptr(), g.data_ptr(), _lib.ptr(weight), _lib.ptr(bias),
                      mean.data_ptr(), invstd.data_ptr(), int(ctx.relu), int(ctx.has_res), float(ctx.scale),
                      grad_x.data_ptr(), _lib.ptr(grad_res), _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(g2))
        return grad_x, gw, gb, None, None, grad_res, None, None, None, None, None, None


def _residual_operand(residual, like):
    """The residual as the kernels read it: in place when it is a [B,V,C] fp32 tensor or a column slice of a wider
    row-major one (row stride = its leading dimension), a contiguous copy otherwise.  Returns (tensor, row stride)."""
    if residual is None:
        return None, like.shape[2]
    res = residual
    nv, c = like.shape[1], like.shape[2]
    ok = (res.dim() == 3 and res.shape == like.shape and res.stride(2) == 1 and res.stride(1) >= c
          and res.stride(0) == nv * res.stride(1) and res.dtype == torch.float32 and res.is_cuda)
    if not ok:
        res = res.contiguous()
    return res, res.stride(1)


class _InputTap(torch.autograd.Function):
    """full = cat(features, pooled) and, as a second output, a contiguous copy of its leading `width` columns -- the
    residual of the block's first layer pair (models.py:252: `features[:, :, :self.hidden]`).  As two autograd ops the
    slice's gradient comes back as a zero-filled tensor of the full width (35.6 MB at the training shape) that is then
    added to the first layer's input gradient; here the narrow gradient is added into the two input gradients directly."""

    @staticmethod
    def forward(ctx, features, pooled, width, lead_view=False):
        from . import ops as _ops
        nf = features.shape[-1]
        ctx.lead_view = bool(lead_view)
        # `pooled` allocated with room for the coordinates in front (utils.batched_pooling(headroom=...), utils.concat_features):
        # they are copied into the free columns and the wide buffer IS the input -- no 35 MB concatenation
        ctx.in_place = (features.dim() == 3 and features.is_cuda and features.dtype == torch.float32 and pooled.dtype == torch.float32
                        and features.shape[:2] == pooled.shape[:2] and _ops.headroom_of(pooled) == nf)
        if ctx.in_place:
            buf = _ops._headroom[pooled.untyped_storage().data_ptr()][0]()
            if not _ops._already_placed(buf, 0, features):      # (the pooling launch may have copied them: batched_pooling(fronts=))
                buf[..., :nf].copy_(features)
            _ops._register_headroom(buf, 0)
            full = buf
        else:
            full = torch.cat((features, pooled), dim=-1)
        if width > full.shape[-1]:
            raise RuntimeError("the block input has %d columns, fewer than the %d hidden ones" % (full.shape[-1], width))
        ctx.nf, ctx.width = nf, width
        ctx.shapes = (features.shape, pooled.shape)
        # the first residual = the input's leading columns: handed to the one-launch layers as a VIEW (they read any row pitch);
        # the separate operators' vector BatchNorm kernel wants 16-byte rows: a contiguous copy
        lead = full[..., :width]
        return full, (lead if ctx.lead_view else lead.contiguous())

    @staticmethod
    def backward(ctx, g_full, g_res):
        nf, width = ctx.nf, ctx.width
        lead = min(nf, width)
        gf = gp = None
        if g_full is not None and ctx.in_place and g_full.is_contiguous() and ctx.needs_input_grad[1]:
            # the consumers (pooling backward, the previous block's last layer, the sum of the coordinates' gradients) read their
            # column slices of the ONE input gradient in place; the residual's share is added into it with one launch
            # (nothing else reads g_full)
            if g_res is not None:
                g_full[..., :width].add_(g_res)
            return (g_full[..., :nf] if ctx.needs_input_grad[0] else None), g_full[..., nf:], None, None
        if ctx.needs_input_grad[0]:
            gf = g_full[..., :nf].clone() if g_full is not None else g_res.new_zeros(ctx.shapes[0])
            if g_res is not None and lead:
                gf[..., :lead] += g_res[..., :lead]
        if ctx.needs_input_grad[1]:
            gp = g_full[..., nf:].contiguous() if g_full is not None else g_res.new_zeros(ctx.shapes[1])
            if g_res is not None and width > nf:
                gp[..., :width - nf] += g_res[..., nf:width]
        return gf, gp, None, None


class _SyncVertexBN(torch.autograd.Function):
    """nn.BatchNorm1d(verts) over the GLOBAL batch of a data-parallel job: N shards of B/N meshes normalise exactly as ONE
    process holding all B meshes -- the reference's semantics (it is single-GPU: models.py:237-297 sees the whole batch).
    ONE all-reduce forward, ONE backward, both of per-vertex scalars; only tensor ops and collectives, no host reads: with
    the nccl (= RCCL) backend the whole thing is captured into a step's HIP graph like the gradient bucket's all-reduce.

    Forward statistics without the E[x^2] - mean^2 cancellation and without a second round trip: every rank forms its OWN
    mean and centred second moment M2 (two local passes, as the fused kernel does), and the ranks exchange
    (n mean, M2, n mean^2, n) per vertex -- the global moment is sum(M2) + [sum(n mean^2) - N mean_g^2], whose bracket is the
    BETWEEN-rank spread of the means; the packed statistics travel and are combined in float64 (a few KB), so that bracket
    is exact for the fp32 means that went in (activations with a mean of 30 and unit spread: tests/test_dist_gloo.py)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        import torch.distributed as dist
        nv = x.shape[1]
        n_local = float(x.shape[0] * x.shape[2])
        mean_l = x.mean(dim=(0, 2))
        m2_l = ((x - mean_l.view(1, -1, 1)) ** 2).sum(dim=(0, 2))
        mean64 = mean_l.double()
        packed = torch.cat((n_local * mean64, m2_l.double(), n_local * mean64 * mean64,
                            torch.full((1,), n_local, dtype=torch.float64, device=x.device)))
        dist.all_reduce(packed)
        n = packed[-1]
        s1, m2, q = packed[:nv], packed[nv:2 * nv], packed[2 * nv:3 * nv]
        mean_g = s1 / n
        var_g = (m2 + (q - n * mean_g * mean_g)).clamp_min(0.0) / n                         # biased, as used for normalisation
        mean, var = mean_g.to(x.dtype), var_g.to(x.dtype)
        invstd = torch.rsqrt(var + eps)
        with torch.no_grad():
            running_mean.mul_(1 - momentum).add_(momentum * mean)
            running_var.mul_(1 - momentum).add_(momentum * (var_g * (n / (n - 1).clamp_min(1.0))).to(x.dtype))
        xhat = (x - mean.view(1, -1, 1)) * invstd.view(1, -1, 1)
        ctx.save_for_backward(xhat, weight, invstd, n.to(x.dtype))
        return xhat * weight.view(1, -1, 1) + bias.view(1, -1, 1)

    @staticmethod
    def backward(ctx, grad_out):
        import torch.distributed as dist
        xhat, weight, invstd, n = ctx.saved_tensors
        local = torch.stack((grad_out.sum(dim=(0, 2)), (grad_out * xhat).sum(dim=(0, 2))), dim=1)
        grad_weight, grad_bias = local[:, 1].clone(), local[:, 0].clone()                  # local shard: DDP-style, reduced with the other gradients
        tot = local.clone()
        dist.all_reduce(tot)
        mg, mgx = (tot[:, 0] / n).view(1, -1, 1), (tot[:, 1] / n).view(1, -1, 1)
        grad_x = (weight * invstd).view(1, -1, 1) * (grad_out - mg - xhat * mgx)
        return grad_x, grad_weight, grad_bias, None, None, None, None


class VertexBatchNorm(nn.Module):
    """nn.BatchNorm1d(verts) for [B,V,C] activations with the ReLU and the block's residual average
    `(residual + relu(bn(x))) / 2` folded into the same kernel.  Parameters and buffers are named as in
    nn.BatchNorm1d, so `bnN.*` checkpoint entries load unchanged."""

    def __init__(self, verts, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = verts, eps, momentum
        self.weight = nn.Parameter(torch.ones(verts))
        self.bias = nn.Parameter(torch.zeros(verts))
        self.register_buffer("running_mean", torch.zeros(verts))
        self.register_buffer("running_var", torch.ones(verts))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending_batches = 0   # counted on the host; folded into the buffer when the state is saved

    # False (default): local-shard statistics on the fused kernels -- torch DDP's semantics, no collective inside forward /
    # backward.  True: under torch.distributed with > 1 rank the statistics are those of the GLOBAL batch (the reference is
    # single-GPU and normalises over its whole batch: exact N-shard == 1-process equivalence), at the price of two small
    # all-reduces per layer and step (one forward, one backward: _SyncVertexBN) and of the separate operators instead of the
    # one-launch layers; tensor ops and collectives only, so with RCCL the step is still captured into ONE HIP graph
    # (tests/test_dist_step_gpu.py).
    sync_across_ranks = False

    _warned_local_statistics = False

    _sync_single_rank_groups = False    # tests: take the synchronised route in a 1-rank group too (RCCL capture on one GPU)

    def _synchronised(self):
        many = (self.training and torch.distributed.is_available() and torch.distributed.is_initialized()
                and (torch.distributed.get_world_size() > 1 or VertexBatchNorm._sync_single_rank_groups))
        if many and not self.sync_across_ranks and not VertexBatchNorm._warned_local_statistics:
            VertexBatchNorm._warned_local_statistics = True       # once per process: the default changed in round 3
            import warnings
            warnings.warn("geometrics_amd VertexBatchNorm: %d ranks are active and sync_across_ranks is False -- every rank "
                          "normalises with the statistics of its OWN shard (DDP semantics), which is not what the single-GPU "
                          "reference computes over its whole batch; set VertexBatchNorm.sync_across_ranks = True for "
                          "global-batch statistics" % torch.distributed.get_world_size(), stacklevel=3)
        return many and self.sync_across_ranks

    MAX_VALUES_PER_VERTEX = 4096   # b * c held in the registers of one workgroup (csrc/vertex_bn.hip)

    def fused_kernel_serves(self, x):
        """True when the register-resident HIP kernels serve this call: device tensors with b*c <= 4096 values per
        vertex, local statistics, and not a gradient through an eval()'d block (the fused backward is written for batch
        statistics; frozen-BN fine-tuning takes the library ops)."""
        b, _, c = x.shape
        eval_grad = not self.training and torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)
        return x.is_cuda and b * c <= self.MAX_VALUES_PER_VERTEX and not eval_grad and not self._synchronised()

    def forward(self, x, relu=False, residual=None, scale=0.5, tap=False):
        """tap=True returns the result twice: on the HIP kernel two tensor objects over the same memory whose gradients
        are added inside the backward kernel (give one to each consumer); on the library routes the same tensor twice."""
        if self._synchronised() or not self.fused_kernel_serves(x):
            if self.training:
                self._pending_batches += 1      # nn.BatchNorm1d's num_batches_tracked, folded in when the state is saved
            if self._synchronised():
                y = _SyncVertexBN.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum, self.eps)
            else:                               # library ops, same maths
                y = nn.functional.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                                             self.training, self.momentum, self.eps)
            y = torch.relu(y) if relu else y
            y = (residual + y) * scale if residual is not None else y
            return (y, y) if tap else y
        if self.training:
            self._pending_batches += 1
        return _VertexBN.apply(x, self.weight, self.bias, self.running_mean, self.running_var, residual,
                               self.training, self.momentum, self.eps, relu, scale, tap)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._pending_batches:
            self.num_batches_tracked += self._pending_batches
            self._pending_batches = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)


class BatchMeshDeformationBlock(nn.Module):
    """Reference models.py:203-297: 13 hidden 0N-GCN layers (each followed by BatchNorm1d(verts) + ReLU,
    a residual average after every pair) and a coordinate head `gc15` (hidden -> output_features)."""

    def __init__(self, input_features, verts, hidden=192, output_features=3):
        super().__init__()
        self.hidden = hidden
        self.gc1 = Batch_Image_ZERON_GCNGCN(input_features, hidden)
        for i in range(2, 14):
            setattr(self, "gc%d" % i, Batch_Image_ZERON_GCNGCN(hidden, hidden))
        self.gc15 = Batch_Image_ZERON_GCNGCN(hidden, output_features)
        for i in range(1, 15):                       # bn14 exists in the reference (unused) -- kept for the keys
            setattr(self, "bn%d" % i, VertexBatchNorm(verts))

    def _layer(self, i, x, adj, residual=None, tap=False):
        """gc_i -> bn_i -> ReLU (-> residual average); tap: see VertexBatchNorm.forward."""
        return getattr(self, "bn%d" % i)(getattr(self, "gc%d" % i)(x, adj, _identity), relu=True, residual=residual, tap=tap)

    # How the twelve equal hidden layers get their weight gradients: True = ONE strided-batched library product at the end of
    # the backward pass (layers.weight_gradient_batching), False = the matrix-core pair launch per layer (input gradient and
    # weight-gradient partials together, csrc/dense_gemm.hip) + the shared end-of-pass reduction.  Measured at the reference's
    # training shape (batch 16 x 482 vertices) and at the BASELINE shard: profiles/r04_training_shape_variants.txt.
    batch_weight_gradients = True

    def forward(self, features, pooled, adj):
        import contextlib
        from . import deform as _deform
        batching = _layers.weight_gradient_batching(depth=14) if self.batch_weight_gradients else contextlib.nullcontext()
        csr = _layers.adjacency_csr(adj) if (torch.is_tensor(adj) and adj.dim() == 2 and features.is_cuda) else None
        if csr is not None and _deform.serves(self, features, pooled, csr):
            # ONE launch per hidden layer and direction (csrc/deform_block.hip): aggregation + BatchNorm1d(verts) + ReLU +
            # residual average + the next layer's product; the first layer's product and the coordinate head stay the layers'
            with batching:
                full, lead = _InputTap.apply(features, pooled, self.hidden, True)
                s1 = _layers._dense(full, self.gc1.weight1)
                if tuple(self.gc15.weight1.shape[-2:]) == (192, 3) and self.gc15.bias is not None:
                    # the coordinate head's product (and its two gradients) ride in the last / first layer launch
                    feats_out, s15 = _deform.hidden_chain(self, s1, lead, csr, head=self.gc15)
                    coords = _layers.zero_n_aggregate(s15, adj, self.gc15.bias, 3 // self.gc15.split, None)
                else:
                    feats, feats_out = _deform.hidden_chain(self, s1, lead, csr)
                    coords = self.gc15(feats, adj, _identity)
            return feats_out, coords
        with batching:
            full, lead = _InputTap.apply(features, pooled, self.hidden)
            x = self._layer(1, full, adj)
            feats, feats_r = self._layer(2, x, adj, residual=lead, tap=True)
            for i in (3, 5, 7, 9, 11):
                x = self._layer(i, feats, adj)
                feats, feats_r = self._layer(i + 1, x, adj, residual=feats_r, tap=True)
            feats, feats_r = self._layer(13, feats, adj, residual=feats_r, tap=True)
            coords = self.gc15(feats, adj, _identity)
        return feats_r, coords


class MeshEncoder(nn.Module):
    """Reference models.py:299-348: 16 unbatched 0N-GCN layers (3 -> 60 ... -> 300, ELU) and a GCNMax head that
    max-pools the vertices into the latent vector.  Layer names, widths and state_dict keys are the reference's.

    `forward(positions, adj)` is the reference call for ONE mesh (adj = its dense normalised adjacency, converted
    to CSR once and cached).  `encode_batch(batch)` takes a geometrics_amd.ragged.RaggedMeshBatch and returns
    [B, latent] for all meshes at once -- the replacement for the per-mesh python loop of auto_encoder.py:71-76:
    every layer is one GEMM over sum(V) rows + one aggregation launch on the block-diagonal CSR."""

    _WIDTHS = (("h1", 3, 60), ("h21", 60, 60), ("h22", 60, 60), ("h23", 60, 60), ("h24", 60, 120), ("h3", 120, 120),
               ("h4", 120, 120), ("h41", 120, 150), ("h5", 150, 200), ("h6", 200, 210), ("h7", 210, 250),
               ("h8", 250, 300), ("h81", 300, 300), ("h9", 300, 300), ("h10", 300, 300), ("h11", 300, 300))

    def __init__(self, latent_length):
        super().__init__()
        for name, cin, cout in self._WIDTHS:
            setattr(self, name, ZERON_GCN(cin, cout))
        self.reduce = GCNMax(300, latent_length)

    def _trunk(self, positions, adj):
        features = positions
        for name, _, _ in self._WIDTHS:
            features = getattr(self, name)(features, adj, F.elu)
        return features

    def forward(self, positions, adj, play=False):
        return self.reduce(self._trunk(positions, adj), adj, F.elu)

    def encode_batch(self, batch):
        return self.reduce(self._trunk(batch.verts, batch), batch, F.elu)

    def graphed_encode(self, batch, warmup=3):
        """encode_batch for batches of a FIXED topology (the sizes and connectivity of `batch`; positions vary), with the
        forward and the backward pass each replayed as ONE HIP graph: `f = enc.graphed_encode(batch); latents = f(verts)`
        with verts [sum(V), 3] is differentiable like the eager call (parameters and, when verts requires grad, verts).
        The eager path is launch-bound -- 17 layers x (product, aggregation) + the segmented max, forward and backward:
        ~90 launches of a few microseconds each from python (tools/time_encoder.py) -- which is what the capture removes.
        torch.cuda.make_graphed_callables does the capture (static input / output / gradient buffers, autograd-aware)."""
        encoder = self

        class _Encode(nn.Module):     # the encoder's parameters are this module's: their gradients flow through the graph
            def __init__(self):
                super().__init__()
                self.encoder = encoder

            def forward(self, verts):
                return self.encoder.reduce(self.encoder._trunk(verts, batch), batch, F.elu)

        sample = batch.verts.detach().clone().requires_grad_(batch.verts.requires_grad)
        return torch.cuda.make_graphed_callables(_Encode(), (sample,), num_warmup_iters=warmup)
