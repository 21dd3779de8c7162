"""Functional + autograd layer over the C ABI for the HBM-bound stages (sampling, losses).

Every function launches on torch's current stream, allocates its outputs with torch and
never synchronises with the host, so a whole training step can be captured in a HIP graph.
"""
import ctypes
import weakref

import torch

from . import _lib
from . import chamfer_distance as _chamfer
from .chamfer_distance import chamfer_nn
from .tri_distance import face_order, faces_in_order, morton_order, tri_distance_indexed


def _f32(t, name, ndim, last=None):
    return _lib.require(t, name, torch.float32, ndim, last)


def face_areas(verts, faces):
    """areas[B,F] = 0.5*|(v0-v1) x (v1-v2)|  (reference utils.py:596-602, un-normalised)."""
    verts = _f32(verts.detach(), "verts", 3, 3)
    faces = _lib.require(faces, "faces", torch.int64, 2, 3)
    b, nv, _ = verts.shape
    out = torch.empty(b, faces.shape[0], dtype=torch.float32, device=verts.device)
    with torch.cuda.device(verts.device):
        _lib.call("geom_face_areas_f32", b, nv, verts.data_ptr(), faces.shape[0], faces.data_ptr(), out.data_ptr())
    return out


def device_sum(x, scale=1.0):
    """0-dim tensor scale*sum(x) with a fixed reduction tree (bit-reproducible)."""
    x = x.contiguous()
    out = torch.empty((), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("geom_sum_f32", x.numel(), x.data_ptr(), float(scale), out.data_ptr())
    return out


class SampleFaces(torch.autograd.Function):
    """points[B,S,3] from pre-drawn (choices [B,S] int64 face ids, u [B,S] (sqrt'ed), v [B,S]):
    fused gather + barycentric combination (reference utils.py:615-631); backward scatters
    grad_points into grad_verts with the same weights (what autograd does through the three
    index_selects of the reference, in one kernel)."""

    @staticmethod
    def forward(ctx, verts, faces, choices, u, v):
        verts_c = _f32(verts, "verts", 3, 3)
        faces = _lib.require(faces, "faces", torch.int64, 2, 3)
        choices = _lib.require(choices, "choices", torch.int64, 2)
        u = _f32(u, "u", 2)
        v = _f32(v, "v", 2)
        b, nv, _ = verts_c.shape
        num = choices.shape[1]
        if choices.shape[0] != b or u.shape != choices.shape or v.shape != choices.shape:
            raise RuntimeError("choices/u/v must all be [B,num]")
        points = torch.empty(b, num, 3, dtype=torch.float32, device=verts_c.device)
        with torch.cuda.device(verts_c.device):
            _lib.call("geom_sample_faces_fwd_f32", b, nv, verts_c.data_ptr(), faces.shape[0], faces.data_ptr(),
                      num, choices.data_ptr(), u.data_ptr(), v.data_ptr(), points.data_ptr())
        ctx.save_for_backward(faces, choices, u, v)
        ctx.nv = nv
        return points

    @staticmethod
    def backward(ctx, grad_points):
        faces, choices, u, v = ctx.saved_tensors
        grad_points = grad_points.contiguous()
        b, num, _ = grad_points.shape
        grad_verts = torch.zeros(b, ctx.nv, 3, dtype=torch.float32, device=grad_points.device)
        with torch.cuda.device(grad_points.device):
            _lib.call("geom_sample_faces_bwd_f32", b, ctx.nv, faces.shape[0], faces.data_ptr(), num,
                      choices.data_ptr(), u.data_ptr(), v.data_ptr(), grad_points.data_ptr(), grad_verts.data_ptr())
        return grad_verts, None, None, None, None


class GatherSqDistSum(torch.autograd.Function):
    """sum_j |dst[b, idx[b,j]] - src[b,j]|^2 given the squared distances `sq` the NN scan already
    produced for exactly these pairs (reference utils.py:416-417, 462 recompute them from two
    index_selects).  Differentiable in src and dst."""

    @staticmethod
    def forward(ctx, src, dst, idx, sq):
        ctx.save_for_backward(src, dst, idx)
        return device_sum(sq)

    @staticmethod
    def backward(ctx, grad):
        src, dst, idx = ctx.saved_tensors
        src_c, dst_c = src.contiguous(), dst.contiguous()
        b, n, _ = src_c.shape
        m = dst_c.shape[1]
        need_src, need_dst = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_src = torch.empty_like(src_c) if need_src else None
        g_dst = torch.zeros_like(dst_c) if need_dst else None
        grad = grad.contiguous()
        with torch.cuda.device(src_c.device):
            _lib.call("geom_chamfer_grad_f32", b, n, src_c.data_ptr(), m, dst_c.data_ptr(), idx.data_ptr(),
                      grad.data_ptr(), 1.0, _lib.ptr(g_src), 0, _lib.ptr(g_dst))
        return g_src, g_dst, None, None


class PointToTriangleSum(torch.autograd.Function):
    """sum_j |q_j - p_j|^2 with q_j the closest point selected by option[b,j] on triangle
    index[b,j] (reference calc_point_to_line, utils.py:506-550).  Differentiable in verts."""

    @staticmethod
    def forward(ctx, xyz, verts, faces, option, index):
        xyz_c = _f32(xyz, "xyz", 3, 3)
        verts_c = _f32(verts, "verts", 3, 3)
        b, n, _ = xyz_c.shape
        nv = verts_c.shape[1]
        dev = xyz_c.device
        sq = torch.empty(b, n, dtype=torch.float32, device=dev)
        closest = torch.empty(b, n, 3, dtype=torch.float32, device=dev)
        weights = torch.empty(b, n, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("geom_p2tri_loss_fwd_f32", b, n, xyz_c.data_ptr(), nv, verts_c.data_ptr(), faces.shape[0],
                      faces.data_ptr(), option.data_ptr(), index.data_ptr(), sq.data_ptr(), closest.data_ptr(),
                      weights.data_ptr())
        ctx.save_for_backward(xyz_c, faces, index, closest, weights)
        ctx.nv = nv
        return device_sum(sq)

    @staticmethod
    def backward(ctx, grad):
        xyz, faces, index, closest, weights = ctx.saved_tensors
        b, n, _ = xyz.shape
        grad_verts = torch.zeros(b, ctx.nv, 3, dtype=torch.float32, device=xyz.device)
        grad = grad.contiguous()
        with torch.cuda.device(xyz.device):
            _lib.call("geom_p2tri_loss_bwd_f32", b, n, xyz.data_ptr(), ctx.nv, faces.shape[0], faces.data_ptr(),
                      index.data_ptr(), closest.data_ptr(), weights.data_ptr(), grad.data_ptr(), 1.0,
                      grad_verts.data_ptr())
        return None, grad_verts, None, None, None


_vf_cache = {}   # id(faces) -> (weakref, version, nv, vf_ptr, vf_item)


def vertex_faces(faces, nv):
    """Static CSR vertex -> incident (face << 2 | corner), ascending per vertex: what the gather backward walks.
    Built once per faces TENSOR OBJECT / in-place version / vertex count (same caching policy as the adjacency CSR)."""
    key = id(faces)
    hit = _vf_cache.get(key)
    if hit is not None and hit[0]() is faces and hit[1] == faces._version and hit[2] == nv:
        return hit[3], hit[4]
    with torch.no_grad():
        flat = faces.reshape(-1)
        if flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= nv):
            raise RuntimeError("faces refer to vertices outside [0, %d)" % nv)
        order = torch.argsort(flat, stable=True)                      # position p = 3*face + corner
        vf_item = (((order // 3) << 2) | (order % 3)).to(torch.int32).contiguous()
        vf_ptr = torch.zeros(nv + 1, dtype=torch.int64, device=faces.device)
        vf_ptr[1:] = torch.cumsum(torch.bincount(flat, minlength=nv), 0)
        vf_ptr = vf_ptr.to(torch.int32).contiguous()
    _vf_cache[key] = (weakref.ref(faces, lambda _ref, k=key: _vf_cache.pop(k, None)), faces._version, nv, vf_ptr, vf_item)
    return vf_ptr, vf_item


class GtIndex:
    """What the culled Chamfer scan of the surface step keeps per ground-truth cloud (static data: built once per batch of
    objects, e.g. by the data loader): a spatially coherent visiting order [B,N] int32 and the index written from it by
    geom_nn_cull_index_f32 (run spheres + the cloud in that order).  `order=None`: a 30-bit Morton order made on the
    device.  Any order gives the same results, bit for bit; a bad one only costs speed."""

    def __init__(self, gt_points, order=None):
        gt = _f32(gt_points.detach(), "gt_points", 3, 3)
        b, n, _ = gt.shape
        if order is None:
            order = torch.stack([morton_order(gt[i]) for i in range(b)]) if b else torch.empty(0, n, dtype=torch.int32, device=gt.device)
        order = _lib.require(order, "order", torch.int32, 2)
        if order.shape != (b, n) or order.device != gt.device:
            raise RuntimeError("order must be an int32 [B,N] permutation per cloud on the clouds' device")
        self.order = order
        self.index = torch.empty(max(int(_lib.lib().geom_nn_cull_index_floats(b, n)), 4), dtype=torch.float32, device=gt.device)
        self.shape, self.device = (b, n), gt.device
        self.source = (gt.data_ptr(), gt._version)
        # the index HOLDS the cloud's memory: while it lives the allocator cannot hand that address to another cloud (a
        # recycled address at version 0 would pass the check below and be scanned through the OLD cloud's index)
        self._cloud = gt
        with torch.cuda.device(gt.device):
            _lib.call("geom_nn_cull_index_f32", b, n, gt.data_ptr(), order.data_ptr(), self.index.data_ptr())

    def check(self, gt):
        if tuple(gt.shape[:2]) != self.shape or gt.device != self.device or (gt.data_ptr(), gt._version) != self.source:
            raise RuntimeError("this GtIndex was built for another ground-truth tensor (or the tensor was modified since)")


class ScanPrep:
    """What the draw launch of a step prepared for the scan of the same step (ops.draw_samples(prepare_scan_for=...)): the
    triangle records and, on the culled route, the index of the sampled points (which were generated in visiting order)."""
    __slots__ = ("tri_ws", "sample_index")

    def __init__(self, tri_ws, sample_index=None):
        self.tri_ws, self.sample_index = tri_ws, sample_index


# The finalize pass of the one-sided surface loss as extra (role) workgroups of the fused scan launch (csrc/tri_distance.hip:
# ScanTail) instead of a launch of its own; same outputs bit for bit.  False: always the separate launch (the A/B switch).
scan_finalize_tail = True
# FAIL-SAFE of the in-launch roles (they wait for tiles of their own launch; a role that waits in vain gives up, the loss is
# NaN and the backward writes NaN gradients for the meshes whose ordering is missing -- csrc/surface_layout.h: status
# words).  Every eager forward that used the roles sends its status words to pinned host memory behind the launch (no
# synchronisation); the NEXT forward looks at the copies that have landed, and the first time one says "gave up" the roles
# are switched off for the rest of the process (the stand-alone finalize launch takes over: 1 % of a step) with a warning.
# A captured step cannot look (no python runs between replays): it stays loud through the NaN loss and gradients, and
# `finalize_roles_gave_up()` lets a training loop ask at the points where it synchronises anyway.
_roles_watch = []          # [(event, pinned int32 [b + 1])] of recent forwards
_roles_gave_up = False


def _watch_roles(order, b, nf, cap):
    global _roles_gave_up, scan_finalize_tail
    if torch.cuda.is_current_stream_capturing():      # (an event query would invalidate the capture)
        return
    done = [w for w in _roles_watch if w[0].query()]
    for w in done:
        _roles_watch.remove(w)
        if int(w[1].abs().sum()) != 0 and not _roles_gave_up:
            _roles_gave_up = True
            scan_finalize_tail = False
            import warnings
            warnings.warn("geometrics_amd: a finalize role of the fused surface scan gave up waiting (status %s); its loss and "
                          "gradients were NaN.  The roles are switched off for the rest of this process (ops.scan_finalize_tail"
                          " = False): the finalize pass runs as its own launch again." % w[1].tolist())
    if order is None or len(_roles_watch) >= 8:
        return
    words = (b + 4) & ~3
    host = torch.empty(b + 1, dtype=torch.int32, pin_memory=True)
    host.copy_(order[order.numel() - words:order.numel() - words + b + 1], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _roles_watch.append((ev, host))


def finalize_roles_gave_up():
    """True once a finalize role of the fused scan launch has given up in this process (checked without synchronising: call
    it behind a point where the step is known to be complete, e.g. after reading the loss)."""
    _watch_roles(None, 0, 0, 0)
    return _roles_gave_up


# Inspection hook (tests, bench.py's parity spot check): a dict that receives the arg-min outputs of the next SurfaceLoss
# forward passes as they sit on the device -- idx_gt (nearest sampled point of every gt point), idx_pred (nearest gt point of
# every sampled point), the draws (choices, u, v) and sampled points it ran on and, one-sided loss, tri_index / tri_option /
# tri_dist of the point-to-triangle scan.  None: off.
scan_capture = None


class SurfaceLoss(torch.autograd.Function):
    """The whole sampled-surface loss of the reference in one autograd node.

      two_sided=False  batch_point_to_surface (utils.py:441-502):
            3000 * ( mean_s |gt[nn(s)] - pred_s|^2  +  mean_g |closest_on_mesh(g) - g|^2 )
      two_sided=True   batch_point_to_point (utils.py:393-438):
            3000 * ( mean_s |gt[nn(s)] - pred_s|^2  +  mean_g |pred[nn(g)] - g|^2 )

    Forward: [tri scan -> closest point] -> sample -> NN (both directions) -> one two-segment sum, one stream.
    Backward: a gather -- the points are binned by face (integer atomics) and every vertex sums the points on its
    incident faces in a fixed order: no float atomics, no zero-fill, bit-reproducible (csrc/surface_gather.hip); the
    gradient of the sampled points is never materialised.  Returns (loss, sq_gt, sq_pred); the squared NN
    distances feed the F1 score and are not differentiable."""

    @staticmethod
    def forward(ctx, verts, faces, gt, choices, u, v, two_sided, scale, points=None, tri_ws=None, loss_out=None, gt_index=None,
                mesh_weight=None):
        """mesh_weight (optional, [B] fp32 on the meshes' device): per-mesh factors -- loss = sum_m w[m] * (mesh m's share of the
        loss above), mesh m's gradient times w[m] (geom_surface_finalize_w_f32 / geom_surface_gather_w_f32)."""
        verts_c = _f32(verts.detach(), "verts", 3, 3)
        gt_c = _f32(gt.detach(), "gt_points", 3, 3)
        faces = _lib.require(faces, "faces", torch.int64, 2, 3)
        b, nv, _ = verts_c.shape
        choices = _lib.require(choices.reshape(b, -1), "choices", torch.int64, 2)
        u = _f32(u.reshape(b, -1), "u", 2)
        v = _f32(v.reshape(b, -1), "v", 2)
        num, n_gt, nf, dev = choices.shape[1], gt_c.shape[1], faces.shape[0], verts_c.device
        if gt_c.shape[0] != b:
            raise RuntimeError("gt_points and verts batch sizes differ")
        if mesh_weight is not None:
            mesh_weight = _f32(mesh_weight.detach(), "mesh_weight", 1)
            if mesh_weight.shape[0] != b or mesh_weight.device != verts_c.device:
                raise RuntimeError("mesh_weight must be [B] on the meshes' device")
        L = _lib.lib()
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        have_points = points is not None      # drawn and gathered by one kernel (ops.draw_samples(with_points=True))
        points = _f32(points.detach(), "points", 3, 3) if have_points else torch.empty(b, num, 3, **f32)
        if points.shape != (b, num, 3):
            raise RuntimeError("points must be [B,num,3] for the given draws")
        if loss_out is None:
            out = torch.empty((), **f32)
        else:     # the caller's memory receives the loss (a data-parallel step: the tail of its all-reduce bucket -- no copy launch)
            if not (loss_out.is_cuda and loss_out.device == dev and loss_out.dtype == torch.float32 and loss_out.numel() == 1):
                raise RuntimeError("loss_out must be ONE fp32 element on the mesh batch's device")
            out = loss_out.detach().view(())
        sq_gt, sq_pred = torch.empty(b, n_gt, **f32), torch.empty(b, num, **f32)
        idx_p, idx_g = torch.empty(b, n_gt, **i32), torch.empty(b, num, **i32)
        prep = tri_ws if isinstance(tri_ws, ScanPrep) else None
        if prep is not None:
            tri_ws = prep.tri_ws
        cull = None
        if gt_index is not None and prep is not None and prep.sample_index is not None and not two_sided:
            gt_index.check(gt_c)    # the culled Chamfer tiles: both clouds' indices are there
            cull = _lib.SurfaceCull(gt_index.order.data_ptr(), gt_index.index.data_ptr(), prep.sample_index.data_ptr(), None)
        if not two_sided:
            ws_bytes = L.geom_tri_distance_workspace_bytes(b, n_gt, nf)
            ws_ready = tri_ws is not None and have_points      # written by the draw launch (ops.draw_samples(prepare_scan_for=...))
            if ws_ready and (tri_ws.numel() * 4 < ws_bytes or tri_ws.device != dev):
                raise RuntimeError("tri_ws does not belong to this mesh batch / query count")
            ws = tri_ws if ws_ready else torch.empty(max(ws_bytes, 16) // 4, **f32)
            tri_order = face_order(verts_c, faces)      # cached k-d leaf order of the faces: two-level scan
            tri_d, option, index = torch.empty(b, n_gt, **f32), torch.empty(b, n_gt, **i32), torch.empty(b, n_gt, **i32)
            sq, closest, weights = torch.empty(b, n_gt, **f32), torch.empty(b, n_gt, 3, **f32), torch.empty(b, n_gt, 3, **f32)
        with torch.cuda.device(dev):
            if not have_points:
                _lib.call("geom_sample_faces_fwd_f32", b, nv, verts_c.data_ptr(), nf, faces.data_ptr(), num,
                          choices.data_ptr(), u.data_ptr(), v.data_ptr(), points.data_ptr())
            # Both arg-min scans -- Chamfer NN in both directions and (one-sided loss) the point-to-triangle scan with
            # the closest point / weights / squared distance of the winner -- in ONE call: one heterogeneous launch
            # behind the triangle-record prep.  (Round 1 ran them back to back: two streams inside a captured graph cost
            # 5-10 us per fork/join edge.)  When a gradient is wanted the scans also write every point's gradient record
            # into the backward scratch.
            want = bool(ctx.needs_input_grad[0])
            order = torch.empty(L.geom_surface_order_words(b, nf, num, n_gt), dtype=torch.int32, device=dev)
            coef_s, coef_o = scale / (b * num), scale / (b * n_gt)
            wrote = ctypes.c_int(0)
            flags = _chamfer.default_flags()
            if not two_sided and ws_ready:
                flags |= _lib.FLAG_TRI_WS_READY
            if two_sided:
                tri_args = (nv, None, nf, None, None, None, None, None, None, None, None)    # nf still sizes the scratch layout
                ws_ptr, ws_len = None, 0
            else:
                tri_args = (nv, verts_c.data_ptr(), nf, faces.data_ptr(), _lib.ptr(tri_order), tri_d.data_ptr(),
                            option.data_ptr(), index.data_ptr(), sq.data_ptr(), closest.data_ptr(), weights.data_ptr())
                ws_ptr, ws_len = ws.data_ptr(), ws_bytes
            # one-sided loss: the finalize pass (below) rides in the scan launch as extra (role) workgroups where the launch is
            # the fused one and a mesh's faces + points fit its LDS (tail.finalized says so): each mesh is ordered as soon
            # as ITS triangle tiles are through instead of in a launch of its own behind the slowest tile
            tail = None
            if not two_sided and scan_finalize_tail and mesh_weight is None:    # (the in-launch roles know no per-mesh factors)
                tail = _lib.SurfaceTail(choices.data_ptr(), scale / sq_pred.numel(), scale / sq.numel(), int(want), out.data_ptr(), 0)
            _lib.check(L.geom_surface_scan_f32(
                b, n_gt, gt_c.data_ptr(), num, points.data_ptr(), sq_gt.data_ptr(), idx_p.data_ptr(), sq_pred.data_ptr(),
                idx_g.data_ptr(), *tri_args, u.data_ptr(), v.data_ptr(), coef_s, coef_o, order.data_ptr() if want else None,
                flags, ws_ptr, ws_len, ctypes.byref(wrote), ctypes.byref(cull) if cull is not None else None,
                ctypes.byref(tail) if tail is not None else None, _lib.stream_ptr()), "geom_surface_scan_f32")
            # finalize: the loss reduction AND, when a gradient is wanted, the backward's preparation (points counting-sorted
            # by face in ascending id order; the records too when the scans could not write them) in one launch; the
            # backward is then a single gather launch
            other_sq = sq_gt if two_sided else sq
            args = (b, nf, num, choices.data_ptr(), u.data_ptr(), v.data_ptr(), points.data_ptr(), n_gt, gt_c.data_ptr(),
                    idx_g.data_ptr(), idx_p.data_ptr() if two_sided else None, None if two_sided else index.data_ptr(),
                    None if two_sided else closest.data_ptr(), None if two_sided else weights.data_ptr(),
                    sq_pred.data_ptr(), other_sq.data_ptr(), scale / sq_pred.numel(), scale / other_sq.numel(), coef_s, coef_o)
            if tail is None or not tail.finalized:
                wptr = _lib.ptr(mesh_weight)
                code = L.geom_surface_finalize_w_f32(*args, int(want), wrote.value, order.data_ptr(), out.data_ptr(), wptr,
                                                     _lib.stream_ptr())
                if code == _lib.EUNSUPPORTED:       # too many faces + points for the in-LDS ordering: loss only, scatter backward
                    if mesh_weight is not None and want:
                        raise RuntimeError("per-mesh weights need the ordered backward (faces + points of a mesh within ~38 000)")
                    want = False
                    code = L.geom_surface_finalize_w_f32(*args, 0, 0, order.data_ptr(), out.data_ptr(), wptr, _lib.stream_ptr())
                _lib.check(code, "geom_surface_finalize_w_f32")
            ctx.order = order if want else None
            if tail is not None and tail.finalized and want:
                _watch_roles(order, b, nf, num + n_gt)
            if scan_capture is not None:
                scan_capture.update(idx_gt=idx_p, idx_pred=idx_g, sq_gt=sq_gt, sq_pred=sq_pred, choices=choices, u=u, v=v, points=points)
                if not two_sided:
                    scan_capture.update(tri_index=index, tri_option=option, tri_dist=tri_d)
            if two_sided:
                ctx.save_for_backward(faces, choices, u, v, points, gt_c, idx_g, idx_p)
            else:
                ctx.save_for_backward(faces, choices, u, v, points, gt_c, idx_g, index, closest, weights)
        ctx.two_sided, ctx.scale, ctx.nv = two_sided, scale, nv
        ctx.mesh_weight = mesh_weight
        ctx.mark_non_differentiable(sq_gt, sq_pred)
        ctx.set_materialize_grads(False)    # no zero tensors (two fill launches) for the two distance outputs
        return out, sq_gt, sq_pred

    @staticmethod
    def backward(ctx, grad, _g1, _g2):
        saved = ctx.saved_tensors
        faces, choices, u, v, points, gt = saved[:6]
        b, num, _ = points.shape
        n_gt, nf, nv = gt.shape[1], faces.shape[0], ctx.nv
        dev = points.device
        grad = grad.contiguous()
        with torch.cuda.device(dev):
            if ctx.order is not None:           # prepared by the forward's finalize launch: one gather, no atomics
                vf_ptr, vf_item = vertex_faces(faces, nv)
                grad_verts = torch.empty(b, nv, 3, dtype=torch.float32, device=dev)
                _lib.call("geom_surface_gather_w_f32", b, nv, nf, vf_ptr.data_ptr(), vf_item.data_ptr(), num, n_gt, 1,
                          ctx.order.data_ptr(), grad.data_ptr(), _lib.ptr(ctx.mesh_weight), grad_verts.data_ptr())
            elif ctx.mesh_weight is not None:
                raise RuntimeError("per-mesh weights need the ordered backward")
            else:
                # a mesh whose faces + points do not fit the ordering pass's LDS (> ~38 000 together): scatter formulation
                # (fp32 atomics into a zeroed gradient; same values up to summation order)
                grad_verts = torch.zeros(b, nv, 3, dtype=torch.float32, device=dev)
                if ctx.two_sided:
                    for other_idx, via_nn, coef in ((saved[6], 0, ctx.scale / (b * num)), (saved[7], 1, ctx.scale / (b * n_gt))):
                        _lib.call("geom_sample_chamfer_bwd_f32", b, nv, nf, faces.data_ptr(), num, choices.data_ptr(),
                                  u.data_ptr(), v.data_ptr(), points.data_ptr(), n_gt, gt.data_ptr(), other_idx.data_ptr(),
                                  via_nn, grad.data_ptr(), coef, grad_verts.data_ptr())
                else:
                    index, closest, weights = saved[7:10]
                    _lib.call("geom_surface_loss_bwd_f32", b, nv, nf, faces.data_ptr(), num, choices.data_ptr(),
                              u.data_ptr(), v.data_ptr(), points.data_ptr(), n_gt, gt.data_ptr(), saved[6].data_ptr(),
                              index.data_ptr(), closest.data_ptr(), weights.data_ptr(), grad.data_ptr(),
                              ctx.scale / (b * num), ctx.scale / (b * n_gt), grad_verts.data_ptr())
        return grad_verts, None, None, None, None, None, None, None, None, None, None, None, None


class Laplacian(torch.autograd.Function):
    """lap = p - mean(neighbours(p)) over the CSR of the binary adjacency (reference
    batch_get_lap_info, utils.py:654-662, a dense [V,V] @ [B,V,3] there)."""

    @staticmethod
    def forward(ctx, positions, rowptr, col, inv_deg):
        x = _f32(positions, "positions", 3, 3)
        b, nv, _ = x.shape
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.call("geom_laplacian_f32", b, nv, rowptr.data_ptr(), col.data_ptr(), inv_deg.data_ptr(),
                      x.data_ptr(), 0, out.data_ptr())
        ctx.save_for_backward(rowptr, col, inv_deg)
        return out

    @staticmethod
    def backward(ctx, grad):
        rowptr, col, inv_deg = ctx.saved_tensors
        g = grad.contiguous()
        b, nv, _ = g.shape
        out = torch.empty_like(g)
        with torch.cuda.device(g.device):
            _lib.call("geom_laplacian_f32", b, nv, rowptr.data_ptr(), col.data_ptr(), inv_deg.data_ptr(),
                      g.data_ptr(), 1, out.data_ptr())
        return out, None, None, None


class EdgeSqLenSum(torch.autograd.Function):
    """sum over meshes and faces of the three squared edge lengths (reference batch_calc_edge,
    utils.py:636-651, before its means)."""

    @staticmethod
    def forward(ctx, verts, faces):
        v = _f32(verts, "verts", 3, 3)
        faces = _lib.require(faces, "faces", torch.int64, 2, 3)
        b, nv, _ = v.shape
        per_face = torch.empty(b, faces.shape[0], dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            _lib.call("geom_edge_sqlen_fwd_f32", b, nv, v.data_ptr(), faces.shape[0], faces.data_ptr(),
                      per_face.data_ptr())
        ctx.save_for_backward(v, faces)
        return device_sum(per_face)

    @staticmethod
    def backward(ctx, grad):
        v, faces = ctx.saved_tensors
        b, nv, _ = v.shape
        grad = grad.contiguous()
        grad_verts = torch.zeros_like(v)
        with torch.cuda.device(v.device):
            _lib.call("geom_edge_sqlen_bwd_f32", b, nv, v.data_ptr(), faces.shape[0], faces.data_ptr(),
                      grad.data_ptr(), 1.0, grad_verts.data_ptr())
        return grad_verts, None


class FanOut(torch.autograd.Function):
    """n tensor objects over x's memory, one per consumer: their gradients meet in ONE launch (geom_sum_tensors_f32, fixed
    order) instead of autograd's n - 1 accumulation launches.  A stage's positions have six consumers in the reference's
    driver (GEOMetrics.py:118-161)."""

    @staticmethod
    def forward(ctx, x, n):
        from .layers import _alias
        xc = x if x.is_contiguous() else x.contiguous()
        ctx.n = n
        return tuple(_alias(xc) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        import ctypes

        def rows(g):      # a [.., width] column slice of a wider row-major buffer is read in place
            if g.is_contiguous():
                return g, g.shape[-1]
            if g.dim() >= 2 and g.stride(-1) == 1 and g.stride(-2) >= g.shape[-1] and \
                    all(g.stride(d) == g.stride(d + 1) * g.shape[d + 1] for d in range(g.dim() - 2)):
                return g, g.stride(-2)
            return g.contiguous(), g.shape[-1]
        given = [rows(g) for g in grads if g is not None]
        if not given:
            return None, None
        first = given[0][0]
        fast = first.is_cuda and first.dtype == torch.float32 and len(given) <= 8 and first.dim() >= 1
        if len(given) == 1 or not fast:
            total = first.contiguous()
            for g, _ in given[1:]:
                total = total + g
            return total, None
        out = torch.empty(first.shape, dtype=torch.float32, device=first.device)
        ptrs = (ctypes.c_void_p * len(given))(*[g.data_ptr() for g, _ in given])
        width = first.shape[-1]
        with torch.cuda.device(out.device):
            if all(ld == width for _, ld in given):
                _lib.call("geom_sum_tensors_f32", len(given), ptrs, out.numel(), out.data_ptr())
            else:
                lds = (ctypes.c_int64 * len(given))(*[ld for _, ld in given])
                _lib.call("geom_sum_tensors_rows_f32", len(given), ptrs, lds, out.numel() // width, width, out.data_ptr())
        return out, None


class SumScalars(torch.autograd.Function):
    """((t0 + t1) + t2) + ... of up to 8 zero-dimensional fp32 tensors in ONE launch (geom_sum_tensors_f32); backward: every term
    receives the incoming gradient itself (no launch).  The driver's `loss = edge_loss + surface_loss + lap_loss + ...`
    (GEOMetrics.py:164) costs a launch per `+` forward and one per term backward."""

    @staticmethod
    def forward(ctx, *terms):
        import ctypes
        ts = [t.reshape(()) for t in terms]
        out = torch.empty((), dtype=torch.float32, device=ts[0].device)
        ctx.n = len(ts)
        if not all(t.is_cuda and t.dtype == torch.float32 for t in ts) or len(ts) > 8:
            total = ts[0]
            for t in ts[1:]:
                total = total + t
            return total
        ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        with torch.cuda.device(out.device):
            _lib.call("geom_sum_tensors_f32", len(ts), ptrs, 1, out.data_ptr())
        return out

    @staticmethod
    def backward(ctx, grad):
        return (grad,) * ctx.n


class StageRegularisers(torch.autograd.Function):
    """The regularisers of ONE deformation stage as one scalar (GEOMetrics.py:147-161 on utils.py:636-662):
        w_edge * mean over (mesh, face) of (|e1|^2 + |e2|^2 + |e3|^2)(cur) / 3
      + w_lap  * mean over (mesh, vertex) of |lap(prev) - lap(cur)|^2      + w_move * mean over (mesh, vertex) of |prev - cur|^2
    ONE launch forward (+ the fixed-order sum of its partials), ONE backward (csrc/regularizers.hip) instead of ~45 eager
    launches; prev [B,V,3] or the [V,3] template."""

    @staticmethod
    def forward(ctx, prev, cur, faces, rowptr, col, inv_deg, w_lap, w_move, w_edge):
        c = _f32(cur, "cur", 3, 3)
        b, nv, _ = c.shape
        batched = prev.dim() == 3
        p = _f32(prev, "prev", 3 if batched else 2, 3)
        if batched and p.shape != c.shape or not batched and p.shape[0] != nv:
            raise RuntimeError("prev must be [B,V,3] like cur or one [V,3] mesh")
        faces = _lib.require(faces, "faces", torch.int64, 2, 3)
        nf = faces.shape[0]
        c_lap, c_move = float(w_lap) / (b * nv), float(w_move) / (b * nv)
        c_edge = float(w_edge) / (3.0 * b * nf) if nf else 0.0
        lapd = torch.empty_like(c)
        partial = torch.empty(int(_lib.lib().geom_stage_regularisers_blocks(b, nv, nf)), dtype=torch.float32, device=c.device)
        with torch.cuda.device(c.device):
            _lib.call("geom_stage_regularisers_fwd_f32", b, nv, p.data_ptr(), int(batched), c.data_ptr(), nf, faces.data_ptr(),
                      rowptr.data_ptr(), col.data_ptr(), inv_deg.data_ptr(), c_lap, c_move, c_edge, lapd.data_ptr(), partial.data_ptr())
        ctx.save_for_backward(p, c, faces, rowptr, col, inv_deg, lapd)
        ctx.coef, ctx.batched = (c_lap, c_move, c_edge), batched
        return device_sum(partial)

    @staticmethod
    def backward(ctx, grad):
        p, c, faces, rowptr, col, inv_deg, lapd = ctx.saved_tensors
        b, nv, _ = c.shape
        c_lap, c_move, c_edge = ctx.coef
        g = grad.contiguous()
        vf_ptr, vf_item = vertex_faces(faces, nv) if c_edge != 0.0 else (None, None)
        want_prev = ctx.needs_input_grad[0]
        grad_cur = torch.empty_like(c)
        grad_prev = torch.empty_like(c) if want_prev else None
        with torch.cuda.device(c.device):
            _lib.call("geom_stage_regularisers_bwd_f32", b, nv, p.data_ptr(), int(ctx.batched), c.data_ptr(), faces.shape[0],
                      faces.data_ptr(), rowptr.data_ptr(), col.data_ptr(), inv_deg.data_ptr(), _lib.ptr(vf_ptr), _lib.ptr(vf_item),
                      c_lap, c_move, c_edge, lapd.data_ptr(), g.data_ptr(), _lib.ptr(grad_prev if ctx.batched else None),
                      grad_cur.data_ptr())
        if want_prev and not ctx.batched:      # a template that asks for its gradient: the sum over the batch of -grad_cur's lap/move part
            raise RuntimeError("StageRegularisers: the gradient with respect to an unbatched [V,3] prev is not implemented")
        return grad_prev, (grad_cur if ctx.needs_input_grad[1] else None), None, None, None, None, None, None, None


# ---- feature buffers with HEADROOM: concatenation without torch.cat ------------------------------------------------------------
# The driver builds a deformation block's input as cat(positions, cat(previous features, pooled features)) (GEOMetrics.py:118-131,
# models.py:241): two copies of ~35 MB forward per stage, and slicing copies of the same size when the gradient of the
# concatenation is taken apart again.  A pooling call that is told how many columns will be put IN FRONT of its output
# (`headroom`) allocates the wide buffer itself and writes its features at that column offset (geom_pool_features_fwd_ld_f32); the
# later "concatenations" (`concat_in_front`) only copy the narrow operand into the free columns and widen the view, and in the
# backward pass every consumer reads its column slice of the ONE wide gradient in place.
_headroom = {}     # storage address -> (weak reference to the wide buffer, columns still free in front of the view handed out,
#                      {first column: (address, version, width) of a tensor the pooling launch already copied there})


def _register_headroom(buf, free, placed=None):
    import weakref
    key = buf.untyped_storage().data_ptr()
    if placed is None:
        old = _headroom.get(key)
        placed = old[2] if old is not None and old[0]() is buf else {}
    _headroom[key] = (weakref.ref(buf, lambda _r, k=key: _headroom.pop(k, None)), free, placed)


def _already_placed(buf, col, t):
    """Whether columns [col, col + width) of the wide buffer already hold `t` (PoolFeatures(fronts=...) copied it there)."""
    hit = _headroom.get(buf.untyped_storage().data_ptr())
    return hit is not None and hit[0]() is buf and hit[2].get(col) == (t.data_ptr(), t._version, t.shape[2])


def headroom_of(t):
    """Free columns in front of `t` [B,V,C] inside a wide buffer made by PoolFeatures(headroom=...) / concat_in_front, or 0."""
    if not (torch.is_tensor(t) and t.dim() == 3 and t.is_cuda and t.stride(2) == 1):
        return 0
    hit = _headroom.get(t.untyped_storage().data_ptr())
    if hit is None or hit[0]() is None:
        return 0
    buf, free = hit[0](), hit[1]
    ld = buf.shape[2]
    if t.stride(1) != ld or t.stride(0) != t.shape[1] * ld or t.shape[:2] != buf.shape[:2] or t.storage_offset() != free:
        return 0
    return free if free + t.shape[2] == ld else 0


class _ConcatInFront(torch.autograd.Function):
    """cat((front, wide_view), -1) where wide_view has headroom: `front` is copied into the free columns right in front of the
    view and the view grows to cover them -- no copy of the wide operand; backward: the two column slices of the gradient,
    as views."""

    @staticmethod
    def forward(ctx, front, view):
        hit = _headroom[view.untyped_storage().data_ptr()]
        buf, free = hit[0](), hit[1]
        cf = front.shape[2]
        if not _already_placed(buf, free - cf, front):
            buf[..., free - cf:free].copy_(front)
        out = buf[..., free - cf:]
        _register_headroom(buf, free - cf)
        ctx.cf = cf
        return out

    @staticmethod
    def backward(ctx, g):
        return g[..., :ctx.cf], g[..., ctx.cf:]


def concat_in_front(front, view):
    """torch.cat((front, view), dim=-1) for [B,V,*] tensors -- without touching `view` when it was allocated with headroom for
    `front` (PoolFeatures(headroom=...)); a plain torch.cat otherwise."""
    if (torch.is_tensor(front) and front.dim() == 3 and front.is_cuda and front.dtype == torch.float32 and view.dtype == torch.float32
            and front.shape[:2] == view.shape[:2] and 0 < front.shape[2] <= headroom_of(view)):
        return _ConcatInFront.apply(front, view)
    return torch.cat((front, view), dim=-1)


class PoolFeatures(torch.autograd.Function):
    """Bilinear pooling of the image feature maps at the projected vertex positions (reference
    batched_pooling, utils.py:316-389): one kernel forward, one backward (texel scatters + closed-form
    chain to the vertex positions)."""

    @staticmethod
    def forward(ctx, verts, cam_mat, cam_pos, headroom, *blocks):
        """headroom: columns left free IN FRONT of the pooled features (0: a plain contiguous result; see concat_in_front), or
        (headroom, fronts): also up to two [B,V,*] tensors, left to right as they will stand in front of the features (their
        widths add up to the headroom), which the launch copies there -- concat_in_front / the deformation block then find them
        in place (the gradient still flows through those nodes: the tensors are not inputs of this one)."""
        fronts = ()
        if isinstance(headroom, (tuple, list)):
            headroom, fronts = headroom
        headroom = int(headroom or 0)
        import ctypes
        v = _f32(verts, "verts_pos", 3, 3)
        cam_mat = _f32(cam_mat, "cam_mat", 3, 3)
        cam_pos = _f32(cam_pos, "cam_pos", 2, 3)
        blks = [_f32(b, "block", 4) for b in blocks]
        b, nv, _ = v.shape
        for blk in blks:
            if blk.shape[0] != b or blk.shape[2] != blk.shape[3]:
                raise RuntimeError("feature maps must be [B, C, dim, dim] with the vertex batch size")
        n = len(blks)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in blks])
        chans = (ctypes.c_int * n)(*[t.shape[1] for t in blks])
        dims = (ctypes.c_int * n)(*[t.shape[2] for t in blks])
        ctot = sum(t.shape[1] for t in blks)
        placed, srcs = {}, []
        if fronts:
            col = 0
            for t in fronts:
                if not (torch.is_tensor(t) and t.dim() == 3 and tuple(t.shape[:2]) == (b, nv) and t.is_cuda and t.device == v.device
                        and t.dtype == torch.float32):
                    raise RuntimeError("fronts must be fp32 [B,V,*] tensors on the meshes' device")
                c = t.detach().contiguous()
                srcs.append((c, col))
                if c.data_ptr() == t.data_ptr():       # (a copy made here is not what the caller will hand to the concatenation)
                    placed[col] = (t.data_ptr(), t._version, t.shape[2])
                col += t.shape[2]
            if col != headroom or len(srcs) > 2:
                raise RuntimeError("at most two fronts, whose widths add up to the headroom")
        if headroom > 0:      # the features as the trailing columns of a wider buffer: see concat_in_front
            buf = torch.empty(b, nv, headroom + ctot, dtype=torch.float32, device=v.device)
            out = buf[..., headroom:]
            _register_headroom(buf, headroom, placed)
        else:
            out = torch.empty(b, nv, ctot, dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            if srcs:
                m = len(srcs)
                _lib.call("geom_pool_features_fwd_fronts_f32", b, nv, v.data_ptr(), cam_mat.data_ptr(), cam_pos.data_ptr(), n,
                          ptrs, chans, dims, out.data_ptr(), headroom + ctot, m, (ctypes.c_void_p * m)(*[t.data_ptr() for t, _ in srcs]),
                          (ctypes.c_int * m)(*[t.shape[2] for t, _ in srcs]), (ctypes.c_int * m)(*[c for _, c in srcs]), buf.data_ptr())
            else:
                _lib.call("geom_pool_features_fwd_ld_f32", b, nv, v.data_ptr(), cam_mat.data_ptr(), cam_pos.data_ptr(), n,
                          ptrs, chans, dims, out.data_ptr(), headroom + ctot if headroom > 0 else 0)
        ctx.save_for_backward(v, cam_mat, cam_pos, *blks)
        ctx.meta = (ptrs, chans, dims, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        import ctypes
        v, cam_mat, cam_pos = ctx.saved_tensors[:3]
        blks = ctx.saved_tensors[3:]
        ptrs, chans, dims, n = ctx.meta
        b, nv, _ = v.shape
        ctot = grad_out.shape[2]
        # the gradient in place when it is a column slice of a wider row-major buffer (the block's input gradient), else contiguous
        if grad_out.stride(2) == 1 and grad_out.stride(1) >= ctot and grad_out.stride(0) == nv * grad_out.stride(1) and grad_out.is_cuda \
                and grad_out.dtype == torch.float32:
            g, g_ld = grad_out, grad_out.stride(1)
        else:
            g, g_ld = grad_out.contiguous(), ctot
        need_blocks = [ctx.needs_input_grad[4 + i] for i in range(n)]
        gblks = [torch.empty_like(t) if need else None for t, need in zip(blks, need_blocks)]
        gptrs = (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in gblks])
        gverts = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        ws, ws_bytes = None, 0
        if any(need_blocks) or gverts is not None:
            # texel -> (vertex, weight) lists for the gather formulation of the map gradient + the per-chunk partial sums
            # of the vertex gradient
            ws_bytes = _lib.lib().geom_pool_features_bwd_workspace_bytes(b, nv, n, dims)
            ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            _lib.call("geom_pool_features_bwd_ld_f32", b, nv, v.data_ptr(), cam_mat.data_ptr(), cam_pos.data_ptr(), n,
                      ptrs, chans, dims, g.data_ptr(), g_ld if g_ld != ctot else 0, gptrs, _lib.ptr(gverts), _lib.ptr(ws), ws_bytes)
        return (gverts, None, None, None) + tuple(gblks)


class SegmentMax(torch.autograd.Function):
    """Column maximum per mesh of a ragged batch: x [sum(V), C], offsets [B+1] int64 (device) -> [B, C]
    (GCNMax's `torch.max(i_s, dim=0)[0]`, reference layers.py:78, for every mesh at once).  The gradient goes to
    the arg-max row only (lowest row on ties), written by a gather -- grad_x needs no zero fill."""

    @staticmethod
    def forward(ctx, x, offsets, max_len):
        x_ = _f32(x, "x", 2)
        offsets = _lib.require(offsets, "offsets", torch.int64, 1)
        nseg, c = offsets.numel() - 1, x_.shape[1]
        out = torch.empty(nseg, c, dtype=torch.float32, device=x_.device)
        arg = torch.empty(nseg, c, dtype=torch.int32, device=x_.device)
        ws_bytes = _lib.lib().geom_segment_max_workspace_bytes(nseg, c, int(max_len))
        ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=x_.device)
        with torch.cuda.device(x_.device):
            _lib.call("geom_segment_max_fwd_f32", nseg, offsets.data_ptr(), int(max_len), c, x_.data_ptr(),
                      out.data_ptr(), arg.data_ptr(), ws.data_ptr(), ws_bytes)
        ctx.save_for_backward(offsets, arg)
        ctx.rows = x_.shape[0]
        ctx.mark_non_differentiable(arg)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        offsets, arg = ctx.saved_tensors
        g = grad_out.contiguous()
        nseg, c = arg.shape
        grad_x = torch.empty(ctx.rows, c, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.call("geom_segment_max_bwd_f32", nseg, offsets.data_ptr(), ctx.rows, c, g.data_ptr(), arg.data_ptr(),
                      grad_x.data_ptr())
        return grad_x, None, None


_rng_states = {}


def manual_seed(seed, device=None, mesh_offset=0):
    """(Re)seed the in-kernel sampler stream of `device` (default: current).  `mesh_offset` = the global index of
    the first mesh this process holds: the generator is keyed on (seed, stream position, global mesh index, sample),
    so data-parallel ranks that pass the same seed and their shard's offset draw exactly what one process holding the
    whole batch would draw.  Without a call the stream is seeded from torch's default generator the first time it is
    used, with a rank-distinct offset under torch.distributed.  One sampler stream per device: calls that draw
    from it must be issued on one HIP stream at a time (the position is advanced inside the kernel)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    state = torch.tensor([int(seed) & (2 ** 63 - 1), 0, 0, int(mesh_offset)], dtype=torch.int64).to(dev)   # seed, position, arrivals, first mesh
    _rng_states[key] = state
    return state


def set_rng_state(state, device=None):
    """Make `state` (a tensor manual_seed returned) the sampler stream of `device`: lets one process alternate
    between several independent streams (e.g. two shards stepped in turn)."""
    dev = state.device if device is None else torch.device(device)
    _rng_states[dev.index if dev.index is not None else torch.cuda.current_device()] = state


def _rng_state(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _rng_states.get(key)
    if st is None:
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        # ranks seeded alike (the reference seeds 41 everywhere) must not draw the same samples for different meshes
        st = manual_seed(torch.initial_seed() ^ 0x5DEECE66D, dev, mesh_offset=rank << 24)
    return st


def draw_samples(verts, faces, num, generator=None, with_points=False, prepare_scan_for=None, gt_index=None):
    """The random part of batch_sample (reference utils.py:604-612, 627-628): choices [B,num] ~
    area-weighted with replacement, u = sqrt(U1), v = U2, in ONE kernel: per-mesh face-area CDF in LDS,
    binary search per sample, uniforms from an in-kernel Philox stream whose position lives on the device
    (graph replays draw fresh numbers; no generator bookkeeping launches).  With an explicit torch
    `generator` the uniforms come from torch.rand instead; meshes with more than 16384 faces take the
    torch.multinomial route.  with_points=True also returns the sampled points [B,num,3] (or None when the
    kernel could not produce them), which ops.SurfaceLoss accepts in place of its own gather launch.
    prepare_scan_for = n_gt (with with_points=True): the same launch also writes the triangle records of the surface scan
    of n_gt query points per mesh; a fifth return value is then the prepared workspace tensor (or None when the scan
    will not take the fused route), to be handed to ops.SurfaceLoss as `tri_ws`.  gt_index (an ops.GtIndex of the step's
    ground-truth clouds): the launch also lists the samples in a visiting order and writes their index for the culled
    Chamfer scan; the fifth return value is then an ops.ScanPrep carrying both."""
    verts_c = _f32(verts.detach(), "verts", 3, 3)
    faces = _lib.require(faces, "faces", torch.int64, 2, 3)
    b, nv, _ = verts_c.shape
    dev = verts_c.device
    choices = torch.empty(b, num, dtype=torch.int64, device=dev)
    u = torch.empty(b, num, dtype=torch.float32, device=dev)
    v = torch.empty(b, num, dtype=torch.float32, device=dev)
    uniforms = points = None
    with torch.cuda.device(dev):
        if generator is None:
            if with_points:
                points = torch.empty(b, num, 3, dtype=torch.float32, device=dev)
            if with_points and prepare_scan_for:
                n_gt, nf = int(prepare_scan_for), faces.shape[0]
                ws_bytes = _lib.lib().geom_tri_distance_workspace_bytes(b, n_gt, nf)
                tri_ws = torch.empty(max(ws_bytes, 16) // 4, dtype=torch.float32, device=dev)
                prepared = ctypes.c_int(0)
                cull = s_index = None
                if gt_index is not None:
                    s_index = torch.empty(max(int(_lib.lib().geom_nn_cull_index_floats(b, num)), 4), dtype=torch.float32, device=dev)
                    cull = _lib.SurfaceCull(None, None, s_index.data_ptr(), _lib.ptr(faces_in_order(verts_c, faces)))
                code = _lib.lib().geom_surface_prepare_f32(
                    b, nv, verts_c.data_ptr(), nf, faces.data_ptr(), num, _rng_state(dev).data_ptr(), choices.data_ptr(),
                    u.data_ptr(), v.data_ptr(), points.data_ptr(), n_gt, _lib.ptr(face_order(verts_c, faces)), 0,
                    tri_ws.data_ptr(), ws_bytes, ctypes.byref(prepared), ctypes.byref(cull) if cull is not None else None,
                    _lib.stream_ptr())
                if code == 0:
                    if prepared.value & 2:
                        return choices, u, v, points, ScanPrep(tri_ws, s_index)
                    return choices, u, v, points, (tri_ws if prepared.value else None)
            else:
                code = _lib.lib().geom_draw_samples_rng_f32(b, nv, verts_c.data_ptr(), faces.shape[0], faces.data_ptr(), num,
                                                            _rng_state(dev).data_ptr(), choices.data_ptr(), u.data_ptr(),
                                                            v.data_ptr(), _lib.ptr(points), _lib.stream_ptr())
        else:
            uniforms = torch.rand(3, b, num, device=dev, generator=generator)
            code = _lib.lib().geom_draw_samples_f32(b, nv, verts_c.data_ptr(), faces.shape[0], faces.data_ptr(), num,
                                                    uniforms.data_ptr(), choices.data_ptr(), u.data_ptr(),
                                                    v.data_ptr(), _lib.stream_ptr())
    if code == _lib.EUNSUPPORTED:
        if uniforms is None:
            uniforms = torch.rand(3, b, num, device=dev)
        choices = torch.multinomial(face_areas(verts_c, faces), num, True, generator=generator)
        out = (choices, torch.sqrt(uniforms[1]), uniforms[2])
        if with_points:
            out = out + ((None, None) if prepare_scan_for else (None,))
        return out
    _lib.check(code, "geom_draw_samples_f32")
    if with_points:
        return (choices, u, v, points, None) if prepare_scan_for else (choices, u, v, points)
    return choices, u, v


class VertexHead(torch.autograd.Function):
    """pos = base + scale * feat[..., :3] in one kernel; the adjoint writes grad_feat = [scale*grad_pos | 0]
    in one pass (a slice + mul + add costs two launches forward and mul + zero-fill + strided copy backward)."""

    @staticmethod
    def forward(ctx, base, feat, scale):
        b_ = _f32(base, "base", 3, 3)
        f_ = _f32(feat, "feat", 3)
        if f_.shape[:2] != b_.shape[:2] or f_.shape[2] < 4 or f_.shape[2] % 4:
            raise RuntimeError("feat must be [B,V,C] with C a multiple of 4 matching base [B,V,3]")
        pos = torch.empty_like(b_)
        rows = b_.shape[0] * b_.shape[1]
        with torch.cuda.device(b_.device):
            _lib.call("geom_vertex_head_fwd_f32", rows, f_.shape[2], b_.data_ptr(), f_.data_ptr(), float(scale),
                      pos.data_ptr())
        ctx.scale, ctx.c = float(scale), f_.shape[2]
        return pos

    @staticmethod
    def backward(ctx, grad_pos):
        g = grad_pos.contiguous()
        rows = g.shape[0] * g.shape[1]
        grad_feat = None
        if ctx.needs_input_grad[1]:
            grad_feat = torch.empty(g.shape[0], g.shape[1], ctx.c, dtype=torch.float32, device=g.device)
            with torch.cuda.device(g.device):
                _lib.call("geom_vertex_head_bwd_f32", rows, ctx.c, g.data_ptr(), ctx.scale, grad_feat.data_ptr())
        return (g if ctx.needs_input_grad[0] else None), grad_feat, None


__all__ = ["face_areas", "device_sum", "SampleFaces", "GatherSqDistSum", "PointToTriangleSum", "SurfaceLoss",
           "Laplacian", "EdgeSqLenSum", "PoolFeatures", "VertexHead", "SegmentMax", "manual_seed", "set_rng_state",
           "draw_samples", "GtIndex", "ScanPrep",
           "chamfer_nn", "tri_distance_indexed"]
