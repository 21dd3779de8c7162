"""Functional + autograd layer over the C ABI for the HBM-bound stages (sampling, losses).

Every function launches on torch's current stream, allocates its outputs with torch and
never synchronises with the host, so a whole training step can be captured in a HIP graph.
"""
import torch

from . import _lib
from .chamfer_distance import chamfer_nn
from .tri_distance import tri_distance_indexed


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


def draw_samples(verts, faces, num, generator=None):
    """The random part of batch_sample (reference utils.py:604-612, 627-628) in three batched
    calls instead of a python loop of B multinomials: choices [B,num] ~ area-weighted with
    replacement, u = sqrt(U1), v = U2."""
    areas = face_areas(verts, faces)
    choices = torch.multinomial(areas, num, True, generator=generator)
    uv = torch.rand(2, areas.shape[0], num, device=areas.device, generator=generator)
    return choices, torch.sqrt(uv[0]), uv[1]


__all__ = ["face_areas", "device_sum", "SampleFaces", "GatherSqDistSum", "PointToTriangleSum", "draw_samples",
           "chamfer_nn", "tri_distance_indexed"]
