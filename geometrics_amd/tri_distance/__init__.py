"""Drop-in for the reference package `tri_distance` (tri_distance/tri_distance.py:9-43).

    TriDistance()(xyz1, tri1, tri2, tri3) -> (dist, point, index)

dist [B,N] f32 squared distance to the chosen closest point, point [B,N] i32 region code 0..6,
index [B,N] i32 winning triangle; all non-differentiable like the reference's.
`tri_distance_indexed` takes (verts, faces) and gathers the corners in-kernel.
"""
import torch

from .. import _lib


def _outputs(b, n, dev):
    return (torch.empty(b, n, dtype=torch.float32, device=dev),
            torch.empty(b, n, dtype=torch.int32, device=dev),
            torch.empty(b, n, dtype=torch.int32, device=dev))


def _workspace(b, n, m, dev):
    """Scratch for the per-triangle {sphere, corners} records (64 B per triangle); torch's caching
    allocator makes this free after the first call and keeps it hipGraph-capturable."""
    nbytes = _lib.lib().geom_tri_distance_workspace_bytes(b, n, m)
    return torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=dev), nbytes


def forward_cuda(xyz1, tri1, tri2, tri3, dist, point, index, flags=0, use_workspace=True):
    """Same call shape as the reference's pybind `tri.forward_cuda` (tri_distance.cpp:16-30,34-36)."""
    b, n, _ = xyz1.shape
    m = tri1.shape[1]
    with torch.cuda.device(xyz1.device):
        if use_workspace:
            ws, nbytes = _workspace(b, n, m, xyz1.device)
            code = _lib.lib().geom_tri_distance_ws_f32(
                b, n, xyz1.data_ptr(), m, tri1.data_ptr(), tri2.data_ptr(), tri3.data_ptr(),
                dist.data_ptr(), point.data_ptr(), index.data_ptr(), flags, ws.data_ptr(), nbytes,
                _lib.stream_ptr())
        else:
            code = _lib.lib().geom_tri_distance_f32(
                b, n, xyz1.data_ptr(), m, tri1.data_ptr(), tri2.data_ptr(), tri3.data_ptr(),
                dist.data_ptr(), point.data_ptr(), index.data_ptr(), flags, _lib.stream_ptr())
    _lib.check(code, "geom_tri_distance_f32")


def tri_distance(xyz1, tri1, tri2, tri3, flags=0, use_workspace=True):
    xyz1 = _lib.require(xyz1.detach(), "xyz1", torch.float32, 3, 3)
    tris = [_lib.require(t.detach(), "tri%d" % (i + 1), torch.float32, 3, 3) for i, t in enumerate((tri1, tri2, tri3))]
    dev = _lib.same_device(xyz1, *tris)
    b, n, _ = xyz1.shape
    for t in tris:
        if t.shape != tris[0].shape or t.shape[0] != b:
            raise RuntimeError("tri1/tri2/tri3 must share one [B,M,3] shape with xyz1's batch")
    dist, point, index = _outputs(b, n, dev)
    forward_cuda(xyz1, tris[0], tris[1], tris[2], dist, point, index, flags, use_workspace)
    return dist, point, index


def tri_distance_indexed(xyz1, verts, faces, flags=0, use_workspace=True):
    """Corners gathered in-kernel from verts [B,V,3] through faces [F,3] (int64)."""
    xyz1 = _lib.require(xyz1.detach(), "xyz1", torch.float32, 3, 3)
    verts = _lib.require(verts.detach(), "verts", torch.float32, 3, 3)
    faces = _lib.require(faces, "faces", torch.int64, 2, 3)
    dev = _lib.same_device(xyz1, verts, faces)
    b, n, _ = xyz1.shape
    if verts.shape[0] != b:
        raise RuntimeError("verts and xyz1 batch sizes differ")
    dist, point, index = _outputs(b, n, dev)
    with torch.cuda.device(dev):
        if use_workspace:
            ws, nbytes = _workspace(b, n, faces.shape[0], dev)
            code = _lib.lib().geom_tri_distance_indexed_ws_f32(
                b, n, xyz1.data_ptr(), verts.shape[1], verts.data_ptr(), faces.shape[0], faces.data_ptr(),
                dist.data_ptr(), point.data_ptr(), index.data_ptr(), flags, ws.data_ptr(), nbytes,
                _lib.stream_ptr())
        else:
            code = _lib.lib().geom_tri_distance_indexed_f32(
                b, n, xyz1.data_ptr(), verts.shape[1], verts.data_ptr(), faces.shape[0], faces.data_ptr(),
                dist.data_ptr(), point.data_ptr(), index.data_ptr(), flags, _lib.stream_ptr())
    _lib.check(code, "geom_tri_distance_indexed_f32")
    return dist, point, index


class TriDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, tri1, tri2, tri3):
        dist, point, index = tri_distance(xyz1, tri1, tri2, tri3)
        ctx.mark_non_differentiable(dist, point, index)
        return dist, point, index

    @staticmethod
    def backward(ctx, *grads):
        return None, None, None, None


class TriDistance(torch.nn.Module):
    def forward(self, xyz1, tri1, tri2, tri3):
        return TriDistanceFunction.apply(xyz1, tri1, tri2, tri3)
