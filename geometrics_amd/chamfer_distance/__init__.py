"""Drop-in for the reference package `chamfer_distance` (chamfer_distance/chamfer_distance.py:9-38).

    ChamferDistance()(xyz1, xyz2) -> (idx1, idx2)      int32, non-differentiable

idx1[b,i] = arg-min_j |xyz2[b,j]-xyz1[b,i]|^2, idx2 the other direction.  Differences from the
reference wrapper: outputs are allocated on the device (no four host allocations + H2D
copies per call), the launch goes to torch's current stream, inputs are validated and
launch failures raise instead of printing.  `chamfer_nn` additionally returns the
squared distances the kernel computes anyway.
"""
import torch

from .. import _lib


_arithmetic_flags = 0


def set_arithmetic(mode):
    """Canonical arithmetic of the squared distance in every NN scan issued without explicit flags:
    "unfused" (default) = (dx*dx + dy*dy) + dz*dz, bit-identical to the reference's CPU nnsearch as shipped;
    "fma" = fma(dz, dz, fma(dx, dx, dy*dy)), bit-identical to the same source built with FMA contraction (what
    gcc -mfma -ffp-contract=fast and nvcc's default produce).  Indices agree between the two except on ties within
    one rounding; distances differ by round-off (<= 1e-7 relative)."""
    global _arithmetic_flags
    if mode not in ("unfused", "fma"):
        raise ValueError("arithmetic must be 'unfused' or 'fma'")
    _arithmetic_flags = _lib.FLAG_NN_FMA if mode == "fma" else 0


def default_flags():
    """Flags of an NN scan issued without flags of its own: the package arithmetic + the reference-quirk mode
    (geometrics_amd.set_reference_quirks / GEOM_REF_QUIRKS: the shipped CUDA kernel's tail truncation)."""
    quirks = _lib.quirk_flags()
    if quirks and _arithmetic_flags:
        raise RuntimeError("reference-quirk mode reproduces the shipped kernel's tile structure in the un-fused arithmetic "
                           "only: chamfer_distance.set_arithmetic('unfused') or switch set_reference_quirks(False)")
    return _arithmetic_flags | quirks


# Clouds with at least this many (query, target) pairs per mesh go through the culled scan on Morton orders made on the
# device (a dozen small launches): same outputs, bit for bit; measured break-even incl. the ordering ~30 000 x 30 000
# points, 7x at 100 000 x 100 000 (profiles/r03_culled_chamfer.txt).  Never during a HIP-graph capture (the ordering is
# a chain of library launches) and never in the reference-tail-truncation mode.  0 switches the dispatch off.
AUTO_CULL_PAIRS = 2_000_000_000


def chamfer_nn(xyz1, xyz2, flags=None):
    """(dist1 [B,N] f32, idx1 [B,N] i32, dist2 [B,M] f32, idx2 [B,M] i32)."""
    if flags is None:
        flags = default_flags()
    xyz1 = _lib.require(xyz1.detach(), "xyz1", torch.float32, 3, 3)
    xyz2 = _lib.require(xyz2.detach(), "xyz2", torch.float32, 3, 3)
    dev = _lib.same_device(xyz1, xyz2)
    b, n, _ = xyz1.shape
    b2, m, _ = xyz2.shape
    if b != b2:
        raise RuntimeError("batch sizes differ: %d vs %d" % (b, b2))
    if (AUTO_CULL_PAIRS and n * m >= AUTO_CULL_PAIRS and b > 0 and not (flags & ~_lib.FLAG_NN_FMA)
            and not torch.cuda.is_current_stream_capturing()):
        return chamfer_nn_culled(xyz1, xyz2, "morton", "morton", flags)
    dist1 = torch.empty(b, n, dtype=torch.float32, device=dev)
    dist2 = torch.empty(b, m, dtype=torch.float32, device=dev)
    idx1 = torch.empty(b, n, dtype=torch.int32, device=dev)
    idx2 = torch.empty(b, m, dtype=torch.int32, device=dev)
    forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2, flags)
    return dist1, idx1, dist2, idx2


def chamfer_nn_culled(xyz1, xyz2, order1="morton", order2="morton", flags=None, validate=True):
    """chamfer_nn through the culled scan (csrc/nn_scan.h nn_culled_body): the same four tensors, bit for bit, for ANY
    visiting orders that are PERMUTATIONS of the clouds (int32 [B,N] / [B,M], "morton" = made here on the device, None = the
    clouds' own order); coherent orders let the scan skip most (query tile, target run) pairs -- an incoherent permutation
    only costs speed.  A tensor that is NOT a permutation (an entry out of range, or one repeated) is refused: the kernels
    index the clouds and the outputs with its entries unchecked, so it would read and write out of bounds or leave outputs
    unwritten.  The check (sort + compare, one host sync per supplied order) is skipped only with validate=False or while
    the stream is being captured -- then the caller vouches for the order.  The surface step has its own route to this
    scan (ops.GtIndex); this entry serves stand-alone Chamfer calls on clouds that are reused or already ordered."""
    if flags is None:
        flags = default_flags()
    xyz1 = _lib.require(xyz1.detach(), "xyz1", torch.float32, 3, 3)
    xyz2 = _lib.require(xyz2.detach(), "xyz2", torch.float32, 3, 3)
    dev = _lib.same_device(xyz1, xyz2)
    b, n, _ = xyz1.shape
    b2, m, _ = xyz2.shape
    if b != b2:
        raise RuntimeError("batch sizes differ: %d vs %d" % (b, b2))

    def visiting(order, cloud, name):
        if isinstance(order, str):
            if order != "morton":
                raise ValueError("order must be a tensor, None or 'morton'")
            from ..tri_distance import morton_order
            return torch.stack([morton_order(cloud[i]) for i in range(b)]) if b else None
        if order is None:
            return None
        order = _lib.require(order, name, torch.int32, 2)
        if order.shape != cloud.shape[:2] or order.device != dev:
            raise RuntimeError("%s must be an int32 [B,N] permutation per cloud on the clouds' device" % name)
        if validate and order.numel() and not torch.cuda.is_current_stream_capturing():
            want = torch.arange(order.shape[1], dtype=torch.int32, device=dev)
            if not bool((order.sort(dim=1).values == want).all()):
                raise ValueError("%s is not a permutation of 0..%d in every row (out-of-range or repeated entries)"
                                 % (name, order.shape[1] - 1))
        return order

    o1, o2 = visiting(order1, xyz1, "order1"), visiting(order2, xyz2, "order2")
    dist1 = torch.empty(b, n, dtype=torch.float32, device=dev)
    dist2 = torch.empty(b, m, dtype=torch.float32, device=dev)
    idx1 = torch.empty(b, n, dtype=torch.int32, device=dev)
    idx2 = torch.empty(b, m, dtype=torch.int32, device=dev)
    L = _lib.lib()
    ws = torch.empty(max(int(L.geom_chamfer_nn_culled_workspace_floats(b, n, m)), 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("geom_chamfer_nn_culled_f32", b, n, xyz1.data_ptr(), m, xyz2.data_ptr(), _lib.ptr(o1), _lib.ptr(o2),
                  dist1.data_ptr(), idx1.data_ptr(), dist2.data_ptr(), idx2.data_ptr(), flags, ws.data_ptr())
    return dist1, idx1, dist2, idx2


def forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2, flags=None):
    """Same call shape as the reference's pybind `cd.forward_cuda` (chamfer_distance.cpp:15-27,36-38):
    caller-allocated outputs, filled in place."""
    if flags is None:
        flags = default_flags()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    with torch.cuda.device(xyz1.device):
        code = _lib.lib().geom_chamfer_nn_f32(
            b, n, xyz1.data_ptr(), m, xyz2.data_ptr(),
            dist1.data_ptr(), idx1.data_ptr(), dist2.data_ptr(), idx2.data_ptr(),
            flags, _lib.stream_ptr())
    _lib.check(code, "geom_chamfer_nn_f32")


class ChamferDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        _, idx1, _, idx2 = chamfer_nn(xyz1, xyz2)
        ctx.mark_non_differentiable(idx1, idx2)
        return idx1, idx2

    @staticmethod
    def backward(ctx, *grads):  # integer outputs: nothing flows (the reference defines no backward)
        return None, None


class ChamferDistance(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        return ChamferDistanceFunction.apply(xyz1, xyz2)
