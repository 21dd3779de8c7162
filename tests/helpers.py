import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def fill_parameters(module, seed, gain=2.0):
    """Deterministic parameter values independent of torch's RNG stream (so a fixture only stores the seed):
    matrices ~ U(+-gain/sqrt(sum(shape))), vectors ~ U(+-0.1), in sorted-name order."""
    import zlib

    import torch
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
            bound = gain / np.sqrt(sum(p.shape)) if p.dim() >= 2 else 0.1
            p.copy_(torch.from_numpy(rng.uniform(-bound, bound, tuple(p.shape)).astype(np.float32)))
    return module


def tri_true_case(name):
    """Inputs of a tests/golden/tri_true_*.npz fixture, regenerated from their seeds (the fixture stores the expected
    per-point true squared distances + a checksum of the inputs): (verts [B,V,3], faces [F,3], points [B,N,3], true [B,N])."""
    from geometrics_amd import meshgen
    g = golden(name)
    V, F = meshgen.icosphere(int(g["level"]))
    first = int(g["first"]) if "first" in g else 0
    verts = meshgen.jittered_batch(V, int(g["batch"]), first=first)
    pts = meshgen.gt_cloud(int(g["batch"]), int(g["num"]), first=first, cube=bool(g["cube"]))
    assert float(verts.astype(np.float64).sum() + pts.astype(np.float64).sum()) == float(g["checksum"])
    return verts, F, pts, g["true_sqdist"]


# ---- per-row gradient bounds (round-3 review: the surface backward is a deterministic gather, so it can be held to more than
# a tensor-wide max-norm) --------------------------------------------------------------------------------------------------
def fp64_surface_gradient(verts, faces, gt, choices, u, v, two_sided, scale=3000.0, tri_flags=0):
    """The gradient of batch_point_to_point / batch_point_to_surface (reference utils.py:393-502) with respect to the
    vertices, evaluated in FLOAT64 from the closed form, TERM BY TERM: every loss term (one per sampled point, one per gt
    point) owns a private copy of the 3 corners it depends on, autograd gives the term's corner gradients, and a scatter-add
    of them is the exact gradient `grad` [B,V,3]; the scatter-add of their absolute values is `mass` [B,V,3], the sum of
    |contributions| that meet in a row -- the scale an fp32 evaluation's round-off is proportional to, whatever cancels.
    Arg-min indices come from the oracle on the fp32 sampled points (what the HIP path reproduces bit for bit).
    Inputs are numpy arrays; returns (loss, grad, mass) as float64 numpy."""
    import torch
    import oracle
    from oracle import ref_ops
    tv, tf = torch.from_numpy(np.asarray(verts)), torch.from_numpy(np.asarray(faces)).long()
    tg = torch.from_numpy(np.asarray(gt))
    ch = torch.from_numpy(np.asarray(choices)).long()
    tu, tw = torch.from_numpy(np.asarray(u)), torch.from_numpy(np.asarray(v))
    b, nv, _ = tv.shape
    num, n_gt = ch.shape[1], tg.shape[1]
    pred32 = ref_ops.sample_points(tv, tf, ch, tu, tw)                    # fp32, bitwise what the kernel produces
    _, idx_p, _, idx_g = oracle.chamfer_nn(tg.numpy(), pred32.numpy())
    idx_p, idx_g = torch.from_numpy(idx_p).long(), torch.from_numpy(idx_g).long()
    dv, dg, du, dw = tv.double(), tg.double(), tu.double(), tw.double()

    def corners_of(face_ids):                                             # [B,N] face ids -> vertex ids [B,N,3], leaf corners [B,N,3,3]
        vid = tf[face_ids]
        c = torch.gather(dv, 1, vid.reshape(b, -1, 1).expand(-1, -1, 3)).reshape(b, -1, 3, 3)
        return vid, c.clone().requires_grad_(True)

    def sample(c, uu, ww):
        return (1 - uu)[..., None] * c[:, :, 0] + (uu * (1 - ww))[..., None] * c[:, :, 1] + (uu * ww)[..., None] * c[:, :, 2]

    groups = []
    # terms of the sampled points: |gt[nn(s)] - pred_s|^2
    vid_s, c_s = corners_of(ch)
    near = torch.gather(dg, 1, idx_g.unsqueeze(-1).expand(-1, -1, 3))
    loss = (scale / (b * num)) * ((near - sample(c_s, du, dw)) ** 2).sum()
    groups.append((vid_s, c_s))
    if two_sided:     # terms of the gt points: |pred[nn(g)] - g|^2 -- the corners of the face sample nn(g) was drawn on
        face_g = torch.gather(ch, 1, idx_p)
        vid_g, c_g = corners_of(face_g)
        ug, wg = torch.gather(du, 1, idx_p), torch.gather(dw, 1, idx_p)
        loss = loss + (scale / (b * n_gt)) * ((sample(c_g, ug, wg) - dg) ** 2).sum()
    else:             # |closest_on_triangle(g) - g|^2 for the winner of the point-to-triangle scan (its region code selects the formula)
        _, opt, tri = oracle.tri_scan_indexed(tg.numpy(), tv.numpy(), tf.numpy(), tri_flags)
        vid_g, c_g = corners_of(torch.from_numpy(tri).long())
        flat = c_g.reshape(-1, 3, 3)
        closest = ref_ops.closest_point(dg.reshape(-1, 3), flat[:, 0], flat[:, 1], flat[:, 2], torch.from_numpy(opt).reshape(-1))
        loss = loss + (scale / (b * n_gt)) * ((closest - dg.reshape(-1, 3)) ** 2).sum()
    groups.append((vid_g, c_g))
    loss.backward()
    grad = torch.zeros(b, nv, 3, dtype=torch.float64)
    mass = torch.zeros(b, nv, 3, dtype=torch.float64)
    floor = torch.zeros(b, nv, 3, dtype=torch.float64)
    # Every term is 2 * coef * (a DIFFERENCE of two fp32 coordinates) * (a weight <= 1): the difference carries the absolute
    # rounding of the coordinates themselves (ulps of max|coordinate|), however small it is -- a gt point lying 1e-5 from the
    # surface has a term a thousand times smaller than that rounding.  `floor` = one coordinate ulp through every term that
    # meets in the element; the bound is rtol * mass + floor_ulps * floor.
    ulp = float(np.finfo(np.float32).eps) * float(max(np.abs(verts).max(), np.abs(gt).max()))
    for (vid, c), coef in zip(groups, (scale / (b * num), scale / (b * n_gt))):
        index = vid.reshape(b, -1, 1).expand(-1, -1, 3)
        g = c.grad.reshape(b, -1, 3)
        grad.scatter_add_(1, index, g)
        mass.scatter_add_(1, index, g.abs())
        floor.scatter_add_(1, index, torch.full_like(g, 2.0 * coef * ulp))
    return float(loss.detach()), grad.numpy(), mass.numpy(), floor.numpy()


def rows_close(actual, exact, mass, rtol, what="", floor=None, floor_ulps=0.0):
    """|actual - exact| <= rtol * mass + floor_ulps * floor ELEMENT BY ELEMENT, where mass = the sum of the absolute
    contributions that meet in the element (and floor = one coordinate ulp through each of them, see
    fp64_surface_gradient): a wrong neighbour or a dropped term on a low-gradient vertex fails here even when the tensor's
    largest entry is 1000x bigger (the max-norm `close` of test_ops_parity_gpu.py would let it pass)."""
    actual, exact, mass = (np.asarray(x, np.float64) for x in (actual, exact, mass))
    err = np.abs(actual - exact)
    bound = rtol * mass + (0.0 if floor is None else floor_ulps * np.asarray(floor, np.float64)) + 1e-30
    worst = float((err / bound).max())
    log = os.environ.get("GEOM_MARGIN_LOG")
    if log:                       # margins of a run, for choosing / reporting the bounds (LAB_NOTES.md quotes them)
        with open(log, "a") as f:
            f.write("%s: worst element at %.3g of its bound (rtol %g, floor %g ulp), max abs err %.3g, max |exact| %.3g\n"
                    % (what, worst, rtol, floor_ulps, err.max(), np.abs(exact).max()))
    assert worst <= 1.0, "%s: worst element is %.2fx its bound (rtol %g of the row's term mass + %g coordinate ulps per term); " \
                         "max err %g" % (what, worst, rtol, floor_ulps, err.max())
    return worst
