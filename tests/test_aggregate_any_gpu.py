"""The any-width table aggregation (csrc/zn_gcn.hip: zn_aggregate_ell_any_kernel) -- the unbatched ZERON_GCN layers of the
mesh encoder (reference layers.py:34-41 at models.py:299-348's widths: k = c // 10 is 6 ... 30, not a multiple of 4, c not
a multiple of k) -- against the generic CSR kernel: same neighbour order, so the SAME BITS, forward and backward; the bias
gradient (another partial layout) against float64.  Meshes: an icosphere, the block-diagonal ragged batch of the encoder
step, and the reference's 482.obj template whose two 33-entry pole rows continue in the CSR tail."""
import numpy as np
import pytest
import torch

from helpers import golden
from geometrics_amd import _lib as L
from geometrics_amd import layers, meshgen, utils

pytestmark = pytest.mark.gpu


def dev(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def _csr_for(kind, gpu):
    if kind == "482.obj":
        faces = dev(golden("adj_482")["faces"], gpu)
        return layers.adjacency_csr(utils.adj_init(faces)["adj"])
    if kind == "icosphere":
        return layers.adjacency_csr(utils.adj_init(dev(meshgen.icosphere(3)[1], gpu))["adj"])
    from geometrics_amd import ragged
    verts, faces = [], []
    for i, lv in enumerate([2, 3, 2, 4, 3]):
        V, Fc = meshgen.icosphere(lv)
        verts.append(torch.from_numpy(meshgen.jittered_batch(V, 1, first=i)[0]).to(gpu))
        faces.append(torch.from_numpy(np.ascontiguousarray(Fc)).to(gpu))
    return ragged.RaggedMeshBatch.from_faces(verts, faces).csr


@pytest.mark.parametrize("kind", ["icosphere", "ragged", "482.obj"])
@pytest.mark.parametrize("c,k", [(300, 30), (60, 6), (250, 25), (210, 21), (150, 15), (50, 5), (7, 3), (192, 20), (64, 64), (48, 0)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_any_width_table_kernel_equals_the_csr_kernel(gpu, kind, c, k, act):
    csr = _csr_for(kind, gpu)
    assert csr.ell_w == 8 and (kind != "482.obj" or csr.over is not None)
    b, nv = (3, csr.nv) if kind != "ragged" else (1, csr.nv)
    g = torch.Generator(device="cpu").manual_seed(c * 7 + k + act)
    sup = torch.randn(b, nv, c, generator=g).to(gpu)
    bias = torch.randn(c, generator=g).to(gpu)
    gout = torch.randn(b, nv, c, generator=g).to(gpu)
    over, over_t = csr.over or (None, None, None), csr.over_t or (None, None, None)
    want, got = torch.empty_like(sup), torch.full_like(sup, float("nan"))
    L.call("geom_zn_gcn_aggregate_fwd_f32", b, nv, c, k, csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.val.data_ptr(),
           sup.data_ptr(), bias.data_ptr(), act, want.data_ptr())
    L.call("geom_zn_gcn_aggregate_ell_fwd_f32", b, nv, c, k, csr.ell_w, csr.ell_col.data_ptr(), csr.ell_val.data_ptr(),
           L.ptr(over[0]), L.ptr(over[1]), L.ptr(over[2]), sup.data_ptr(), bias.data_ptr(), act, got.data_ptr(), None)
    assert torch.equal(got, want)
    scr = torch.empty(L.lib().geom_zn_gcn_bwd_scratch_floats(b, nv, c), device=gpu)
    gs_want, gb_want = torch.empty_like(sup), torch.empty(c, device=gpu)
    L.call("geom_zn_gcn_aggregate_bwd_f32", b, nv, c, k, csr.rowptr_t.data_ptr(), csr.col_t.data_ptr(), csr.val_t.data_ptr(),
           gout.data_ptr(), want.data_ptr(), act, gs_want.data_ptr(), gb_want.data_ptr(), scr.data_ptr())
    gs, gb = torch.full_like(sup, float("nan")), torch.full((c,), float("nan"), device=gpu)
    scr2 = torch.full_like(scr, float("nan"))
    L.call("geom_zn_gcn_aggregate_ell_bwd_f32", b, nv, c, k, csr.ell_w, csr.ell_col_t.data_ptr(), csr.ell_val_t.data_ptr(),
           L.ptr(over_t[0]), L.ptr(over_t[1]), L.ptr(over_t[2]), gout.data_ptr(), want.data_ptr(), None, act, gs.data_ptr(),
           gb.data_ptr(), scr2.data_ptr())
    assert torch.equal(gs, gs_want)
    # bias gradient: column sums of g' in another (fixed) order
    gp = gout.double()
    if act == 1:
        gp = gp * (want > 0)
    elif act == 2:
        gp = torch.where(want > 0, gp, gp * (want.double() + 1))
    ref = gp.view(-1, c).sum(0)
    mass = gp.abs().view(-1, c).sum(0)
    assert bool(((gb.double() - ref).abs() <= 1e-6 * mass + 1e-30).all())
    # the partial rows the deferred reduction is told about are the rows the launch wrote
    rows = int(L.lib().geom_zn_gcn_bwd_partial_rows(b, nv, c, k, csr.ell_w))
    assert rows > 0 and rows * c <= scr2.numel()
    part = scr2[:rows * c].view(rows, c)
    assert not bool(torch.isnan(part).any()) and (rows * c == scr2.numel() or bool(torch.isnan(scr2[rows * c:]).all()))
    assert bool(((part.double().sum(0) - ref).abs() <= 1e-6 * mass + 1e-30).all())


def test_encoder_layers_take_the_table_kernel_through_autograd(gpu):
    """A ZERON_GCN layer at an encoder width on the ragged batch, forward + backward through the module: equal to the same
    layer with the table switched off (generic CSR kernel) -- aggregation bits identical, so everything is."""
    import torch.nn.functional as F
    csr = _csr_for("ragged", gpu)
    torch.manual_seed(4)
    layer = layers.ZERON_GCN(120, 150).to(gpu)
    x = torch.randn(csr.nv, 120, device=gpu, requires_grad=True)
    seed = torch.randn(csr.nv, 150, device=gpu)
    res = []
    for table in (True, False):
        keep = csr.ell_w
        try:
            if not table:
                csr.ell_w = 0
            layer.zero_grad()
            x.grad = None
            out = layer(x, csr, F.elu)
            out.backward(seed)
            res.append([out.detach().clone(), x.grad.clone(), layer.weight.grad.clone(), layer.bias.grad.clone()])
        finally:
            csr.ell_w = keep
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b)
    assert (res[0][3] - res[1][3]).abs().max().item() <= 1e-5 * res[1][3].abs().max().item()
