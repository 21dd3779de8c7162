"""The any-shape matrix-core product (csrc/dense_any.hip, `geom_gemm_f32`) against float64: the three operand forms of a
layer's `torch.mm(input, weight)` (reference layers.py:30) and its two gradients, at the mesh encoder's widths (reference
models.py:299-348: 3, 60, 120, 150, 200, 210, 250, 300; latent 50), at ragged row counts, with odd pitches and 4-byte
aligned views.  Exact fp32 arithmetic in another summation order than any other product: every element within
(summed extent) * eps * sum_k |a||b| of the float64 value -- the bound that holds for ANY order of fp32 additions."""
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -24


def _check(got, a64, b64):
    want = a64 @ b64
    mass = a64.abs() @ b64.abs()
    k = a64.shape[1]
    bound = (k + 8) * EPS * mass + 1e-30
    err = (got.double().cpu() - want).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, "worst element at %.3f of its bound" % worst
    # and far inside it on average: the bound is a worst case, a wrong summed index would not hide in it
    assert float(err.norm()) <= 4 * EPS * (k ** 0.5) * float(mass.norm()) + 1e-30


SHAPES = [(18432, 300, 300), (18432, 3, 60), (18432, 300, 50), (4099, 210, 250), (1000, 150, 200), (777, 61, 77), (64, 64, 64),
          (257, 963, 192), (300, 5, 3), (129, 1, 1)]


@pytest.mark.parametrize("rows,cin,c", SHAPES)
def test_forward_form(gpu, rows, cin, c):
    from geometrics_amd import dense
    g = torch.Generator(device="cpu").manual_seed(rows + cin + c)
    x, w = torch.randn(rows, cin, generator=g), torch.randn(cin, c, generator=g)
    out = dense.gemm(x.cuda(), w.cuda())
    _check(out, x.double(), w.double())
    assert torch.equal(out, dense.gemm(x.cuda(), w.cuda()))          # bit-reproducible


@pytest.mark.parametrize("rows,cin,c", SHAPES)
def test_input_gradient_form(gpu, rows, cin, c):
    from geometrics_amd import dense
    g = torch.Generator(device="cpu").manual_seed(rows + cin + c + 1)
    gr, w = torch.randn(rows, c, generator=g), torch.randn(cin, c, generator=g)
    out = dense.gemm(gr.cuda(), w.cuda(), trans_b=True)
    _check(out, gr.double(), w.double().t())


@pytest.mark.parametrize("rows,cin,c", SHAPES)
def test_weight_gradient_form_with_and_without_the_split(gpu, rows, cin, c):
    from geometrics_amd import _lib, dense
    g = torch.Generator(device="cpu").manual_seed(rows + cin + c + 2)
    x, gr = torch.randn(rows, cin, generator=g), torch.randn(rows, c, generator=g)
    xd, gd = x.cuda(), gr.cuda()
    out = dense.gemm(xd, gd, trans_a=True)
    _check(out, x.double().t(), gr.double())
    assert torch.equal(out, dense.gemm(xd, gd, trans_a=True))        # the split's fixed order: bit-reproducible
    # without a workspace the same entry point sums in one pass
    one = torch.empty(cin, c, device="cuda")
    with torch.cuda.device(0):
        _lib.call("geom_gemm_f32", cin, c, rows, xd.data_ptr(), cin, 1, gd.data_ptr(), c, 1, one.data_ptr(), c, None, 0)
    _check(one, x.double().t(), gr.double())
    if rows >= 4096:
        assert int(_lib.lib().geom_gemm_workspace_floats(cin, c, rows)) > 0


def test_pitched_and_misaligned_views(gpu):
    """Row pitches larger than the extents, bases at odd 4-byte offsets, an output written into a wider matrix: only the
    addressed elements are read and written."""
    from geometrics_amd import dense
    g = torch.Generator(device="cpu").manual_seed(11)
    big_a = torch.randn(500, 131, generator=g).cuda()
    big_b = torch.randn(90, 77, generator=g).cuda()
    a = big_a[:, 1:88]                 # [500, 87], pitch 131, base + 4 bytes
    b = big_b[1:88, 3:70]              # [87, 67], pitch 77
    frame = torch.full((500, 101), 7.0, device="cuda")
    out = frame[:, 5:72]
    dense.gemm(a, b, out=out)
    _check(out, a.double().cpu(), b.double().cpu())
    keep = torch.ones(500, 101, dtype=torch.bool)
    keep[:, 5:72] = False
    assert bool((frame.cpu()[keep] == 7.0).all())
    # non-finite values outside the addressed extents must not leak in (zero fill of the ragged edge, not clamped reads)
    big_a2 = big_a.clone()
    big_a2[:, 88:] = float("nan")
    big_a2[:, 0] = float("inf")
    out2 = dense.gemm(big_a2[:, 1:88], b)
    assert torch.equal(out2, out)


def test_empty_extents_and_argument_errors(gpu):
    from geometrics_amd import _lib, dense
    assert dense.gemm(torch.zeros(0, 5, device="cuda"), torch.zeros(5, 4, device="cuda")).shape == (0, 4)
    z = dense.gemm(torch.zeros(6, 0, device="cuda"), torch.zeros(0, 4, device="cuda"), out=torch.ones(6, 4, device="cuda"))
    assert bool((z == 0).all())                                         # an empty sum
    with pytest.raises(ValueError):
        dense.gemm(torch.zeros(6, 3, device="cuda"), torch.zeros(4, 4, device="cuda"))
    x = torch.zeros(16, device="cuda")
    assert _lib.lib().geom_gemm_f32(4, 4, 4, x.data_ptr(), 2, 0, x.data_ptr(), 4, 1, x.data_ptr(), 4, None, 0, None) == -1   # lda < k


def test_layer_through_autograd_matches_the_library_route(gpu):
    """A ZERON_GCN layer at encoder widths: forward and all gradients with the any-shape kernels against the library's
    products (torch.matmul + autograd) -- the same values within fp32 summation order."""
    import numpy as np
    import torch.nn.functional as F
    from geometrics_amd import layers, meshgen, utils
    V, Fc = meshgen.icosphere(3)
    adj = utils.normalize_adj(utils.calc_adj(torch.from_numpy(np.ascontiguousarray(Fc)).cuda()))
    torch.manual_seed(3)
    layer = layers.ZERON_GCN(210, 250).cuda()
    x = torch.randn(V.shape[0], 210, device="cuda", requires_grad=True)
    seed = torch.randn(V.shape[0], 250, device="cuda")
    res = {}
    for mode in (True, False):
        layers.use_any_shape_products = mode
        try:
            layer.zero_grad()
            x.grad = None
            out = layer(x, adj, F.elu)
            out.backward(seed)
            res[mode] = [out.detach().clone(), x.grad.clone(), layer.weight.grad.clone(), layer.bias.grad.clone()]
        finally:
            layers.use_any_shape_products = True
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()


def test_capturable_and_allocation_free(gpu):
    """The entry point allocates nothing and synchronises nothing: a product captured into a HIP graph (workspace supplied by
    the caller, as in a captured training step) replays to the bits of the eager call -- forward and split weight gradient."""
    from geometrics_amd import _lib, dense
    g = torch.Generator(device="cpu").manual_seed(5)
    x, w, gr = torch.randn(9000, 150, generator=g).cuda(), torch.randn(150, 210, generator=g).cuda(), torch.randn(9000, 210, generator=g).cuda()
    want_f, want_w = dense.gemm(x, w), dense.gemm(x, gr, trans_a=True)
    out_f, out_w = torch.zeros_like(want_f), torch.zeros_like(want_w)
    ws = torch.empty(int(_lib.lib().geom_gemm_workspace_floats(150, 210, 9000)), device="cuda")
    assert ws.numel() > 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph):
            dense.gemm(x, w, out=out_f)
            dense.gemm(x, gr, trans_a=True, out=out_w, workspace=ws)
    torch.cuda.current_stream().wait_stream(side)
    out_f.zero_(), out_w.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_f, want_f) and torch.equal(out_w, want_w)


def test_layer_with_odd_width_and_a_strided_input(gpu):
    """ZERON_GCN at a width that is neither a multiple of 4 nor of 2 (75 columns, k = 7: scalar column groups in the table
    aggregation, scalar stores in the product) fed with a NON-contiguous input view: forward and every gradient against a
    float64 restatement of the layer (reference layers.py:34-41)."""
    import numpy as np
    import torch.nn.functional as F
    from geometrics_amd import layers, meshgen, utils
    V, Fc = meshgen.icosphere(3)
    adj = utils.normalize_adj(utils.calc_adj(torch.from_numpy(np.ascontiguousarray(Fc)).cuda()))
    torch.manual_seed(8)
    layer = layers.ZERON_GCN(33, 75).cuda()
    base = torch.randn(33, V.shape[0], device="cuda")
    x = base.t().requires_grad_(True)                       # [V, 33] with strides (1, V): not contiguous
    assert not x.is_contiguous()
    seed = torch.randn(V.shape[0], 75, device="cuda")
    out = layer(x, adj, F.elu)
    out.backward(seed)
    xd = base.t().double().cpu().requires_grad_(True)
    wd, bd = layer.weight.detach().double().cpu().requires_grad_(True), layer.bias.detach().double().cpu().requires_grad_(True)
    s = xd @ wd
    k = 75 // 10
    ref = F.elu(torch.cat((adj.double().cpu() @ s[:, :k], s[:, k:]), dim=1) + bd)
    ref.backward(seed.double().cpu())
    for got, want in ((out, ref), (x.grad, xd.grad), (layer.weight.grad, wd.grad), (layer.bias.grad, bd.grad)):
        assert (got.detach().double().cpu() - want.detach()).abs().max().item() <= 2e-5 * want.detach().abs().max().item()
