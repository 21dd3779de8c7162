"""The fp32 matrix-core products of the 0N-GCN layers (csrc/dense_gemm.hip) against float64 products of the same operands.

Reference semantics: `support = torch.matmul(input, weight)` (layers.py:30, 107, 140) and autograd's two gradients of it.
Tolerance: an fp32 fma chain over K terms against float64 is ~1e-7 * sum|a*b| (MI355X guide, FP32-input MFMA); the checks
below allow 2e-6 of the row/column scale -- far inside the 1e-5 the layer fixtures are held to."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    # rows, cin, c
    (20496, 963, 192),   # BASELINE shard, layer 1 (4-byte aligned rows)
    (20496, 192, 192),   # hidden layers
    (7712, 1155, 192),   # reference training shape (16 x 482), block input
    (7712, 192, 192),
    (2562, 963, 192),    # one mesh: fewer tiles than CUs
    (1000, 37, 48),      # odd everything
    (16, 4, 16),
    (83, 192, 192),
]


def _close(got, want, scale_axis=None, tol=5e-6):
    got = got.double().cpu()
    want = want.cpu()
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= tol * scale * 50, (err, scale)      # max-norm
    # relative Frobenius: catches a transposed / shifted tile that a max-norm of similar magnitudes would not
    assert (got - want).norm().item() <= tol * want.norm().item() + 1e-30


# Per-ELEMENT bound (round-4 review: the layer / GEMM gradients are deterministic fixed-order sums and were held to a
# max-norm only): |got - exact| <= ROW_RTOL * sum_t |a_t| |b_t|, the mass of the products that meet in the element.  An fp32
# chain of K terms carries ~sqrt(K) eps of that mass (K = 20 496 rows for a weight gradient: 8e-6; 963 for the first layer's
# products: 2e-6); a dropped or doubled tile row, a wrong split boundary or a mis-indexed leftover column is a whole term --
# 1 / K of the mass at the very least -- in one element, which a max-norm over a 963 x 192 tensor lets through.
# Measured (GEOM_MARGIN_LOG, MI355X): the worst element of any shape sits at 4.1e-7 of its mass (forward 20 496 x 192 x 192),
# the weight gradients at 8e-8 (their 20 496 terms are added as 25-64 partial sums of MFMA chains): the bound is 3.5x that.
ROW_RTOL_GEMM = 1.5e-6


def _rows_close(got, a64, b64, what):
    import os
    exact = a64 @ b64
    mass = a64.abs() @ b64.abs()
    err = (got.double().cpu() - exact.cpu()).abs()
    worst = float((err / (ROW_RTOL_GEMM * mass.cpu() + 1e-30)).max())
    log = os.environ.get("GEOM_MARGIN_LOG")
    if log:
        with open(log, "a") as f:
            f.write("%s: worst element at %.3g of its bound (rtol %g of the element's product mass)\n" % (what, worst, ROW_RTOL_GEMM))
    assert worst <= 1.0, "%s: worst element is %.2fx its bound" % (what, worst)


def _operands(rows, cin, c, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(rows, cin, generator=g)
    w = torch.randn(cin, c, generator=g) * 0.1
    gr = torch.randn(rows, c, generator=g)
    return x.cuda(), w.cuda(), gr.cuda()


@pytest.mark.parametrize("rows,cin,c", SHAPES)
def test_forward_matches_float64(rows, cin, c):
    from geometrics_amd import dense
    x, w, _ = _operands(rows, cin, c)
    out = dense.forward(x, w)
    _close(out, x.double() @ w.double())
    _rows_close(out, x.double(), w.double(), "forward %dx%dx%d" % (rows, cin, c))


@pytest.mark.parametrize("rows,cin,c", SHAPES)
def test_input_gradient_matches_float64(rows, cin, c):
    from geometrics_amd import dense
    x, w, g = _operands(rows, cin, c, 1)
    out = dense.backward_input(g, w)
    _close(out, g.double() @ w.double().t())
    _rows_close(out, g.double(), w.double().t(), "input gradient %dx%dx%d" % (rows, cin, c))


@pytest.mark.parametrize("rows,cin,c", [s if s[2] % 12 == 0 else (s[0], s[1], 48) for s in SHAPES] + [(4000, 100, 96)])
def test_weight_and_bias_gradient_match_float64(rows, cin, c):
    from geometrics_amd import dense
    x, w, g = _operands(rows, cin, c, 2)
    want_bias = cin >= 96          # the column sums ride on the first full output tile
    gw, gb = dense.backward_weight(x, g, want_bias)
    _close(gw, x.double().t() @ g.double())
    _rows_close(gw, x.double().t(), g.double(), "weight gradient %dx%dx%d" % (rows, cin, c))
    if want_bias:
        _close(gb, g.double().sum(0))
    # bit-reproducible: fixed split order
    gw2, _ = dense.backward_weight(x, g, False)
    assert torch.equal(gw, gw2)


@pytest.mark.parametrize("rows,cin", [(20496, 963), (20496, 192), (2562, 192), (83, 192)])
def test_split_epilogue_finishes_the_pass_through_columns(rows, cin):
    """ksplit mode (layers.py:108-116, ReLU): aggregated columns raw and compact, pass-through columns with bias + ReLU,
    sign bits of the pass-through columns."""
    from geometrics_amd import dense
    c, k = 192, 64
    x, w, _ = _operands(rows, cin, c, 3)
    bias = torch.randn(c, device="cuda") * 0.5
    out = torch.full((rows, c), float("nan"), device="cuda")
    sup = torch.empty(rows, k, device="cuda")
    mask = torch.zeros(rows, c // 16, dtype=torch.int16, device="cuda")
    dense.forward_split(x, w, bias, k, out, sup, mask)
    full = dense.forward(x, w)
    assert torch.equal(sup, full[:, :k])                       # same arithmetic, same bits
    assert torch.equal(out[:, k:], torch.relu(full[:, k:] + bias[k:]))
    assert bool(torch.isnan(out[:, :k]).all())                 # the aggregated columns of `out` are not touched
    bits = (mask.view(rows, c // 16, 1).int() & 0xffff) >> torch.arange(16, device="cuda").view(1, 1, 16) & 1
    want = (out[:, k:] > 0).view(rows, (c - k) // 16, 16).int()
    assert torch.equal(bits[:, k // 16:], want)
    assert int(bits[:, :k // 16].sum()) == 0


@pytest.mark.parametrize("rows,cin,c", [(20496, 192, 192), (7712, 192, 192), (2562, 192, 192), (20496, 963, 192), (83, 192, 192)])
def test_pair_launch_equals_the_two_separate_launches(rows, cin, c):
    """geom_dense_bwd_f32 (one launch, two workgroups per CU) runs the same two bodies: same bits as the separate launches."""
    from geometrics_amd import dense
    x, w, g = _operands(rows, cin, c, 4)
    ws = dense.weight_workspace(rows, cin, c, x.device)
    gx = torch.empty(rows, cin, device="cuda")
    dense.backward_pair(x, g, w, gx, ws, want_colsum=True)
    gw = torch.empty(cin, c, device="cuda")
    gb = torch.empty(c, device="cuda")
    dense.reduce([(rows, cin, c, ws, gw, gb)])
    assert torch.equal(gx, dense.backward_input(g, w))
    gw2, gb2 = dense.backward_weight(x, g, True)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    _close(gx, g.double() @ w.double().t())
    _close(gw, x.double().t() @ g.double())
    _rows_close(gx, g.double(), w.double().t(), "pair launch, input gradient %dx%dx%d" % (rows, cin, c))
    _rows_close(gw, x.double().t(), g.double(), "pair launch, weight gradient %dx%dx%d" % (rows, cin, c))


def test_unsupported_shapes_are_refused():
    from geometrics_amd import _lib
    x = torch.zeros(4, 4, device="cuda")
    code = _lib.lib().geom_dense_fwd_f32(4, 4, 200, x.data_ptr(), x.data_ptr(), 0, None, x.data_ptr(), None, None, None)
    assert code == _lib.EUNSUPPORTED
    code = _lib.lib().geom_dense_bwd_weight_f32(4, 4, 16, x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, None)
    assert code == _lib.EUNSUPPORTED          # 16 output columns: not whole 12-column groups
    from geometrics_amd import dense
    assert dense.plan(20496, 192, 64)["dw"] == "lib" and dense.plan(20496, 192, 192)["pair"]


def test_head_mode_equals_aggregation_plus_vertex_head():
    """layers.zero_n_aggregate_head (positions out of the aggregation launches, gradient of the layer output never
    materialised) against zero_n_aggregate + ops.VertexHead: same positions bit for bit, same gradients bit for bit."""
    import torch.nn.functional as F
    from geometrics_amd import layers, meshgen, ops, utils
    for gen, act in ((lambda: meshgen.icosphere(2), F.relu), (meshgen.uv_sphere, F.relu), (lambda: meshgen.icosphere(2), None)):
        V, Fc = gen()
        adj = utils.adj_init(torch.from_numpy(Fc).cuda())["adj"]
        torch.manual_seed(5)
        b, nv, c, k = 3, V.shape[0], 48, 16
        sup = torch.randn(b, nv, c, device="cuda", requires_grad=True)
        bias = torch.randn(c, device="cuda", requires_grad=True)
        base = torch.randn(b, nv, 3, device="cuda", requires_grad=True)
        gpos = torch.randn(b, nv, 3, device="cuda")
        ref = ops.VertexHead.apply(base, layers.zero_n_aggregate(sup, adj, bias, k, act), 0.01)
        ref_g = torch.autograd.grad(ref, [sup, bias, base], gpos)
        got = layers.zero_n_aggregate_head(sup, adj, bias, k, act, base, 0.01)
        got_g = torch.autograd.grad(got, [sup, bias, base], gpos)
        assert torch.equal(got, ref)
        for a, b_ in zip(got_g, ref_g):
            assert torch.equal(a, b_)


def test_fused_adam_graph_replays_track_torch_adam():
    """The in-kernel step advance (arrival tree, state read through an atomic load + LDS broadcast) under HIP-graph replay:
    N replays of ONE captured FusedAdam launch against N steps of torch.optim.Adam on the same gradients -- a wrong bias
    correction on any replay (a workgroup reading the state after another one advanced it) shows up at the 1e-3 level."""
    from geometrics_amd import optim
    torch.manual_seed(11)
    shapes = [(963, 192), (192,), (192, 192), (5,), (1, 192, 192), (3000, 7)]
    ours = [torch.randn(*s, device="cuda").requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in ours]
    grads = [torch.randn(*s, device="cuda") for s in shapes]
    opt = optim.FusedAdam(ours, lr=1e-3)
    ref_opt = torch.optim.Adam(ref, lr=1e-3)
    for p, g in zip(ours, grads):
        p.grad = g
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt.step()                      # warm-up step 1 (eager)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.step()                      # captured, not executed
    n_replays = 9
    for _ in range(n_replays):
        graph.replay()
    torch.cuda.synchronize()
    assert opt.step_count == 1 + n_replays
    for _ in range(1 + n_replays):
        for p, g in zip(ref, grads):
            p.grad = g.clone()
        ref_opt.step()
    for a, b in zip(ours, ref):
        assert torch.allclose(a.detach(), b.detach(), rtol=2e-5, atol=2e-6)


def test_adam_inside_the_backward_pass_equals_the_separate_launch():
    """optim.FusedAdam.in_backward(): the step applied by the end-of-pass reduction launch (geom_dense_reduce_adam_f32)
    against the same iterations with the stand-alone optimiser launch: parameters, moments and step count bit for bit, over
    several iterations (bias corrections advance), eager and as a replayed HIP graph; and a pass that does not cover every
    parameter leaves the step to `step()`."""
    import torch.nn.functional as F
    from geometrics_amd import layers, meshgen, optim, utils
    V, Fc = meshgen.uv_sphere()
    adj = utils.adj_init(torch.from_numpy(Fc).cuda())["adj"]
    x = torch.randn(4, V.shape[0], 40, device="cuda")
    target = torch.randn(4, V.shape[0], 48, device="cuda")

    def make():
        torch.manual_seed(3)
        stack = torch.nn.ModuleList([layers.Batch_Image_ZERON_GCNGCN(40, 48), layers.Batch_Image_ZERON_GCNGCN(48, 48),
                                     layers.Batch_Image_ZERON_GCNGCN(48, 48)]).cuda()
        return stack, optim.FusedAdam(stack.parameters(), lr=1e-2)

    def iteration(stack, opt, fused):
        import contextlib
        opt.zero_grad()
        with layers.deferred_parameter_gradients(), (opt.in_backward() if fused else contextlib.nullcontext()):
            h = x
            for layer in stack:
                h = layer(h, adj, F.relu)
            ((h - target) ** 2).mean().backward()
        stepped = bool(getattr(opt, "_stepped_in_backward", False))
        opt.step()
        return stepped

    a, oa = make()
    b, ob = make()
    for _ in range(4):
        assert iteration(a, oa, True) is True          # the launch covered all six parameters: step() was a no-op
        assert iteration(b, ob, False) is False
    assert oa.step_count == ob.step_count == 4
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q) and torch.equal(p.grad, q.grad)
    for m1, m2 in zip(oa.exp_avg + oa.exp_avg_sq, ob.exp_avg + ob.exp_avg_sq):
        assert torch.equal(m1, m2)
    # an optimiser that ALSO owns a parameter this pass does not produce a gradient for: nothing is fused
    extra = torch.nn.Parameter(torch.zeros(5, device="cuda"))
    oc = optim.FusedAdam(list(a.parameters()) + [extra], lr=1e-2)
    od = optim.FusedAdam(list(b.parameters()) + [torch.nn.Parameter(torch.zeros(5, device="cuda"))], lr=1e-2)
    assert iteration(a, oc, True) is False and iteration(b, od, False) is False
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
    # replayed as a HIP graph
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        iteration(a, oa, True)
        iteration(b, ob, False)
    torch.cuda.current_stream().wait_stream(side)
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
        iteration(a, oa, True)
    with torch.cuda.graph(gb):
        iteration(b, ob, False)
    for _ in range(3):
        ga.replay()
        gb.replay()
    torch.cuda.synchronize()
    assert oa.step_count == ob.step_count
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
