"""The fused layer-boundary launches (csrc/zn_stack.hip) against the two separate operators they replace.

Reference semantics: layers.py:107-116 across two consecutive `Batch_Image_ZERON_GCNGCN` layers.  The aggregated operand
(forward: the activated input of the second layer + its sign words; backward: the gradient of the support) is produced by
the same thread mapping and arithmetic as `zn_aggregate_ell_kernel` -> compared BIT FOR BIT with that kernel; the products
(exact fp32 MFMA, another summation order than any other kernel) against float64 products of those operands."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

K, C = 64, 192


def _mesh(level):
    from geometrics_amd import layers, meshgen, utils
    V, Fc = meshgen.icosphere(level)
    faces = torch.from_numpy(np.ascontiguousarray(Fc)).cuda()
    info = utils.adj_init(faces)
    return V.shape[0], layers.adjacency_csr(info["adj"])


def _close(got, want, tol=5e-6):
    got, want = got.double().cpu(), want.double().cpu()
    scale = want.abs().max().item() + 1e-30
    assert (got - want).abs().max().item() <= tol * scale * 50
    assert (got - want).norm().item() <= tol * want.norm().item() + 1e-30


@pytest.mark.parametrize("level,b,act,n_out", [(4, 8, 1, 192), (4, 3, 1, 192), (2, 5, 1, 192), (2, 2, 2, 96), (3, 1, 0, 192),
                                               (4, 8, 2, 192), (4, 64, 1, 192)])
def test_forward_boundary(gpu, level, b, act, n_out):
    from geometrics_amd import fused, layers
    nv, csr = _mesh(level)
    assert fused.supported(csr, C, K, n_out)
    g = torch.Generator(device="cpu").manual_seed(level * 100 + b)
    s_prev = torch.randn(b, nv, C, generator=g).cuda()
    bias = (torch.randn(C, generator=g) * 0.3).cuda()
    w = (torch.randn(C, n_out, generator=g) * 0.1).cuda()
    # the separate operators
    want_x = torch.empty_like(s_prev)
    want_mask = layers.aggregate_forward(s_prev, bias, csr, K, act, want_x, want_mask=True)
    mask = torch.zeros(b * nv * 16, dtype=torch.int16, device="cuda") if act == 1 else None
    wt = torch.full((n_out, C), float("nan"), device="cuda")
    x, s = fused.layer_forward(s_prev, bias, csr, K, act, w, mask=mask, wt_out=wt)
    assert torch.equal(x, want_x)
    if act == 1:
        assert torch.equal(mask, want_mask)
    assert torch.equal(wt, w.t().contiguous())
    _close(s.view(-1, n_out), want_x.view(-1, C).double() @ w.double())
    # bit-reproducible
    x2, s2 = fused.layer_forward(s_prev, bias, csr, K, act, w)
    assert torch.equal(s, s2)


@pytest.mark.parametrize("level,b,act,head", [(4, 8, 1, False), (4, 8, 1, True), (4, 3, 1, False), (2, 5, 1, True), (2, 2, 2, False),
                                              (3, 1, 0, False), (3, 2, 0, True), (4, 64, 1, False)])
def test_backward_boundary(gpu, level, b, act, head):
    from geometrics_amd import fused, layers
    nv, csr = _mesh(level)
    g = torch.Generator(device="cpu").manual_seed(level * 100 + b + 7)
    out = torch.randn(b, nv, C, generator=g).cuda()          # the layer's activated output (only its signs / values matter)
    if act == 1:
        out = torch.relu(out)
    w = (torch.randn(C, C, generator=g) * 0.1).cuda()       # [cin, c]
    wt = w.t().contiguous()
    mask = None
    if act == 1:   # sign words in the aggregation kernel's layout, from the forward kernel itself
        s_raw = torch.randn(b, nv, C, generator=g).cuda()
        out = torch.empty_like(s_raw)
        mask = layers.aggregate_forward(s_raw, None, csr, K, 1, out, want_mask=True)
    scale = 0.01
    if head:
        gp = torch.randn(b, nv, 3, generator=g).cuda()
        grad_out = torch.zeros(b, nv, C, device="cuda")
        grad_out[..., :3] = scale * gp
    else:
        gp = None
        grad_out = torch.randn(b, nv, C, generator=g).cuda()
    want_g, want_bias = layers.aggregate_backward(grad_out, csr, K, act, out if act == 2 else None, mask, True,
                                                  bias=torch.zeros(C, device="cuda"))
    rows = fused.partial_rows(b, nv)
    partial = torch.full((rows, C), float("nan"), device="cuda")
    g_out, grad_in = fused.layer_backward(None if head else grad_out, out if act == 2 else None, mask, csr, K, act, wt,
                                          colsum_partial=partial, grad_pos=gp, head_scale=scale, shape=(b, nv, C))
    assert torch.equal(g_out, want_g)
    _close(grad_in.view(-1, C), want_g.view(-1, C).double() @ w.double().t())
    # the bias gradient's partial sums: another (fixed) order than the aggregation kernel's
    got_bias = partial.double().sum(0)
    gprime = grad_out.double()
    if act == 1:
        gprime = gprime * (out > 0)
    elif act == 2:
        gprime = torch.where(out > 0, gprime, gprime * (out.double() + 1))
    want = gprime.view(-1, C).sum(0)
    assert (got_bias.cpu() - want.cpu()).abs().max().item() <= 1e-5 * (gprime.abs().view(-1, C).sum(0).max().item() + 1e-30)
    _ = want_bias


def test_unsupported_shapes_are_refused(gpu):
    from geometrics_amd import _lib
    x = torch.zeros(64, device="cuda")
    p = x.data_ptr()
    L = _lib.lib()
    assert L.geom_zn_layer_fwd_f32(1, 4, 96, 32, 8, p, p, p, None, 0, p, 96, p, None, p, None, None) == _lib.EUNSUPPORTED
    assert L.geom_zn_layer_fwd_f32(1, 4, 192, 64, 16, p, p, p, None, 0, p, 192, p, None, p, None, None) == _lib.EUNSUPPORTED
    assert L.geom_zn_layer_fwd_f32(1, 4, 192, 64, 8, p, p, p, None, 0, p, 100, p, None, p, None, None) == _lib.EUNSUPPORTED


# ---- the stack operator (layers.zero_n_stack_positions): boundaries as single launches, against the layer-by-layer route ----
def _stack_run(force, b, level, act_fn, seed=5):
    import torch.nn.functional as F  # noqa: F401
    from geometrics_amd import fused, layers
    nv, csr = _mesh(level)
    torch.manual_seed(seed)
    stack = torch.nn.ModuleList([layers.Batch_Image_ZERON_GCNGCN(i, o) for i, o in ((99, C), (C, C), (C, C))]).cuda()
    for layer in stack:
        layer.bias.data.uniform_(-0.05, 0.05)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    x = torch.randn(b, nv, 99, generator=g).cuda().requires_grad_(True)
    base = torch.randn(b, nv, 3, generator=g).cuda().requires_grad_(True)
    seed_grad = torch.randn(b, nv, 3, generator=g).cuda()
    fused.force = force
    try:
        if force is None:      # the layer-by-layer route: the reference's call sequence
            h = x
            for layer in stack[:-1]:
                h = layer(h, csr, act_fn)
            pos = stack[-1].forward_positions(h, csr, act_fn, base, 0.01)
        else:
            pos = layers.zero_n_stack_positions(x, csr, list(stack), act_fn, base, 0.01)
        pos.backward(seed_grad)
    finally:
        fused.force = None
    torch.cuda.synchronize()
    grads = [x.grad, base.grad] + [p.grad for p in stack.parameters()]
    return pos.detach(), [t.detach().clone() for t in grads]


@pytest.mark.parametrize("mode", ["fwd", "fwd+bwd"])
@pytest.mark.parametrize("act", ["relu", "elu", "none"])
@pytest.mark.parametrize("b,level", [(2, 3), (3, 2)])
def test_the_stack_operator_matches_the_layer_by_layer_route(gpu, b, level, act, mode):
    """Same parameters and inputs through (a) the layers one by one and (b) the stack operator with its boundary launches
    forced on (forward only / forward and backward): positions and every gradient agree within the fp32 summation order of
    the products.  (The smooth activation checks the whole chain; under ReLU a unit whose pre-activation is within rounding
    of 0 may switch, so hidden-layer gradients get the looser bound there -- the bound of tests/test_bench_step_gpu.py.)"""
    import torch.nn.functional as F
    act_fn = {"relu": F.relu, "elu": F.elu, "none": None}[act]
    want_pos, want = _stack_run(None, b, level, act_fn)
    got_pos, got = _stack_run({"fwd": True, "bwd": mode == "fwd+bwd"}, b, level, act_fn)
    tol = 2e-3 if act == "relu" else 2e-5
    assert (got_pos - want_pos).abs().max().item() <= 2e-6 * want_pos.abs().max().item()
    for gt, wt_ in zip(got, want):
        assert gt.shape == wt_.shape
        scale = wt_.abs().max().item() + 1e-30
        assert (gt - wt_).abs().max().item() <= tol * scale, (act, mode, gt.shape, (gt - wt_).abs().max().item() / scale)


def test_the_stack_operator_uses_the_boundary_launch(gpu):
    """The route is the fused one when forced (the link of the last boundary is there and carries the transposed weight)."""
    import torch.nn.functional as F
    from geometrics_amd import fused, layers
    nv, csr = _mesh(3)
    stack = torch.nn.ModuleList([layers.Batch_Image_ZERON_GCNGCN(i, o) for i, o in ((48, C), (C, C), (C, C))]).cuda()
    x = torch.randn(2, nv, 48, device="cuda")
    fused.force = {"fwd": True, "bwd": True}
    try:
        s, link = layers._stack_supports(x, csr, list(stack), {"activation": F.relu})
        assert link is not None and torch.equal(link.wt, stack[-1].weight1[0].t())
        fused.force = {"fwd": False, "bwd": False}
        s2, link2 = layers._stack_supports(x, csr, list(stack), {"activation": F.relu})
        assert link2 is None
    finally:
        fused.force = None
    assert (s - s2).abs().max().item() <= 2e-5 * s2.abs().max().item()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("act", ["elu", "none"])     # none: the head's fused backward (aggregation backward + input gradient) too
def test_the_plan_switches_the_boundary_launches_on_by_itself_at_a_large_shard(gpu, act):
    """32 meshes of 2562 vertices = 81 984 rows: above fused.plan's threshold, so the stack operator takes the boundary
    launches in BOTH directions without being forced (what bench.py's 64-mesh whole-batch figure runs); positions and every
    gradient against the layer-by-layer route with the plan forced off."""
    import torch.nn.functional as F
    from geometrics_amd import fused
    assert fused.plan(32 * 2562) == {"fwd": True, "bwd": True}
    b, level = 32, 4
    act_fn = F.elu if act == "elu" else None
    want_pos, want = _stack_run({"fwd": False, "bwd": False}, b, level, act_fn)
    nv, csr = _mesh(level)
    from geometrics_amd import layers
    torch.manual_seed(5)
    stack = torch.nn.ModuleList([layers.Batch_Image_ZERON_GCNGCN(i, o) for i, o in ((99, C), (C, C), (C, C))]).cuda()
    s, link = layers._stack_supports(torch.randn(b, nv, 99, device="cuda"), csr, list(stack), {"activation": F.relu})
    assert link is not None and link.wt is not None                         # the un-forced plan took the boundary launch
    del s, link, stack
    got_pos, got = _stack_auto(b, level, act_fn)
    assert (got_pos - want_pos).abs().max().item() <= 2e-6 * want_pos.abs().max().item()
    for gt, wt_ in zip(got, want):
        scale = wt_.abs().max().item() + 1e-30
        assert (gt - wt_).abs().max().item() <= 2e-5 * scale, (gt.shape, (gt - wt_).abs().max().item() / scale)


def _stack_auto(b, level, act_fn, seed=5):
    """_stack_run through layers.zero_n_stack_positions with fused.force left alone (the plan decides)."""
    from geometrics_amd import fused, layers
    nv, csr = _mesh(level)
    torch.manual_seed(seed)
    stack = torch.nn.ModuleList([layers.Batch_Image_ZERON_GCNGCN(i, o) for i, o in ((99, C), (C, C), (C, C))]).cuda()
    for layer in stack:
        layer.bias.data.uniform_(-0.05, 0.05)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    x = torch.randn(b, nv, 99, generator=g).cuda().requires_grad_(True)
    base = torch.randn(b, nv, 3, generator=g).cuda().requires_grad_(True)
    seed_grad = torch.randn(b, nv, 3, generator=g).cuda()
    assert fused.force is None
    pos = layers.zero_n_stack_positions(x, csr, list(stack), act_fn, base, 0.01)
    pos.backward(seed_grad)
    torch.cuda.synchronize()
    grads = [x.grad, base.grad] + [p.grad for p in stack.parameters()]
    return pos.detach(), [t.detach().clone() for t in grads]
