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
