"""-m gpu: the hidden layers of the mesh deformation block as one launch per layer and direction (csrc/deform_block.hip,
geometrics_amd/deform.py; reference models.py:237-297 on layers.py:107-116).

* a single forward / backward launch against the separate operators it replaces (aggregation bits identical; the per-vertex
  BatchNorm and the products within fp32 summation order of a float64 evaluation);
* the whole block (13 fused hidden layers) against a FLOAT64 restatement of the reference block on the host: features,
  coordinates, running statistics, every parameter gradient and both input gradients;
* the same against the block on the separate operators (`deform.enabled = False`), batch 16 on the 482-vertex template with its
  two 33-entry poles (table + CSR tail) and batch 5 on a pole-free icosphere (rows beyond the batch are zero rows of the tile);
* the step inside a HIP graph; shapes the launches do not serve fall back to the separate operators;
* the thirteen launches of a direction as ONE (a vertex's workgroup waits for its neighbours' rows inside the launch): bit for
  bit the layer-by-layer launches; only the stream that owns a device's chain launches issues them; a mesh whose workgroups are
  not all resident at once takes the layer-by-layer launches."""
import numpy as np
import pytest
import torch

from geometrics_amd import deform, layers, meshgen, models, utils

pytestmark = pytest.mark.gpu


def _mesh(name, gpu):
    V, Fc = meshgen.uv_sphere() if name == "uv_sphere_482" else meshgen.icosphere(2)
    adj = utils.adj_init(torch.from_numpy(Fc).to(gpu))["adj"]
    return V.shape[0], adj, layers.adjacency_csr(adj)


def _maxrel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _bn64(z, gamma, beta, eps):
    """nn.BatchNorm1d(verts) on [B,V,C] in training mode, float64: one statistic per vertex over (B, C)."""
    mean = z.mean(dim=(0, 2), keepdim=True)
    var = ((z - mean) ** 2).mean(dim=(0, 2), keepdim=True)
    return (z - mean) / torch.sqrt(var + eps) * gamma.view(1, -1, 1) + beta.view(1, -1, 1), mean.flatten(), var.flatten()


@pytest.mark.parametrize("mesh,batch", [("uv_sphere_482", 16), ("icosphere_162", 5)])
def test_one_forward_launch_against_the_separate_operators(gpu, mesh, batch):
    nv, adj, csr = _mesh(mesh, gpu)
    torch.manual_seed(3)
    c = 192
    s = torch.randn(batch, nv, c, device=gpu)
    bias = torch.randn(c, device=gpu) * 0.1
    gamma, beta = torch.rand(nv, device=gpu) + 0.5, torch.randn(nv, device=gpu) * 0.2
    res = torch.randn(batch, nv, c, device=gpu)
    w = torch.randn(c, c, device=gpu) / 14
    rm, rv = torch.zeros(nv, device=gpu), torch.ones(nv, device=gpu)
    z, x, s_next = (torch.empty_like(s) for _ in range(3))
    mean, invstd = torch.empty(nv, device=gpu), torch.empty(nv, device=gpu)
    packed, _ = deform.pack_weights([w])
    deform.layer_forward(s, bias, csr, gamma, beta, rm, rv, True, 0.1, 1e-5, True, res, 0.5, z, x, mean, invstd, w_next=packed[0], s_out=s_next)
    # aggregation: the bits of the stand-alone operator
    z_ref = layers.zero_n_aggregate(s, adj, bias, 64, None)
    assert torch.equal(z, z_ref)
    # BatchNorm + ReLU + residual average, and the product, against float64
    y64, m64, v64 = _bn64(z.double().cpu(), gamma.double().cpu(), beta.double().cpu(), 1e-5)
    x64 = (res.double().cpu() + torch.relu(y64)) * 0.5
    assert _maxrel(x, x64) <= 2e-6
    assert _maxrel(mean, m64) <= 1e-5 and _maxrel(invstd, 1.0 / torch.sqrt(v64 + 1e-5)) <= 1e-5
    n = batch * c
    assert _maxrel(rm, 0.1 * m64) <= 1e-5 and _maxrel(rv, 0.9 + 0.1 * v64 * n / (n - 1)) <= 1e-5
    s64 = x.double().cpu() @ w.double().cpu()
    bound = (c + 8) * 2.0 ** -24 * (x.double().cpu().abs() @ w.double().cpu().abs())
    assert bool(((s_next.double().cpu() - s64).abs() <= bound + 1e-30).all())
    # no product: the last hidden layer
    x2 = torch.empty_like(s)
    deform.layer_forward(s, bias, csr, gamma, beta, rm.clone(), rv.clone(), True, 0.1, 1e-5, True, None, 0.5, None, x2, mean, invstd)
    assert _maxrel(x2, torch.relu(y64)) <= 2e-6


@pytest.mark.parametrize("mesh,batch", [("uv_sphere_482", 16), ("icosphere_162", 5)])
def test_one_backward_launch_against_the_separate_operators(gpu, mesh, batch):
    nv, adj, csr = _mesh(mesh, gpu)
    torch.manual_seed(4)
    c = 192
    shape = (batch, nv, c)
    dz_up, z, g2 = (torch.randn(*shape, device=gpu) for _ in range(3))
    wt = torch.randn(c, c, device=gpu) / 14
    gamma, beta = torch.rand(nv, device=gpu) + 0.5, torch.randn(nv, device=gpu) * 0.2
    z64 = z.double().cpu()
    mean64 = z64.mean(dim=(0, 2))
    var64 = ((z64 - mean64.view(1, -1, 1)) ** 2).mean(dim=(0, 2))
    mean, invstd = mean64.float().to(gpu), (1.0 / torch.sqrt(var64 + 1e-5)).float().to(gpu)
    ds, dz, gres = (torch.empty(*shape, device=gpu) for _ in range(3))
    gbw, gbb = torch.empty(nv, device=gpu), torch.empty(nv, device=gpu)
    colsum = torch.empty(nv, c, device=gpu)
    _, packed_t = deform.pack_weights([wt.t().contiguous()])      # the launch reads the layer's weight transposed, packed
    deform.layer_backward(shape, csr, z, gamma, beta, mean, invstd, True, True, 0.5, dz, gbw, gbb, dz_up=dz_up, ds_up=ds, wt_up=packed_t[0],
                          g2=g2, grad_res=gres, colsum=colsum)
    ds_ref, _ = layers.aggregate_backward(dz_up, csr, 64, layers._ACT_NONE, None, None, False)
    assert torch.equal(ds, ds_ref)
    # float64 from here: dX = dS . W^T (wt IS the transposed weight), + g2, residual scale, ReLU mask, BatchNorm backward
    gx = (ds.double().cpu() @ wt.double().cpu() + g2.double().cpu()) * 0.5
    assert _maxrel(gres, gx) <= 2e-6
    m, i = mean.double().cpu().view(1, -1, 1), invstd.double().cpu().view(1, -1, 1)
    xh = (z64 - m) * i
    ga, be = gamma.double().cpu().view(1, -1, 1), beta.double().cpu().view(1, -1, 1)
    # (an element whose normalised value sits within rounding of the ReLU kink may fall on either side in float64: the mask is
    # formed in fp32, operation by operation as the kernel forms it -- and as the forward launch formed its ReLU)
    on = ((((z - mean.view(1, -1, 1)) * invstd.view(1, -1, 1)) * gamma.view(1, -1, 1) + beta.view(1, -1, 1)) > 0).cpu()
    gy = torch.where(on, gx, torch.zeros_like(gx))
    sg, sgx = gy.sum(dim=(0, 2)), (gy * xh).sum(dim=(0, 2))
    n = batch * c
    dz64 = ga * i * (gy - sg.view(1, -1, 1) / n - xh * sgx.view(1, -1, 1) / n)
    assert _maxrel(gbb, sg) <= 1e-4 and _maxrel(gbw, sgx) <= 1e-4
    assert _maxrel(dz, dz64) <= 1e-4
    assert _maxrel(colsum, dz64.sum(dim=0)) <= 1e-4
    # no product: the gradient of the output is read from memory (+ the second one)
    g = torch.randn(*shape, device=gpu)
    dz2 = torch.empty(*shape, device=gpu)
    deform.layer_backward(shape, csr, z, gamma, beta, mean, invstd, True, False, 0.5, dz2, gbw, gbb, g=g, g2=g2)
    gy = torch.where(on, (g + g2).double().cpu(), torch.zeros_like(gx))
    sg, sgx = gy.sum(dim=(0, 2)), (gy * xh).sum(dim=(0, 2))
    assert _maxrel(dz2, ga * i * (gy - sg.view(1, -1, 1) / n - xh * sgx.view(1, -1, 1) / n)) <= 1e-4


def _block64(block, feats, pooled, adj, relu=True):
    """models.py:237-297 restated in float64 on the host (dense adjacency, torch ops): returns (features, coords, parameters)."""
    p = {k: v.detach().double().cpu().requires_grad_(v.requires_grad) for k, v in block.named_parameters()}
    adj = adj.double().cpu()

    def gc(i, x):
        sup = x @ p["gc%d.weight1" % i][0]
        k = sup.shape[-1] // 3
        return torch.cat((adj @ sup[..., :k], sup[..., k:]), dim=-1) + p["gc%d.bias" % i]

    def layer(i, x):
        y, _, _ = _bn64(gc(i, x), p["bn%d.weight" % i], p["bn%d.bias" % i], 1e-5)
        return torch.relu(y) if relu else y
    f = torch.cat((feats, pooled), dim=-1)
    x = layer(1, f)
    x = layer(2, x)
    f = (f[..., :block.hidden] + x) / 2
    for i in (3, 5, 7, 9, 11):
        x = layer(i, f)
        x = layer(i + 1, x)
        f = (f + x) / 2
    x = layer(13, f)
    f = (f + x) / 2
    return f, gc(15, f), p


def _l2rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm()) / max(float(b.norm()), 1e-30)


@pytest.mark.parametrize("mesh,batch", [("uv_sphere_482", 16), ("icosphere_162", 5)])
@pytest.mark.parametrize("relu", [False, True])
def test_the_fused_block_against_float64_and_against_the_separate_operators(gpu, mesh, batch, relu):
    """relu=False (deform.relu, tests only): the chain without its ReLUs is smooth, and EVERY gradient -- 13 fused backward
    launches deep -- is held to a max-norm bound against float64.  relu=True: the block as the reference runs it; a
    pre-activation within rounding of zero falls on either side in any two fp32 evaluations (expected: about one element in
    the block's 19 M per pass) and switches one unit's term, which moves single entries of a gradient by 1e-3 .. 1e-2 of its
    scale whatever the implementation -- so there the gradients are held in the L2 norm (a wrong mask, a dropped residual
    or a transposed tile is an O(1) error in it; one switched unit is ~1e-3), the forward in max-norm as before."""
    import copy
    nv, adj, csr = _mesh(mesh, gpu)
    torch.manual_seed(5)
    block = models.BatchMeshDeformationBlock(3 + 200, nv).to(gpu).train()
    with torch.no_grad():
        for i in range(1, 14):
            getattr(block, "bn%d" % i).weight.uniform_(0.5, 1.5)
            getattr(block, "bn%d" % i).bias.uniform_(-0.3, 0.3)
    twin = copy.deepcopy(block)
    feats = torch.randn(batch, nv, 3, device=gpu)
    pooled = torch.randn(batch, nv, 200, device=gpu)
    g_f, g_c = torch.randn(batch, nv, 192, device=gpu), torch.randn(batch, nv, 3, device=gpu)

    def run(blk, fused):
        deform.enabled, deform.relu = fused, relu
        try:
            f, p = feats.clone().requires_grad_(True), pooled.clone().requires_grad_(True)
            assert deform.serves(blk, f, p, csr) == fused
            out_f, coords = blk(f, p, adj)
            ((out_f * g_f).sum() + (coords * g_c).sum()).backward()
        finally:
            deform.enabled, deform.relu = True, True
        return out_f, coords, f.grad, p.grad
    out_f, coords, gf, gp = run(block, True)
    f64, p64 = feats.double().cpu().requires_grad_(True), pooled.double().cpu().requires_grad_(True)
    e_f, e_c, params64 = _block64(block, f64, p64, adj, relu)
    ((e_f * g_f.double().cpu()).sum() + (e_c * g_c.double().cpu()).sum()).backward()
    # forward: 2e-5 of scale against float64 (the bar of the reference-fixture test of the block)
    assert _maxrel(out_f, e_f) <= 2e-5 and _maxrel(coords, e_c) <= 2e-5
    named = dict(block.named_parameters())
    pairs = [(n, p.grad, params64[n].grad) for n, p in named.items() if not n.startswith("bn14")]
    pairs += [("features", gf, f64.grad), ("pooled", gp, p64.grad)]
    assert all(named[n].grad is None for n in named if n.startswith("bn14"))
    if not relu:
        for name, got, want in pairs:
            assert _maxrel(got, want) <= 5e-5, "%s: %.2e of scale from float64" % (name, _maxrel(got, want))
        return
    ref_f, ref_c, rgf, rgp = run(twin, False)          # the separate operators on the same parameters
    assert _maxrel(out_f, ref_f) <= 2e-5 and _maxrel(coords, ref_c) <= 2e-5
    for name, got, want in pairs:
        assert _l2rel(got, want) <= 5e-3, "%s: %.2e from float64 in the L2 norm" % (name, _l2rel(got, want))
    for i in range(1, 14):
        a, b = getattr(block, "bn%d" % i), getattr(twin, "bn%d" % i)
        assert _maxrel(a.running_mean, b.running_mean) <= 1e-4 and _maxrel(a.running_var, b.running_var) <= 1e-4
    sd = block.state_dict()
    assert int(sd["bn1.num_batches_tracked"]) == 1 and int(sd["bn14.num_batches_tracked"]) == 0


def test_the_fused_block_replays_inside_a_hip_graph(gpu):
    nv, adj, csr = _mesh("uv_sphere_482", gpu)
    torch.manual_seed(6)
    block = models.BatchMeshDeformationBlock(3 + 197, nv).to(gpu).train()
    feats = torch.randn(16, nv, 3, device=gpu, requires_grad=True)
    pooled = torch.randn(16, nv, 197, device=gpu, requires_grad=True)
    params = list(block.parameters())

    def step():
        for p in params:
            p.grad = None
        feats.grad = pooled.grad = None
        f, c = block(feats, pooled, adj)
        (f.sum() + c.sum()).backward()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager = [p.grad.clone() for p in params if p.grad is not None] + [feats.grad.clone()]
    for i in range(1, 14):      # the running statistics move with every step: rewind so that the replay starts where the eager step did
        getattr(block, "bn%d" % i).running_mean.zero_(), getattr(block, "bn%d" % i).running_var.fill_(1.0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        step()
    g.replay()
    torch.cuda.synchronize()
    replayed = [p.grad for p in params if p.grad is not None] + [feats.grad]
    assert len(eager) == len(replayed)
    for a, b in zip(eager, replayed):
        assert torch.equal(a, b)          # same launches, fixed reduction orders: bit-reproducible


def test_shapes_the_launches_do_not_serve_take_the_separate_operators(gpu):
    nv, adj, csr = _mesh("icosphere_162", gpu)
    block = models.BatchMeshDeformationBlock(200, nv).to(gpu).train()
    narrow = models.BatchMeshDeformationBlock(200, nv, hidden=48).to(gpu).train()
    f, p = torch.randn(4, nv, 3, device=gpu), torch.randn(4, nv, 197, device=gpu)
    assert deform.serves(block, f, p, csr)
    assert not deform.serves(narrow, f, p, csr)                                        # another width
    assert not deform.serves(block, torch.randn(17, nv, 3, device=gpu), torch.randn(17, nv, 197, device=gpu), csr)   # more than one tile of meshes
    block.eval()
    assert not deform.serves(block, f, p, csr)                                         # running statistics: the library route
    with torch.no_grad():
        out_f, coords = block(f, p, adj)
    assert out_f.shape == (4, nv, 192) and coords.shape == (4, nv, 3)
    block.train()
    big = torch.randn(17, nv, 3, device=gpu, requires_grad=True)
    out_f, coords = block(big, torch.randn(17, nv, 197, device=gpu), adj)
    (out_f.sum() + coords.sum()).backward()
    assert torch.isfinite(big.grad).all()


def test_packed_weight_slices_hold_the_matrix_and_its_transpose(gpu):
    """geom_deform_pack_weights_f32: [wave][e / 4][lane][e % 4] with element e = (4 jp + c) * 3 + u of lane (x, g) =
    W[48 g + 4 jp + c][48 wave + 3 x + u] (and the same of W^T): a permutation of the matrix, checked index by index."""
    torch.manual_seed(8)
    ws = [torch.randn(1, 192, 192, device=gpu), torch.randn(192, 192, device=gpu)]
    fwd, bwd = deform.pack_weights(ws)
    r = np.arange(192 * 192)
    wave, rem = r // 9216, r % 9216
    i, lane, t = rem >> 8, (rem & 255) >> 2, rem & 3
    e = 4 * i + t
    jp, c, u = e // 12, (e // 3) & 3, e % 3
    g, x = lane >> 4, lane & 15
    k, col = 48 * g + 4 * jp + c, 48 * wave + 3 * x + u
    for n, w in enumerate(ws):
        m = w.reshape(192, 192).cpu().numpy()
        assert np.array_equal(fwd[n].cpu().numpy(), m[k, col]) and np.array_equal(bwd[n].cpu().numpy(), m[col, k])


def test_block_input_assembled_in_place_equals_the_concatenations(gpu):
    """The driver builds a block's input as cat(positions, cat(previous features, pooled)) (GEOMetrics.py:123-129, models.py:241).
    With `utils.batched_pooling(..., headroom=3 + 192)` + `utils.concat_features` the pooled features are written straight into
    the wide buffer, the fronts are copied into its free columns, and in the backward pass the pooling and the previous layer
    read their column slices of the ONE input gradient in place (row pitch 1155).  Against the same computation with plain
    torch.cat: features, coordinates and every gradient (maps, positions, previous features, parameters) bit for bit."""
    import copy
    from geometrics_amd import ops
    nv, adj, csr = _mesh("uv_sphere_482", gpu)
    torch.manual_seed(14)
    b = 16
    block = models.BatchMeshDeformationBlock(3 + 192 + 96, nv).to(gpu).train()
    twin = copy.deepcopy(block)
    maps = [torch.randn(b, c, d, d, device=gpu) for c, d in ((64, 14), (32, 7))]
    pos = (torch.from_numpy(meshgen.uv_sphere()[0]).to(gpu).unsqueeze(0) + 0.02 * torch.randn(b, nv, 3, device=gpu))
    prev = torch.randn(b, nv, 192, device=gpu)
    img = torch.tensor([[30.0 + 7 * i, 20.0, 1.2] for i in range(b)], device=gpu)
    g_f, g_c = torch.randn(b, nv, 192, device=gpu), torch.randn(b, nv, 3, device=gpu)

    def run(blk, in_place, fronts=False):
        ms = [m.clone().requires_grad_(True) for m in maps]
        p, f_prev = pos.clone().requires_grad_(True), prev.clone().requires_grad_(True)
        pooled = utils.batched_pooling(ms, p, img, headroom=3 + 192 if in_place else 0, fronts=(p, f_prev) if fronts else None)
        if fronts:      # the pooling launch placed both: the concatenations below copy nothing
            buf = ops._headroom[pooled.untyped_storage().data_ptr()][0]()
            assert ops._already_placed(buf, 0, p) and ops._already_placed(buf, 3, f_prev)
            assert torch.equal(buf[..., :3], p.detach()) and torch.equal(buf[..., 3:195], f_prev.detach())
        assert (ops.headroom_of(pooled) == 195) == in_place
        f = utils.concat_features(f_prev, pooled) if in_place else torch.cat((f_prev, pooled), dim=-1)
        assert f.shape == (b, nv, 192 + 96) and (ops.headroom_of(f) == 3) == in_place
        out_f, coords = blk(p, f, adj)
        ((out_f * g_f).sum() + (coords * g_c).sum()).backward()
        return [out_f.detach(), coords.detach(), p.grad, f_prev.grad] + [m.grad for m in ms] + [q.grad for q in blk.parameters() if q.grad is not None]
    want = run(twin, False)
    for fronts in (False, True):     # (True: positions and previous features copied into place by the pooling launch itself)
        got = run(copy.deepcopy(twin), True, fronts)
        assert len(got) == len(want) and len(got) >= 55
        for i, (a, w) in enumerate(zip(got, want)):
            if i in (4, 5):     # the maps' gradients: per-texel lists are filled in ticket order (same terms, fp32 order may differ: DESIGN 5)
                assert float((a - w).abs().max()) <= 1e-5 * float(w.abs().max())
            else:
                assert torch.equal(a, w), i
    with pytest.raises(RuntimeError):
        utils.batched_pooling([m.clone() for m in maps], pos, img, headroom=10, fronts=(pos,))      # widths must add up
    # pooling alone: the pitched forward and a pitched upstream gradient give the plain call's bits
    ms = [m.clone().requires_grad_(True) for m in maps]
    p = pos.clone().requires_grad_(True)
    wide = utils.batched_pooling(ms, p, img, headroom=5)
    plain = utils.batched_pooling([m.detach() for m in ms], p.detach(), img)
    assert torch.equal(wide, plain) and not wide.is_contiguous()
    g_wide = torch.randn(b, nv, 5 + 96, device=gpu)
    wide.backward(g_wide[..., 5:])
    ms2 = [m.clone().requires_grad_(True) for m in maps]
    p2 = pos.clone().requires_grad_(True)
    utils.batched_pooling(ms2, p2, img).backward(g_wide[..., 5:].contiguous())
    assert torch.equal(p.grad, p2.grad)
    assert all(float((x.grad - y.grad).abs().max()) <= 1e-5 * float(y.grad.abs().max()) for x, y in zip(ms, ms2))


def _block_step(block, feats, pooled, adj):
    for p in block.parameters():
        p.grad = None
    feats.grad = pooled.grad = None
    for i in range(1, 14):
        getattr(block, "bn%d" % i).running_mean.zero_(), getattr(block, "bn%d" % i).running_var.fill_(1.0)
    f, c = block(feats, pooled, adj)
    (f * torch.linspace(-1, 1, f.shape[-1], device=f.device)).sum().add((c * c).sum()).backward()
    out = [f.detach().clone(), c.detach().clone(), feats.grad.clone(), pooled.grad.clone()]
    out += [p.grad.clone() for p in block.parameters() if p.grad is not None]
    out += [getattr(block, "bn%d" % i).running_var.clone() for i in range(1, 14)]
    return out


@pytest.mark.parametrize("mesh,batch", [("uv_sphere_482", 16), ("icosphere_162", 5)])
def test_one_launch_per_direction_equals_the_launch_per_layer(gpu, mesh, batch):
    """deform.chain: forward and backward of the block's thirteen hidden layers as ONE launch each against the thirteen launches
    (same bodies, the rows between layers travel through memory with agent-scope accesses behind per-vertex counters): every
    output, gradient and running statistic bit for bit, five times over (a race would not repeat)."""
    nv, adj, csr = _mesh(mesh, gpu)
    from geometrics_amd import _lib
    assert _lib.lib().geom_deform_chain_fits(nv) == 1
    torch.manual_seed(16)
    block = models.BatchMeshDeformationBlock(3 + 197, nv).to(gpu).train()
    feats = torch.randn(batch, nv, 3, device=gpu, requires_grad=True)
    pooled = torch.randn(batch, nv, 197, device=gpu, requires_grad=True)
    try:
        deform.chain = False
        ref = _block_step(block, feats, pooled, adj)
        deform.chain = True
        for _ in range(5):
            got = _block_step(block, feats, pooled, adj)
            assert len(got) == len(ref)
            for a, b in zip(got, ref):
                assert torch.equal(a, b)
    finally:
        deform.chain = True


def test_chain_launches_belong_to_one_stream_and_need_a_resident_grid(gpu):
    nv, adj, csr = _mesh("uv_sphere_482", gpu)
    from geometrics_amd import _lib
    lib = _lib.lib()
    assert lib.geom_deform_chain_fits(482) == 1 and lib.geom_deform_chain_fits(100000) == 0 and lib.geom_deform_chain_fits(0) == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(a):
        assert deform.chain_fits(nv, gpu)                    # (everything idle: the launches move to stream a)
        busy = torch.empty(1 << 28, device=gpu)
        for _ in range(20):
            busy.fill_(1.0)                                   # stream a has work in flight ...
        with torch.cuda.stream(b):
            assert not deform.chain_fits(nv, gpu)            # ... so stream b issues its layers one by one
        assert deform.chain_fits(nv, gpu)
    torch.cuda.synchronize()
    with torch.cuda.stream(b):
        assert deform.chain_fits(nv, gpu)                    # the owner is idle: the launches move
    # a mesh with more vertices than workgroups fit the chip: the block runs on the layer-by-layer launches, same results as ever
    V, Fc = meshgen.icosphere(3)                              # 642 vertices
    assert lib.geom_deform_chain_fits(V.shape[0]) == 0
    adj642 = utils.adj_init(torch.from_numpy(Fc).to(gpu))["adj"]
    torch.manual_seed(17)
    block = models.BatchMeshDeformationBlock(3 + 197, V.shape[0]).to(gpu).train()
    feats = torch.randn(4, V.shape[0], 3, device=gpu, requires_grad=True)
    pooled = torch.randn(4, V.shape[0], 197, device=gpu, requires_grad=True)
    got = _block_step(block, feats, pooled, adj642)
    try:
        deform.enabled = False
        ref = _block_step(block, feats, pooled, adj642)
    finally:
        deform.enabled = True
    assert _maxrel(got[0], ref[0]) < 5e-5 and _maxrel(got[1], ref[1]) < 5e-5


def test_a_second_backward_over_a_retained_graph_takes_the_launches_per_layer(gpu):
    """The chain launch's counters are good for one pass (they count up from zero): backward(retain_graph=True) twice gives
    the gradients of one pass twice -- the second pass on the launches per layer, bit for bit the same."""
    nv, adj, csr = _mesh("uv_sphere_482", gpu)
    torch.manual_seed(18)
    block = models.BatchMeshDeformationBlock(3 + 197, nv).to(gpu).train()
    feats = torch.randn(16, nv, 3, device=gpu, requires_grad=True)
    pooled = torch.randn(16, nv, 197, device=gpu, requires_grad=True)
    f, c = block(feats, pooled, adj)
    loss = (f * f).sum() + c.sum()
    loss.backward(retain_graph=True)
    once = [p.grad.clone() for p in block.parameters() if p.grad is not None] + [pooled.grad.clone()]
    for p in block.parameters():
        p.grad = None
    pooled.grad = feats.grad = None
    loss.backward()
    again = [p.grad for p in block.parameters() if p.grad is not None] + [pooled.grad]
    for a, b in zip(once, again):
        assert torch.equal(a, b)


def test_chain_launches_under_a_busy_neighbour_stream(gpu):
    """The chain launches while another stream keeps the chip busy with ordinary kernels (their workgroups come and go, so the
    chain's are not all resident from the first cycle: late workgroups are waited for, never deadlocked on), forty steps:
    every step's outputs and gradients bit for bit those of the launches per layer."""
    nv, adj, csr = _mesh("uv_sphere_482", gpu)
    torch.manual_seed(19)
    block = models.BatchMeshDeformationBlock(3 + 197, nv).to(gpu).train()
    feats = torch.randn(16, nv, 3, device=gpu, requires_grad=True)
    pooled = torch.randn(16, nv, 197, device=gpu, requires_grad=True)
    try:
        deform.chain = False
        ref = _block_step(block, feats, pooled, adj)
    finally:
        deform.chain = True
    torch.cuda.synchronize()
    noise = torch.cuda.Stream()
    junk = torch.empty(1 << 26, device=gpu)
    a, b = torch.randn(2048, 2048, device=gpu), torch.randn(2048, 2048, device=gpu)
    for step in range(40):
        with torch.cuda.stream(noise):
            for _ in range(6):
                junk.fill_(float(step))
                torch.mm(a, b)
        got = _block_step(block, feats, pooled, adj)
        for x, y in zip(got, ref):
            assert torch.equal(x, y), "step %d" % step
    torch.cuda.synchronize()
