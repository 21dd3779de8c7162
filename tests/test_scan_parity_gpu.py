"""-m gpu parity of the two arg-min scans (through the C ABI) against the CPU oracle.

Bar: indices, region codes AND distances bit-exact (integer/index work; the distances are
produced by the same un-fused fp32 operation sequence on both sides)."""
import numpy as np
import pytest
import torch

from geometrics_amd import meshgen
from geometrics_amd._lib import FLAG_FIX_REGION6, FLAG_NN_FMA, FLAG_REF_TAIL_TRUNC, FLAG_TRI_BRUTE_FORCE
from geometrics_amd.chamfer_distance import ChamferDistance, chamfer_nn
from geometrics_amd.tri_distance import TriDistance, morton_order, tri_distance, tri_distance_indexed

pytestmark = pytest.mark.gpu


def _dev(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def _check_nn(oracle_mod, gpu, a, b, flags=0):
    d1, i1, d2, i2 = chamfer_nn(_dev(a, gpu), _dev(b, gpu), flags)
    e1, j1, e2, j2 = oracle_mod.chamfer_nn(a, b, flags)
    np.testing.assert_array_equal(i1.cpu().numpy(), j1)
    np.testing.assert_array_equal(i2.cpu().numpy(), j2)
    np.testing.assert_array_equal(d1.cpu().numpy().view(np.uint32), e1.view(np.uint32))
    np.testing.assert_array_equal(d2.cpu().numpy().view(np.uint32), e2.view(np.uint32))


@pytest.mark.parametrize("b,n,m", [(1, 500, 500), (2, 3000, 3000), (3, 1, 1), (2, 7, 2466), (1, 2466, 5),
                                   (2, 63, 65), (1, 1025, 1023), (4, 300, 4097)])
def test_nn_random(oracle_mod, gpu, b, n, m):
    rng = np.random.default_rng(b * 1000003 + n * 131 + m)
    a = rng.standard_normal((b, n, 3)).astype(np.float32)
    c = rng.standard_normal((b, m, 3)).astype(np.float32)
    _check_nn(oracle_mod, gpu, a, c)


def test_nn_ties_lowest_index(oracle_mod, gpu):
    rng = np.random.default_rng(5)
    a = rng.integers(-3, 4, (2, 777, 3)).astype(np.float32)       # integer grid: masses of exact ties
    c = rng.integers(-3, 4, (2, 1300, 3)).astype(np.float32)
    c[:, 600:900] = c[:, :300]                                      # duplicated targets
    _check_nn(oracle_mod, gpu, a, c)


def test_nn_config_shapes(oracle_mod, gpu):
    V, F = meshgen.icosphere(4)
    verts = meshgen.jittered_batch(V, 2)
    gt = meshgen.gt_cloud(2, 3000)
    _check_nn(oracle_mod, gpu, gt, verts)      # 3000 vs 2562 (2562 % 4 == 2)
    _check_nn(oracle_mod, gpu, gt, meshgen.gt_cloud(2, 3000, first=7, cube=True))


@pytest.mark.parametrize("m", [4, 5, 6, 7, 511, 512, 513, 516, 2466, 1, 3, 1027])
def test_nn_reference_tail_truncation(oracle_mod, gpu, m):
    rng = np.random.default_rng(m)
    a = rng.standard_normal((2, 130, 3)).astype(np.float32)
    c = rng.standard_normal((2, m, 3)).astype(np.float32)
    _check_nn(oracle_mod, gpu, a, c, FLAG_REF_TAIL_TRUNC)


def test_package_wide_reference_quirk_mode_at_the_validation_shape(oracle_mod, gpu):
    """geometrics_amd.set_reference_quirks(True) / GEOM_REF_QUIRKS=1: every call that passes no flags -- the module objects
    an unmodified driver holds (ChamferDistance(), TriDistance()), the fused losses, the compiled forward_cuda entry points --
    reproduces the SHIPPED CUDA kernels' tail truncation (chamfer_distance.cu:31-33, tri_distance.cu:129,134).  Shape: the
    reference's validation, 2466 points (GEOMetrics.py:227,349): 2466 = 4*512 + 418 and 418 % 4 = 2, so the last 2 targets /
    triangles of the final tile are never seen.  Checked against the tiled restatement of the kernel (oracle.nn_tiled /
    tri_scan with FLAG_REF_TAIL_TRUNC) bit for bit, and against the reference-emitted fixture nn_ragged_m2466 (the FULL scan
    of the reference's CPU nnsearch): identical wherever the winner is not one of the 2 dropped targets."""
    import geometrics_amd
    from helpers import golden
    from geometrics_amd import _shim, utils
    from geometrics_amd.chamfer_distance import ChamferDistance
    from geometrics_amd.tri_distance import TriDistance
    from oracle import ref_ops
    g = golden("nn_ragged_m2466")
    a, c = g["xyz1"], g["xyz2"]                                    # [1, 97, 3] queries, [1, 2466, 3] targets
    assert c.shape[1] == 2466 and not geometrics_amd.reference_quirks()
    full1, _ = ChamferDistance()(_dev(a, gpu), _dev(c, gpu))
    np.testing.assert_array_equal(full1.cpu().numpy(), g["idx1"])   # default mode = the reference's CPU scan
    rng = np.random.default_rng(2466)
    V, Fc = meshgen.icosphere(3)                                    # 642 vertices, 1280 faces: 1280 % 512 = 256 -> no tri tail;
    tris = rng.standard_normal((3, 2, 2466, 3)).astype(np.float32)  # ... so a 2466-triangle soup for the tri scan
    q = rng.standard_normal((2, 300, 3)).astype(np.float32)
    for k, t in enumerate((2464, 2465)):                            # the two triangles of the dropped tail hug a query each
        tris[:, :, t] = q[:, k][None] + 0.01 * rng.standard_normal((3, 2, 3)).astype(np.float32)
    verts = meshgen.jittered_batch(V, 2)
    gt = meshgen.gt_cloud(2, 2466)
    ch, u, v = meshgen.sampling_draws(verts, Fc, 2466)
    try:
        assert geometrics_amd.set_reference_quirks(True) and geometrics_amd.reference_quirks()
        i1, i2 = ChamferDistance()(_dev(a, gpu), _dev(c, gpu))
        e1, j1, e2, j2 = oracle_mod.chamfer_nn(a, c, FLAG_REF_TAIL_TRUNC)
        np.testing.assert_array_equal(i1.cpu().numpy(), j1)
        np.testing.assert_array_equal(i2.cpu().numpy(), j2)
        seen = g["idx1"] < 2464                                     # the reference's full-scan winner survives the truncation
        assert (~seen).sum() < seen.sum()
        np.testing.assert_array_equal(i1.cpu().numpy()[seen], g["idx1"][seen])
        assert (i1.cpu().numpy()[~seen] < 2464).all()
        # the compiled entry points (what the reference's own wrappers would import) follow the same switch
        d1, d2 = torch.empty(1, 97, device=gpu), torch.empty(1, 2466, device=gpu)
        k1, k2 = torch.empty(1, 97, dtype=torch.int32, device=gpu), torch.empty(1, 2466, dtype=torch.int32, device=gpu)
        _shim.cd.forward_cuda(_dev(a, gpu), _dev(c, gpu), d1, d2, k1, k2)
        np.testing.assert_array_equal(k1.cpu().numpy(), j1)
        np.testing.assert_array_equal(d1.cpu().numpy().view(np.uint32), e1.view(np.uint32))
        # TriDistance(): 2466 triangles
        dist, point, index = TriDistance()(_dev(q, gpu), *(_dev(t, gpu) for t in tris))
        ed, ep, ei = oracle_mod.tri_scan(q, tris[0], tris[1], tris[2], FLAG_REF_TAIL_TRUNC)
        fd, _, fi = oracle_mod.tri_scan(q, tris[0], tris[1], tris[2], 0)
        np.testing.assert_array_equal(index.cpu().numpy(), ei)
        np.testing.assert_array_equal(point.cpu().numpy(), ep)
        np.testing.assert_array_equal(dist.cpu().numpy().view(np.uint32), ed.view(np.uint32))
        assert (ei < 2464).all() and (fi >= 2464).any()             # the case does exercise the dropped triangles
        # the fused losses at the validation shape: loss and F1 of the truncated scans
        tv, tf, tg = torch.from_numpy(verts), torch.from_numpy(Fc), torch.from_numpy(gt)
        draws = tuple(torch.from_numpy(x) for x in (ch, u, v))
        info = utils.adj_init(tf.to(gpu))
        loss, f1 = utils.batch_point_to_point(tv.to(gpu), info, tg.to(gpu), num=2466, f1=True, draws=tuple(x.to(gpu) for x in draws))
        want, want_f1 = ref_ops.point_to_point(tv, tf, tg, *draws, f1=True, nn_flags=FLAG_REF_TAIL_TRUNC)
        full = ref_ops.point_to_point(tv, tf, tg, *draws)
        assert abs(float(loss) - float(want)) <= 1e-5 * abs(float(want)) and abs(f1 - want_f1) <= 1e-9
        assert float(want) != float(full)                           # the mode changes the number, as in the CUDA build
    finally:
        geometrics_amd.set_reference_quirks(False)
    again, _ = ChamferDistance()(_dev(a, gpu), _dev(c, gpu))
    np.testing.assert_array_equal(again.cpu().numpy(), g["idx1"])


def test_nn_nan_and_inf_follow_the_sequential_scan(oracle_mod, gpu):
    rng = np.random.default_rng(9)
    a = rng.standard_normal((1, 70, 3)).astype(np.float32)
    c = rng.standard_normal((1, 300, 3)).astype(np.float32)
    c[0, 0, 1] = np.nan          # NaN seed sticks
    a[0, 3, 0] = np.inf          # every distance inf/nan for this query
    c[0, 17] = np.nan            # NaN later: never chosen
    d1, i1, d2, i2 = chamfer_nn(_dev(a, gpu), _dev(c, gpu))
    e1, j1, e2, j2 = oracle_mod.chamfer_nn(a, c)
    np.testing.assert_array_equal(i1.cpu().numpy(), j1)
    np.testing.assert_array_equal(i2.cpu().numpy(), j2)
    np.testing.assert_array_equal(np.isnan(d1.cpu().numpy()), np.isnan(e1))


@pytest.mark.parametrize("b,n,m", [(1, 500, 500), (2, 3000, 3000), (3, 1, 1), (2, 7, 2466), (2, 63, 65), (4, 300, 4097)])
def test_nn_fma_arithmetic_random(oracle_mod, gpu, b, n, m):
    """GEOM_FLAG_NN_FMA: the HIP scan in the contracted arithmetic against oracle_nn_scan_fma (itself bit-identical to
    the reference nnsearch built with FMA contraction), indices and distances bit for bit."""
    rng = np.random.default_rng(b * 7 + n * 31 + m)
    a = rng.standard_normal((b, n, 3)).astype(np.float32)
    c = rng.standard_normal((b, m, 3)).astype(np.float32)
    _check_nn(oracle_mod, gpu, a, c, FLAG_NN_FMA)


def test_nn_fma_reference_vectors_and_mode_switch(oracle_mod, gpu):
    from helpers import golden, golden_names
    from geometrics_amd import chamfer_distance
    fx = golden("nnfma_outputs")
    for name in golden_names("nn_"):
        if name == "nn_config2_outputs":
            g = golden(name)
            a, c = meshgen.gt_cloud(2, 3000, first=int(g["gt_first"])), meshgen.gt_cloud(2, 3000, first=int(g["pred_first"]))
        else:
            g = golden(name)
            a, c = g["xyz1"], g["xyz2"]
        d1, i1, d2, i2 = chamfer_nn(_dev(a, gpu), _dev(c, gpu), FLAG_NN_FMA)
        np.testing.assert_array_equal(i1.cpu().numpy(), fx[name + ".idx1"])
        np.testing.assert_array_equal(i2.cpu().numpy(), fx[name + ".idx2"])
        np.testing.assert_array_equal(d1.cpu().numpy().view(np.uint32), fx[name + ".dist1"].view(np.uint32))
        np.testing.assert_array_equal(i1.cpu().numpy(), g["idx1"])       # and the un-fused reference's winners
    # NaN / inf rules are those of the sequential scan in this arithmetic too
    rng = np.random.default_rng(9)
    a = rng.standard_normal((1, 70, 3)).astype(np.float32)
    c = rng.standard_normal((1, 300, 3)).astype(np.float32)
    c[0, 0, 1], a[0, 3, 0], c[0, 17] = np.nan, np.inf, np.nan
    d1, i1, d2, i2 = chamfer_nn(_dev(a, gpu), _dev(c, gpu), FLAG_NN_FMA)
    e1, j1, e2, j2 = oracle_mod.chamfer_nn(a, c, FLAG_NN_FMA)
    np.testing.assert_array_equal(i1.cpu().numpy(), j1)
    np.testing.assert_array_equal(i2.cpu().numpy(), j2)
    # the package-wide switch routes flag-less calls (ChamferDistance, the fused loss) to the FMA kernel
    chamfer_distance.set_arithmetic("fma")
    try:
        d3, i3, _, _ = chamfer_nn(_dev(fx_a := golden("nn_config1")["xyz1"], gpu), _dev(golden("nn_config1")["xyz2"], gpu))
        np.testing.assert_array_equal(d3.cpu().numpy().view(np.uint32), fx["nn_config1.dist1"].view(np.uint32))
    finally:
        chamfer_distance.set_arithmetic("unfused")
    with pytest.raises(RuntimeError):
        chamfer_nn(_dev(a, gpu), _dev(c, gpu), FLAG_NN_FMA | FLAG_REF_TAIL_TRUNC)


def _orders(kind, b, n, gpu, seed=0):
    if kind == "reversed":
        return torch.arange(n - 1, -1, -1, dtype=torch.int32, device=gpu).repeat(b, 1).contiguous()
    if kind == "random":
        gen = torch.Generator().manual_seed(seed)
        return torch.stack([torch.randperm(n, generator=gen) for _ in range(b)]).to(torch.int32).to(gpu)
    return kind        # "morton" / None: made by the wrapper


def _check_culled(oracle_mod, gpu, a, c, flags=0, kinds=("morton", None, "reversed", "random")):
    from geometrics_amd.chamfer_distance import chamfer_nn_culled
    e1, j1, e2, j2 = oracle_mod.chamfer_nn(a, c, flags)
    for kind in kinds:
        d1, i1, d2, i2 = chamfer_nn_culled(_dev(a, gpu), _dev(c, gpu), _orders(kind, a.shape[0], a.shape[1], gpu, 1),
                                           _orders(kind, c.shape[0], c.shape[1], gpu, 2), flags)
        np.testing.assert_array_equal(i1.cpu().numpy(), j1, err_msg=str(kind))
        np.testing.assert_array_equal(i2.cpu().numpy(), j2, err_msg=str(kind))
        np.testing.assert_array_equal(d1.cpu().numpy().view(np.uint32), e1.view(np.uint32), err_msg=str(kind))
        np.testing.assert_array_equal(d2.cpu().numpy().view(np.uint32), e2.view(np.uint32), err_msg=str(kind))


@pytest.mark.parametrize("b,n,m", [(1, 500, 500), (2, 3000, 3000), (3, 1, 1), (2, 7, 2466), (1, 2466, 5), (2, 63, 65),
                                   (1, 1025, 1023), (4, 300, 4097), (1, 16, 16), (1, 15, 17), (2, 9000, 130)])
@pytest.mark.parametrize("flags", [0, FLAG_NN_FMA])
def test_culled_nn_equals_the_oracle_for_any_order(oracle_mod, gpu, b, n, m, flags):
    """The culled scan (visiting orders, run spheres, packed evaluation, exact tie path) against the CPU restatement of
    the reference's sequential scan: indices and distances bit for bit, whatever the orders -- coherent (Morton), the
    clouds' own, reversed, random -- and in both arithmetics; ragged sizes, fewer than one run, more than one sphere chunk."""
    rng = np.random.default_rng(b * 1000003 + n * 131 + m)
    a = rng.standard_normal((b, n, 3)).astype(np.float32)
    c = rng.standard_normal((b, m, 3)).astype(np.float32)
    _check_culled(oracle_mod, gpu, a, c, flags)


def test_culled_nn_refuses_an_order_that_is_not_a_permutation(gpu):
    """The kernels index the clouds and the outputs with the order's entries unchecked (nn_scan.h nn_cull_prep_point,
    nn_culled_body), so a supplied order with an out-of-range or a repeated entry must be refused on the host (round-3
    advice), for either cloud; a proper permutation passes."""
    from geometrics_amd.chamfer_distance import chamfer_nn_culled
    rng = np.random.default_rng(1)
    a, c = _dev(rng.standard_normal((2, 130, 3)).astype(np.float32), gpu), _dev(rng.standard_normal((2, 90, 3)).astype(np.float32), gpu)
    good1, good2 = _orders("random", 2, 130, gpu), _orders("random", 2, 90, gpu)
    chamfer_nn_culled(a, c, good1, good2)
    for bad_value in (130, -1, 5):                    # past the end, negative, a duplicate of another entry
        bad = good1.clone()
        bad[1, 7] = bad_value if bad_value != 5 else bad[1, 8]
        with pytest.raises(ValueError, match="not a permutation"):
            chamfer_nn_culled(a, c, bad, good2)
    bad2 = good2.clone()
    bad2[0, 0] = 90
    with pytest.raises(ValueError, match="order2"):
        chamfer_nn_culled(a, c, good1, bad2)


def test_culled_nn_ties_nan_inf_and_the_reference_vectors(oracle_mod, gpu):
    from helpers import golden, golden_names
    from geometrics_amd.chamfer_distance import chamfer_nn_culled
    rng = np.random.default_rng(5)
    a = rng.integers(-3, 4, (2, 777, 3)).astype(np.float32)       # integer grid: masses of exact ties, within and across runs
    c = rng.integers(-3, 4, (2, 1300, 3)).astype(np.float32)
    c[:, 600:900] = c[:, :300]                                      # duplicated targets
    for flags in (0, FLAG_NN_FMA):
        _check_culled(oracle_mod, gpu, a, c, flags)
    rng = np.random.default_rng(9)
    a = rng.standard_normal((1, 70, 3)).astype(np.float32)
    c = rng.standard_normal((1, 300, 3)).astype(np.float32)
    c[0, 0, 1], a[0, 3, 0], c[0, 17] = np.nan, np.inf, np.nan      # NaN seed sticks; an all-inf query; a NaN that never wins
    a[0, 40], c[0, 200, 2] = np.nan, -np.inf
    for flags in (0, FLAG_NN_FMA):
        e1, j1, e2, j2 = oracle_mod.chamfer_nn(a, c, flags)
        for kind in ("morton", None, "random"):
            d1, i1, d2, i2 = chamfer_nn_culled(_dev(a, gpu), _dev(c, gpu), _orders(kind, 1, 70, gpu), _orders(kind, 1, 300, gpu), flags)
            np.testing.assert_array_equal(i1.cpu().numpy(), j1)
            np.testing.assert_array_equal(i2.cpu().numpy(), j2)
            np.testing.assert_array_equal(np.isnan(d1.cpu().numpy()), np.isnan(e1))
            np.testing.assert_array_equal(np.isnan(d2.cpu().numpy()), np.isnan(e2))
            ok = ~np.isnan(e1)
            np.testing.assert_array_equal(d1.cpu().numpy()[ok].view(np.uint32), e1[ok].view(np.uint32))
    # the reference's own outputs (tests/golden: generated from its nnsearch, both builds)
    fx = golden("nnfma_outputs")
    for name in golden_names("nn_"):
        g = golden(name)
        if name == "nn_config2_outputs":
            a, c = meshgen.gt_cloud(2, 3000, first=int(g["gt_first"])), meshgen.gt_cloud(2, 3000, first=int(g["pred_first"]))
        else:
            a, c = g["xyz1"], g["xyz2"]
        d1, i1, d2, i2 = chamfer_nn_culled(_dev(a, gpu), _dev(c, gpu))
        np.testing.assert_array_equal(i1.cpu().numpy(), g["idx1"])
        np.testing.assert_array_equal(i2.cpu().numpy(), g["idx2"])
        d1, i1, d2, i2 = chamfer_nn_culled(_dev(a, gpu), _dev(c, gpu), flags=FLAG_NN_FMA)
        np.testing.assert_array_equal(i1.cpu().numpy(), fx[name + ".idx1"])
        np.testing.assert_array_equal(d1.cpu().numpy().view(np.uint32), fx[name + ".dist1"].view(np.uint32))
    with pytest.raises(RuntimeError):
        chamfer_nn_culled(_dev(a, gpu), _dev(c, gpu), flags=FLAG_REF_TAIL_TRUNC)      # the truncation mode stays on the plain scan
    with pytest.raises(RuntimeError):
        chamfer_nn_culled(_dev(a, gpu), _dev(c, gpu), torch.zeros(1, 3, dtype=torch.int32, device=gpu))


def test_large_clouds_take_the_culled_scan_with_the_same_results(oracle_mod, gpu, monkeypatch):
    """chamfer_nn / ChamferDistance dispatch clouds above AUTO_CULL_PAIRS to the culled scan (Morton orders made on the
    device): same four tensors, bit for bit -- checked against the plain scan and the oracle with the threshold lowered."""
    from geometrics_amd import chamfer_distance as cd
    rng = np.random.default_rng(77)
    a = rng.standard_normal((2, 2100, 3)).astype(np.float32)
    c = rng.standard_normal((2, 1900, 3)).astype(np.float32)
    c[:, 1000:1100] = c[:, :100]
    plain = chamfer_nn(_dev(a, gpu), _dev(c, gpu))
    calls = []
    real = cd.chamfer_nn_culled
    monkeypatch.setattr(cd, "chamfer_nn_culled", lambda *args, **kw: (calls.append(1), real(*args, **kw))[1])
    monkeypatch.setattr(cd, "AUTO_CULL_PAIRS", 1_000_000)
    auto = cd.chamfer_nn(_dev(a, gpu), _dev(c, gpu))
    i1, i2 = ChamferDistance()(_dev(a, gpu), _dev(c, gpu))
    assert len(calls) == 2
    for x, y in zip(plain, auto):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    assert torch.equal(i1, plain[1]) and torch.equal(i2, plain[3])
    e1, j1, e2, j2 = oracle_mod.chamfer_nn(a, c)
    np.testing.assert_array_equal(auto[1].cpu().numpy(), j1)
    np.testing.assert_array_equal(auto[2].cpu().numpy().view(np.uint32), e2.view(np.uint32))
    calls.clear()
    cd.chamfer_nn(_dev(a, gpu), _dev(c, gpu), FLAG_REF_TAIL_TRUNC)       # the truncation mode never dispatches
    monkeypatch.setattr(cd, "AUTO_CULL_PAIRS", 0)
    cd.chamfer_nn(_dev(a, gpu), _dev(c, gpu))
    assert not calls


def test_chamfer_module_contract(gpu):
    x = torch.rand(2, 100, 3, device=gpu, requires_grad=True)
    y = torch.rand(2, 80, 3, device=gpu)
    i1, i2 = ChamferDistance()(x, y)
    assert i1.dtype == torch.int32 and i2.dtype == torch.int32
    assert i1.shape == (2, 100) and i2.shape == (2, 80)
    assert not i1.requires_grad and i1.device == x.device
    with pytest.raises(RuntimeError):
        ChamferDistance()(x.cpu(), y.cpu())
    with pytest.raises(RuntimeError):
        ChamferDistance()(x.double(), y.double())


def _mesh_case(b, level, npts, seed=0, cube=False):
    V, F = meshgen.icosphere(level)
    verts = meshgen.jittered_batch(V, b, first=seed)
    pts = meshgen.gt_cloud(b, npts, first=seed, cube=cube)
    return verts, F, pts


def _check_tri(oracle_mod, gpu, pts, verts, F, flags=0):
    t1, t2, t3 = (np.ascontiguousarray(verts[:, F[:, k]]) for k in range(3))
    ed, ep, ei = oracle_mod.tri_scan(pts, t1, t2, t3, flags)
    d, p, i = tri_distance(_dev(pts, gpu), _dev(t1, gpu), _dev(t2, gpu), _dev(t3, gpu), flags)
    np.testing.assert_array_equal(i.cpu().numpy(), ei)
    np.testing.assert_array_equal(p.cpu().numpy(), ep)
    np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), ed.view(np.uint32))
    d2, p2, i2 = tri_distance_indexed(_dev(pts, gpu), _dev(verts, gpu), _dev(F, gpu), flags)
    assert torch.equal(i2, i) and torch.equal(p2, p) and torch.equal(d2.view(torch.int32), d.view(torch.int32))
    # the in-library brute-force scan (every pair through the full decision tree) must agree too
    d3, p3, i3 = tri_distance_indexed(_dev(pts, gpu), _dev(verts, gpu), _dev(F, gpu), flags | FLAG_TRI_BRUTE_FORCE)
    assert torch.equal(i3, i) and torch.equal(p3, p) and torch.equal(d3.view(torch.int32), d.view(torch.int32))
    d4, p4, i4 = tri_distance(_dev(pts, gpu), _dev(t1, gpu), _dev(t2, gpu), _dev(t3, gpu), flags | FLAG_TRI_BRUTE_FORCE)
    assert torch.equal(i4, i) and torch.equal(p4, p) and torch.equal(d4.view(torch.int32), d.view(torch.int32))
    # the reference-shaped entry points without a workspace (in-kernel staging) as well
    d5, p5, i5 = tri_distance(_dev(pts, gpu), _dev(t1, gpu), _dev(t2, gpu), _dev(t3, gpu), flags, use_workspace=False)
    assert torch.equal(i5, i) and torch.equal(p5, p) and torch.equal(d5.view(torch.int32), d.view(torch.int32))
    d6, p6, i6 = tri_distance_indexed(_dev(pts, gpu), _dev(verts, gpu), _dev(F, gpu), flags, use_workspace=False)
    assert torch.equal(i6, i) and torch.equal(p6, p) and torch.equal(d6.view(torch.int32), d.view(torch.int32))
    # two-level (grouped) scan: the visiting order decides speed only -- a Morton order (what "auto" above used),
    # no order (flat workspace scan), a random permutation (useless group spheres) and a reversed one all give
    # the same bits, with `index` in the ORIGINAL numbering
    nf = F.shape[0]
    rng = np.random.default_rng(nf)
    orders = [None, morton_order(_dev(verts[0][F].mean(1), gpu)), _dev(rng.permutation(nf).astype(np.int32), gpu),
              _dev(np.arange(nf - 1, -1, -1, dtype=np.int32), gpu)]
    for order in orders:
        d7, p7, i7 = tri_distance_indexed(_dev(pts, gpu), _dev(verts, gpu), _dev(F, gpu), flags, order=order)
        assert torch.equal(i7, i) and torch.equal(p7, p) and torch.equal(d7.view(torch.int32), d.view(torch.int32))
        d8, p8, i8 = tri_distance(_dev(pts, gpu), _dev(t1, gpu), _dev(t2, gpu), _dev(t3, gpu), flags, order=order)
        assert torch.equal(i8, i) and torch.equal(p8, p) and torch.equal(d8.view(torch.int32), d.view(torch.int32))
    return ep


def test_tri_config1(oracle_mod, gpu):
    verts, F, pts = _mesh_case(2, 2, 500)
    _check_tri(oracle_mod, gpu, pts, verts, F)


def test_tri_config3(oracle_mod, gpu):
    verts, F, pts = _mesh_case(1, 4, 3000)
    codes = _check_tri(oracle_mod, gpu, pts, verts, F)
    assert (codes == 0).mean() > 0.5   # interior wins dominate on a closed surface


def test_tri_cube_points_and_fix6(oracle_mod, gpu):
    verts, F, pts = _mesh_case(2, 2, 700, seed=3, cube=True)
    codes = _check_tri(oracle_mod, gpu, pts, verts, F)
    assert set(np.unique(codes)) >= {0, 1, 2, 3, 4, 5}
    _check_tri(oracle_mod, gpu, pts, verts, F, FLAG_FIX_REGION6)


@pytest.mark.parametrize("nf", [1, 3, 4, 5, 511, 513, 516, 320])
def test_tri_ragged_and_truncation(oracle_mod, gpu, nf):
    verts, F, pts = _mesh_case(2, 3, 130, seed=nf)
    F = F[:nf]
    _check_tri(oracle_mod, gpu, pts, verts, F)
    _check_tri(oracle_mod, gpu, pts, verts, F, FLAG_REF_TAIL_TRUNC)


def test_tri_degenerate_triangles(oracle_mod, gpu):
    verts, F, pts = _mesh_case(1, 2, 200, seed=11)
    F = F.copy()
    F[5] = [7, 7, 9]      # zero-length edge -> inf/nan projections
    F[0] = [3, 3, 3]      # degenerate FIRST triangle: NaN seed semantics
    F[40] = [1, 2, 2]
    t1, t2, t3 = (np.ascontiguousarray(verts[:, F[:, k]]) for k in range(3))
    ed, ep, ei = oracle_mod.tri_scan(pts, t1, t2, t3)
    order = morton_order(_dev(verts[0][F].mean(1), gpu))
    for flags, ordr in ((0, None), (FLAG_TRI_BRUTE_FORCE, None), (0, order), (0, torch.flip(order, [0]))):
        d, p, i = tri_distance(_dev(pts, gpu), _dev(t1, gpu), _dev(t2, gpu), _dev(t3, gpu), flags, order=ordr)
        np.testing.assert_array_equal(i.cpu().numpy(), ei)
        np.testing.assert_array_equal(p.cpu().numpy(), ep)
        np.testing.assert_array_equal(np.isnan(d.cpu().numpy()), np.isnan(ed))


@pytest.mark.parametrize("case", ["on_vertices", "far_away", "offset_1000", "tiny_scale", "huge_scale", "slivers",
                                  "duplicate_faces", "points_on_surface", "batch8_baseline"])
def test_tri_culling_is_exact_under_stress(oracle_mod, gpu, case):
    """Inputs chosen to break a sloppy culling bound: zero distances, bounds much larger than the
    mesh, catastrophic-cancellation offsets, extreme scales, near-degenerate triangles, exact ties."""
    rng = np.random.default_rng(hash(case) % 2 ** 31)
    verts, F, pts = _mesh_case(2, 3, 400, seed=21)
    if case == "on_vertices":
        pts[:, :300] = verts[:, rng.integers(0, verts.shape[1], 300)]
    elif case == "far_away":
        pts = pts * 40 + 7
    elif case == "offset_1000":
        verts, pts = verts + 1000.0, pts + 1000.0
    elif case == "tiny_scale":
        verts, pts = verts * 1e-4, pts * 1e-4
    elif case == "huge_scale":
        verts, pts = verts * 1e5, pts * 1e5
    elif case == "slivers":
        verts = verts.copy()
        verts[:, F[::7, 2]] = verts[:, F[::7, 1]] + 1e-6 * rng.standard_normal((2, len(F[::7]), 3)).astype(np.float32)
    elif case == "duplicate_faces":
        F = np.concatenate([F, F[:200], F[::-1][:100]])
    elif case == "points_on_surface":
        ch, u, v = meshgen.sampling_draws(verts, F, 400)
        x, y, z = (np.take_along_axis(verts, F[ch][..., k][..., None].repeat(3, -1), 1) for k in range(3))
        pts = ((1 - u)[..., None] * x + (u * (1 - v))[..., None] * y + (u * v)[..., None] * z).astype(np.float32)
    elif case == "batch8_baseline":
        verts, F, pts = _mesh_case(8, 4, 3000, seed=0)
        d, p, i = tri_distance_indexed(_dev(pts, gpu), _dev(verts, gpu), _dev(F, gpu))
        d2, p2, i2 = tri_distance_indexed(_dev(pts, gpu), _dev(verts, gpu), _dev(F, gpu), FLAG_TRI_BRUTE_FORCE)
        assert torch.equal(i2, i) and torch.equal(p2, p) and torch.equal(d2.view(torch.int32), d.view(torch.int32))
        ed, ep, ei = oracle_mod.tri_scan_indexed(pts[:1], verts[:1], F)
        np.testing.assert_array_equal(i[:1].cpu().numpy(), ei)
        return
    _check_tri(oracle_mod, gpu, np.ascontiguousarray(pts, np.float32), np.ascontiguousarray(verts, np.float32), F)
    _check_tri(oracle_mod, gpu, np.ascontiguousarray(pts, np.float32), np.ascontiguousarray(verts, np.float32), F,
               FLAG_FIX_REGION6)


def test_tri_module_contract(gpu):
    verts, F, pts = _mesh_case(1, 2, 64)
    t = [_dev(np.ascontiguousarray(verts[:, F[:, k]]), gpu) for k in range(3)]
    dist, point, index = TriDistance()(_dev(pts, gpu), *t)
    assert dist.dtype == torch.float32 and point.dtype == torch.int32 and index.dtype == torch.int32
    assert dist.shape == (1, 64) and not dist.requires_grad
    assert int(point.min()) >= 0 and int(point.max()) <= 6 and int(index.max()) < F.shape[0]


def test_size_independent_properties_at_the_config5_shard(gpu):
    """BASELINE config 5 shard (8 meshes, 3000 vs 3000 points, 5120 faces): properties that need no
    oracle run -- self-consistency of (distance, index), optimality against random candidates,
    equivariance under permutation of queries and of targets/triangles, idempotence."""
    B, N = 8, 3000
    V, F = meshgen.icosphere(4)
    verts = _dev(meshgen.jittered_batch(V, B), gpu)
    faces = _dev(F, gpu)
    gt = _dev(meshgen.gt_cloud(B, N), gpu)
    pred = _dev(meshgen.gt_cloud(B, N, first=100), gpu)
    g = torch.Generator(device="cpu").manual_seed(0)

    d1, i1, d2, i2 = chamfer_nn(gt, pred)
    # (distance, index) are consistent: the distance is the squared distance to the indexed target
    picked = torch.gather(pred, 1, i1.long().unsqueeze(-1).expand(-1, -1, 3))
    assert torch.allclose(((picked - gt) ** 2).sum(-1), d1, rtol=1e-6, atol=0)
    # optimality: no random other target is closer
    for _ in range(4):
        r = torch.randint(0, N, (B, N), generator=g).to(gpu)
        other = torch.gather(pred, 1, r.unsqueeze(-1).expand(-1, -1, 3))
        assert bool((((other - gt) ** 2).sum(-1) >= d1 * (1 - 1e-6)).all())
    # symmetry of the two directions: the pair found from one side bounds the other side's distance
    assert bool((torch.gather(d2, 1, i1.long()) <= d1).all())
    # permuting the queries permutes the outputs; permuting the targets relabels the indices
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(gpu)
    gt_p = torch.gather(gt, 1, perm.unsqueeze(-1).expand(-1, -1, 3))
    d1p, i1p, _, _ = chamfer_nn(gt_p, pred)
    assert torch.equal(d1p, torch.gather(d1, 1, perm)) and torch.equal(i1p, torch.gather(i1, 1, perm))
    pred_p = torch.gather(pred, 1, perm.unsqueeze(-1).expand(-1, -1, 3))
    d1q, i1q, _, _ = chamfer_nn(gt, pred_p)
    assert torch.equal(d1q, d1)                                  # the minimum does not depend on the order
    assert torch.equal(torch.gather(perm, 1, i1q.long()), i1.long()) or bool(
        (torch.gather(pred_p, 1, i1q.long().unsqueeze(-1).expand(-1, -1, 3)) == picked).all())
    # idempotence / determinism
    d1r, i1r, d2r, i2r = chamfer_nn(gt, pred)
    assert torch.equal(d1r, d1) and torch.equal(i1r, i1) and torch.equal(d2r, d2) and torch.equal(i2r, i2)

    dist, option, index = tri_distance_indexed(gt, verts, faces)
    assert int(index.min()) >= 0 and int(index.max()) < F.shape[0] and int(option.min()) >= 0 and int(option.max()) <= 6
    # the chosen point is never farther than the nearest corner of the winning triangle
    corners = verts[torch.arange(B, device=gpu)[:, None, None], faces[index.long()]]          # [B,N,3,3]
    corner_d = ((corners - gt.unsqueeze(2)) ** 2).sum(-1).min(-1)[0]
    keep = option != 6                                                                           # Q2: region 6 is off-triangle
    assert bool((dist[keep] <= corner_d[keep] * (1 + 1e-5) + 1e-12).all())
    # query permutation equivariance and triangle relabelling invariance of the distance
    dp, op, ip = tri_distance_indexed(gt_p, verts, faces)
    assert torch.equal(dp, torch.gather(dist, 1, perm)) and torch.equal(ip, torch.gather(index, 1, perm))
    fperm = torch.randperm(F.shape[0], generator=g).to(gpu)
    df, of, jf = tri_distance_indexed(gt, verts, faces[fperm])
    assert torch.equal(df, dist)
    # winners on a shared edge / vertex tie bit-exactly between the adjacent triangles, and the tie goes to
    # the lowest index of the CURRENT order: the distance is order-independent, the label need not be
    same = fperm[jf.long()] == index.long()
    assert float(same.float().mean()) > 0.85
    assert bool(same[option == 0].all())                        # interior winners are unique
    d_again, o_again, i_again = tri_distance_indexed(gt, verts, faces)
    assert torch.equal(d_again, dist) and torch.equal(i_again, index) and torch.equal(o_again, option)


def test_empty_and_degenerate_shapes(gpu):
    """Empty batch is a no-op; a direction without targets has no arg-min and is rejected (the reference
    silently leaves zero-initialised outputs); single points and single triangles work."""
    e = torch.empty(0, 5, 3, device=gpu)
    d1, i1, d2, i2 = chamfer_nn(e, torch.empty(0, 7, 3, device=gpu))
    assert d1.shape == (0, 5) and i2.shape == (0, 7)
    with pytest.raises(RuntimeError):
        chamfer_nn(torch.rand(1, 4, 3, device=gpu), torch.empty(1, 0, 3, device=gpu))
    d, p, i = tri_distance_indexed(torch.empty(0, 3, 3, device=gpu), torch.empty(0, 4, 3, device=gpu),
                                   torch.zeros(1, 3, dtype=torch.int64, device=gpu))
    assert d.shape == (0, 3)
    one = torch.tensor([[[0.2, 0.2, 1.0]]], device=gpu)
    tri = torch.tensor([[[0., 0., 0.], [1., 0., 0.], [0., 1., 0.]]], device=gpu)
    d, p, i = tri_distance_indexed(one, tri, torch.tensor([[0, 1, 2]], device=gpu))
    assert float(d) == 1.0 and int(p) == 0 and int(i) == 0
    d1, i1, d2, i2 = chamfer_nn(one, tri)
    assert int(i1) == 0 and int(i2[0, 0]) == 0 and float(d1) == pytest.approx(1.08, rel=1e-6)


@pytest.mark.parametrize("level,npts,b", [(5, 700, 2), (6, 200, 1)])
def test_tri_two_level_scan_multi_chunk_meshes(oracle_mod, gpu, level, npts, b):
    """20 480 / 81 920 faces: the group spheres no longer fit one LDS staging pass (512 groups = 8192 triangles per
    chunk), so the seed / cull / evaluate phases run per chunk with the bound carried over.  Grouped (k-d order and a
    random order), flat and brute-force scans must agree bitwise; mesh 0 is also checked against the C oracle."""
    verts, F, pts = _mesh_case(b, level, npts, seed=level)
    dv, df, dp = _dev(verts, gpu), _dev(F, gpu), _dev(pts, gpu)
    d, p, i = tri_distance_indexed(dp, dv, df)                                   # k-d order (cached), two-level
    rnd = _dev(np.random.default_rng(level).permutation(F.shape[0]).astype(np.int32), gpu)
    for kw in (dict(order=None), dict(order=rnd), dict(flags=FLAG_TRI_BRUTE_FORCE), dict(flags=FLAG_REF_TAIL_TRUNC)):
        d2, p2, i2 = tri_distance_indexed(dp, dv, df, **kw)
        if kw.get("flags") == FLAG_REF_TAIL_TRUNC:       # a different function of the input: compare its own variants
            d3, p3, i3 = tri_distance_indexed(dp, dv, df, FLAG_REF_TAIL_TRUNC, order=None)
            assert torch.equal(i2, i3) and torch.equal(p2, p3) and torch.equal(d2.view(torch.int32), d3.view(torch.int32))
            continue
        assert torch.equal(i2, i) and torch.equal(p2, p) and torch.equal(d2.view(torch.int32), d.view(torch.int32))
    ed, ep, ei = oracle_mod.tri_scan_indexed(pts[:1, :100], verts[:1], F)
    np.testing.assert_array_equal(i[:1, :100].cpu().numpy(), ei)
    np.testing.assert_array_equal(p[:1, :100].cpu().numpy(), ep)
    np.testing.assert_array_equal(d[:1, :100].cpu().numpy().view(np.uint32), ed.view(np.uint32))


def test_large_point_sets_are_exact(gpu):
    """Sizes far beyond the configs (65 536 queries x 100 000 targets; 100 000 points x 5120 faces): the NN result is
    checked against a chunked torch evaluation of the SAME un-fused arithmetic -- distances bitwise, index = the first
    target attaining the minimum -- and the tri scans (two-level, flat, brute force) against each other."""
    g = torch.Generator(device="cpu").manual_seed(3)
    a = (torch.rand(1, 65536, 3, generator=g) - 0.5).to(gpu)
    b = (torch.rand(1, 100000, 3, generator=g) - 0.5).to(gpu)
    b[0, 70000:70050] = b[0, 10:60]                                  # exact duplicates: ties go to the lower index
    d1, i1, d2, i2 = chamfer_nn(a, b)
    for q, t, d, i in ((a[0], b[0], d1[0], i1[0]), (b[0], a[0], d2[0], i2[0])):
        for s in range(0, q.shape[0], 8192):
            qq = q[s:s + 8192]
            dx, dy, dz = (t[None, :, k] - qq[:, None, k] for k in range(3))
            dist = (dx * dx + dy * dy) + dz * dz                     # same operation order, elementwise = un-fused
            best = dist.min(dim=1)[0]
            first = (dist == best[:, None]).float().argmax(dim=1)
            assert torch.equal(best.view(torch.int32), d[s:s + 8192].view(torch.int32))
            assert torch.equal(first.int(), i[s:s + 8192])
    V, F = meshgen.icosphere(4)
    verts, faces = _dev(meshgen.jittered_batch(V, 1), gpu), _dev(F, gpu)
    pts = (torch.rand(1, 100000, 3, generator=g) - 0.5).to(gpu)
    ref = tri_distance_indexed(pts, verts, faces, FLAG_TRI_BRUTE_FORCE)
    for kw in (dict(), dict(order=None)):
        got = tri_distance_indexed(pts, verts, faces, **kw)
        assert all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(got, ref))


@pytest.mark.parametrize("name", ["tri_true_config1", "tri_true_config3", "tri_true_cube"])
@pytest.mark.parametrize("scan", ["grouped", "flat", "culled", "brute"])
def test_tri_fix6_distances_equal_the_legacy_eberly_truth(gpu, name, scan):
    """The HIP scans against the reference's legacy Eberly point_to_line DIRECTLY (not via the oracle): with the
    region-6 delta corrected every distance is the exact point-to-mesh squared distance (fixture:
    tests/golden/tri_true_*.npz, float64 from old_GEOMetrics/utils.py:734-1026); in the reference's quirk mode it is
    never below it."""
    from helpers import tri_true_case
    verts, faces, pts, true = tri_true_case(name)
    kw = {"grouped": dict(order="auto"), "flat": dict(order=None), "culled": dict(use_workspace=False),
          "brute": dict(use_workspace=False)}[scan]
    extra = FLAG_TRI_BRUTE_FORCE if scan == "brute" else 0
    d6, o6, i6 = tri_distance_indexed(_dev(pts, gpu), _dev(verts, gpu), _dev(faces, gpu), FLAG_FIX_REGION6 | extra, **kw)
    d6 = d6.cpu().numpy().astype(np.float64)
    assert (np.abs(d6 - true) <= 1e-5 * true + 1e-9).all()
    assert int((o6 == 6).sum()) > 10
    dq, oq, iq = tri_distance_indexed(_dev(pts, gpu), _dev(verts, gpu), _dev(faces, gpu), extra, **kw)
    assert (dq.cpu().numpy() >= true * (1 - 1e-5) - 1e-9).all()


def test_compiled_forward_cuda_entry_points_equal_the_ctypes_binding(oracle_mod, gpu):
    """cd.forward_cuda / tri.forward_cuda (the compiled pybind shim with the reference's call shapes) write exactly what
    the ctypes-bound operators return; config-1 size, oracle-checked."""
    from geometrics_amd import _shim
    V, F = meshgen.icosphere(2)
    verts, gt = meshgen.jittered_batch(V, 2), meshgen.gt_cloud(2, 500)
    pred = meshgen.gt_cloud(2, 500, first=9)
    a, c = _dev(gt, gpu), _dev(pred, gpu)
    d1, d2 = torch.empty(2, 500, device=gpu), torch.empty(2, 500, device=gpu)
    i1, i2 = torch.empty(2, 500, dtype=torch.int32, device=gpu), torch.empty(2, 500, dtype=torch.int32, device=gpu)
    assert _shim.cd.forward_cuda(a, c, d1, d2, i1, i2) is None
    e1, j1, e2, j2 = oracle_mod.chamfer_nn(gt, pred)
    np.testing.assert_array_equal(i1.cpu().numpy(), j1)
    np.testing.assert_array_equal(i2.cpu().numpy(), j2)
    np.testing.assert_array_equal(d1.cpu().numpy().view(np.uint32), e1.view(np.uint32))
    tri = [_dev(np.ascontiguousarray(verts[:, F[:, k]]), gpu) for k in range(3)]
    dist, point, index = torch.empty(2, 500, device=gpu), torch.empty(2, 500, dtype=torch.int32, device=gpu), \
        torch.empty(2, 500, dtype=torch.int32, device=gpu)
    assert _shim.tri.forward_cuda(a, *tri, dist, point, index) is None
    et, ept, eit = oracle_mod.tri_scan(gt, *(np.ascontiguousarray(verts[:, F[:, k]]) for k in range(3)))
    np.testing.assert_array_equal(index.cpu().numpy(), eit)
    np.testing.assert_array_equal(point.cpu().numpy(), ept)
    np.testing.assert_array_equal(dist.cpu().numpy().view(np.uint32), et.view(np.uint32))
    with pytest.raises(RuntimeError):
        _shim.cd.forward_cuda(a.double(), c, d1, d2, i1, i2)


def test_reference_shaped_tri_module_takes_the_two_level_scan_with_a_cached_order(gpu):
    """TriDistance()(xyz, tri1, tri2, tri3) (tri_distance.py:9-43; no face list): the module keeps one Morton order per
    triangle count and runs the two-level scan with it -- same bits as the flat scan and as the oracle, also with an order
    cached from ANOTHER geometry of the same size (a stale order may cost speed, never a result), and through the
    compiled `tri.forward_cuda` entry point."""
    import oracle
    from geometrics_amd import _shim, tri_distance as td
    V, Fc = meshgen.icosphere(3)
    B, n = 2, 700
    faces = torch.from_numpy(Fc).to(gpu)
    gt = torch.from_numpy(meshgen.gt_cloud(B, n)).to(gpu)
    td._soup_orders.clear()
    for seed in (0, 7):                 # second pass: the cached order belongs to the first geometry
        verts = torch.from_numpy(meshgen.jittered_batch(V, B, first=seed)).to(gpu)
        if seed:
            verts = verts.flip(1).contiguous()      # same triangle count, every triangle somewhere else
        tris = [verts[:, faces[:, i]].contiguous() for i in range(3)]
        d, p, i = td.TriDistance()(gt, *tris)
        d0, p0, i0 = td.tri_distance(gt, *tris, order=None)
        assert torch.equal(d, d0) and torch.equal(p, p0) and torch.equal(i, i0)
        ed, ep, ei = oracle.tri_scan(gt.cpu().numpy(), *(t.cpu().numpy() for t in tris))
        assert np.array_equal(i.cpu().numpy(), ei) and np.array_equal(p.cpu().numpy(), ep)
        assert np.array_equal(d.cpu().numpy().view(np.uint32), ed.view(np.uint32))
        dist, point, index = (torch.empty(B, n, dtype=t, device=gpu) for t in (torch.float32, torch.int32, torch.int32))
        _shim.tri.forward_cuda(gt, *tris, dist, point, index)
        assert torch.equal(dist, d) and torch.equal(point, p) and torch.equal(index, i)
    key = (Fc.shape[0], gt.device)
    assert key in td._soup_orders and td._soup_orders[key][1] == 2       # built once, used twice
    td._soup_orders[key][1] = td.SOUP_REFRESH                             # due for a rebuild
    old = td._soup_orders[key][0]
    td.TriDistance()(gt, *tris)
    assert td._soup_orders[key][1] == 1 and td._soup_orders[key][0] is not old
