"""-m gpu: what the in-kernel samplers draw, as DISTRIBUTIONS (reference utils.py:590-633: `torch.multinomial(areas, num, True)`
per mesh, `u = sqrt(U1)`, `v = U2`).  The draws themselves cannot equal torch's (another generator), so parity of this stage is
parity in law: both generators -- independent draws (ops.draw_samples) and the visiting-order generation of the culled route
(sorted uniforms from exponential spacings, `gt_index=`) -- are held to

  * Kolmogorov-Smirnov on U1 = u^2 and on v against U(0, 1), and a chi-square on the JOINT law of (u^2, v) over a 16 x 16 grid
    (uniform on the unit square <=> the point is uniform in its triangle under the reference's barycentric map);
  * a chi-square of the face counts against the normalised areas (faces pooled to expected counts >= 50);
  * independence: of (u^2, v) from the face drawn; between two meshes of one call; between two calls (replays) of one mesh;
    between two data-parallel shards (mesh_offset) -- correlations within 5 / sqrt(n), a chi-square on the contingency table of
    the two meshes' (area-balanced) face bins; and no serial correlation inside a stream.

Every bound is a p-value floor of 1e-6 or a 5-sigma band at n = 200 000 (fixed seeds: the tests are deterministic; a generator
whose law is off in the third digit fails them)."""
import numpy as np
import pytest
import torch
from scipy import stats

from geometrics_amd import meshgen, ops

pytestmark = pytest.mark.gpu

P_FLOOR = 1e-6


def _setup(gpu, meshes=2):
    V, Fc = meshgen.icosphere(3)                      # 642 vertices / 1280 faces, jittered: unequal areas
    verts = torch.from_numpy(np.ascontiguousarray(meshgen.jittered_batch(V, meshes))).to(gpu)
    faces = torch.from_numpy(Fc).to(gpu)
    return verts, faces


N_GT_SORTED = 2750     # (a step the fused scan route takes: tests/test_ops_parity_gpu.py::test_sorted_draws_at_ragged_sizes)


def _draw(gpu, verts, faces, num, rounds, sorted_route, seed, mesh_offset=0):
    """rounds x num samples per mesh: (choices, u, v) as numpy [rounds, B, num]."""
    gi = None
    if sorted_route:
        gt = torch.from_numpy(np.ascontiguousarray(meshgen.gt_cloud(verts.shape[0], N_GT_SORTED))).to(gpu)
        gi = ops.GtIndex(gt)
    ops.manual_seed(seed, gpu, mesh_offset=mesh_offset)
    cs, us, vs = [], [], []
    for _ in range(rounds):
        if sorted_route:
            d = ops.draw_samples(verts, faces, num, with_points=True, prepare_scan_for=N_GT_SORTED, gt_index=gi)
            assert isinstance(d[4], ops.ScanPrep) and d[4].sample_index is not None, "the visiting-order generation did not run"
        else:
            d = ops.draw_samples(verts, faces, num)
        cs.append(d[0].cpu().numpy()), us.append(d[1].cpu().numpy()), vs.append(d[2].cpu().numpy())
    ops.manual_seed(0, gpu)
    return np.stack(cs), np.stack(us).astype(np.float64), np.stack(vs).astype(np.float64)


def _corr_ok(a, b, sigmas=5.0):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    r = np.corrcoef(a, b)[0, 1]
    return abs(r) <= sigmas / np.sqrt(a.size), r


@pytest.mark.parametrize("sorted_route", [False, True], ids=["independent", "visiting_order"])
def test_marginal_and_joint_laws(gpu, sorted_route):
    verts, faces = _setup(gpu, 6 if sorted_route else 2)
    num, rounds = (1000, 200) if sorted_route else (200000, 1)
    c, u, v = _draw(gpu, verts, faces, num, rounds, sorted_route, seed=101)
    u1 = (u * u)[:, 0].ravel()
    vv = v[:, 0].ravel()
    assert u1.size == 200000
    assert stats.kstest(u1, "uniform").pvalue > P_FLOOR, "u^2 is not uniform"
    assert stats.kstest(vv, "uniform").pvalue > P_FLOOR, "v is not uniform"
    grid, _, _ = np.histogram2d(np.clip(u1, 0, 1 - 1e-12), np.clip(vv, 0, 1 - 1e-12), bins=16, range=[[0, 1], [0, 1]])
    assert stats.chisquare(grid.ravel()).pvalue > P_FLOOR, "(u^2, v) is not uniform on the unit square"
    # faces against the normalised areas
    areas = ops.face_areas(verts, faces)[0].double().cpu().numpy()
    p = areas / areas.sum()
    counts = np.bincount(c[:, 0].ravel(), minlength=p.size).astype(np.float64)
    order = np.argsort(p)
    # faces pooled in ascending area until a cell expects >= 50 draws (the jittered mesh has slivers)
    groups, cell, acc = np.empty(p.size, dtype=np.int64), 0, 0.0
    for k, pk in enumerate(p[order] * counts.sum()):
        groups[k] = cell
        acc += pk
        if acc >= 50:
            cell, acc = cell + 1, 0.0
    if acc > 0 and cell > 0:
        groups[groups == cell] = cell - 1      # a short last cell joins its neighbour
    obs = np.bincount(groups, weights=counts[order])
    exp = np.bincount(groups, weights=p[order]) * counts.sum()
    assert exp.min() >= 50 and obs.size > 100
    assert stats.chisquare(obs, exp).pvalue > P_FLOOR, "face frequencies do not follow the areas"
    # (u^2, v) independent of the face drawn: against the face's area rank
    rank = np.empty(p.size)
    rank[order] = np.arange(p.size)
    ok, r = _corr_ok(rank[c[:, 0].ravel()], u1)
    assert ok, "u^2 correlates with the face drawn (r = %.4f)" % r
    ok, r = _corr_ok(rank[c[:, 0].ravel()], vv)
    assert ok, "v correlates with the face drawn (r = %.4f)" % r
    ok, r = _corr_ok(u1, vv)
    assert ok, "u^2 and v correlate (r = %.4f)" % r


@pytest.mark.parametrize("sorted_route", [False, True], ids=["independent", "visiting_order"])
def test_independence_across_meshes_replays_and_shards(gpu, sorted_route):
    verts, faces = _setup(gpu, 6 if sorted_route else 2)
    num, rounds = (1000, 200) if sorted_route else (100000, 2)
    c, u, v = _draw(gpu, verts, faces, num, rounds, sorted_route, seed=202)
    u1 = u * u
    # two meshes of one call
    for name, x in (("u^2", u1), ("v", v)):
        ok, r = _corr_ok(x[:, 0], x[:, 1])
        assert ok, "%s of two meshes of one call correlate (r = %.4f)" % (name, r)
    # two calls (replays) of one mesh: round k against round k + 1
    for name, x in (("u^2", u1), ("v", v)):
        ok, r = _corr_ok(x[:-1, 0], x[1:, 0])
        assert ok, "%s of consecutive calls correlate (r = %.4f)" % (name, r)
    # no serial correlation inside a stream (sample i against sample i + 1)
    for name, x in (("u^2", u1), ("v", v)):
        ok, r = _corr_ok(x[:, 0, :-1], x[:, 0, 1:])
        assert ok, "%s is serially correlated (r = %.4f)" % (name, r)
    # the faces two meshes draw: contingency table over 8 area-balanced bins of each mesh's faces
    if not sorted_route:          # (in visiting order sample i of both meshes sits at the same quantile by construction)
        areas = ops.face_areas(verts, faces).double().cpu().numpy()
        bins = []
        for m in range(2):
            order = np.argsort(areas[m])
            cum = np.cumsum(areas[m][order]) / areas[m].sum()
            b = np.empty(order.size, dtype=np.int64)
            b[order] = np.minimum((cum * 8).astype(np.int64), 7)
            bins.append(b[c[:, m].ravel()])
        table = np.zeros((8, 8))
        np.add.at(table, (bins[0], bins[1]), 1)
        assert stats.chi2_contingency(table)[1] > P_FLOOR, "the faces of two meshes of one call are not independent"
    # two data-parallel shards: the same seed, another mesh offset -> other global meshes
    c2, u2, v2 = _draw(gpu, verts, faces, num, rounds, sorted_route, seed=202, mesh_offset=verts.shape[0])
    assert not np.array_equal(u, u2)
    for name, x, y in (("u^2", u1, u2 * u2), ("v", v, v2)):
        ok, r = _corr_ok(x[:, 0], y[:, 0])
        assert ok, "%s of two shards correlate (r = %.4f)" % (name, r)
    # ... and a shard at offset 0 IS the first call again (keyed on the global mesh index, not on launch order)
    c3, u3, v3 = _draw(gpu, verts, faces, num, rounds, sorted_route, seed=202, mesh_offset=0)
    assert np.array_equal(c, c3) and np.array_equal(u, u3) and np.array_equal(v, v3)
