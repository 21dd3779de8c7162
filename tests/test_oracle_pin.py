"""CPU: pin the oracle against the golden vectors the REFERENCE produced
(tests/golden/make_golden.py) and, when it is present, against the reference's own
nnsearch binary (oracle/_ref)."""

import numpy as np
import pytest
import torch

import oracle
from oracle import ref_ops
from helpers import bits, golden, golden_names, tri_true_case
from geometrics_amd import meshgen

NN_CASES = [n for n in golden_names("nn_") if n != "nn_config2_outputs"]


@pytest.mark.parametrize("name", NN_CASES)
def test_nn_scan_matches_reference_vectors(name):
    g = golden(name)
    d1, i1, d2, i2 = oracle.chamfer_nn(g["xyz1"], g["xyz2"])
    np.testing.assert_array_equal(i1, g["idx1"])
    np.testing.assert_array_equal(i2, g["idx2"])
    np.testing.assert_array_equal(bits(d1), bits(g["dist1"]))
    np.testing.assert_array_equal(bits(d2), bits(g["dist2"]))
    # the un-truncated tiled kernel semantics is the same scan
    e1, j1 = oracle.nn_tiled(g["xyz1"], g["xyz2"], 0)
    np.testing.assert_array_equal(j1, g["idx1"])
    np.testing.assert_array_equal(bits(e1), bits(g["dist1"]))


def test_nn_scan_config2_reference_outputs():
    g = golden("nn_config2_outputs")
    gt = meshgen.gt_cloud(2, 3000, first=int(g["gt_first"]))
    pr = meshgen.gt_cloud(2, 3000, first=int(g["pred_first"]))
    assert float(gt.sum() + pr.sum()) == float(g["checksum"])   # regenerated inputs are the fixture's inputs
    d1, i1, d2, i2 = oracle.chamfer_nn(gt, pr)
    np.testing.assert_array_equal(i1, g["idx1"])
    np.testing.assert_array_equal(i2, g["idx2"])
    np.testing.assert_array_equal(bits(d1), bits(g["dist1"]))


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref (reference nnsearch) not built here")
@pytest.mark.parametrize("seed", range(4))
def test_nn_scan_equals_live_reference_binary(seed):
    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    a = rng.standard_normal((2, n, 3)).astype(np.float32)
    b = (rng.integers(-2, 3, (2, m, 3)) if seed % 2 else rng.standard_normal((2, m, 3))).astype(np.float32)
    d, i = oracle.nn_scan(a, b)
    dr, ir = oracle.ref_nnsearch(a, b)
    np.testing.assert_array_equal(i, ir)
    np.testing.assert_array_equal(bits(d), bits(dr))


def _nn_case_inputs(name):
    g = golden(name)
    if name == "nn_config2_outputs":
        return (meshgen.gt_cloud(2, 3000, first=int(g["gt_first"])), meshgen.gt_cloud(2, 3000, first=int(g["pred_first"]))), g
    return (g["xyz1"], g["xyz2"]), g


@pytest.mark.parametrize("name", NN_CASES + ["nn_config2_outputs"])
def test_nn_fma_arithmetic_matches_the_fma_built_reference_vectors(name):
    """GEOM_FLAG_NN_FMA, the second pinned arithmetic: oracle_nn_scan_fma against the outputs of the
    reference's nnsearch built with -mfma -ffp-contract=fast (tests/golden/nnfma_outputs.npz), bit for bit; and the
    index-equality property between the two arithmetics on every NN fixture (ties on the integer grid are exact in
    both; random clouds never put two candidates within one rounding of each other)."""
    (a, b), g = _nn_case_inputs(name)
    fx = golden("nnfma_outputs")
    d1, i1, d2, i2 = oracle.chamfer_nn(a, b, oracle.FLAG_NN_FMA)
    np.testing.assert_array_equal(i1, fx[name + ".idx1"])
    np.testing.assert_array_equal(i2, fx[name + ".idx2"])
    np.testing.assert_array_equal(bits(d1), bits(fx[name + ".dist1"]))
    np.testing.assert_array_equal(bits(d2), bits(fx[name + ".dist2"]))
    np.testing.assert_array_equal(i1, g["idx1"])            # same winners as the un-fused reference
    np.testing.assert_array_equal(i2, g["idx2"])
    np.testing.assert_allclose(d1, g["dist1"], rtol=1e-6, atol=1e-12)


@pytest.mark.skipif(not oracle.have_ref_fma(), reason="oracle/_ref FMA build (or an FMA-capable host) not available")
@pytest.mark.parametrize("seed", range(4))
def test_nn_fma_scan_equals_live_fma_built_reference_binary(seed):
    rng = np.random.default_rng(40 + seed)
    n, m = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    a = rng.standard_normal((2, n, 3)).astype(np.float32)
    b = (rng.integers(-2, 3, (2, m, 3)) if seed % 2 else rng.standard_normal((2, m, 3))).astype(np.float32)
    d, i = oracle.nn_scan_fma(a, b)
    dr, ir = oracle.ref_nnsearch_fma(a, b)
    np.testing.assert_array_equal(i, ir)
    np.testing.assert_array_equal(bits(d), bits(dr))


def test_tail_truncation_quirks():
    rng = np.random.default_rng(0)
    a = rng.standard_normal((1, 50, 3)).astype(np.float32)
    for m in (3, 4, 5, 513, 515, 516, 2466):
        b = rng.standard_normal((1, m, 3)).astype(np.float32)
        d, i = oracle.nn_tiled(a, b, oracle.FLAG_REF_TAIL_TRUNC)
        last = ((m - 1) // 512) * 512
        if m - last < 4:                       # Q3: empty final tile merges (0.0, 0)
            assert (i == 0).all() and (d == 0).all()
        else:                                  # Q1: truncated targets are never candidates
            skipped = {k for k in range(m) if (k - (k // 512) * 512) >= min(512, m - (k // 512) * 512) // 4 * 4}
            assert not (set(i.ravel().tolist()) & skipped)
            keep = np.array([k for k in range(m) if k not in skipped])
            d2, i2 = oracle.nn_scan(a, b[:, keep])
            np.testing.assert_array_equal(keep[i2], i)


def test_tri_scan_agrees_with_reference_closest_point_formulas():
    """The reference has no CPU tri_distance; its python calc_point_to_line, evaluated on the
    restatement's (index, option), must give the same squared distance for options 0-5
    (option 6 differs by construction: kernel quirk Q2)."""
    g = golden("tri_vs_ref_formulas")
    tri = [np.ascontiguousarray(g["verts"][:, g["faces"][:, k]]) for k in range(3)]
    d, opt, idx = oracle.tri_scan(g["points"], *tri)
    np.testing.assert_array_equal(opt, g["option"])
    np.testing.assert_array_equal(idx, g["index"])
    ok = g["option"][0] != 6
    assert ok.sum() > 500
    np.testing.assert_allclose(d[0][ok], g["ref_sqdist"][ok], rtol=2e-4, atol=1e-9)
    d2, o2, i2 = oracle.tri_scan_indexed(g["points"], g["verts"], g["faces"])
    np.testing.assert_array_equal(bits(d2), bits(d))
    np.testing.assert_array_equal(i2, idx)
    # true distance sanity: the chosen point is never farther than the nearest corner
    corner = np.min([((g["points"][0][:, None] - t[0][None]) ** 2).sum(-1).min(1) for t in tri], axis=0)
    assert (d[0][ok] <= corner[ok] * (1 + 1e-5) + 1e-12).all()


@pytest.mark.parametrize("name", ["tri_true_config1", "tri_true_config3", "tri_true_cube"])
def test_tri_scan_pinned_by_the_legacy_eberly_implementation(name):
    """Independent pin of the decision tree AND the arg-min of oracle_tri_scan: the reference's legacy
    `point_to_line` (old_GEOMetrics/utils.py:734-1026, Eberly's region formulation -- no code or structure in common
    with tri_distance.cu) gives the exact point-to-mesh squared distance of every query (float64, fixture made by
    tests/golden/make_golden.py --tri-true).  With the region-6 delta corrected the scan must reproduce it for EVERY
    point: a wrong region classification, a wrong candidate formula or a wrong arg-min would all show up as a
    larger distance.  In the reference's quirk mode (region 6 walks along AB, tri_distance.cu:180) the result can
    only be >= the true distance, and is the same wherever the same triangle wins outside region 6."""
    verts, faces, pts, true = tri_true_case(name)
    d6, o6, i6 = oracle.tri_scan_indexed(pts, verts, faces, oracle.FLAG_FIX_REGION6)
    assert (np.abs(d6 - true) <= 1e-5 * true + 1e-9).all()
    assert (o6 == 6).sum() > 10                      # region 6 really occurs in the corrected mode
    dq, oq, iq = oracle.tri_scan_indexed(pts, verts, faces)
    assert (dq >= true * (1 - 1e-5) - 1e-9).all()
    same = (oq != 6) & (iq == i6)
    assert same.sum() > 0.8 * same.size
    np.testing.assert_array_equal(bits(dq[same]), bits(d6[same]))
    # the direct (pre-gathered corners) entry point is the same scan
    tri = [np.ascontiguousarray(verts[:, faces[:, k]]) for k in range(3)]
    d2, o2, i2 = oracle.tri_scan(pts, *tri, oracle.FLAG_FIX_REGION6)
    np.testing.assert_array_equal(bits(d2), bits(d6))
    np.testing.assert_array_equal(i2, i6)


def test_tri_pair_option_codes():
    A, B, C = np.float32([0, 0, 0]), np.float32([1, 0, 0]), np.float32([0, 1, 0])
    cases = {(-1, -1, 0): 1, (2, -1, 0): 2, (-1, 2, 0): 3, (.5, -1, 0): 4, (1, 1, 0): 5, (-1, .5, 0): 6, (.2, .2, 3): 0}
    for p, code in cases.items():
        d, opt = oracle.tri_pair(np.float32(p), A, B, C)
        assert opt == code
    d, _ = oracle.tri_pair(np.float32([.2, .2, 3]), A, B, C)
    assert d == np.float32(9)
    d6, _ = oracle.tri_pair(np.float32([-1, .5, 0]), A, B, C)                          # Q2: C + uca*(B-A)
    d6f, _ = oracle.tri_pair(np.float32([-1, .5, 0]), A, B, C, oracle.FLAG_FIX_REGION6)
    assert d6f == np.float32(1.0) and d6 != d6f


# --------------------------------------------------- python stages vs reference ----
def test_sampling_restatement_bitwise():
    g = golden("sample_v162")
    verts = torch.from_numpy(g["verts"]).requires_grad_(True)
    pts = ref_ops.sample_points(verts, torch.from_numpy(g["faces"]), torch.from_numpy(g["choices"]),
                                torch.from_numpy(g["u"]), torch.from_numpy(g["v"]))
    np.testing.assert_array_equal(bits(pts.detach().numpy()), bits(g["points"]))
    pts.backward(torch.from_numpy(g["grad_points"]))
    np.testing.assert_allclose(verts.grad.numpy(), g["grad_verts"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,fn", [("p2p_v162", ref_ops.point_to_point), ("p2s_v162", ref_ops.point_to_surface)])
def test_loss_restatements(name, fn):
    g = golden(name)
    verts = torch.from_numpy(g["verts"]).requires_grad_(True)
    loss, f1 = fn(verts, torch.from_numpy(g["faces"]), torch.from_numpy(g["gt"]), torch.from_numpy(g["choices"]),
                  torch.from_numpy(g["u"]), torch.from_numpy(g["v"]), f1=True)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    assert abs(f1 - float(g["f1"])) < 1e-12
    np.testing.assert_allclose(verts.grad.numpy(), g["grad_verts"], rtol=1e-4, atol=1e-6)


def test_point_to_line_restatement():
    g = golden("p2line_options")
    a, b, c = (torch.from_numpy(g[k]).requires_grad_(True) for k in "abc")
    loss = ref_ops.point_to_line(torch.from_numpy(g["p"]), a, b, c, torch.from_numpy(g["option"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    for t, k in ((a, "grad_a"), (b, "grad_b"), (c, "grad_c")):
        np.testing.assert_allclose(t.grad.numpy(), g[k], rtol=1e-3, atol=1e-6)


def test_adjacency_restatement():
    g = golden("adj_ico162")
    adj_orig = ref_ops.calc_adj(torch.from_numpy(g["faces"]))
    np.testing.assert_array_equal(adj_orig.numpy(), g["adj_orig"])
    np.testing.assert_array_equal(bits(ref_ops.normalize_adj(adj_orig).numpy()), bits(g["adj"]))


LAYER_SPLIT = {"ZERON_GCN": 10, "BatchZERON_GCN": 10, "Batch_Image_ZERON_GCNGCN": 3, "Batch_Image_ZERON_GCNGCN_out3": 3}
LAYER_ACT = {"ZERON_GCN": torch.nn.functional.elu, "BatchZERON_GCN": torch.nn.functional.elu,
             "Batch_Image_ZERON_GCNGCN": torch.relu, "Batch_Image_ZERON_GCNGCN_out3": lambda t: t,
             "GCNMax": torch.nn.functional.elu, "BatchGCNMax": torch.nn.functional.elu}


@pytest.mark.parametrize("name", sorted(LAYER_ACT))
def test_layer_restatements(name):
    g = golden("layer_" + name)
    adj = torch.from_numpy(golden("layer_adj")["adj"])
    x = torch.from_numpy(g["x"])
    if name in LAYER_SPLIT:
        wkey = "param.weight1" if "Image" in name else "param.weight"
        out = ref_ops.zero_n_layer(x, adj, torch.from_numpy(g[wkey]), torch.from_numpy(g["param.bias"]),
                                   LAYER_SPLIT[name], LAYER_ACT[name])
    else:
        out = ref_ops.gcn_max(x, adj, torch.from_numpy(g["param.weight_Ws.0"]), torch.from_numpy(g["param.weight_Bs.0"]),
                              LAYER_ACT[name], batched=name.startswith("Batch"))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-5, atol=1e-6)


def test_regulariser_restatements():
    g = golden("regularisers_v162")
    verts = torch.from_numpy(g["verts"]).requires_grad_(True)
    faces = torch.from_numpy(g["faces"])
    adj_orig = ref_ops.calc_adj(faces)
    lap = ref_ops.lap_info(verts, adj_orig)
    np.testing.assert_allclose(lap.detach().numpy(), g["lap"], rtol=1e-5, atol=1e-7)
    lap.backward(torch.from_numpy(g["grad_lap"]))
    np.testing.assert_allclose(verts.grad.numpy(), g["grad_verts_lap"], rtol=1e-5, atol=1e-6)
    v2 = torch.from_numpy(g["verts"]).requires_grad_(True)
    edge = ref_ops.calc_edge(v2, faces)
    edge.backward()
    np.testing.assert_allclose(edge.item(), g["edge"], rtol=1e-6)
    np.testing.assert_allclose(v2.grad.numpy(), g["grad_verts_edge"], rtol=1e-4, atol=1e-8)


def test_nn_grad_restatement_equals_autograd_of_the_distance_form():
    """oracle_nn_grad follows the reference's nnd_backward (my_lib.c:49-111): it is the gradient of
    sum(w1*dist1) + sum(w2*dist2) with the NN indices held fixed."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal((2, 40, 3)).astype(np.float32)
    b = rng.standard_normal((2, 55, 3)).astype(np.float32)
    w1 = rng.standard_normal((2, 40)).astype(np.float32)
    w2 = rng.standard_normal((2, 55)).astype(np.float32)
    _, i1, _, i2 = oracle.chamfer_nn(a, b)
    g1, g2 = oracle.nn_grad(a, b, w1, w2, i1, i2)
    ta, tb = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    d1 = ((ta - torch.gather(tb, 1, torch.from_numpy(i1).long().unsqueeze(-1).expand(-1, -1, 3))) ** 2).sum(-1)
    d2 = ((tb - torch.gather(ta, 1, torch.from_numpy(i2).long().unsqueeze(-1).expand(-1, -1, 3))) ** 2).sum(-1)
    ((d1 * torch.from_numpy(w1)).sum() + (d2 * torch.from_numpy(w2)).sum()).backward()
    np.testing.assert_allclose(g1, ta.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g2, tb.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_mesh_encoder_restatement_matches_reference_fixture():
    """oracle.ref_ops.mesh_encoder (the per-mesh checker of the ragged encoder path) against latents the imported
    reference MeshEncoder produced (tests/golden/make_golden.py --encoder)."""
    import torch
    from geometrics_amd import models
    from helpers import fill_parameters
    from oracle import ref_ops
    fx = golden("mesh_encoder_ragged")
    enc = fill_parameters(models.MeshEncoder(50), int(fx["seed"]))
    params = {k: v.detach() for k, v in enc.state_dict().items()}
    verts = np.split(fx["verts"], np.cumsum(fx["sizes"])[:-1])
    faces = np.split(fx["faces"], np.cumsum(fx["face_counts"])[:-1])
    for i, (v, f) in enumerate(zip(verts, faces)):
        adj = ref_ops.normalize_adj(ref_ops.calc_adj(torch.from_numpy(f)))
        lat = ref_ops.mesh_encoder(params, torch.from_numpy(v), adj)
        assert np.allclose(lat.numpy(), fx["latents"][i], rtol=1e-5, atol=1e-6)
    # fp32 summation-order noise of the 17-layer stack, measured by the reference's own float64 run
    assert np.abs(fx["latents"] - fx["latents_f64"]).max() < 1e-4 * np.abs(fx["latents_f64"]).max()
