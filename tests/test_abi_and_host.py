"""CPU: the C-ABI library loads and exports every symbol include/geom_hip.h declares; host
logic (mesh generator, adjacency, CSR, argument validation) -- no compute calls on a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import bits, golden
from geometrics_amd import _lib, meshgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "geom_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(geom_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 14
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libgeom_hip.so does not export %s" % n
    assert sorted(_lib.declared_symbols()) == names, "ctypes table and header disagree"
    assert _lib.lib().geom_abi_version() == _lib.ABI_VERSION


def test_error_strings_and_argument_rejection_without_a_gpu():
    L = _lib.lib()
    assert b"invalid" in L.geom_strerror(-1)
    assert L.geom_strerror(0) == b"success"
    # negative sizes / null pointers are rejected before any launch
    assert L.geom_chamfer_nn_f32(-1, 1, None, 1, None, None, None, None, None, 0, None) == -1
    assert L.geom_chamfer_nn_f32(1, 4, None, 4, None, None, None, None, None, 0, None) == -1
    assert L.geom_tri_distance_f32(1, 4, None, 0, None, None, None, None, None, None, 0, None) == -1
    assert L.geom_chamfer_nn_f32(0, 4, None, 4, None, None, None, None, None, 0, None) == 0   # empty batch: no-op
    assert L.geom_zn_gcn_aggregate_fwd_f32(1, 4, 8, 9, None, None, None, None, None, 0, None, None) == -1  # k > c


def test_ops_refuse_cpu_tensors():
    from geometrics_amd.chamfer_distance import ChamferDistance
    from geometrics_amd.tri_distance import TriDistance
    from geometrics_amd import utils, layers
    x = torch.zeros(1, 4, 3)
    with pytest.raises(RuntimeError, match="HIP device"):
        ChamferDistance()(x, x)
    with pytest.raises(RuntimeError, match="HIP device"):
        TriDistance()(x, x, x, x)
    with pytest.raises(RuntimeError, match="HIP device"):
        utils.batch_sample(x, torch.zeros(2, 3, dtype=torch.int64), 5,
                           draws=(torch.zeros(1, 5, dtype=torch.int64), torch.zeros(1, 5), torch.zeros(1, 5)))
    with pytest.raises(RuntimeError, match="HIP device"):
        layers.ZERON_GCN(4, 20)(torch.zeros(4, 4), torch.eye(4), torch.relu)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgeom_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.lib()


def test_icosphere_sizes_and_winding():
    for level, (nv, nf, ne) in {2: (162, 320, 480), 4: (2562, 5120, 7680)}.items():
        V, F = meshgen.icosphere(level)
        assert V.shape == (nv, 3) and F.shape == (nf, 3) and V.dtype == np.float32 and F.dtype == np.int64
        edges = {tuple(sorted((f[i], f[(i + 1) % 3]))) for f in F for i in range(3)}
        assert len(edges) == ne
        n = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])
        assert (np.einsum("ij,ij->i", n, V[F].mean(1)) > 0).all()
        np.testing.assert_allclose(np.linalg.norm(V, axis=1), meshgen.RADIUS, rtol=1e-6)


def test_synthetic_inputs_are_deterministic():
    V, F = meshgen.icosphere(2)
    a = meshgen.jittered_batch(V, 3)
    np.testing.assert_array_equal(a[1:], meshgen.jittered_batch(V, 2, first=1))
    ch, u, v = meshgen.sampling_draws(a, F, 100)
    assert ch.shape == (3, 100) and ch.max() < F.shape[0] and (u >= 0).all() and (u <= 1).all()
    np.testing.assert_array_equal(meshgen.gt_cloud(2, 50), meshgen.gt_cloud(2, 50))


def test_adjacency_matches_reference_vectors():
    from geometrics_amd import utils
    g = golden("adj_ico162")
    info = utils.adj_init(torch.from_numpy(g["faces"]))
    np.testing.assert_array_equal(info["adj_orig"].numpy(), g["adj_orig"])
    np.testing.assert_array_equal(bits(info["adj"].numpy()), bits(g["adj"]))
    assert info["faces"].dtype == torch.int64
    g = golden("adj_482")                                # the reference's own template mesh
    info = utils.adj_init(torch.from_numpy(g["faces"]))
    r, c = np.nonzero(info["adj"].numpy())
    np.testing.assert_array_equal(r, g["nnz_rows"])
    np.testing.assert_array_equal(c, g["nnz_cols"])
    np.testing.assert_array_equal(bits(info["adj"].numpy()[r, c]), bits(g["nnz_vals"]))
    assert len(r) == 3362 and info["adj"].shape == (482, 482)


def test_csr_builder_on_host():
    from geometrics_amd import layers, utils
    g = golden("adj_482")
    adj = utils.adj_init(torch.from_numpy(g["faces"]))["adj"]
    rowptr, col, val = layers._to_csr(adj)
    assert rowptr.dtype == torch.int32 and col.dtype == torch.int32 and val.dtype == torch.float32
    assert int(rowptr[-1]) == 3362 and int((rowptr[1:] - rowptr[:-1]).max()) == 33     # the two degree-32 poles
    dense = torch.zeros_like(adj)
    rows = torch.repeat_interleave(torch.arange(482), (rowptr[1:] - rowptr[:-1]).long())
    dense[rows, col.long()] = val
    assert torch.equal(dense, adj)
    rp_t, col_t, val_t = layers._to_csr(adj.t())
    dense_t = torch.zeros_like(adj)
    rows_t = torch.repeat_interleave(torch.arange(482), (rp_t[1:] - rp_t[:-1]).long())
    dense_t[rows_t, col_t.long()] = val_t
    assert torch.equal(dense_t, adj.t())


def test_layer_parameter_names_and_shapes():
    from geometrics_amd import layers
    assert list(dict(layers.ZERON_GCN(7, 20).named_parameters())) == ["weight", "bias"]
    assert list(dict(layers.BatchZERON_GCN(7, 20).named_parameters())) == ["weight", "bias"]
    p = dict(layers.Batch_Image_ZERON_GCNGCN(963, 192).named_parameters())
    assert list(p) == ["weight1", "bias"] and tuple(p["weight1"].shape) == (1, 963, 192)
    # (1 + 1e-6): uniform_(-b, b) draws in fp32 and may return fp32(b), one rounding above the double b (seen once in ~30 runs)
    assert float(p["weight1"].detach().abs().max()) <= 0.3 * 6 / (964 ** 0.5) * (1 + 1e-6) and float(p["bias"].detach().abs().max()) <= 0.1 * (1 + 1e-6)
    for cls in (layers.GCNMax, layers.BatchGCNMax):
        assert list(dict(cls(30, 50).named_parameters())) == ["weight_Ws.0", "weight_Bs.0"]
    assert layers.ZERON_GCN(7, 20, bias=False).bias is None
    assert float(layers.ZERON_GCN(7, 20).bias.abs().max()) == 0.0


def test_triangle_visiting_orders_are_permutations_with_compact_leaves():
    """kd_order / morton_order (host + torch code, no kernel): valid permutations; every run of 16 consecutive
    entries of the k-d order is a spatially compact cell (what the two-level tri scan's group spheres rely on)."""
    import numpy as np
    import torch
    from geometrics_amd import meshgen
    from geometrics_amd.tri_distance import kd_order, morton_order
    V, F = meshgen.icosphere(3)
    cent = torch.from_numpy(V[F].mean(1))
    rng = np.random.default_rng(0)
    shuffled = cent[torch.from_numpy(rng.permutation(len(F)))]          # destroy the generator's own coherence
    for fn in (kd_order, morton_order):
        order = fn(shuffled)
        assert order.dtype == torch.int32 and sorted(order.tolist()) == list(range(len(F)))
    def mean_cell_radius(order):
        c = shuffled[order.long()][: len(F) // 16 * 16].reshape(-1, 16, 3)
        return float((c - c.mean(1, keepdim=True)).norm(dim=-1).max(1)[0].mean())
    identity = torch.arange(len(F), dtype=torch.int32)
    assert mean_cell_radius(kd_order(shuffled)) < 0.25 * mean_cell_radius(identity)
    assert mean_cell_radius(morton_order(shuffled)) < 0.5 * mean_cell_radius(identity)
    # degenerate inputs: fewer points than a leaf, non-finite coordinates
    assert kd_order(cent[:5]).tolist() == sorted(kd_order(cent[:5]).tolist(), key=lambda i: i) or len(kd_order(cent[:5])) == 5
    bad = cent[:40].clone()
    bad[3] = float("nan")
    assert sorted(kd_order(bad).tolist()) == list(range(40)) and sorted(morton_order(bad).tolist()) == list(range(40))


def test_compiled_forward_cuda_shim_loads_and_validates():
    """The pybind module with the reference's `forward_cuda` call shapes (chamfer_distance.cpp:36-38,
    tri_distance.cpp:34-36): built, importable without a GPU, and it refuses CPU tensors instead of faulting."""
    import torch
    from geometrics_amd import _shim
    m = _shim.module()
    assert "chamfer_distance.cpp" in m.chamfer_forward_cuda.__doc__ and "tri_distance.cpp" in m.tri_forward_cuda.__doc__
    z = torch.zeros(1, 2, 3)
    i = torch.zeros(1, 2, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="HIP device"):
        _shim.cd.forward_cuda(z, z, torch.zeros(1, 2), torch.zeros(1, 2), i, i)
    with pytest.raises(RuntimeError, match="HIP device"):
        _shim.tri.forward_cuda(z, z, z, z, torch.zeros(1, 2), i, i)


def test_reference_quirk_mode_switch_reaches_every_flagless_default(monkeypatch):
    """geometrics_amd.set_reference_quirks / GEOM_REF_QUIRKS: the default flags of the NN and tri entry points become
    GEOM_FLAG_REF_TAIL_TRUNC; explicit flags are never touched; the FMA arithmetic does not combine with it."""
    import geometrics_amd
    from geometrics_amd import chamfer_distance as cd
    assert not geometrics_amd.reference_quirks() and cd.default_flags() == 0 and _lib.quirk_flags() == 0
    try:
        geometrics_amd.set_reference_quirks(True)
        assert cd.default_flags() == _lib.FLAG_REF_TAIL_TRUNC == _lib.quirk_flags() == 1
        cd.set_arithmetic("fma")
        with pytest.raises(RuntimeError, match="un-fused"):
            cd.default_flags()
    finally:
        cd.set_arithmetic("unfused")
        geometrics_amd.set_reference_quirks(False)
    assert cd.default_flags() == 0
    import importlib
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", "import geometrics_amd as g; print(int(g.reference_quirks()))"],
                         env=dict(os.environ, GEOM_REF_QUIRKS="1"), cwd=ROOT, stdout=subprocess.PIPE, text=True, timeout=120)
    assert out.stdout.strip() == "1"


def test_bound_gradient_bucket_gathers_only_what_did_not_land_in_it():
    """dist.GradBucket(bind=True): the per-parameter views are registered as gradient targets (the layers' launches write
    there and autograd adopts the tensor); pack() then copies only gradients that were produced elsewhere, zero-fills a
    missing one, and places the trailing scalars."""
    import torch
    from geometrics_amd import dist as gdist, layers
    a, b = torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(4))
    c = torch.nn.Parameter(torch.zeros(5))
    bucket = gdist.GradBucket([a, b, c], extra=2, bind=True)
    assert layers._gradient_buffer(a, a).data_ptr() == bucket.views[0].data_ptr()       # a launch would write here
    assert layers._gradient_buffer(None, a).data_ptr() != bucket.views[0].data_ptr()
    a.grad = bucket.views[0].detach()              # what autograd does with the tensor a backward returned
    a.grad.fill_(1.5)
    assert layers._gradient_buffer(a, a).data_ptr() != bucket.views[0].data_ptr()       # accumulating: never onto the old gradient
    b.grad = torch.full((4,), 2.5)                 # a gradient from somewhere else (library fallback)
    assert bucket.pack(torch.tensor(7.0), torch.tensor(8.0)) is True      # it had to launch copies: reported to the caller
    assert bucket.flat.tolist() == [1.5] * 6 + [2.5] * 4 + [0.0] * 5 + [7.0, 8.0]
    b.grad, c.grad = bucket.views[1].detach(), bucket.views[2].detach()
    assert bucket.pack() is False                                          # everything in place: nothing to launch
    with pytest.raises(ValueError):
        layers.bind_gradient_targets([a], [torch.zeros(3, 2)])
    layers.bind_gradient_targets([a, b, c], [None, None, None])
    assert layers._gradient_buffer(a, a).data_ptr() != bucket.views[0].data_ptr()


def test_a_library_built_from_other_sources_is_refused(monkeypatch):
    """The loader compares the digest the build stamped the library with against the kernel sources on disk: an edit or
    a checkout without a rebuild must not run silently (it did once: a whole set of timings taken with a stale kernel)."""
    from geometrics_amd import build
    assert build.built_digest() == build.source_digest()            # the library under test is the current one
    _lib._refuse_stale_library()
    monkeypatch.setattr(build, "built_digest", lambda: "0" * 64)
    with pytest.raises(RuntimeError, match="rebuild"):
        _lib._refuse_stale_library()
    monkeypatch.setenv("GEOM_ALLOW_STALE_LIB", "1")
    _lib._refuse_stale_library()
    monkeypatch.delenv("GEOM_ALLOW_STALE_LIB")
    monkeypatch.setattr(build, "built_digest", lambda: None)        # no stamp: nothing to compare
    _lib._refuse_stale_library()


def test_culled_chamfer_entry_points_validate_before_launching():
    """geom_nn_cull_index_* / geom_chamfer_nn_culled_*: sizes of the index (run spheres + the cloud in rows of whole runs,
    16-byte aligned), argument checks, empty batches -- nothing here reaches a launch."""
    L = _lib.lib()
    assert L.geom_nn_cull_index_floats(1, 3000) == 4 * 187 + 9000
    assert L.geom_nn_cull_index_floats(2, 17) == 2 * (4 * 1 + 52)             # 51 floats -> a 52-float row
    assert L.geom_nn_cull_index_floats(0, 10) == 0 and L.geom_nn_cull_index_floats(3, 0) == 0
    assert L.geom_chamfer_nn_culled_workspace_floats(2, 3000, 17) == L.geom_nn_cull_index_floats(2, 3000) + L.geom_nn_cull_index_floats(2, 17)
    buf = (ctypes.c_float * 64)()
    p16 = (ctypes.addressof(buf) + 15) & ~15
    assert L.geom_nn_cull_index_f32(-1, 4, p16, None, p16, None) == -1
    assert L.geom_nn_cull_index_f32(0, 4, None, None, None, None) == 0         # empty batch
    assert L.geom_nn_cull_index_f32(1, 4, p16, None, p16 + 4, None) == -1      # index not 16-byte aligned
    assert L.geom_nn_cull_index_f32(1, 4, None, None, p16, None) == -1
    nn = lambda b, n, m, flags, ws: L.geom_chamfer_nn_culled_f32(b, n, p16, m, p16, None, None, p16, p16, p16, p16, flags, ws, None)
    assert nn(-1, 4, 4, 0, p16) == -1 and nn(1, 0, 4, 0, p16) == -1 and nn(1, 4, 4, 0, None) == -1 and nn(1, 4, 4, 0, p16 + 8) == -1
    assert nn(0, 4, 4, 0, None) == 0 and nn(3, 0, 0, 0, None) == 0
    assert nn(1, 4, 4, _lib.FLAG_REF_TAIL_TRUNC, p16) == _lib.EUNSUPPORTED      # the truncation mode stays on the plain scan


def test_new_entry_points_reject_bad_arguments_without_a_gpu():
    """The round-2 entry points (fused scan, prepare, finalize, gather, Adam with in-kernel step advance): sizes and
    pointers are validated before any launch, empty batches are no-ops."""
    L = _lib.lib()
    one = ctypes.c_int(7)
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    # geom_surface_scan_f32(b, n_gt, gt, num, points, sq_gt, idx_p, sq_pred, idx_g, nv, verts, nf, faces, order, ... )
    scan = lambda b, n_gt, num, gt: L.geom_surface_scan_f32(b, n_gt, gt, num, None, None, None, None, None, 0, None, 0, None, None,
                                                            None, None, None, None, None, None, None, None, 1.0, 1.0, None, 0, None,
                                                            0, ctypes.byref(one), None, None, None)
    assert scan(-1, 4, 4, None) == -1 and scan(1, 0, 4, None) == -1 and scan(1, 4, 4, None) == -1
    assert scan(0, 4, 4, None) == 0 and one.value == 0            # empty batch; *records_written cleared
    assert L.geom_surface_prepare_f32(1, 4, None, 4, None, 8, None, None, None, None, None, 8, None, 0, None, 0, None, None, None) == -1
    assert L.geom_surface_prepare_f32(0, 4, None, 4, None, 8, None, None, None, None, None, 8, None, 0, None, 0, None, None, None) == 0
    assert L.geom_surface_prepare_f32(1, 4, None, 20000, None, 8, None, None, None, None, None, 8, None, 0, None, 0, None, None, None) \
        == _lib.EUNSUPPORTED                                       # the face-area CDF lives in LDS: <= 16384 faces
    fin = lambda b, order, loss: L.geom_surface_finalize_f32(b, 8, 4, None, None, None, None, 4, None, None, None, None, None, None,
                                                             p, p, 1.0, 1.0, 1.0, 1.0, 0, 0, order, loss, None)
    assert fin(1, None, p) == -1 and fin(1, p16, None) == -1 and fin(-1, p16, p) == -1 and fin(0, p16, p) == 0
    # off | seg | pface | slot (padded to 4 words) | two float4 records per point | status words [b + 1] (padded to 4)
    assert L.geom_surface_order_words(2, 10, 5, 7) == ((2 * 11 + 3 * 2 * 12 + 3) // 4 * 4) + 2 * 12 * 8 + 4
    assert L.geom_surface_gather_f32(1, 4, 8, None, None, 4, 4, 1, None, None, None, None) == -1
    assert L.geom_surface_gather_f32(0, 4, 8, None, None, 4, 4, 1, None, None, None, None) == 0
    assert L.geom_adam_step_f32(_lib.ADAM_MAX_TENSORS + 1, None, None, None, None, None, 1e-3, .9, .999, 1e-8, 1.0, None, 1, None) == -2   # > 64 tensors per launch
    assert L.geom_adam_step_f32(1, None, None, None, None, None, 1e-3, .9, .999, 1e-8, 1.0, None, 1, None) == -1
    assert L.geom_adam_step_f32(0, None, None, None, None, None, 1e-3, .9, .999, 1e-8, 1.0, None, 1, None) == 0
    assert L.geom_chamfer_nn_f32(1, 4, p, 4, p, p, p, p, p, _lib.FLAG_NN_FMA | _lib.FLAG_REF_TAIL_TRUNC, None) == -1


def test_dense_plan_routes_each_product_by_shape():
    """geometrics_amd.dense.plan (host logic, no GPU): which of a layer's three products go to the matrix-core kernels.
    Forward stays with the library; 192-wide layers get both gradients in the pair launch; the 963-wide layer gets the
    split weight-gradient kernel only; narrow / odd / tiny shapes stay with the library altogether."""
    from geometrics_amd import dense
    hidden = dense.plan(20496, 192, 192)
    assert hidden == {"fwd": "lib", "dx": "mfma", "dw": "mfma", "pair": True}
    first = dense.plan(20496, 963, 192)
    assert first["dw"] == "mfma" and first["dx"] == "lib" and not first["pair"] and first["fwd"] == "lib"
    assert dense.plan(7712, 192, 192)["pair"]                       # the reference's training batch (16 x 482 vertices)
    assert dense.plan(324, 192, 192)["dw"] == "lib"                 # two 162-vertex meshes: launch-bound either way
    assert dense.plan(20496, 192, 3)["dw"] == "lib"                 # the 3-channel output layer of a deformation block
    assert dense.plan(20496, 192, 64)["dw"] == "lib"                # not whole 12-column groups per wave
    assert dense.plan(2 ** 21, 963, 192)["dw"] == "lib"             # beyond the kernels' 32-bit byte offsets
    assert dense.supported(963, 192, 20496) and not dense.supported(963, 200, 20496)


# ---- round 5: host-side plans of the boundary launches and the any-shape product (no device needed) -------------------------
def test_fused_plan_thresholds_and_overrides(monkeypatch):
    from geometrics_amd import fused
    monkeypatch.delenv("GEOM_FUSED_PLAN", raising=False)
    assert fused.force is None
    assert fused.plan(8 * 2562) == {"fwd": False, "bwd": False}          # the BASELINE shard keeps the separate operators
    assert fused.plan(32 * 2562) == {"fwd": True, "bwd": True} and fused.plan(64 * 2562)["bwd"]
    monkeypatch.setenv("GEOM_FUSED_PLAN", "fwd")
    assert fused.plan(10) == {"fwd": True, "bwd": False}
    monkeypatch.setenv("GEOM_FUSED_PLAN", "off")
    assert fused.plan(10 ** 6) == {"fwd": False, "bwd": False}
    fused.force = {"fwd": True, "bwd": True}
    try:
        assert fused.plan(1) == {"fwd": True, "bwd": True}                  # tests force it whatever the environment says
    finally:
        fused.force = None

    class Csr:
        ell_w, over, over_t = 8, None, None
    assert fused.supported(Csr, 192, 64, 192) and fused.supported(Csr, 192, 64, 96)
    assert not fused.supported(Csr, 192, 64, 100) and not fused.supported(Csr, 96, 32, 96) and not fused.supported(Csr, 192, 48, 192)
    Csr.over = (1, 2, 3)
    assert not fused.supported(Csr, 192, 64, 192)                           # long rows: the table kernel with its CSR tail


def test_which_kernel_takes_a_product():
    import torch
    from geometrics_amd import dense, layers
    # the 192-column kernels where they apply, the any-shape kernel for every other width, the library for tiny inputs
    assert dense.plan(20496, 963, 192)["dw"] == "mfma" and not dense.plan(20496, 963, 192)["pair"]
    assert dense.plan(20496, 192, 192)["pair"]
    assert dense.plan(18432, 300, 300)["dw"] == "lib" and dense.any_supported(18432, 300, 300)
    assert layers._takes_any_shape_kernel(18432, 300, 300) and layers._takes_any_shape_kernel(7712, 192, 3)
    assert not layers._takes_any_shape_kernel(20496, 192, 192)              # dense_gemm.hip's pair launch
    assert not layers._takes_any_shape_kernel(100, 60, 60)                  # launch-bound: the library
    keep = layers.use_any_shape_products
    layers.use_any_shape_products = False
    try:
        assert not layers._takes_any_shape_kernel(18432, 300, 300)
    finally:
        layers.use_any_shape_products = keep
    with pytest.raises(ValueError):
        dense.gemm(torch.zeros(4, 3), torch.zeros(5, 2))                    # summed extents differ: refused before any launch
    with pytest.raises(ValueError):
        dense.gemm(torch.zeros(4, 6)[:, ::2], torch.zeros(3, 2))            # rows must have unit stride
    link = layers._StackLink()
    assert link.wt is None and link.dx is None and link.g_ref is None and not link.wanted
