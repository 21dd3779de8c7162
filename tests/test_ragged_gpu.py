"""Ragged-batch auto-encoder path on the MI355X: segmented max, block-diagonal CSR
builders, and MeshEncoder.encode_batch against latents / gradients the imported reference produced."""
import numpy as np
import pytest
import torch

from helpers import fill_parameters, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _meshes(fx, gpu):
    verts = [torch.from_numpy(v).to(gpu) for v in np.split(fx["verts"], np.cumsum(fx["sizes"])[:-1])]
    faces = [torch.from_numpy(f).to(gpu) for f in np.split(fx["faces"], np.cumsum(fx["face_counts"])[:-1])]
    return verts, faces


@pytest.mark.parametrize("sizes,c", [([5, 1, 300, 64], 50), ([2562, 642, 7], 300), ([1], 3), ([4100], 65)])
def test_segment_max_matches_torch(gpu, sizes, c):
    from geometrics_amd import ops
    from oracle import ref_ops
    torch.manual_seed(sum(sizes) + c)
    x = torch.randn(sum(sizes), c, device=gpu)
    x[::7] = x[::7].round()                                   # ties inside segments
    x.requires_grad_(True)
    offsets = torch.tensor([0] + sizes, dtype=torch.int64).cumsum(0).to(gpu)
    out = ops.SegmentMax.apply(x, offsets, max(sizes))
    ref = ref_ops.segment_max(x.detach().cpu(), sizes)
    assert torch.equal(out.cpu(), ref)                         # a max is exact
    g = torch.randn_like(out)
    out.backward(g)
    # gradient: all of g[s, col] lands on ONE row of the segment (the lowest row holding the max), zeros elsewhere
    parts, gparts = torch.split(x.detach(), sizes), torch.split(x.grad, sizes)
    for s, (part, gpart) in enumerate(zip(parts, gparts)):
        first = (part == part.max(dim=0, keepdim=True)[0]).float().argmax(dim=0)     # lowest arg-max row
        expect = torch.zeros_like(part)
        expect[first, torch.arange(c, device=gpu)] = g[s]
        assert torch.equal(gpart, expect)


def test_segment_max_nan_and_split_independence(gpu):
    from geometrics_amd import ops
    x = torch.randn(3000, 8, device=gpu)
    x[1234, 3] = float("nan")
    x[2000, 3] = float("nan")
    offsets = torch.tensor([0, 3000], dtype=torch.int64, device=gpu)
    a = ops.SegmentMax.apply(x, offsets, 3000)                 # 47 row splits
    b = ops.SegmentMax.apply(x, offsets, 1)                    # claimed max_len 1 -> a single split scans everything
    assert torch.isnan(a[0, 3]) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(x.max(dim=0, keepdim=True)[0]))


def test_block_diagonal_csr_builders_agree_with_dense(gpu):
    """from_faces (no dense matrix) == from_dense(normalize_adj(calc_adj)) == the block-diagonal dense matrix."""
    from geometrics_amd import ragged, utils
    fx = golden("mesh_encoder_ragged")
    verts, faces = _meshes(fx, gpu)
    adjs = [utils.normalize_adj(utils.calc_adj(f)) for f in faces]
    a = ragged.RaggedMeshBatch.from_faces(verts, faces)
    b = ragged.RaggedMeshBatch.from_dense(verts, adjs)
    dense = torch.block_diag(*adjs)
    for batch in (a, b):
        c = batch.csr
        assert c.nv == dense.shape[0] == batch.total and batch.sizes == [int(s) for s in fx["sizes"]]
        rebuilt = torch.zeros_like(dense)
        rows = torch.repeat_interleave(torch.arange(c.nv, device=gpu), (c.rowptr[1:] - c.rowptr[:-1]).long())
        rebuilt[rows, c.col.long()] = c.val
        assert torch.equal(rebuilt, dense)
        rebuilt_t = torch.zeros_like(dense)
        rows_t = torch.repeat_interleave(torch.arange(c.nv, device=gpu), (c.rowptr_t[1:] - c.rowptr_t[:-1]).long())
        rebuilt_t[rows_t, c.col_t.long()] = c.val_t
        assert torch.equal(rebuilt_t, dense.t())
    assert torch.equal(a.csr.col, b.csr.col) and torch.equal(a.csr.val, b.csr.val)
    assert torch.equal(a.offsets.cpu(), torch.tensor([0] + a.sizes).cumsum(0))


@pytest.mark.parametrize("builder", ["faces", "dense"])
def test_encode_batch_matches_reference_fixture(gpu, builder):
    """One ragged launch sequence == the reference's per-mesh loop (auto_encoder.py:71-76): latents, gradient
    w.r.t. the vertex positions and parameter gradients, against the imported reference (fp32 summation order of the
    CSR rows differs from the dense mm, amplified through 17 layers: 2e-4 of the largest magnitude)."""
    from geometrics_amd import models, ragged, utils
    fx = golden("mesh_encoder_ragged")
    verts, faces = _meshes(fx, gpu)
    enc = fill_parameters(models.MeshEncoder(50), int(fx["seed"])).to(gpu)
    if builder == "faces":
        batch = ragged.RaggedMeshBatch.from_faces(verts, faces)
    else:
        batch = ragged.RaggedMeshBatch.from_dense(verts, [utils.normalize_adj(utils.calc_adj(f)) for f in faces])
    batch.verts.requires_grad_(True)
    latents = enc.encode_batch(batch)
    (latents * torch.from_numpy(fx["g"]).to(gpu)).sum().backward()

    def close(got, want, tol=2e-4):
        want = torch.from_numpy(want)
        return float((got.detach().cpu() - want).abs().max()) <= tol * float(want.abs().max())

    assert latents.shape == (3, 50) and close(latents, fx["latents"])
    assert close(latents, fx["latents_f64"].astype(np.float32))
    assert close(batch.verts.grad, fx["grad_verts"], 1e-3)
    params = dict(enc.named_parameters())
    for key in [k for k in fx if k.startswith("grad.")]:
        assert close(params[key[5:]].grad, fx[key], 1e-3), key


def test_per_mesh_forward_equals_ragged_batch(gpu):
    """The reference call `encoder(mesh, adj)` on one mesh (dense adjacency, cached CSR) and the ragged batch run
    the same kernels row for row: identical latents bit for bit on meshes with the same row order."""
    from geometrics_amd import models, ragged, utils
    fx = golden("mesh_encoder_ragged")
    verts, faces = _meshes(fx, gpu)
    enc = fill_parameters(models.MeshEncoder(50), 3).to(gpu)
    with torch.no_grad():
        batch = ragged.RaggedMeshBatch.from_faces(verts, faces)
        together = enc.encode_batch(batch)
        alone = torch.stack([enc(v, utils.normalize_adj(utils.calc_adj(f))) for v, f in zip(verts, faces)])
    assert float((together - alone).abs().max()) <= 2e-5 * float(alone.abs().max())   # GEMM tiling may differ with M


def test_graphed_encode_equals_the_eager_ragged_batch_for_new_positions(gpu):
    """MeshEncoder.graphed_encode(batch): forward and backward of encode_batch replayed as HIP graphs for a FIXED
    topology.  For positions the capture never saw it must return what the eager call returns -- latents, parameter
    gradients and the gradient of the positions, bit for bit (the same kernels on the same shapes) -- twice in a row (static
    buffers are reused) and with a different upstream gradient."""
    from geometrics_amd import models, ragged
    fx = golden("mesh_encoder_ragged")
    verts, faces = _meshes(fx, gpu)
    enc = fill_parameters(models.MeshEncoder(50), 3).to(gpu)
    batch = ragged.RaggedMeshBatch.from_faces(verts, faces)
    batch.verts.requires_grad_(True)
    graphed = enc.graphed_encode(batch)
    gen = torch.Generator(device="cpu").manual_seed(5)
    for trial in range(3):
        moved = (batch.verts.detach() + 0.01 * torch.randn(batch.verts.shape, generator=gen).to(gpu)).requires_grad_(True)
        upstream = torch.randn(3, 50, generator=gen).to(gpu)
        enc.zero_grad(set_to_none=True)
        batch_moved = ragged.RaggedMeshBatch(moved, batch.sizes, batch.csr)
        want = enc.encode_batch(batch_moved)
        (want * upstream).sum().backward()
        want_grads = [p.grad.clone() for p in enc.parameters()]
        want_dv = moved.grad.clone()
        enc.zero_grad(set_to_none=True)
        moved2 = moved.detach().clone().requires_grad_(True)
        got = graphed(moved2)
        (got * upstream).sum().backward()
        assert torch.equal(got, want), trial
        assert torch.equal(moved2.grad, want_dv), trial
        for p, g in zip(enc.parameters(), want_grads):
            assert torch.equal(p.grad, g), trial


def test_ragged_rejects_bad_input(gpu):
    from geometrics_amd import ragged
    v = torch.zeros(4, 3, device=gpu)
    f = torch.tensor([[0, 1, 2], [1, 2, 3]], device=gpu)
    with pytest.raises(RuntimeError):
        ragged.RaggedMeshBatch.from_faces([v], [f, f])
    with pytest.raises(RuntimeError):
        ragged.RaggedMeshBatch.from_dense([v], [torch.zeros(3, 3, device=gpu)])
    with pytest.raises(RuntimeError):
        ragged.RaggedMeshBatch.from_faces([v.cpu()], [f.cpu()])
