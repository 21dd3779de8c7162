"""Ragged batches of meshes for the auto-encoder path.

The reference encodes a batch one mesh at a time -- `for mesh, adj in zip(batch['verts'], batch['adjs'])`
(auto_encoder.py:71-76) -- because every ShapeNet mesh has its own vertex count and its own dense
normalised adjacency (utils.py:251-258).  On MI355X that is ~17 x 5 small launches per mesh.  Here the
meshes are concatenated along the vertex axis and their adjacencies become ONE block-diagonal CSR, so a
0N-GCN layer over the whole batch is one GEMM over sum(V) rows plus one aggregation launch, and GCNMax's
per-mesh maximum is a segmented reduction (csrc/segment.hip).
"""
import torch

from . import layers


class RaggedMeshBatch:
    """verts [sum(V),3] fp32, offsets [B+1] int64 (device), sizes (host), csr = block-diagonal normalised
    adjacency in the layout the 0N-GCN kernels consume (pass it wherever a layer takes `adj`)."""

    def __init__(self, verts, sizes, csr):
        self.verts = verts
        self.sizes = [int(s) for s in sizes]
        self.num_meshes = len(self.sizes)
        self.max_len = max(self.sizes) if self.sizes else 0
        host = torch.tensor([0] + self.sizes, dtype=torch.int64).cumsum(0)
        self.total = int(host[-1])
        self.offsets = host.to(verts.device)
        self.csr = csr

    # ---------------------------------------------------------------- builders ----
    @classmethod
    def from_faces(cls, verts_list, faces_list):
        """Straight from the triangle lists: the normalised adjacency D^-1 (A + I) of utils.py:96-131 per mesh,
        assembled directly in CSR -- no dense [V,V] matrix is ever formed and there is one host sync (the
        number of distinct edges) for the whole batch."""
        if len(verts_list) != len(faces_list) or not verts_list:
            raise RuntimeError("need one face list per mesh and at least one mesh")
        dev = verts_list[0].device
        if dev.type != "cuda":
            raise RuntimeError("ragged batches live on a HIP device; geometrics_amd has no CPU path")
        sizes = [int(v.shape[0]) for v in verts_list]
        total = sum(sizes)
        base, shifted = 0, []
        for v, f in zip(verts_list, faces_list):
            if v.dim() != 2 or v.shape[1] != 3 or f.dim() != 2 or f.shape[1] != 3:
                raise RuntimeError("meshes are (verts [V,3], faces [F,3]) pairs")
            shifted.append(f.to(dev, torch.int64) + base)
            base += v.shape[0]
        faces = torch.cat(shifted)
        a, b, c = faces[:, 0], faces[:, 1], faces[:, 2]
        loops = torch.arange(total, device=dev)
        rows = torch.cat([a, a, b, b, c, c, loops])
        cols = torch.cat([b, c, a, c, a, b, loops])
        keys = torch.unique(rows * total + cols)          # sorted: row-major == CSR order
        rows, cols = keys // total, keys % total
        counts = torch.bincount(rows, minlength=total)
        rowptr = torch.zeros(total + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(counts, 0)
        r_inv = 1.0 / counts.to(torch.float32)            # normalize_adj: 1 / row sum of the binary matrix
        rowptr, col = rowptr.to(torch.int32), cols.to(torch.int32).contiguous()
        # the pattern is symmetric, so CSR^T shares rowptr/col and only swaps which row's scale an entry carries
        csr = layers.csr_from_parts(rowptr, col, r_inv[rows].contiguous(), rowptr, col, r_inv[cols].contiguous())
        verts = torch.cat([v.to(dev, torch.float32) for v in verts_list]).contiguous()
        return cls(verts, sizes, csr)

    @classmethod
    def from_dense(cls, verts_list, adjs_list):
        """From the per-mesh dense adjacencies the reference's data loader hands over (`batch['adjs']`,
        utils.py:256-257).  Values are taken as they are (any normalisation)."""
        if len(verts_list) != len(adjs_list) or not verts_list:
            raise RuntimeError("need one adjacency per mesh and at least one mesh")
        dev = verts_list[0].device
        if dev.type != "cuda":
            raise RuntimeError("ragged batches live on a HIP device; geometrics_amd has no CPU path")
        parts, parts_t, base = [], [], 0
        for v, adj in zip(verts_list, adjs_list):
            if adj.shape != (v.shape[0], v.shape[0]):
                raise RuntimeError("adjacency %s does not match %d vertices" % (tuple(adj.shape), v.shape[0]))
            dense = adj.detach().to(dev, torch.float32)
            parts.append(layers._to_csr(dense) + (base,))
            parts_t.append(layers._to_csr(dense.t()) + (base,))
            base += v.shape[0]

        def join(ps):
            nnz_base, rowptrs, cols, vals = 0, [torch.zeros(1, dtype=torch.int32, device=dev)], [], []
            for rowptr, col, val, vbase in ps:
                rowptrs.append(rowptr[1:] + nnz_base)
                cols.append(col + vbase)
                vals.append(val)
                nnz_base += int(col.numel())
            return torch.cat(rowptrs).contiguous(), torch.cat(cols).contiguous(), torch.cat(vals).contiguous()

        csr = layers.csr_from_parts(*join(parts), *join(parts_t))
        verts = torch.cat([v.to(dev, torch.float32) for v in verts_list]).contiguous()
        return cls(verts, [v.shape[0] for v in verts_list], csr)

    # ------------------------------------------------------------------ helpers ----
    def split(self, x):
        """[sum(V), ...] -> list of per-mesh views."""
        return list(torch.split(x, self.sizes, dim=0))


__all__ = ["RaggedMeshBatch"]
