"""What bounds the any-width table aggregation at an encoder shape (18 432 rows x 300 columns): the launch with and without
the activation, with and without the gathers (k = 0: a pure copy + bias), forward and backward.  us per launch, graph replay."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geometrics_amd import _lib as L  # noqa: E402
from geometrics_amd import meshgen, ragged  # noqa: E402

dev = torch.device("cuda:0")
verts, faces = [], []
for i, lv in enumerate([2, 3, 4, 3, 3, 4, 2, 3, 4, 3, 3, 2, 4, 3, 3, 4]):
    V, Fc = meshgen.icosphere(lv)
    verts.append(torch.from_numpy(meshgen.jittered_batch(V, 1, first=i)[0]).to(dev))
    faces.append(torch.from_numpy(np.ascontiguousarray(Fc)).to(dev))
csr = ragged.RaggedMeshBatch.from_faces(verts, faces).csr
nv = csr.nv


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * reps)


for c in (300, 200, 120, 60):
    sup, bias, out = torch.randn(1, nv, c, device=dev), torch.randn(c, device=dev), torch.empty(1, nv, c, device=dev)
    g = torch.randn(1, nv, c, device=dev)
    gs, gb = torch.empty_like(sup), torch.empty(c, device=dev)
    scr = torch.empty(L.lib().geom_zn_gcn_bwd_scratch_floats(1, nv, c), device=dev)
    mb = 2 * nv * c * 4 / 1e6
    for k in (c // 10, 0):
        for act in (2, 0):
            f = timed(lambda: L.call("geom_zn_gcn_aggregate_ell_fwd_f32", 1, nv, c, k, csr.ell_w, csr.ell_col.data_ptr(), csr.ell_val.data_ptr(),
                                     None, None, None, sup.data_ptr(), bias.data_ptr(), act, out.data_ptr(), None))
            bw = timed(lambda: L.call("geom_zn_gcn_aggregate_ell_bwd_f32", 1, nv, c, k, csr.ell_w, csr.ell_col_t.data_ptr(), csr.ell_val_t.data_ptr(),
                                      None, None, None, g.data_ptr(), out.data_ptr(), None, act, gs.data_ptr(), None, scr.data_ptr()))
            print("c %3d k %2d act %d: fwd %5.1f us (%4.2f TB/s)   bwd %5.1f us   [%5.1f MB read + written forward]" % (c, k, act, f, mb / f, bw, mb))
