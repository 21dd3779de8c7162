"""The practical roof of the headline aggregation launch: the same bytes (8 meshes x 2562 rows x 192 columns in, the same out)
as a pure copy + bias + ReLU in the same launch geometry class (k = 0: no gathers) against the real launch (k = 64), and a
torch copy of the tensor.  us per launch, back to back in a HIP graph (operands partly cache-resident: an upper bound on speed)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geometrics_amd import _lib as L  # noqa: E402
from geometrics_amd import layers, meshgen, utils  # noqa: E402

dev = torch.device("cuda:0")
V, Fc = meshgen.icosphere(4)
csr = layers.adjacency_csr(utils.adj_init(torch.from_numpy(np.ascontiguousarray(Fc)).to(dev))["adj"])
b, nv, c = 8, csr.nv, 192


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) * 1e3 / (5 * reps)


sup, bias, out = torch.randn(b, nv, c, device=dev), torch.randn(c, device=dev), torch.empty(b, nv, c, device=dev)
mb = 2 * b * nv * c * 4 / 1e6
for k in (64, 0):
    t = timed(lambda: L.call("geom_zn_gcn_aggregate_ell_fwd_f32", b, nv, c, k, csr.ell_w, csr.ell_col.data_ptr(), csr.ell_val.data_ptr(),
                             None, None, None, sup.data_ptr(), bias.data_ptr(), 1, out.data_ptr(), None))
    print("aggregation launch, k = %2d: %5.1f us = %4.2f TB/s of %4.1f MB" % (k, t, mb / t, mb))
t = timed(lambda: out.copy_(sup))
print("torch copy of the same tensor: %5.1f us = %4.2f TB/s" % (t, mb / t))
t = timed(lambda: torch.relu(sup + bias))
print("torch relu(x + bias) (two launches, one temporary): %5.1f us" % t)
