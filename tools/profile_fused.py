"""Eager launches of the fused layer-boundary kernels at the BASELINE shard (for rocprofv3 --pmc passes: tools/prof_fused.sh)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometrics_amd import fused, layers, meshgen, utils  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
V, Fc = meshgen.icosphere(4)
nv = V.shape[0]
csr = layers.adjacency_csr(utils.adj_init(torch.from_numpy(np.ascontiguousarray(Fc)).to(dev))["adj"])
K, C = 64, 192
sp = torch.randn(b, nv, C, device=dev)
xs, ss = torch.empty_like(sp), torch.empty_like(sp)
bias = torch.randn(C, device=dev) * 0.1
w = torch.randn(C, C, device=dev) * 0.1
wt = w.t().contiguous()
mask = torch.zeros(b * nv * 16, dtype=torch.int16, device=dev)
go = torch.randn(b, nv, C, device=dev)
part = torch.empty(fused.partial_rows(b, nv), C, device=dev)
for _ in range(10):
    fused.layer_forward(sp, bias, csr, K, 1, w, x_out=xs, mask=mask, s_out=ss)
    fused.layer_backward(go, None, mask, csr, K, 1, wt, g_out=xs, grad_in=ss, colsum_partial=part)
torch.cuda.synchronize()
