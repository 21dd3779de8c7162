"""Phase stamps of the fused layer-boundary launch (probe build libgeom_zs_stamps.so, tools/probe/zs_variants.sh build):
   GEOM_ALLOW_STALE_LIB=1 GEOM_LIB_OVERRIDE=tools/probe/bin/libgeom_zs_stamps.so python tools/probe/zs_stamps.py [fwd|bwd]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geometrics_amd import _lib, fused, layers, meshgen, utils  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
b = 8
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
part = torch.empty(fused.partial_rows(b, nv), C, device=dev)
for _ in range(5):
    if mode == "fwd":
        fused.layer_forward(sp, bias, csr, K, 1, w, x_out=xs, mask=mask, s_out=ss)
    else:
        fused.layer_backward(sp, None, mask, csr, K, 1, wt, g_out=xs, grad_in=ss, colsum_partial=part)
torch.cuda.synchronize()
n = 256 * 64
buf = (ctypes.c_ulonglong * n)()
L = _lib.lib()
L.geom_zs_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.geom_zs_probe_read(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 2, 32).astype(np.int64)
t0 = st[:, :, 0].min()
print("first wave starts: spread over workgroups %d ticks (MFMA role), %d (gather role)" % (st[:, 0, 0].max() - t0, st[:, 1, 0].max() - t0))
last = st.max(axis=2).max(axis=1)
print("last stamp of a workgroup, from the first start on the chip: median %d, p90 %d, max %d ticks" % (np.median(last - t0), np.percentile(last - t0, 90), (last - t0).max()))
print("slowest workgroups:", np.argsort(last)[-6:], (np.sort(last)[-6:] - t0))
for role, label in ((0, "MFMA waves"), (1, "gather waves")):
    t = st[:, role, :]
    names = {0: "start", 1: "weight slice loaded", 2: "row-block 0 finished", 3: "barrier 0"}
    for it in range(5):
        names[4 + 4 * it] = "b%d begin" % it
        names[5 + 4 * it] = "b%d %s" % (it, "MFMAs done" if role == 0 else "loads + stores issued")
        names[6 + 4 * it] = "b%d %s" % (it, "tile staged" if role == 0 else "next row-block finished")
        names[7 + 4 * it] = "b%d barrier" % it
    print(label)
    prev = t[:, 0]
    for i in sorted(names):
        cur = t[:, i]
        if i == 0 or not (cur > 0).all():
            continue
        d = cur - prev
        print("   %-30s median %6d  p10 %6d  p90 %6d   (at %7d)" % (names[i], np.median(d), np.percentile(d, 10), np.percentile(d, 90),
                                                                    np.median(cur - t[:, 0])))
        prev = cur
