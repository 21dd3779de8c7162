"""Phase timeline of the deformation block's layer launches from the stamps of a -DDB_PROBE_STAMPS build (db_stamps.sh):
mean / p90 of every phase over the workgroups, and the launch's first-start -> last-end span, at the reference's training shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from geometrics_amd import _lib, deform, layers, meshgen, utils

gpu = torch.device("cuda:0")
V, Fc = meshgen.uv_sphere()
nv = V.shape[0]
adj = utils.adj_init(torch.from_numpy(Fc).to(gpu))["adj"]
csr = layers.adjacency_csr(adj)
b, c = 16, 192
torch.manual_seed(0)
s = torch.randn(b, nv, c, device=gpu)
bias = torch.randn(c, device=gpu) * 0.1
gamma, beta = torch.rand(nv, device=gpu) + 0.5, torch.randn(nv, device=gpu) * 0.2
res = torch.randn(b, nv, c, device=gpu)
w = torch.randn(c, c, device=gpu) / 14
packed, packed_t = deform.pack_weights([w])
rm, rv = torch.zeros(nv, device=gpu), torch.ones(nv, device=gpu)
z, x, s_next, dz, ds, gres, g2 = (torch.randn(b, nv, c, device=gpu) for _ in range(7))
mean, invstd = torch.zeros(nv, device=gpu), torch.ones(nv, device=gpu)
gbw, gbb = torch.empty(nv, device=gpu), torch.empty(nv, device=gpu)
colsum = torch.empty(nv, c, device=gpu)
L = _lib.lib()
L.geom_db_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
grid = 8 * ((nv + 7) // 8)


def read(names):
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 16, dtype=np.uint64)
    assert L.geom_db_probe_read(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(1024, 16)[:grid].astype(np.int64)
    live = st[:, 0] > 0
    st = st[live]
    t0 = st[:, 0].min()
    print("  workgroups %d, launch span (first start -> last end) %.2f us (100 MHz clock)" % (len(st), (st[:, len(names)].max() - t0) / 100.0))
    print("  start offsets: mean %.2f us, max %.2f us" % ((st[:, 0] - t0).mean() / 100.0, (st[:, 0] - t0).max() / 100.0))
    for i, n in enumerate(names):
        d = (st[:, i + 1] - st[:, i]) / 100.0
        print("  %-46s mean %6.2f  p90 %6.2f  max %6.2f us" % (n, d.mean(), np.percentile(d, 90), d.max()))
    d = (st[:, len(names)] - st[:, 0]) / 100.0
    print("  %-46s mean %6.2f  p90 %6.2f  max %6.2f us" % ("workgroup total", d.mean(), np.percentile(d, 90), d.max()))


def fwd():
    deform.layer_forward(s, bias, csr, gamma, beta, rm, rv, True, 0.1, 1e-5, True, res, 0.5, z, x, mean, invstd, w_next=packed[0], s_out=s_next)


def bwd():
    deform.layer_backward((b, nv, c), csr, z, gamma, beta, mean, invstd, True, True, 0.5, dz, gbw, gbb, dz_up=s, ds_up=ds, wt_up=packed_t[0],
                          g2=g2, grad_res=gres, colsum=colsum)


for name, fn, names in (("forward", fwd, ["table + gathers (+ slice requested)", "statistics (2 block sums)", "finish + stores + panel", "barrier",
                                          "wait for the weight slice", "144 MFMAs + staging", "barrier", "output stores issued"]),
                        ("backward", bwd, ["table + gathers + G store + panel", "barrier", "wait for the weight slice", "144 MFMAs + staging", "barrier",
                                           "read back + mask", "block sum", "dZ stores + column sums"])):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    # cold-ish: evict the operands with a large copy, then one launch
    junk = torch.empty(64 * 1024 * 1024, device=gpu)
    junk.fill_(1.0)
    fn()
    print(name, "(one launch behind a 256 MB fill)")
    read(names)
    for _ in range(3):
        fn()
    print(name, "(warm, back to back)")
    read(names)
