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
    end = len(names)
    live = (st[:, end] > st[:, 0]) & (st[:, 0] > 0)          # workgroups that ran (the grid is padded to 8 * ceil(nv / 8))
    # the counters of different XCDs are not synchronised: offsets are taken inside each XCD (workgroup w runs on XCD w % 8)
    xcd = (np.arange(grid) % 8)[live]
    st = st[live]
    for x in range(8):
        st[xcd == x] -= st[xcd == x][:, 0].min()
    t0 = 0
    # s_memtime counts shader-engine clocks (the 144 MFMAs of a wave = 4 608 issue cycles show as ~5 000): cycles, not time
    print("  workgroups %d, first start -> last end %d cycles" % (len(st), st[:, end].max() - t0))
    so = st[:, 0] - t0
    print("  start offsets: mean %d  p90 %d  max %d cycles;  end offsets: p10 %d  mean %d  max %d" % (
        so.mean(), np.percentile(so, 90), so.max(), np.percentile(st[:, end] - t0, 10), (st[:, end] - t0).mean(), (st[:, end] - t0).max()))
    for i, n in enumerate(names):
        d = st[:, i + 1] - st[:, i]
        print("  %-46s mean %6d  p90 %6d  max %6d cycles" % (n, d.mean(), np.percentile(d, 90), d.max()))
    d = st[:, end] - st[:, 0]
    print("  %-46s mean %6d  p90 %6d  max %6d cycles" % ("workgroup total", d.mean(), np.percentile(d, 90), d.max()))


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
