"""Dev helper: the deformation block's layer launches at the reference's training shape (16 x 482 x 192), HIP-graph replay of
a chain of launches (ping-pong buffers, like the block), with pieces switched off to see what each costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geometrics_amd import deform, layers, meshgen, utils

gpu = torch.device("cuda:0")
V, Fc = meshgen.uv_sphere() if "--ico" not in sys.argv else meshgen.icosphere(3)
nv = V.shape[0]
adj = utils.adj_init(torch.from_numpy(Fc).to(gpu))["adj"]
csr = layers.adjacency_csr(adj)
b, c, n = 16, 192, 12
torch.manual_seed(0)
bias = torch.randn(c, device=gpu) * 0.1
gamma, beta = torch.rand(nv, device=gpu) + 0.5, torch.randn(nv, device=gpu) * 0.2
ws = [torch.randn(c, c, device=gpu) / 14 for _ in range(n)]
packed, packed_t = deform.pack_weights(ws)
rm, rv = torch.zeros(nv, device=gpu), torch.ones(nv, device=gpu)
S = [torch.randn(b, nv, c, device=gpu) for _ in range(2)]
Z = torch.randn(n, b, nv, c, device=gpu)
X = torch.randn(n + 1, b, nv, c, device=gpu)
DZ = torch.randn(n + 1, b, nv, c, device=gpu)
DS = torch.randn(n, b, nv, c, device=gpu)
GR = torch.randn(n, b, nv, c, device=gpu)
mean, invstd = torch.zeros(n, nv, device=gpu), torch.ones(n, nv, device=gpu)
gbw, gbb = torch.empty(n, nv, device=gpu), torch.empty(n, nv, device=gpu)
colsum = torch.empty(n, nv, c, device=gpu)
bench.settle_clocks(gpu, 200)


def fwd_chain(z=True, res=True, product=True):
    def run():
        for i in range(n):
            deform.layer_forward(S[i & 1], bias, csr, gamma, beta, rm, rv, True, 0.1, 1e-5, True, X[i] if res else None, 0.5,
                                 Z[i] if z else None, X[i + 1], mean[i], invstd[i], w_next=packed[i] if product else None,
                                 s_out=S[(i + 1) & 1] if product else None)
    return run


def bwd_chain(res=True, g2=True, product=True, colsums=True):
    def run():
        for i in range(n):
            kw = dict(dz_up=DZ[i], ds_up=DS[i], wt_up=packed_t[i]) if product else dict(g=DZ[i])
            deform.layer_backward((b, nv, c), csr, Z[i], gamma, beta, mean[i], invstd[i], True, res, 0.5, DZ[i + 1], gbw[i], gbb[i],
                                  g2=GR[(i + 1) % n] if g2 else None, grad_res=GR[i] if res else None,
                                  colsum=colsum[i] if colsums else None, **kw)
    return run


for name, fn in (("forward  full (Z, residual, product)", fwd_chain()),
                 ("forward  without the Z store", fwd_chain(z=False)),
                 ("forward  without the residual read", fwd_chain(res=False)),
                 ("forward  without Z and residual", fwd_chain(z=False, res=False)),
                 ("forward  no product", fwd_chain(product=False)),
                 ("backward full (residual, second gradient, product, column sums)", bwd_chain()),
                 ("backward without residual / second gradient", bwd_chain(res=False, g2=False)),
                 ("backward without the column sums", bwd_chain(colsums=False)),
                 ("backward no product", bwd_chain(product=False))):
    print("%-70s %6.2f us per launch" % (name, bench.event_time_us(fn, iters=5, warm=2) / n))
