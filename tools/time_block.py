"""Dev helper: the fused deformation block vs the same block written as the reference writes it (eager
torch ops: dense adjacency product, cat, BatchNorm1d, relu, add, div), forward + backward, 8 meshes of
2562 vertices, 963+192 input features, hidden 192; kernel time via HIP-graph replay."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from geometrics_amd import gemm_tuning, meshgen, models, utils
dev = torch.device("cuda:0")
gemm_tuning.enable()
V, Fc = meshgen.icosphere(4)
nv = V.shape[0]
info = utils.adj_init(torch.from_numpy(Fc).to(dev))
adj = info["adj"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
block = models.BatchMeshDeformationBlock(963, nv).to(dev).train()
feats = torch.randn(B, nv, 3, device=dev, requires_grad=True)
pooled = torch.randn(B, nv, 960, device=dev, requires_grad=True)


class RefStyleBlock(torch.nn.Module):
    """The reference's formulation with plain torch ops (models.py:237-297, layers.py:107-116)."""
    def __init__(self, fused):
        super().__init__()
        self.f = fused
        self.bns = torch.nn.ModuleList([torch.nn.BatchNorm1d(nv) for _ in range(13)]).to(dev)

    def gc(self, i, x):
        l = getattr(self.f, "gc%d" % i)
        s = torch.matmul(x, l.weight1)
        k = s.shape[-1] // 3
        return torch.cat((torch.matmul(adj, s[:, :, :k]), s[:, :, k:]), dim=-1) + l.bias

    def forward(self, features, pooled):
        full = torch.cat((features, pooled), dim=-1)
        x = F.relu(self.bns[0](self.gc(1, full)))
        x = F.relu(self.bns[1](self.gc(2, x)))
        f = (full[:, :, :192] + x) / 2
        for i in (3, 5, 7, 9, 11):
            x = F.relu(self.bns[i - 1](self.gc(i, f)))
            x = F.relu(self.bns[i](self.gc(i + 1, x)))
            f = (f + x) / 2
        x = F.relu(self.bns[12](self.gc(13, f)))
        f = (f + x) / 2
        return f, self.gc(15, f)


ref = RefStyleBlock(block).train()


def t(fn, it=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it):
            fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


def fb(mod, *extra):
    def run():
        feats.grad = pooled.grad = None
        for p in block.parameters():
            p.grad = None
        f, c = mod(feats, pooled, *extra)
        (f.sum() + c.sum()).backward()
    return run


a, b_ = t(fb(block, adj)), t(fb(ref))
print("deformation block fwd+bwd, B=%d: fused kernels %.0f us   reference formulation (dense adj, eager BN) %.0f us   x%.2f" % (B, a, b_, b_ / a))
