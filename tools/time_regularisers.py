"""Dev helper: the sparse regularisers vs the reference's dense formulation (same torch ops the
reference runs), 8 meshes of 2562 vertices, forward + backward, kernel time via HIP-graph replay."""
import sys
import torch
sys.path.insert(0, ".")
from geometrics_amd import meshgen, utils
dev = torch.device("cuda:0")
V, F = meshgen.icosphere(4)
faces = torch.from_numpy(F).to(dev)
info = utils.adj_init(faces)
pos = torch.from_numpy(meshgen.jittered_batch(V, 8)).to(dev).requires_grad_(True)


def dense_lap(p):
    orig = info["adj_orig"]
    ns = torch.matmul(orig, p) - p
    return p - ns * (1. / (orig.sum(1) - 1)).view(-1, 1)


def dense_edge(v):
    p1, p2, p3 = (torch.index_select(v, 1, faces[:, k]) for k in range(3))
    return (torch.sum((p2 - p1) ** 2, -1).mean() + torch.sum((p3 - p1) ** 2, -1).mean() + torch.sum((p2 - p3) ** 2, -1).mean()) / 3.


def t(fn, it=50):
    for _ in range(5):
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


def fb(fn):
    def run():
        pos.grad = None
        fn(pos).sum().backward()
    return run


print("laplacian fwd+bwd  sparse kernel %.1f us   dense reference formulation %.1f us" % (t(fb(lambda p: utils.batch_get_lap_info(p, info))), t(fb(dense_lap))))
print("edge loss fwd+bwd  fused kernel  %.1f us   reference formulation       %.1f us" % (t(fb(lambda p: utils.batch_calc_edge(p, info))), t(fb(dense_edge))))
