import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from geometrics_amd import meshgen, ops
dev = torch.device("cuda:0")
V, F = meshgen.icosphere(4)
B = 8
verts = torch.from_numpy(meshgen.jittered_batch(V, B)).to(dev); faces = torch.from_numpy(F).to(dev)
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
print("draw only            %.1f us" % timeit(lambda: ops.draw_samples(verts, faces, 3000, with_points=True)))
print("draw + tri prep      %.1f us" % timeit(lambda: ops.draw_samples(verts, faces, 3000, with_points=True, prepare_scan_for=3000)))
