"""Dev helper: surface loss fwd+bwd run eagerly at the bench shard or the reference training shape (for
rocprofv3 --kernel-trace --stats):  python tools/profile_loss.py bench|train"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from geometrics_amd import meshgen, utils
dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
(V, F), B = (meshgen.icosphere(4), 8) if sys.argv[1] == "bench" else (meshgen.uv_sphere(), 16)
info = utils.adj_init(to(F))
pos = to(meshgen.jittered_batch(V, B)).requires_grad_(True)
gt = to(meshgen.gt_cloud(B, 3000))
for _ in range(12):
    pos.grad = None
    utils.batch_point_to_surface(pos, info, gt, num=3000).backward()
torch.cuda.synchronize()
