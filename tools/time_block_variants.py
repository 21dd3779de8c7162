"""The deformation block's weight gradients two ways (models.BatchMeshDeformationBlock.batch_weight_gradients) at the
reference's training shape (batch 16 x 482 vertices) -- bench.training_shape_times -- and the block alone at the BASELINE
shard (8 x 2562 vertices).  GPU box:  python tools/time_block_variants.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                             # noqa: E402
import torch                                   # noqa: E402
import bench                                   # noqa: E402
from geometrics_amd import gemm_tuning, layers, meshgen, models, utils   # noqa: E402

dev = torch.device("cuda:0")
gemm_tuning.enable()


def block_at(level_verts, faces, batch, cin=1155):
    info = utils.adj_init(torch.from_numpy(faces).to(dev))
    nv = level_verts.shape[0]
    torch.manual_seed(1)
    block = models.BatchMeshDeformationBlock(cin, nv).to(dev).train()
    feats = torch.randn(batch, nv, 3, device=dev, requires_grad=True)
    pooled = torch.randn(batch, nv, cin - 3, device=dev, requires_grad=True)

    def fb():
        for p in block.parameters():
            p.grad = None
        feats.grad = pooled.grad = None
        with layers.deferred_parameter_gradients():
            f, coords = block(feats, pooled, info["adj"])
            (f.sum() + coords.sum()).backward()
    return bench.event_time_us(fb, iters=5, warm=3)


for batched in (True, False):
    models.BatchMeshDeformationBlock.batch_weight_gradients = batched
    t = bench.training_shape_times(dev)
    V, Fc = meshgen.icosphere(4)
    big = block_at(V, Fc, 8)
    print("batch_weight_gradients=%-5s training shape: step %.1f us (block fwd+bwd %.1f us); block at 8 x 2562 vertices: %.1f us"
          % (batched, t["step_us"], t["deformation_block_fwd_bwd_us"], big), flush=True)
