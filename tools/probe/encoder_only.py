"""The ragged encoder step alone (for rocprofv3): MeshEncoder.encode_batch forward + backward on the 16-mesh batch of
tools/time_encoder.py, N eager iterations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geometrics_amd import meshgen, models, ragged  # noqa: E402

dev = torch.device("cuda:0")
levels = [2, 3, 4, 3, 3, 4, 2, 3, 4, 3, 3, 2, 4, 3, 3, 4]
verts, faces = [], []
for i, lv in enumerate(levels):
    V, Fc = meshgen.icosphere(lv)
    verts.append(torch.from_numpy(meshgen.jittered_batch(V, 1, first=i)[0]).to(dev))
    faces.append(torch.from_numpy(Fc).to(dev))
enc = models.MeshEncoder(50).to(dev)
with torch.no_grad():
    for p in enc.parameters():
        if p.dim() == 2:
            p.mul_(1 / 3.0)
batch = ragged.RaggedMeshBatch.from_faces(verts, faces)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(n):
    enc.zero_grad(set_to_none=True)
    enc.encode_batch(batch).square().mean().backward()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    enc.zero_grad(set_to_none=True)
    enc.encode_batch(batch).square().mean().backward()
b.record()
torch.cuda.synchronize()
print("encode_batch fwd+bwd: %.3f ms" % (a.elapsed_time(b) / n))
