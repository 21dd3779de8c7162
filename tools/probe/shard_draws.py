import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from geometrics_amd import meshgen, ops
from geometrics_amd.tri_distance import face_order
dev = torch.device("cuda:0")
V, Fc = meshgen.icosphere(4)
faces = torch.from_numpy(Fc).to(dev)
face_order(torch.from_numpy(V).to(dev).unsqueeze(0), faces)
allv = torch.from_numpy(meshgen.jittered_batch(V, 16)).to(dev)
gt = torch.from_numpy(meshgen.gt_cloud(16, 3000)).to(dev)
def draw(verts, g, first):
    gi = ops.GtIndex(g)
    ops.manual_seed(3041, dev, mesh_offset=first)
    d = ops.draw_samples(verts, faces, 3000, with_points=True, prepare_scan_for=3000, gt_index=gi)
    return d[0].clone(), d[1].clone(), d[2].clone()
a = draw(allv, gt, 0)
b0 = draw(allv[:8].contiguous(), gt[:8].contiguous(), 0)
b1 = draw(allv[8:].contiguous(), gt[8:].contiguous(), 8)
for k, name in enumerate(("choices", "u", "v")):
    whole = a[k]; parts = torch.cat([b0[k], b1[k]])
    print(name, "equal:", bool(torch.equal(whole, parts)), "mismatches:", int((whole != parts).sum()))
