"""Dev helper for rocprofv3 --pmc passes: the scan kernels of the bench shard launched EAGERLY (no HIP graphs: counter
collection cannot trace graph replays) a few times each: NN alone, tri alone, the fused surface scan.
    python tools/pmc_probe.py [meshes]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometrics_amd import _lib as L, meshgen
from geometrics_amd.chamfer_distance import chamfer_nn
from geometrics_amd.tri_distance import face_order, tri_distance_indexed

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
V, F = meshgen.icosphere(4)
verts = torch.from_numpy(meshgen.jittered_batch(V, B)).to(dev)
faces = torch.from_numpy(F).to(dev)
gt = torch.from_numpy(meshgen.gt_cloud(B, 3000)).to(dev)
pred = torch.from_numpy(meshgen.gt_cloud(B, 3000, first=100)).to(dev)
lib = L.lib()
nv, nf, num, n_gt = V.shape[0], F.shape[0], 3000, 3000
f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
o = [torch.empty(B, n_gt, **f32), torch.empty(B, n_gt, **i32), torch.empty(B, num, **f32), torch.empty(B, num, **i32),
     torch.empty(B, n_gt, **f32), torch.empty(B, n_gt, **i32), torch.empty(B, n_gt, **i32), torch.empty(B, n_gt, **f32),
     torch.empty(B, n_gt, 3, **f32), torch.empty(B, n_gt, 3, **f32)]
ws_bytes = lib.geom_tri_distance_workspace_bytes(B, n_gt, nf)
ws = torch.empty(ws_bytes // 4, **f32)
order = torch.empty(lib.geom_surface_order_words(B, nf, num, n_gt), **i32)
uu, vv = torch.rand(B, num, device=dev), torch.rand(B, num, device=dev)
tri_order = face_order(verts, faces)
wrote = ctypes.c_int(0)
for _ in range(4):
    chamfer_nn(gt, pred)
    tri_distance_indexed(gt, verts, faces)
    L.check(lib.geom_surface_scan_f32(B, n_gt, gt.data_ptr(), num, pred.data_ptr(), o[0].data_ptr(), o[1].data_ptr(),
                                      o[2].data_ptr(), o[3].data_ptr(), nv, verts.data_ptr(), nf, faces.data_ptr(),
                                      tri_order.data_ptr(), o[4].data_ptr(), o[5].data_ptr(), o[6].data_ptr(), o[7].data_ptr(),
                                      o[8].data_ptr(), o[9].data_ptr(), uu.data_ptr(), vv.data_ptr(), 1.0, 1.0, order.data_ptr(),
                                      0, ws.data_ptr(), ws_bytes, ctypes.byref(wrote), None, None, L.stream_ptr()), "scan")
    torch.cuda.synchronize()
