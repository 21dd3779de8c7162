"""Dev helper: HIP-event timing of the two arg-min scans at the BASELINE shard (B meshes)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometrics_amd import meshgen
from geometrics_amd.chamfer_distance import chamfer_nn
from geometrics_amd.tri_distance import tri_distance_indexed

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
V, F = meshgen.icosphere(4)
verts = torch.from_numpy(meshgen.jittered_batch(V, B)).to(dev)
faces = torch.from_numpy(F).to(dev)
gt = torch.from_numpy(meshgen.gt_cloud(B, 3000)).to(dev)
pred = torch.from_numpy(meshgen.gt_cloud(B, 3000, first=100)).to(dev)


def timeit(fn, iters=40, warm=5):
    """Kernel-only time: the launches are captured in a HIP graph, so python/ctypes overhead is excluded."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


t_nn = timeit(lambda: chamfer_nn(gt, pred))
t_tri = timeit(lambda: tri_distance_indexed(gt, verts, faces))
t_nn_fma = timeit(lambda: chamfer_nn(gt, pred, 8))
print(f"nn fma mode {t_nn_fma:.1f} us")
pairs_nn = 2 * B * 3000 * 3000
pairs_tri = B * 3000 * 5120
print(f"B={B} nn {t_nn:.1f} us ({pairs_nn / t_nn / 1e6:.2f} Tpair/s)  tri {t_tri:.1f} us ({pairs_tri / t_tri / 1e6:.3f} Tpair/s)")
print(f"per mesh: nn {t_nn / B:.1f} us  tri {t_tri / B:.1f} us")
t_flat = timeit(lambda: tri_distance_indexed(gt, verts, faces, order=None))
perm = torch.randperm(faces.shape[0], device=dev).to(torch.int32)
t_rand = timeit(lambda: tri_distance_indexed(gt, verts, faces, order=perm), iters=10)
from geometrics_amd.tri_distance import morton_order
mort = morton_order(verts[0][faces].mean(dim=1))
t_mort = timeit(lambda: tri_distance_indexed(gt, verts, faces, order=mort))
ident = torch.arange(faces.shape[0], device=dev, dtype=torch.int32)
t_ident = timeit(lambda: tri_distance_indexed(gt, verts, faces, order=ident))
print(f"tri: grouped/k-d order {t_tri:.1f} us  grouped/Morton {t_mort:.1f} us  grouped/identity order {t_ident:.1f} us  flat {t_flat:.1f} us  grouped/random order {t_rand:.1f} us")

# ---- the fused surface scan (NN tiles + tri tiles in one launch) ----
import ctypes
from geometrics_amd import _lib as L, ops
from geometrics_amd.tri_distance import face_order
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


def fused(flags=0, records=True):
    L.check(lib.geom_surface_scan_f32(B, n_gt, gt.data_ptr(), num, pred.data_ptr(), o[0].data_ptr(), o[1].data_ptr(),
                                      o[2].data_ptr(), o[3].data_ptr(), nv, verts.data_ptr(), nf, faces.data_ptr(),
                                      tri_order.data_ptr(), o[4].data_ptr(), o[5].data_ptr(), o[6].data_ptr(), o[7].data_ptr(),
                                      o[8].data_ptr(), o[9].data_ptr(), uu.data_ptr(), vv.data_ptr(), 1.0, 1.0,
                                      order.data_ptr() if records else None, flags, ws.data_ptr(), ws_bytes,
                                      ctypes.byref(wrote), None, None, L.stream_ptr()), "scan")


print(f"fused scan (prep + NN/tri launch): {timeit(fused):.1f} us   without records {timeit(lambda: fused(0, False)):.1f} us   "
      f"FMA NN {timeit(lambda: fused(8)):.1f} us")
