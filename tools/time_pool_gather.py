"""Dev helper: the map gradient of batched_pooling alone (the binning + gather roles of the two backward launches through the C-ABI, no vertex
gradient) at the driver step's shape -- 16 meshes x 482 vertices (meshgen.uv_sphere under the bench's cameras), the four VGG
maps, the gradient read out of a 1155-wide buffer -- and how the texel lists are distributed (entries per texel, per level)."""
import ctypes
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from geometrics_amd import _lib, meshgen, utils

dev = torch.device("cuda:0")
B = 16
V, _ = meshgen.uv_sphere()
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
verts = torch.from_numpy(np.ascontiguousarray(V * scale)).to(dev).unsqueeze(0).expand(B, -1, -1).contiguous()
nv = verts.shape[1]
img_info = torch.tensor([[30.0 + 10 * i, 25.0, 1.1] for i in range(B)], device=dev)
cam_mat, cam_pos = utils.batch_camera_info(img_info)
shapes = ((64, 56), (128, 28), (256, 14), (512, 7))
maps = [torch.randn(B, c, d, d, device=dev) for c, d in shapes]
n = len(maps)
ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in maps])
chans = (ctypes.c_int * n)(*[c for c, _ in shapes])
dims = (ctypes.c_int * n)(*[d for _, d in shapes])
ctot = sum(c for c, _ in shapes)
LD = 195 + ctot
gbuf = torch.randn(B, nv, LD, device=dev)
g = gbuf[..., 195:]
gmaps = [torch.empty_like(t) for t in maps]
gptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in gmaps])
ws_bytes = _lib.lib().geom_pool_features_bwd_workspace_bytes(B, nv, n, dims)
ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)


def run():
    _lib.call("geom_pool_features_bwd_ld_f32", B, nv, verts.data_ptr(), cam_mat.data_ptr(), cam_pos.data_ptr(), n, ptrs, chans, dims,
              g.data_ptr(), LD, gptrs, None, ws.data_ptr(), ws_bytes)


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(it):
            fn()
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / it * 1e3)
    return best


print("bin + gather: %.1f us per call (16 x %d vertices, gradient pitch %d)" % (t(run), nv, LD))

out = torch.empty(B, nv, LD, device=dev)
gverts = torch.empty_like(verts)
no_maps = (ctypes.c_void_p * n)(*[None] * n)


def run_fwd():
    _lib.call("geom_pool_features_fwd_ld_f32", B, nv, verts.data_ptr(), cam_mat.data_ptr(), cam_pos.data_ptr(), n, ptrs, chans, dims,
              out[..., 195:].data_ptr(), LD)


def run_verts():
    _lib.call("geom_pool_features_bwd_ld_f32", B, nv, verts.data_ptr(), cam_mat.data_ptr(), cam_pos.data_ptr(), n, ptrs, chans, dims,
              g.data_ptr(), LD, no_maps, gverts.data_ptr(), ws.data_ptr(), ws_bytes)


print("forward: %.1f us per call" % t(run_fwd))
print("vertex gradient (two launches): %.1f us per call" % t(run_verts))
# list lengths: pooling identity maps gives P; nonzeros per texel column
for c, d in shapes:
    eye = torch.eye(d * d, device=dev).view(1, d * d, d, d).expand(B, -1, -1, -1).contiguous()
    P = utils.batched_pooling([eye], verts, img_info.clone())
    cnt = (P != 0).sum(1).float()            # [B, texels]
    print("  %2d x %2d: entries per texel mean %.2f  max %d  empty texels %.0f %%  entries in the 10 %% fullest texels %.0f %%" % (
        d, d, float(cnt.mean()), int(cnt.max()), 100 * float((cnt == 0).float().mean()),
        100 * float(cnt.sort(dim=1, descending=True)[0][:, :max(1, d * d // 10)].sum() / cnt.sum())))
