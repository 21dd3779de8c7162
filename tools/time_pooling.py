"""Dev helper: fused batched_pooling vs the reference's formulation (same eager torch ops), 8 meshes x 2562
vertices, VGG map sizes (64x56^2, 128x28^2, 256x14^2, 512x7^2), forward + backward, kernel time."""
import sys
import torch
sys.path.insert(0, ".")
from geometrics_amd import meshgen, utils
dev = torch.device("cuda:0")
V, _ = meshgen.icosphere(4)
B = 8
verts = torch.from_numpy(meshgen.jittered_batch(V, B)).to(dev).requires_grad_(True)
info = torch.tensor([[30.0 * i, 25.0, 1.1] for i in range(B)], device=dev)
blocks = [torch.randn(B, c, d, d, device=dev, requires_grad=True) for c, d in ((64, 56), (128, 28), (256, 14), (512, 7))]


def reference_style(blocks, verts_pos, img_info):
    cam_mat, cam_pos = utils.batch_camera_info(img_info)
    pt = torch.matmul((verts_pos * .57) - cam_pos.unsqueeze(1), cam_mat.permute(0, 2, 1))
    X, Y, Z = pt[:, :, 0], pt[:, :, 1], pt[:, :, 2]
    xs = ((-Y) / (-Z) * 248 + 112.) / 223.
    ys = (X / (-Z) * 248 + 112.) / 223.
    full, bs = None, verts_pos.shape[0]
    for block in blocks:
        dim = block.shape[-1]
        cx, cy = torch.clamp(xs * dim, 0, dim - 1), torch.clamp(ys * dim, 0, dim - 1)
        x1, y1, x2, y2 = torch.floor(cx), torch.floor(cy), torch.ceil(cx), torch.ceil(cy)
        A, Bw, G, H = x2 - cx, cx - x1, y2 - cy, cy - y1
        x1, y1, x2, y2 = x1.long(), y1.long(), x2.long(), y2.long()
        flat = block.permute(1, 0, 2, 3).contiguous().view(block.shape[1], -1)
        up = torch.arange(0, bs, device=dev).unsqueeze(-1).expand(bs, verts_pos.shape[1])
        sel = lambda xx, yy: torch.index_select(flat, 1, ((up * dim * dim) + (xx * dim) + yy).view(-1)).view(-1, bs, verts_pos.shape[1]).permute(1, 0, 2)
        f = (A.unsqueeze(1) * sel(x1, y1) * G.unsqueeze(1) + H.unsqueeze(1) * sel(x1, y2) * A.unsqueeze(1)
             + G.unsqueeze(1) * sel(x2, y1) * Bw.unsqueeze(1) + Bw.unsqueeze(1) * sel(x2, y2) * H.unsqueeze(1)).permute(0, 2, 1)
        full = f if full is None else torch.cat((full, f), dim=2)
    return full


def t(fn, it=10):
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


def fb(fn):
    def run():
        verts.grad = None
        for b in blocks:
            b.grad = None
        fn(blocks, verts, info).sum().backward()
    return run


with torch.no_grad():
    a, b_ = utils.batched_pooling(blocks, verts, info), reference_style(blocks, verts, info)
    print("max abs diff vs reference formulation:", float((a - b_).abs().max()))
x, y = t(fb(utils.batched_pooling)), t(fb(reference_style))
print("batched_pooling fwd+bwd: fused kernels %.0f us   reference formulation %.0f us   x%.1f" % (x, y, y / x))


def fwd_only():
    with torch.no_grad():
        utils.batched_pooling(blocks, verts, info)


verts_ng = verts.detach()
blocks_ng = [b.detach() for b in blocks]


def verts_only():
    verts.grad = None
    utils.batched_pooling(blocks_ng, verts, info).sum().backward()


def maps_only():
    for b in blocks:
        b.grad = None
    utils.batched_pooling(blocks, verts_ng, info).sum().backward()


f = t(fwd_only)
print("forward %.0f us   fwd+verts-bwd %.0f us   fwd+maps-bwd %.0f us" % (f, t(verts_only), t(maps_only)))
