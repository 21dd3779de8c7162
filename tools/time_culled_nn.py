"""Culled Chamfer scan (geom_chamfer_nn_culled_f32) against the brute-force scan on the BASELINE clouds: bitwise equality
of (dist, idx) in both directions and both arithmetics, then the launch times.  GPU box only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometrics_amd import _lib, meshgen                       # noqa: E402
from geometrics_amd.chamfer_distance import chamfer_nn         # noqa: E402
from geometrics_amd.tri_distance import morton_order, kd_order  # noqa: E402


def culled(x1, x2, o1, o2, flags=0, ws=None):
    b, n, _ = x1.shape
    m = x2.shape[1]
    L = _lib.lib()
    if ws is None:
        ws = torch.empty(int(L.geom_chamfer_nn_culled_workspace_floats(b, n, m)), dtype=torch.float32, device=x1.device)
    d1 = torch.empty(b, n, dtype=torch.float32, device=x1.device)
    d2 = torch.empty(b, m, dtype=torch.float32, device=x1.device)
    i1 = torch.empty(b, n, dtype=torch.int32, device=x1.device)
    i2 = torch.empty(b, m, dtype=torch.int32, device=x1.device)
    _lib.check(L.geom_chamfer_nn_culled_f32(b, n, x1.data_ptr(), m, x2.data_ptr(), _lib.ptr(o1), _lib.ptr(o2), d1.data_ptr(),
                                            i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), flags, ws.data_ptr(), _lib.stream_ptr()),
               "geom_chamfer_nn_culled_f32")
    return d1, i1, d2, i2


def orders(x, how):
    f = morton_order if how == "morton" else kd_order
    return torch.stack([f(x[i]) for i in range(x.shape[0])]).contiguous()


def same(a, b):
    return all(torch.equal(p.view(torch.int32) if p.dtype == torch.float32 else p, q.view(torch.int32) if q.dtype == torch.float32 else q)
               for p, q in zip(a, b))


def timeit(f, reps=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


def main():
    dev = torch.device("cuda:0")
    b, n = 8, 3000
    gt = torch.from_numpy(meshgen.gt_cloud(b, n)).to(dev)
    V, F = meshgen.icosphere(4)
    verts = np.repeat(V[None], b, 0).astype(np.float32)
    ch, u, v = meshgen.sampling_draws(verts, F, n)
    Ft = torch.from_numpy(F).to(dev).long()
    vt = torch.from_numpy(verts).to(dev)
    ch = torch.from_numpy(ch).to(dev)
    u = torch.from_numpy(u).to(dev)[..., None]
    v = torch.from_numpy(v).to(dev)[..., None]
    tri = vt[:, Ft]                                                 # [b, F, 3, 3]
    pick = torch.gather(tri, 1, ch[:, :, None, None].expand(-1, -1, 3, 3))
    sm = ((1 - u) * pick[:, :, 0] + u * (1 - v) * pick[:, :, 1] + u * v * pick[:, :, 2]).contiguous()
    for how in ("morton", "kd"):
        o1, o2 = orders(gt, how), orders(sm, how)
        for flags in (0, _lib.FLAG_NN_FMA):
            ref = chamfer_nn(gt, sm, flags)
            got = culled(gt, sm, o1, o2, flags)
            print(how, "flags", flags, "identical" if same(ref, got) else "DIFFERENT", flush=True)
            if not same(ref, got):
                for k, (p, q) in enumerate(zip(ref, got)):
                    print("  out", k, "mismatches", int((p != q).sum()))
        ws = torch.empty(int(_lib.lib().geom_chamfer_nn_culled_workspace_floats(b, n, n)), dtype=torch.float32, device=dev)
        print(how, "brute %.1f us   culled (prep + scan) %.1f us" % (timeit(lambda: chamfer_nn(gt, sm, 0)),
                                                                    timeit(lambda: culled(gt, sm, o1, o2, 0, ws))), flush=True)
    # adversarial: identity order (no locality), ties (duplicated points), NaN / inf members, ragged sizes
    g = torch.Generator(device="cpu").manual_seed(5)
    for (bn, nn_, mm) in ((2, 1000, 777), (1, 64, 15), (3, 17, 260), (1, 16, 16)):
        a = torch.randn(bn, nn_, 3, generator=g).to(dev)
        c = torch.randn(bn, mm, 3, generator=g).to(dev)
        c[:, mm // 2:] = c[:, :mm - mm // 2].clone()                         # exact duplicates: ties across runs
        a[:, ::7] = c[:, :1]                                         # zero distances
        if nn_ > 20:
            a[0, 3, 1] = float("nan")
            c[0, 5, 0] = float("inf")
            c[-1, 9, 2] = float("nan")
        for o in ("identity", "morton", "reverse"):
            if o == "identity":
                o1 = o2 = None
            elif o == "morton":
                o1, o2 = orders(a, "morton"), orders(c, "morton")
            else:
                o1 = torch.arange(nn_ - 1, -1, -1, dtype=torch.int32, device=dev).repeat(bn, 1).contiguous()
                o2 = torch.arange(mm - 1, -1, -1, dtype=torch.int32, device=dev).repeat(bn, 1).contiguous()
            for flags in (0, _lib.FLAG_NN_FMA):
                ref = chamfer_nn(a, c, flags)
                got = culled(a, c, o1, o2, flags)
                ok = all(torch.equal(p.view(torch.int32), q.view(torch.int32)) for p, q in zip(ref, got))
                print((bn, nn_, mm), o, flags, "identical" if ok else "DIFFERENT", flush=True)
                if not ok:
                    for k, (p, q) in enumerate(zip(ref, got)):
                        bad = (p.view(torch.int32) != q.view(torch.int32)).nonzero()
                        print("  out", k, "mismatches", bad.shape[0], bad[:4].tolist(),
                              [(p[tuple(i)].item(), q[tuple(i)].item()) for i in bad[:4]])


if __name__ == "__main__":
    main()
