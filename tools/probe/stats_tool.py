"""Counters and ablation timings of the culled Chamfer scan from the probe build (tools/probe/libcullstats.so =
chamfer_nn.hip with -DNN_CULL_STATS, tools/probe/run_probes.sh).  GPU box only."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, numpy as np
import time_culled_nn as T
from geometrics_amd import _lib, meshgen
L = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libcullstats.so"))
dev = torch.device("cuda:0")
b, n = 8, 3000
gt = torch.from_numpy(meshgen.gt_cloud(b, n)).to(dev)
V, F = meshgen.icosphere(4)
verts = np.repeat(V[None], b, 0).astype(np.float32)
ch, u, v = meshgen.sampling_draws(verts, F, n)
Ft = torch.from_numpy(F).to(dev).long(); vt = torch.from_numpy(verts).to(dev)
ch = torch.from_numpy(ch).to(dev); u = torch.from_numpy(u).to(dev)[..., None]; v = torch.from_numpy(v).to(dev)[..., None]
tri = vt[:, Ft]
pick = torch.gather(tri, 1, ch[:, :, None, None].expand(-1, -1, 3, 3))
sm = ((1 - u) * pick[:, :, 0] + u * (1 - v) * pick[:, :, 1] + u * v * pick[:, :, 2]).contiguous()
KNOB = int(sys.argv[1]) if len(sys.argv) > 1 else None
for how in (("kd",) if KNOB is not None else ("morton", "kd")):
    o1, o2 = T.orders(gt, how), T.orders(sm, how)
    ws = torch.empty(int(_lib.lib().geom_chamfer_nn_culled_workspace_floats(b, n, n)), dtype=torch.float32, device=dev)
    d1 = torch.empty(b, n, device=dev); d2 = torch.empty(b, n, device=dev)
    i1 = torch.empty(b, n, dtype=torch.int32, device=dev); i2 = torch.empty(b, n, dtype=torch.int32, device=dev)
    out = (ctypes.c_ulonglong * 8)()
    L.geom_cull_stats(out, 1)
    vp = ctypes.c_void_p
    L.geom_chamfer_nn_culled_f32.argtypes = [ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint, vp, vp]
    rc = L.geom_chamfer_nn_culled_f32(b, n, gt.data_ptr(), n, sm.data_ptr(), o1.data_ptr(), o2.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), 8 << 16, ws.data_ptr(), None)
    torch.cuda.synchronize()
    L.geom_cull_stats(out, 1)
    def run(fl):
        L.geom_chamfer_nn_culled_f32(b, n, gt.data_ptr(), n, sm.data_ptr(), o1.data_ptr(), o2.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), fl, ws.data_ptr(), None)
    if KNOB is not None:
        for _ in range(100):
            run(KNOB << 16)
        torch.cuda.synchronize()
        continue
    for name, fl in (("full", 0), ("seeds only", 1 << 16), ("+ tests", 2 << 16), ("+ fetches", 4 << 16), ("full, no resolve", 16 << 16), ("seeds only, no resolve", 17 << 16)):
        print("  %-28s %.1f us (prep + scan)" % (name, T.timeit(lambda: run(fl), 200)))
    tiles = out[0]
    print(how, "rc", rc, "tiles", tiles, "admitted by the first test/tile %.1f" % (out[1] / tiles), "evaluated/tile %.1f" % (out[2] / tiles), "of runs", n // 16)
