"""When does each tile of the fused surface-scan launch start and end, and where?  Runs the step's own scan call (triangle
tiles + Chamfer tiles, 8-mesh BASELINE shard) on the instrumented build (-DSCAN_TILE_STAMPS: every workgroup stamps the
100 MHz wall clock at entry and exit) and prints: tile durations per kind, how many tiles run at once, how long the launch's
slots sit idle at its end.      bash tools/probe/run_probes.sh   (build container)
    GEOM_LIB_OVERRIDE=tools/probe/bin/libgeom_scan_stamps.so GEOM_ALLOW_STALE_LIB=1 python tools/probe/scan_tile_stamps.py [--plain]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np                             # noqa: E402
import torch                                   # noqa: E402
import bench                                   # noqa: E402
from geometrics_amd import _lib, ops           # noqa: E402

plain = "--plain" in sys.argv
tail = "--tail" in sys.argv          # the finalize pass inside the launch (its role workgroups lead the grid)
bench.CULLED_CHAMFER = not plain
dev = torch.device("cuda:0")
w = bench.Workload(dev, 0, 8)
L = _lib.lib()
L.geom_probe_read_scan_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
with torch.no_grad():
    pos = w.positions().contiguous()
    for rep in range(4):                       # the last repetition is read: warm caches and clocks
        prep = ops.draw_samples(pos, w.faces, bench.S_PTS, with_points=True, prepare_scan_for=bench.G_PTS, gt_index=w.gt_index)
        if plain:
            call = bench.fused_scan_call(w, pos, prep[3])
            call()
            call = bench.fused_scan_call(w, pos, prep[3], _lib.FLAG_TRI_WS_READY, share=call)
        else:
            call = bench.fused_scan_call(w, pos, prep[3], prep=prep, tail=tail)
        torch.cuda.synchronize()
        call()
        torch.cuda.synchronize()
tri_tiles = 8 * ((bench.G_PTS + 63) // 64)
nn_tiles = 2 * 8 * ((max(bench.G_PTS, bench.S_PTS) + 63) // 64)
rows = tri_tiles + nn_tiles + (16 if tail else 0)
buf = np.zeros((rows, 4), dtype=np.int64)
_lib.check(L.geom_probe_read_scan_stamps(buf.ctypes.data, rows), "geom_probe_read_scan_stamps")
buf = buf[buf[:, 1] > 0]
t0 = buf[:, 0].min()
start, end, kind = (buf[:, 0] - t0) / 100.0, (buf[:, 1] - t0) / 100.0, buf[:, 2]          # microseconds
print("# fused surface scan, %s Chamfer tiles, 8-mesh shard: %d workgroups stamped, launch span %.1f us (first entry to last exit)"
      % ("brute-force" if plain else "culled", len(buf), end.max()))
for k, name in ((0, "triangle tiles"), (1, "Chamfer tiles")):
    d = (end - start)[kind == k]
    s = start[kind == k]
    print("%-15s n %4d   duration us: median %5.1f  p10 %5.1f  p90 %5.1f  max %5.1f   start us: median %5.1f  p90 %5.1f  last %5.1f   last exit %5.1f"
          % (name, len(d), np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max(), np.median(s), np.percentile(s, 90), s.max(),
             end[kind == k].max()))
if tail:
    r = buf[kind == 2]
    for row in r[np.argsort(r[:, 0])]:
        print("role workgroup: entry %5.1f  wait over %5.1f  exit %5.1f us" % ((row[0] - t0) / 100.0, (row[3] - t0) / 100.0 if row[3] > 0 else -1, (row[1] - t0) / 100.0))
    L.geom_probe_read_role_phases.argtypes = [ctypes.c_void_p]
    rp = np.zeros((16, 16), dtype=np.int64)
    _lib.check(L.geom_probe_read_role_phases(rp.ctypes.data), "geom_probe_read_role_phases")
    print("# ordering roles, thread 0's clock (us from the launch's first entry): entry, LDS zeroed, samples binned, wait over, gt points "
          "binned, offsets scanned, ids dropped, placed")
    for row in rp:
        if row[7] > 0:
            print("   " + " ".join("%6.1f" % ((x - t0) / 100.0) for x in row[:8]))
grid = np.arange(0.0, end.max(), 2.0)
print("# tiles in flight (triangle / Chamfer) every 2 us:")
print(" ".join("%d/%d" % (((start <= t) & (end > t) & (kind == 0)).sum(), ((start <= t) & (end > t) & (kind == 1)).sum()) for t in grid))
busy = (end - start).sum()
print("# slot-time used %.0f us over %d workgroups = %.1f us per slot if 768 slots (3 per CU) were packed perfectly; span %.1f us"
      % (busy, len(buf), busy / 768.0, end.max()))

if not plain:
    L.geom_probe_read_tri_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
    ph = np.zeros((min(tri_tiles, 1024), 16), dtype=np.int64)
    _lib.check(L.geom_probe_read_tri_phases(ph.ctypes.data, len(ph)), "geom_probe_read_tri_phases")
    ph = ph[ph[:, 6] > 0]
    names = ["staging of the group spheres", "A1 (smallest upper bound per query)", "A1' (16 literal evaluations of the seed group)",
             "A2 + B + C until ALL waves are through", "final drain of queue B", "epilogue (outputs, point-to-surface records)"]
    print("# phases of a triangle tile (wave 0's clock, us; %d tiles): median  p10  p90" % len(ph))
    for k, name in enumerate(names):
        d = (ph[:, k + 1] - ph[:, k]) / 100.0
        print("%-52s %6.2f %6.2f %6.2f" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    done = (ph[:, 8:16] - ph[:, 3:4]) / 100.0                  # every wave's A2 + B + C time from the end of A1'
    print("%-52s %6.2f %6.2f %6.2f" % ("  A2 + B + C of the FASTEST wave of a tile", np.median(done.min(1)), np.percentile(done.min(1), 10), np.percentile(done.min(1), 90)))
    print("%-52s %6.2f %6.2f %6.2f" % ("  A2 + B + C of the SLOWEST wave of a tile", np.median(done.max(1)), np.percentile(done.max(1), 10), np.percentile(done.max(1), 90)))
    print("%-52s %6.2f %6.2f %6.2f" % ("  mean wave of a tile", np.median(done.mean(1)), np.percentile(done.mean(1), 10), np.percentile(done.mean(1), 90)))
