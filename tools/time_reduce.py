"""dense_reduce_kernel in isolation (HIP-event timing of a captured graph of 30 launches): which jobs cost what."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402
import bench                                   # noqa: E402
from geometrics_amd import dense               # noqa: E402

dev = torch.device("cuda:0")
rows = 8 * 2562
shapes = [(rows, 963, 192), (rows, 192, 192), (rows, 192, 192)]
jobs = []
for r, cin, c in shapes:
    ws = dense.weight_workspace(r, cin, c, dev)
    ws.normal_()
    jobs.append((r, cin, c, ws, torch.empty(cin, c, device=dev), None))
print("workspace floats", [j[3].numel() for j in jobs])
for name, sel in (("963x192 only", jobs[:1]), ("one 192x192", jobs[1:2]), ("all three weights", jobs)):
    print("%-20s %.1f us" % (name, bench.event_time_us(lambda: dense.reduce(sel))))
withb = [(r, cin, c, ws, gw, torch.empty(c, device=dev)) for (r, cin, c, ws, gw, _) in jobs]
print("%-20s %.1f us" % ("+ bias column sums", bench.event_time_us(lambda: dense.reduce(withb))))
x = torch.empty(18_500_000 // 4, device=dev)
y = torch.empty_like(x)
print("%-20s %.1f us" % ("copy of 18.5 MB", bench.event_time_us(lambda: y.copy_(x))))
