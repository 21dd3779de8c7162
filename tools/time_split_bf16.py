"""Dev helper: the split-bf16 product (EXPERIMENT) against the library's and the package's fp32 products for the first layer's
forward shape, HIP-graph replay with operands rotated over three buffers (as tools/time_dense.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometrics_amd import dense, gemm_tuning
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from time_dense import event_time_us

dev = torch.device("cuda")
gemm_tuning.enable()
for rows, k in ((20496, 963), (7712, 1155), (20496, 192)):
    xs = [torch.randn(rows, k, device=dev) for _ in range(3)]
    w = torch.randn(k, 192, device=dev) * 0.05
    planes = dense.split_bf16_planes(w)
    outs = [torch.empty(rows, 192, device=dev) for _ in range(3)]
    fl = 2.0 * rows * k * 192
    t = {}
    t["library (tuned selection)"] = event_time_us([lambda i=i: torch.mm(xs[i], w, out=outs[i]) for i in range(3)])
    if rows >= 512:
        t["fp32 matrix cores (dense_gemm.hip)"] = event_time_us([lambda i=i: dense.forward(xs[i], w, out=outs[i]) for i in range(3)])
    t["split bf16, 6 terms"] = event_time_us([lambda i=i: dense.gemm_split_bf16(xs[i], planes, 6, out=outs[i]) for i in range(3)])
    t["split bf16, 9 terms"] = event_time_us([lambda i=i: dense.gemm_split_bf16(xs[i], planes, 9, out=outs[i]) for i in range(3)])
    t["splitting the weight (once per update)"] = event_time_us([lambda: dense.split_bf16_planes(w)])
    print("rows %d  k %d  (%.2f GFLOP; fp32 MFMA floor %.1f us)" % (rows, k, fl / 1e9, fl / 157.3e6))
    for name, us in t.items():
        print("   %-42s %7.1f us   %6.1f TFLOP/s (fp32-equivalent)" % (name, us, fl / us / 1e6))
