"""Which engine clock does the chip hold under a given kernel?  Runs ONE kernel back to back for ~1.5 s and samples
`rocm-smi --showclocks` meanwhile (sclk).  python tools/probe/kernel_clock.py {split|pair|libfwd|mfma}
The fp32-MFMA-only loop (tools/probe/mfma_rate.cpp) holds 2.36 GHz = 155 TFLOP/s; a product that also streams its operands
through HBM / L2 / LDS draws more power per MFMA cycle."""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                   # noqa: E402
from geometrics_amd import dense, gemm_tuning  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "split"
dev = torch.device("cuda:0")
gemm_tuning.enable()
rows = 20496
x1, xh = torch.randn(rows, 963, device=dev), torch.randn(rows, 192, device=dev)
g = torch.randn(rows, 192, device=dev)
w1, wh = torch.randn(963, 192, device=dev) * 0.1, torch.randn(192, 192, device=dev) * 0.1
ws1, wsh = dense.weight_workspace(rows, 963, 192, dev), dense.weight_workspace(rows, 192, 192, dev)
dxh, out = torch.empty(rows, 192, device=dev), torch.empty(rows, 192, device=dev)
fn = {"split": lambda: dense.backward_weight_partials(x1, g, ws1), "pair": lambda: dense.backward_pair(xh, g, wh, dxh, wsh),
      "libfwd": lambda: torch.mm(x1, w1, out=out)}[which]
flop = {"split": 2.0 * rows * 963 * 192, "pair": 4.0 * rows * 192 * 192, "libfwd": 2.0 * rows * 963 * 192}[which]
for _ in range(10):
    fn()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for _ in range(100):
        fn()
samples, stop = [], False


def sample():
    while not stop:
        out_ = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout
        for line in out_.splitlines():
            if "sclk" in line:
                samples.append(line.split("(")[-1].rstrip(")"))
                break
        time.sleep(0.05)


th = threading.Thread(target=sample)
th.start()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
time.sleep(0.3)
s.record()
n = 0
t_end = time.perf_counter() + 1.5
while time.perf_counter() < t_end:
    graph.replay()
    n += 100
e.record()
torch.cuda.synchronize()
stop = True
th.join()
us = s.elapsed_time(e) * 1e3 / n
print("%s: %d launches back to back, %.1f us each = %.1f TFLOP/s; sclk samples: %s" % (which, n, us, flop / (us * 1e-6) / 1e12, " ".join(samples)))
