"""Probe: the captured driver step with the poles' tail rows switched off (WRONG results at the two poles -- timing only): what the
two 33-neighbour vertices cost the chain launches, whose neighbourhoods advance at the poles' pace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from geometrics_amd import deform, gemm_tuning
dev = torch.device("cuda:0")
gemm_tuning.enable()
bench.settle_clocks(dev, 250)
real = deform._tail_tables
for tail in (True, False, True, False):
    deform._tail_tables = real if tail else (lambda csr: (None, None, None, None))
    r = bench.driver_step_times(dev)
    print("tail rows %-5s  ms_per_step %.4f  blocks %s" % (tail, r["ms_per_step"], [v for k, v in r["stages_us"].items() if "block" in k]))
