"""Dev helper: bench.driver_step_times (HIP-graph replay of the reference driver's step) with the fused deformation-block
launches on and off:   python tools/time_driver_step.py [--zero-edit]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geometrics_amd import deform, gemm_tuning
dev = torch.device("cuda:0")
gemm_tuning.enable()
bench.settle_clocks(dev, 250)
for fused in (True, False, True):
    deform.enabled = fused
    r = bench.driver_step_times(dev)
    print("fused hidden layers %-5s  ms_per_step %.4f  %s" % (fused, r["ms_per_step"], json.dumps(r["stages_us"])))
deform.enabled = True
if "--overlap" in sys.argv:
    for ov in (False, True, False, True):
        r = bench.driver_step_times(dev, overlap_losses=ov)
        print("surface losses on a second stream %-5s  ms_per_step %.4f  final loss %.5f" % (ov, r["ms_per_step"], r["final_loss"]))
if "--stacked" in sys.argv:
    for st in (False, True, False, True):
        bench.DRIVER_STEP_STACKED_LOSSES = st
        r = bench.driver_step_times(dev)
        print("the three surface losses in one call %-5s  ms_per_step %.4f  final loss %.5f" % (st, r["ms_per_step"], r["final_loss"]))
if "--zero-edit" in sys.argv:
    print(json.dumps(bench.driver_step_times(dev, zero_edit=True)))
