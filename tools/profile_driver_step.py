"""Dev helper: replay the captured driver step of bench.py (bench.driver_step_times) for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geometrics_amd import gemm_tuning
dev = torch.device("cuda:0")
gemm_tuning.enable()
print(bench.driver_step_times(dev, profile_replays=20))
