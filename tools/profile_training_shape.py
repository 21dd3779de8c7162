"""Dev helper: run the reference-training-shape step of bench.py eagerly (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geometrics_amd import gemm_tuning
dev = torch.device("cuda:0")
gemm_tuning.enable()
print(bench.training_shape_times(dev))
