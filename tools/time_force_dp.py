"""The N > 1 step sequences on ONE GPU (1-rank RCCL group) beside the N = 1 single-graph step: "two_graphs" (graph A, the
all-reduce issued by the host, graph B beside it) and "captured" (one graph per step with the collective inside it) -- the
fixed cost the data-parallel path adds before any link time (profiles/r05_dp_fixed_cost.txt).
GPU box only:  python tools/time_force_dp.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch                                   # noqa: E402
import bench                                   # noqa: E402
from geometrics_amd import gemm_tuning         # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
gemm_tuning.enable()
for rep in range(2):
    for force, seq in ((False, None), (True, "two_graphs"), (True, "captured")):
        wl = bench.Workload(dev, 0, 8, force_dp=force, dp_sequence=seq)
        try:
            wl.capture()
        except Exception as exc:
            print("force_dp=%s %s: capture failed: %s: %s" % (force, seq, type(exc).__name__, str(exc)[:200]), flush=True)
            from geometrics_amd import _lib
            _lib.clear_hip_error()
            continue
        t = bench.time_steps(wl.run, 300, 30)
        wl.finish()
        print("force_dp=%-5s %-10s: %d graph(s) per step, %.4f ms per step, loss %.6f" % (force, seq or "-", len(wl.graphs), t / 300 * 1e3, wl.mean_loss()),
              flush=True)
        del wl
torch.distributed.destroy_process_group()
