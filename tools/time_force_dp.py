"""The N > 1 step sequence on ONE GPU (1-rank RCCL group): graph A = [Adam on the bucket the previous step all-reduced,
forward, backward, reduction launch -> bucket], the RCCL all-reduce issued from a side stream, graph B = [first layer's
input gradient] beside it.  Prints ms per step beside the N = 1 single-graph step -- the fixed
cost the data-parallel path adds before any link time.  GPU box only:  python tools/time_force_dp.py"""
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
for force in (False, True):
    wl = bench.Workload(dev, 0, 8, force_dp=force)
    wl.capture()
    t = bench.time_steps(wl.run, 300, 30)
    print("force_dp=%s: %d graph(s), overlap=%s, %.4f ms per step" % (force, len(wl.graphs), bool(wl.dp and wl.graphs[1] is not None and not wl.packed_late),
                                                                      t / 300 * 1e3), flush=True)
torch.distributed.destroy_process_group()
