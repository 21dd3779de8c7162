"""Where do the ~35 us go that the two-graph data-parallel step costs over the single-graph step?  Variants of the per-step
host sequence on the SAME two captured graphs (A = Adam + forward + backward + reduction, B = the postponed product), timed
on one GPU with a 1-rank RCCL group.  Timing experiment only (the reduced variants are not correct steps).
GPU box only:  python tools/probe/dp_gaps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch                                   # noqa: E402
import bench                                   # noqa: E402
from geometrics_amd import gemm_tuning         # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
gemm_tuning.enable()
wl = bench.Workload(dev, 0, 8, force_dp=True)
wl.capture()
ga, gb = wl.graphs
main = torch.cuda.current_stream()
side, ev = torch.cuda.Stream(), torch.cuda.Event()


def v_graphs_only():
    ga.replay(); gb.replay()

def v_event():
    ga.replay(); ev.record(); gb.replay()

def v_event_sidewait():
    ga.replay(); ev.record()
    side.wait_event(ev)
    gb.replay()

def v_collective_no_join():
    ga.replay(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        wl.bucket.all_reduce()
    gb.replay()

def v_full():
    ga.replay(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        wl.bucket.all_reduce()
    gb.replay()
    main.wait_stream(side)

def v_collective_on_main():
    ga.replay(); wl.bucket.all_reduce(); gb.replay()

def v_join_only():
    ga.replay(); ev.record()
    side.wait_event(ev)
    gb.replay()
    main.wait_stream(side)

def v_b_first():
    ga.replay(); ev.record(); gb.replay()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        wl.bucket.all_reduce()
    main.wait_stream(side)

def v_async():
    ga.replay()
    work = wl.bucket.all_reduce_async()
    gb.replay()
    work.wait()

prio = os.environ.get("DP_GAPS_PRIORITY")
if prio is not None:      # the launch stream with a priority of its own (-1 = high): does the command processor favour it?
    torch.cuda.set_stream(torch.cuda.Stream(priority=int(prio)))
    main = torch.cuda.current_stream()
    print("launch stream priority", prio)

for name, fn in (("A ; B", v_graphs_only), ("A ; record ; B ; side: wait + all-reduce ; main waits side (B queued BEFORE the host-side collective call)", v_b_first), ("shipped: A ; async all-reduce ; B ; launch stream waits for it", v_async), ("A ; record ; B", v_event), ("A ; record ; side waits ; B", v_event_sidewait),
                 ("A ; record ; side: wait + all-reduce ; B", v_collective_no_join),
                 ("A ; record ; side waits ; B ; main waits side", v_join_only),
                 ("A ; all-reduce on the launch stream ; B", v_collective_on_main),
                 ("full: A ; record ; side: wait + all-reduce ; B ; main waits side", v_full)):
    t = bench.time_steps(fn, 300, 30)
    print("%-70s %.4f ms per step" % (name, t / 300 * 1e3), flush=True)
torch.distributed.destroy_process_group()
