"""Does the event-record NODE that gdist.capture_with_event puts into a captured graph order work on another stream -- and
only behind its place, not behind the rest of the graph?  Graph: [slow producer writes `a = step`] -> event node ->
[long tail].  Side stream: wait(event); b = a.  Correct: b == step every time AND the copy finishes before the tail does.
GPU box only:  python tools/probe/mid_graph_event.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geometrics_amd import dist as gdist  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
big = torch.randn(4096, 4096, device=dev)
out = torch.empty_like(big)
a = torch.zeros(1 << 20, device=dev)
b = torch.zeros_like(a)
step = torch.zeros((), device=dev)
ready = torch.cuda.Event()
side = torch.cuda.Stream()


def body():
    step.add_(1.0)
    torch.mm(big, big, out=out)                # ~1 ms in front of the producer: a wait that does not wait reads stale data
    a.copy_(step.expand_as(a))
    gdist.mark_event_here()
    for _ in range(4):
        torch.mm(big, big, out=out)            # the tail the side stream's work should overlap with


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = gdist.capture_with_event(body, ready)
bad, lead = 0, []
for it in range(50):
    g.replay()
    done_copy, done_tail = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        side.wait_event(ready)
        b.copy_(a)
        done_copy.record()
    done_tail.record()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    want = float(step)
    if float(b[0]) != want or float(b[-1]) != want:
        bad += 1
    lead.append(done_copy.elapsed_time(done_tail))     # > 0: the copy finished before the graph's tail
print("event-record node inside a replayed graph: %d/50 stale reads; the side stream's copy finished %.3f ms before the "
      "graph's tail (median; the tail alone takes ~4 ms)" % (bad, sorted(lead)[len(lead) // 2]))
sys.exit(1 if bad else 0)
