"""Does an EXTERNAL event recorded by a node INSIDE a replayed HIP graph order work on another stream?  (What the
data-parallel step relies on: the gradient all-reduce waits for the end-of-pass reduction launch in the middle of the
captured step while the rest of the graph -- the first layer's input gradient -- keeps running.)
Graph: [slow producer writes `a = step`] -> external record -> [long tail kernel].  Side stream: wait(event); b = a.
Correct ordering: b == step every time, and the copy finishes BEFORE the tail does (overlap).  GPU box only."""
import sys
import time

import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geometrics_amd.dist import GraphEvent  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
big = torch.randn(4096, 4096, device=dev)
out = torch.empty_like(big)
a = torch.zeros(1 << 20, device=dev)
b = torch.zeros_like(a)
step = torch.zeros((), device=dev)
ready = GraphEvent(dev)
side = torch.cuda.Stream()

def body():
    step.add_(1.0)
    torch.mm(big, big, out=out)                # ~1 ms in front of the producer: a wait that does not wait reads stale data
    a.copy_(step.expand_as(a))
    ready.record()
    for _ in range(4):
        torch.mm(big, big, out=out)            # the tail the side stream's work should overlap with

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
torch.cuda.synchronize()
bad = 0
overlap = []
for it in range(50):
    g.replay()
    done_copy, done_tail = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        ready.wait(side)
        b.copy_(a)
        done_copy.record()
    done_tail.record()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    want = float(step)
    if float(b[0]) != want or float(b[-1]) != want:
        bad += 1
    overlap.append(done_copy.elapsed_time(done_tail))     # > 0: the copy finished before the tail
print("external event inside a replayed graph: %d/50 stale reads; copy finished %.3f ms before the graph's tail (median)"
      % (bad, sorted(overlap)[len(overlap) // 2]))
sys.exit(1 if bad else 0)
