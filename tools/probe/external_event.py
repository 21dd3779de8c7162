"""Finding kept reproducible: this stack has no EXTERNAL events for stream capture.  A data-parallel step would like ONE
graph per step with an event-record node right behind the end-of-pass reduction launch (the gradient all-reduce waits for
that node while the rest of the graph keeps running).  torch.cuda.Event(external=True) raises "External events are
disallowed in rocm", and the raw call -- hipEventRecordWithFlags(event, capturing stream, hipEventRecordExternal) --
returns hipErrorInvalidValue during a capture with the HIP runtime PyTorch 2.10+rocm7.0 bundles.  Hence the two graphs of
bench.Workload.capture() (the pass is cut behind the reduction launch).  GPU box only:  python tools/probe/external_event.py"""
import ctypes

import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
try:
    torch.cuda.Event(external=True).record()
    print("torch.cuda.Event(external=True).record(): accepted")
except RuntimeError as exc:
    print("torch.cuda.Event(external=True).record(): %s" % exc)
hip = ctypes.CDLL("libamdhip64.so")
event = ctypes.c_void_p()
assert hip.hipEventCreateWithFlags(ctypes.byref(event), 2) == 0           # hipEventDisableTiming
x = torch.zeros(16, device=dev)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    g.capture_begin()
    x.add_(1.0)
    code = hip.hipEventRecordWithFlags(event, ctypes.c_void_p(s.cuda_stream), 1)    # hipEventRecordExternal
    x.add_(1.0)
    try:
        g.capture_end()
        ended = "capture ended normally"
    except RuntimeError as exc:
        ended = "capture_end: %s" % str(exc).splitlines()[0]
print("hipEventRecordWithFlags(..., hipEventRecordExternal) on a capturing stream -> %d; %s" % (code, ended))
