#!/usr/bin/env python
"""Kernel time of the matrix-core products (csrc/dense_gemm.hip) next to the library's for the same operands:
HIP-graph replay bracketed by HIP events, operands rotated over several buffers so that consecutive launches do not find
their inputs in the infinity cache more than a step would.

    python tools/time_dense.py [rows]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometrics_amd import dense, gemm_tuning  # noqa: E402


def event_time_us(fns, iters=20, warm=3):
    for _ in range(warm):
        for f in fns:
            f()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(iters):
            fns[i % len(fns)]()
    graph.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        torch.cuda.synchronize()
        s.record()
        graph.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / iters)
    return best


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20496
    dev = torch.device("cuda")
    gemm_tuning.enable()
    nbuf = 3
    for cin in (963, 192):
        c = 192
        xs = [torch.randn(rows, cin, device=dev) for _ in range(nbuf)]
        gs = [torch.randn(rows, c, device=dev) for _ in range(nbuf)]
        w = torch.randn(cin, c, device=dev) * 0.1
        bias = torch.randn(c, device=dev)
        outs = [torch.empty(rows, c, device=dev) for _ in range(nbuf)]
        dxs = [torch.empty(rows, cin, device=dev) for _ in range(nbuf)]
        sup = torch.empty(rows, 64, device=dev)
        mask = torch.empty(rows, c // 16, dtype=torch.int16, device=dev)
        ws = dense.weight_workspace(rows, cin, c, dev)
        gw = torch.empty(cin, c, device=dev)
        gb = torch.empty(c, device=dev)
        flop = 2.0 * rows * cin * c
        res = {}
        res["fwd  lib"] = event_time_us([lambda i=i: torch.mm(xs[i], w, out=outs[i]) for i in range(nbuf)])
        res["fwd  mfma"] = event_time_us([lambda i=i: dense.forward(xs[i], w, outs[i]) for i in range(nbuf)])
        res["fwd  mfma split-epilogue"] = event_time_us(
            [lambda i=i: dense.forward_split(xs[i], w, bias, 64, outs[i], sup, mask) for i in range(nbuf)])
        res["dX   lib"] = event_time_us([lambda i=i: torch.mm(gs[i], w.t(), out=dxs[i]) for i in range(nbuf)])
        res["dX   mfma"] = event_time_us([lambda i=i: dense.backward_input(gs[i], w, dxs[i]) for i in range(nbuf)])
        res["dW   lib"] = event_time_us([lambda i=i: torch.mm(xs[i].t(), gs[i], out=gw) for i in range(nbuf)])
        res["dW   mfma partials"] = event_time_us(
            [lambda i=i: dense.backward_weight_partials(xs[i], gs[i], ws, True) for i in range(nbuf)])
        res["dX+dW mfma, one launch (two workgroups per CU)"] = event_time_us(
            [lambda i=i: dense.backward_pair(xs[i], gs[i], w, dxs[i], ws, True) for i in range(nbuf)])
        res["dW   mfma reduce"] = event_time_us([lambda: dense.reduce([(rows, cin, c, ws, gw, gb)])])
        print("rows %d  cin %d  c %d   (%.2f GFLOP per product, fp32 MFMA floor %.1f us)" % (rows, cin, c, flop / 1e9, flop / 157.3e6))
        for k, v in res.items():
            print("   %-48s %7.1f us   %6.1f TFLOP/s" % (k, v, (2 * flop if k.startswith("dX+dW") else flop) / v / 1e6))


if __name__ == "__main__":
    main()
