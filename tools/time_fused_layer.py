#!/usr/bin/env python
"""Kernel time of a fused layer-boundary launch (csrc/zn_stack.hip) next to the two launches it replaces (aggregation +
library product), forward and backward, HIP-graph replay bracketed by HIP events, operands rotated over buffers.

    python tools/time_fused_layer.py [meshes]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometrics_amd import dense, fused, gemm_tuning, layers, meshgen, utils  # noqa: E402
from tools.time_dense import event_time_us  # noqa: E402


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda")
    gemm_tuning.enable()
    V, Fc = meshgen.icosphere(4)
    nv = V.shape[0]
    info = utils.adj_init(torch.from_numpy(np.ascontiguousarray(Fc)).to(dev))
    csr = layers.adjacency_csr(info["adj"])
    K, C = 64, 192
    nbuf = 3
    sp = [torch.randn(b, nv, C, device=dev) for _ in range(nbuf)]
    xs = [torch.empty(b, nv, C, device=dev) for _ in range(nbuf)]
    ss = [torch.empty(b, nv, C, device=dev) for _ in range(nbuf)]
    bias = torch.randn(C, device=dev) * 0.1
    w = torch.randn(C, C, device=dev) * 0.1
    wt = w.t().contiguous()
    masks = [torch.zeros(b * nv * 16, dtype=torch.int16, device=dev) for _ in range(nbuf)]
    res = {}

    def sep_fwd(i):
        layers.aggregate_forward(sp[i], bias, csr, K, 1, xs[i], want_mask=False)
        torch.mm(xs[i].view(-1, C), w, out=ss[i].view(-1, C))
    res["fwd: aggregation + library product"] = event_time_us([lambda i=i: sep_fwd(i) for i in range(nbuf)])
    res["fwd: aggregation alone"] = event_time_us(
        [lambda i=i: layers.aggregate_forward(sp[i], bias, csr, K, 1, xs[i], want_mask=False) for i in range(nbuf)])
    res["fwd: library product alone"] = event_time_us([lambda i=i: torch.mm(xs[i].view(-1, C), w, out=ss[i].view(-1, C)) for i in range(nbuf)])
    res["fwd: fused boundary launch"] = event_time_us(
        [lambda i=i: fused.layer_forward(sp[i], bias, csr, K, 1, w, x_out=xs[i], mask=masks[i], s_out=ss[i]) for i in range(nbuf)])
    # backward
    for i in range(nbuf):
        layers.aggregate_forward(sp[i], bias, csr, K, 1, xs[i], want_mask=False)
    mask = layers.aggregate_forward(sp[0], bias, csr, K, 1, xs[0], want_mask=True)
    go = [torch.randn(b, nv, C, device=dev) for _ in range(nbuf)]
    gs = [torch.empty(b, nv, C, device=dev) for _ in range(nbuf)]
    gi = [torch.empty(b, nv, C, device=dev) for _ in range(nbuf)]
    zero_bias = torch.zeros(C, device=dev)
    part = torch.empty(fused.partial_rows(b, nv), C, device=dev)
    ws = dense.weight_workspace(b * nv, C, C, dev)

    def sep_bwd(i):
        g, _ = layers.aggregate_backward(go[i], csr, K, 1, None, mask, True, bias=zero_bias)
        dense.backward_pair(xs[i].view(-1, C), g.view(-1, C), w, gi[i].view(-1, C), ws)
    res["bwd: aggregation backward + pair launch (dX + dW partials)"] = event_time_us([lambda i=i: sep_bwd(i) for i in range(nbuf)])
    res["bwd: fused boundary launch (G, dX, bias partials)"] = event_time_us(
        [lambda i=i: fused.layer_backward(go[i], None, mask, csr, K, 1, wt, g_out=gs[i], grad_in=gi[i], colsum_partial=part)
         for i in range(nbuf)])
    res["bwd: weight-gradient partials alone"] = event_time_us(
        [lambda i=i: dense.backward_weight_partials(xs[i].view(-1, C), gs[i].view(-1, C), ws) for i in range(nbuf)])
    gp = torch.randn(b, nv, 3, device=dev)
    res["bwd: fused boundary launch, head mode"] = event_time_us(
        [lambda i=i: fused.layer_backward(None, None, mask, csr, K, 1, wt, g_out=gs[i], grad_in=gi[i], colsum_partial=part,
                                          grad_pos=gp, head_scale=0.01, shape=(b, nv, C)) for i in range(nbuf)])
    flop = 2.0 * b * nv * C * C
    print("%d meshes x %d vertices, 192 -> 192 (%.2f GFLOP per product, fp32 MFMA floor %.1f us)" % (b, nv, flop / 1e9, flop / 157.3e6))
    for k, v in res.items():
        print("   %-64s %7.1f us" % (k, v))


if __name__ == "__main__":
    main()
