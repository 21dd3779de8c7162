#!/usr/bin/env python
"""Eager launches of the matrix-core products for rocprofv3 (kernel trace or --pmc): python tools/profile_dense.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometrics_amd import dense  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20496
dev = torch.device("cuda")
for cin in (963, 192):
    c = 192
    x = torch.randn(rows, cin, device=dev)
    g = torch.randn(rows, c, device=dev)
    w = torch.randn(cin, c, device=dev) * 0.1
    out = torch.empty(rows, c, device=dev)
    dx = torch.empty(rows, cin, device=dev)
    ws = dense.weight_workspace(rows, cin, c, dev)
    for _ in range(5):
        dense.forward(x, w, out)
        dense.backward_input(g, w, dx)
        dense.backward_weight_partials(x, g, ws, True)
        if cin <= 192:
            dense.backward_pair(x, g, w, dx, ws)       # dX and the dW partials in ONE launch (what the step runs for the hidden layers)
        torch.mm(x, w, out=out)
    torch.cuda.synchronize()
