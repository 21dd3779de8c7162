"""The any-shape matrix-core product (geom_gemm_f32) at the mesh encoder's layer shapes (16-mesh ragged batch: 18 432 rows):
forward x.w, input gradient g.w^T, weight gradient x^T.g per layer, against the library (torch.mm).  us per launch (HIP-graph
replay of 20 launches), TFLOP/s, and the sum over the encoder's 17 layers.     python tools/time_gemm_any.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometrics_amd import dense  # noqa: E402

WIDTHS = ((3, 60), (60, 60), (60, 60), (60, 60), (60, 120), (120, 120), (120, 120), (120, 150), (150, 200), (200, 210), (210, 250),
          (250, 300), (300, 300), (300, 300), (300, 300), (300, 300), (300, 50))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * reps)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 18432
    only = os.environ.get("SHAPES")          # e.g. "300x300"
    tot = {"any": [0.0, 0.0, 0.0], "lib": [0.0, 0.0, 0.0]}
    seen = {}
    for cin, c in WIDTHS:
        if only and "%dx%d" % (cin, c) not in only.split(","):
            continue
        if (cin, c) not in seen:
            x, w, g = torch.randn(rows, cin, device="cuda"), torch.randn(cin, c, device="cuda"), torch.randn(rows, c, device="cuda")
            out_f, out_x, out_w = torch.empty(rows, c, device="cuda"), torch.empty(rows, cin, device="cuda"), torch.empty(cin, c, device="cuda")
            ws = torch.empty(max(dense._lib.lib().geom_gemm_workspace_floats(cin, c, rows), 4), device="cuda")
            t_any = (timed(lambda: dense.gemm(x, w, out=out_f)), timed(lambda: dense.gemm(g, w, trans_b=True, out=out_x)),
                     timed(lambda: dense.gemm(x, g, trans_a=True, out=out_w, workspace=ws)))
            t_lib = (timed(lambda: torch.mm(x, w, out=out_f)), timed(lambda: torch.mm(g, w.t(), out=out_x)),
                     timed(lambda: torch.mm(x.t(), g, out=out_w)))
            seen[(cin, c)] = (t_any, t_lib)
            fl = 2.0 * rows * cin * c
            print("%5d x %3d -> %3d   fwd %6.1f (%5.1f TF/s; lib %6.1f)   dX %6.1f (%5.1f; lib %6.1f)   dW %6.1f (%5.1f; lib %6.1f)"
                  % (rows, cin, c, t_any[0], fl / t_any[0] / 1e6, t_lib[0], t_any[1], fl / t_any[1] / 1e6, t_lib[1], t_any[2],
                     fl / t_any[2] / 1e6, t_lib[2]), flush=True)
        t_any, t_lib = seen[(cin, c)]
        for i in range(3):
            tot["any"][i] += t_any[i]
            tot["lib"][i] += t_lib[i]
    print("sum over the %d layers: any-shape kernel fwd %.0f + dX %.0f + dW %.0f = %.0f us; library %.0f + %.0f + %.0f = %.0f us"
          % (len(WIDTHS), *tot["any"], sum(tot["any"]), *tot["lib"], sum(tot["lib"])))


if __name__ == "__main__":
    main()
