"""Would splitting a 0N-GCN layer's weight into [W_agg | W_pass] pay?  
    python tools/time_split_gemm.py [meshes] [V]
The idea: let the library write the pass-through columns out[:, k:] = relu(X W_pass + b) straight into the layer
output (bias + ReLU epilogue), so that the aggregation kernel touches only the k aggregated columns.  The price is
two products per direction (N = 128 and N = 64) where there was one (N = 192).  This times, with TunableOp tuning
switched ON for every shape (so neither side runs on an untuned default): forward, dX and dW at N = 192, 128, 64
for both layer widths, plus the epilogue form writing into a column slice of a [M, 192] buffer."""
import os
import sys

os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/split_gemm_tunableop.csv")
import torch  # noqa: E402

torch.cuda.tunable.set_max_tuning_duration(int(os.environ.get("GEOM_TUNE_MS", "100")))
torch.cuda.tunable.set_max_tuning_iterations(int(os.environ.get("GEOM_TUNE_ITERS", "60")))
dev = torch.device("cuda:0")
meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
V = int(sys.argv[2]) if len(sys.argv) > 2 else 2562
M = meshes * V


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("M = %d rows (%d meshes x %d vertices); times in us, back to back" % (M, meshes, V))
for K in (963, 192):
    x = torch.randn(M, K, device=dev)
    rows = {}
    for N in (192, 128, 64):
        w = torch.randn(K, N, device=dev) * 0.05
        g = torch.randn(M, N, device=dev)
        bias = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        wide = torch.empty(M, 192, device=dev)
        view = wide[:, 192 - N:]
        gx, gw = torch.empty(M, K, device=dev), torch.empty(K, N, device=dev)
        wt, xt = w.t(), x.t()
        rows[N] = dict(
            fwd=timed(lambda: torch.mm(x, w, out=y)),
            fwd_bias_relu=timed(lambda: torch._addmm_activation(bias, x, w, use_gelu=False)),
            fwd_into_slice=timed(lambda: torch.mm(x, w, out=view)),
            dX=timed(lambda: torch.mm(g, wt, out=gx)),
            dW=timed(lambda: torch.mm(xt, g, out=gw)),
        )
        # the slice form must really have written in place (no hidden temporary + copy would show in the time only)
        torch.mm(x, w, out=view)
        assert torch.equal(view, torch.mm(x, w)) or torch.allclose(view, torch.mm(x, w), rtol=1e-4, atol=1e-4)
    print("K = %d" % K)
    for N in (192, 128, 64):
        print("   N = %3d  " % N + "  ".join("%s %6.1f" % kv for kv in rows[N].items()))
    for leg in ("fwd", "dX", "dW"):
        one, two = rows[192][leg], rows[128][leg] + rows[64][leg]
        print("   %-3s one product %6.1f   split %6.1f   (%+.1f us)" % (leg, one, two, two - one))
    print("   fwd split with the library's bias+ReLU epilogue on the pass-through part: %6.1f"
          % (rows[128]["fwd_bias_relu"] + rows[64]["fwd"]))
