"""Dev helper: the 12 same-shape weight-gradient products of a deformation block as ONE strided-batched product
(TunableOp-tuned) against 12 separate ones."""
import os
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/dw_batched_tunableop.csv")
import torch
torch.cuda.tunable.set_max_tuning_duration(int(os.environ.get("GEOM_TUNE_MS", "150")))
torch.cuda.tunable.set_max_tuning_iterations(int(os.environ.get("GEOM_TUNE_ITERS", "80")))
dev = torch.device("cuda:0")


def timed(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for rows, layers in ((7712, 12), (7712, 6), (20496, 2), (20496, 12)):
    x = torch.randn(layers, rows, 192, device=dev); g = torch.randn(layers, rows, 192, device=dev)
    out = torch.empty(layers, 192, 192, device=dev)
    xt = x.transpose(1, 2)
    one = torch.empty(192, 192, device=dev)
    t_sep = timed(lambda: [torch.mm(x[l].t(), g[l], out=out[l]) for l in range(layers)])
    t_bmm = timed(lambda: torch.bmm(xt, g, out=out))
    ref = torch.stack([x[l].t() @ g[l] for l in range(layers)])
    err = float((torch.bmm(xt, g) - ref).abs().max() / ref.abs().max())
    print("rows %5d x %2d layers:  separate %7.1f us   one batched product %7.1f us   (rel diff %.1e)" % (rows, layers, t_sep, t_bmm, err))
