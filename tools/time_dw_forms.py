"""Dev helper: the weight-gradient product in its two equivalent forms, both TunableOp-tuned:
   dW = X^T . G   ([Cin x rows] . [rows x Cout], what autograd issues)   vs   dW^T = G^T . X   ([Cout x rows] . [rows x Cin])."""
import os
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/dw_forms_tunableop.csv")
import torch
torch.cuda.tunable.set_max_tuning_duration(int(os.environ.get("GEOM_TUNE_MS", "150")))
torch.cuda.tunable.set_max_tuning_iterations(int(os.environ.get("GEOM_TUNE_ITERS", "80")))
dev = torch.device("cuda:0")


def timed(fn, reps=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for rows in (20496, 7712):
    for cin in (963, 192, 1155):
        x = torch.randn(rows, cin, device=dev); g = torch.randn(rows, 192, device=dev)
        out1 = torch.empty(cin, 192, device=dev); out2 = torch.empty(192, cin, device=dev)
        xt, gt = x.t(), g.t()
        t1 = timed(lambda: torch.mm(xt, g, out=out1))
        t2 = timed(lambda: torch.mm(gt, x, out=out2))
        print("rows %5d  Cin %4d:  X^T.G %6.1f us    G^T.X %6.1f us" % (rows, cin, t1, t2))
