"""Regenerate geometrics_amd/tuning/tunableop_gfx950.csv on an MI355X:
    python tools/tune_gemm.py gpurun_out/tunableop_gfx950.csv [V ...]
Runs PyTorch TunableOp over the GEMM shapes of the 0N-GCN stacks (forward, dW, dX) for the
shard sizes the benchmarks and tests use: V = 2562 (BASELINE icosphere) and 482 (the reference's training template,
GEOMetrics.py:44) by default.  tools/merge_tuning.py folds a partial result into the shipped file."""
import os
import sys

out = sys.argv[1]
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = out
import torch  # noqa: E402

torch.cuda.tunable.set_filename(out)
torch.cuda.tunable.set_max_tuning_duration(int(os.environ.get("GEOM_TUNE_MS", "60")))
torch.cuda.tunable.set_max_tuning_iterations(int(os.environ.get("GEOM_TUNE_ITERS", "40")))
dev = torch.device("cuda:0")
verts = [int(a) for a in sys.argv[2:]] or [2562, 482]
layers = [(963, 192), (192, 192), (1155, 192), (192, 3)]
batches = [int(m) for m in os.environ.get("GEOM_TUNE_MESHES", "1,2,4,8,16").split(",")]
for V, meshes in [(V, m) for V in verts for m in batches] if os.environ.get("GEOM_TUNE_BATCHED_ONLY") != "1" else []:
    M = meshes * V
    for cin, cout in layers:
        x = torch.randn(M, cin, device=dev, requires_grad=True)
        w = torch.randn(cin, cout, device=dev, requires_grad=True)
        y = x @ w
        y.backward(torch.randn_like(y))
        x3 = torch.randn(meshes, V, cin, device=dev, requires_grad=True)      # the [B,V,C] @ [C,O] path
        (x3 @ w).sum().backward()
# the batched weight-gradient product of a run of equal hidden layers (layers.weight_gradient_batching): 2 layers in the
# 3-layer stack of the BASELINE configs, 12 in a deformation block -- operands at the pitch of the stacked buffers
for V, meshes, run in [(V, m, r) for V in verts for m in batches for r in (2, 12)]:
    rows = meshes * V
    x = torch.randn(16, rows, 192, device=dev)[:run]
    g = torch.randn(16, rows, 192, device=dev)[:run]
    torch.bmm(x.transpose(1, 2), g, out=torch.empty(run, 192, 192, device=dev))
torch.cuda.synchronize()
getattr(torch.cuda.tunable, "write_file", lambda: None)()     # older builds write at exit only
print("wrote", out)
