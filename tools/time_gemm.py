"""Dev helper: time the five GEMM shapes of the 3-layer 0N-GCN stack under different BLAS backends."""
import os, sys, time
import torch
dev = torch.device("cuda:0")
M = 8 * 2562
shapes = {  # name: (A shape, B shape, transA, transB)
    "fwd1  [M,963]x[963,192]": ((M, 963), (963, 192), False, False),
    "fwd2  [M,192]x[192,192]": ((M, 192), (192, 192), False, False),
    "dW1   [963,M]x[M,192]": ((M, 963), (M, 192), True, False),
    "dX1   [M,192]x[192,963]": ((M, 192), (963, 192), False, True),
    "dW2   [192,M]x[M,192]": ((M, 192), (M, 192), True, False),
    "dX2   [M,192]x[192,192]": ((M, 192), (192, 192), False, True),
}
def bench(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for lib in sys.argv[1:] or ["default"]:
    if lib in ("hipblaslt", "rocblas", "cublaslt", "cublas"):
        torch.backends.cuda.preferred_blas_library(lib)
    print("== backend", lib, "tunable", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"))
    tot = 0
    for name, (sa, sb, ta, tb) in shapes.items():
        A = torch.randn(*sa, device=dev); B = torch.randn(*sb, device=dev)
        a = A.t() if ta else A; b = B.t() if tb else B
        t = bench(lambda: torch.matmul(a, b))
        fl = 2 * a.shape[0] * a.shape[1] * b.shape[1]
        tot += t * (1 if name[:3] in ("fwd1", "dW1", "dX1") or name.startswith(("fwd1","dW1","dX1")) else 2)
        print(f"  {name:28s} {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s")
    print("  per-step total (1x layer1 + 2x layer2/3 each):", round(tot, 1), "us")
