"""Library-GEMM solution selection for the dense feature GEMMs of the 0N-GCN layers.

The dense `input @ W` products stay library GEMMs (rocBLAS / hipBLASLt through torch.matmul);
for the skinny fp32 shapes of this path ([B*V, 963] x [963, 192] and its two gradients) the
default hipBLASLt heuristic picks kernels that run at 20-40 TFLOP/s, while rocBLAS has
solutions at 90-110 TFLOP/s.  PyTorch's TunableOp can pin the faster solution per shape from a
results file; `geometrics_amd/tuning/tunableop_gfx950.csv` holds the selections measured on
MI355X for this image's rocBLAS/hipBLASLt (tools/tune_gemm.py regenerates it).  Tuning is NOT
run at use time (no warm-up cost, HIP-graph safe); a shape that is not in the file, or a
library-version mismatch, silently falls back to the default heuristic -- results are identical
either way (both are fp32 GEMMs), only the speed differs.
"""
import atexit
import os
import shutil
import sys
import tempfile

import torch

TUNING_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950.csv")


# what the last enable() did: "tunableop file" | "library default (no tuning file)" | "library default (tuning file rejected)"
status = "library default (enable() not called)"


def enable(filename=TUNING_FILE):
    """Use the recorded GEMM selections.  Returns True when the file was found and loaded; `status` says which of the three
    outcomes it was (bench.py prints it as config.gemm_selection; a rejected file is a ~2x slower library product, so
    tests/test_bench_contract_gpu.py asserts that the shipped file loads on the GPU box)."""
    global status
    if os.environ.get("GEOM_TUNING") == "reject":      # tools / tests: behave as after a library update (the file is refused)
        status = "library default (tuning file rejected)"
        return False
    if not (torch.cuda.is_available() and os.path.exists(filename)):
        status = "library default (no tuning file)"
        return False
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.tuning_enable(False)            # never tune here: only replay recorded selections
    tun.record_untuned_enable(False)
    try:
        # read-only use of the shipped file: nothing is written back at exit, and the filename TunableOp holds
        # points into a private 0700 scratch directory (removed at exit), so N ranks never rewrite the shared
        # selections and no predictable path in a shared tmp dir is ever opened for writing
        scratch = tempfile.mkdtemp(prefix="geom_tunableop_")
        atexit.register(shutil.rmtree, scratch, True)
        tun.set_filename(os.path.join(scratch, "selections.csv"))
        if hasattr(tun, "write_file_on_exit"):
            tun.write_file_on_exit(False)
        ok = bool(tun.read_file(filename))
    except Exception as exc:            # malformed / incompatible file: keep the default heuristic, and say so
        print("geometrics_amd.gemm_tuning: could not load %s (%s: %s); library default GEMM selection in use"
              % (filename, type(exc).__name__, exc), file=sys.stderr)
        tun.enable(False)
        status = "library default (tuning file rejected)"
        return False
    status = "tunableop file" if ok else "library default (tuning file rejected)"
    if not ok:
        print("geometrics_amd.gemm_tuning: %s was not accepted by TunableOp (library version mismatch?); "
              "library default GEMM selection in use" % filename, file=sys.stderr)
    return ok


def tune_products(shapes, device=None, max_ms=30, max_iters=30):
    """The shipped selections were refused (another PyTorch / hipBLASLt build than they were recorded with): let TunableOp
    pick the library's solutions for THIS build, once, for the products the layers hand to the library -- shapes = [(meshes,
    vertices, cin, cout)]: the forward `x[B,V,cin] @ w` and the input gradient `g[B*V,cout] @ w^T`, issued exactly as
    geometrics_amd.layers issues them -- each shape timed for at most `max_ms` milliseconds, and replay them from then on (in
    memory, for the life of the process; nothing is written).  Under a second at start-up instead of products on the default
    heuristic for the whole run (the 963-wide ones: 85 us instead of 62); call it BEFORE capturing HIP graphs.  Plain products
    on the current stream only: tuning inside a backward pass of the real step crashed the process (ROCm 7.2 / PyTorch 2.10)."""
    global status
    if not torch.cuda.is_available():
        return False
    tun = torch.cuda.tunable
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    try:
        tun.enable(True)
        tun.tuning_enable(True)
        tun.record_untuned_enable(False)
        tun.set_max_tuning_duration(int(max_ms))
        tun.set_max_tuning_iterations(int(max_iters))
        scratch = tempfile.mkdtemp(prefix="geom_tunableop_")
        atexit.register(shutil.rmtree, scratch, True)
        tun.set_filename(os.path.join(scratch, "selections.csv"))
        if hasattr(tun, "write_file_on_exit"):
            tun.write_file_on_exit(False)
        with torch.no_grad():
            for meshes, nv, cin, cout in shapes:
                x = torch.randn(meshes, nv, cin, device=dev)
                w = torch.randn(cin, cout, device=dev)
                g = torch.randn(meshes * nv, cout, device=dev)
                torch.matmul(x, w)
                torch.matmul(g, w.t())
                torch.mm(g, w.t(), out=torch.empty(meshes * nv, cin, device=dev))
        torch.cuda.synchronize()
    except Exception as exc:
        print("geometrics_amd.gemm_tuning: start-up tuning failed (%s: %s); library default GEMM selection in use"
              % (type(exc).__name__, exc), file=sys.stderr)
        try:
            tun.tuning_enable(False)
            tun.enable(False)
        except Exception:
            pass
        return False
    tun.tuning_enable(False)
    status = "tuned at start-up (the shipped selections were rejected by this library build)"
    return True


def disable():
    if torch.cuda.is_available():
        torch.cuda.tunable.enable(False)
