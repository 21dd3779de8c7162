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
import os
import tempfile

import torch

TUNING_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950.csv")


def enable(filename=TUNING_FILE):
    """Use the recorded GEMM selections.  Returns True when the file was found and loaded."""
    if not (torch.cuda.is_available() and os.path.exists(filename)):
        return False
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.tuning_enable(False)            # never tune here: only replay recorded selections
    tun.record_untuned_enable(False)
    try:
        # read-only use of the shipped file: whatever TunableOp writes when the process exits goes to a private
        # scratch path, so N ranks never rewrite the shared selections concurrently
        tun.set_filename(os.path.join(tempfile.gettempdir(), "geom_tunableop_%d.csv" % os.getpid()))
        return bool(tun.read_file(filename))
    except Exception:                   # malformed / incompatible file: keep the default heuristic
        tun.enable(False)
        return False


def disable():
    if torch.cuda.is_available():
        torch.cuda.tunable.enable(False)
