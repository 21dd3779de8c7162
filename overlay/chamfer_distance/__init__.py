"""Overlay package: `from chamfer_distance import ChamferDistance` in an unmodified reference
checkout resolves here (put `<this repo>/overlay` and `<this repo>` in front of PYTHONPATH).
Replaces the nvcc-JIT module of the reference (chamfer_distance/chamfer_distance.py:1-38)."""
from geometrics_amd.chamfer_distance import (ChamferDistance, ChamferDistanceFunction,  # noqa: F401
                                             chamfer_nn, forward_cuda)
