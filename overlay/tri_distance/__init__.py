"""Overlay package for the reference's `tri_distance` (tri_distance/tri_distance.py:1-43)."""
from geometrics_amd.tri_distance import (TriDistance, TriDistanceFunction, forward_cuda,  # noqa: F401
                                         tri_distance, tri_distance_indexed)
