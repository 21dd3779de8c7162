"""geometrics_amd -- MI355X (gfx950) implementation of the GEOMetrics per-step hot path.

Hand-written HIP kernels behind a C ABI (include/geom_hip.h, geometrics_amd/csrc) and the
python operator surface of the reference on top of it:

    geometrics_amd.chamfer_distance.ChamferDistance      (reference chamfer_distance/chamfer_distance.py)
    geometrics_amd.tri_distance.TriDistance              (reference tri_distance/tri_distance.py)
    geometrics_amd.layers.{ZERON_GCN, GCNMax, Batch_Image_ZERON_GCNGCN, BatchZERON_GCN, BatchGCNMax}
    geometrics_amd.models.{BatchMeshDeformationBlock, MeshEncoder};  geometrics_amd.ragged.RaggedMeshBatch
    geometrics_amd.utils.{batch_sample, batch_point_to_point, batch_point_to_surface, ...}

There is no CPU fallback: every op raises if libgeom_hip.so is missing or a tensor is not
on a HIP device.
"""
__version__ = "0.1.0"

from ._lib import reference_quirks, set_reference_quirks  # noqa: E402,F401  (package-wide GEOM_FLAG_REF_TAIL_TRUNC: GEOM_REF_QUIRKS=1)
