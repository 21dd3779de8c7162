"""Overlay module for the reference's `models.py`: everything stays the user's reference code (VGG,
encoders, decoder) except `BatchMeshDeformationBlock` and `MeshEncoder`, which are replaced by the fused-kernel
versions with identical attribute names and state_dict keys (geometrics_amd/models.py).  Same mechanism as
overlay/utils.py: the reference file further down sys.path is executed and its names re-exported."""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def _find_reference_models():
    for entry in sys.path:
        cand = os.path.join(entry or os.getcwd(), "models.py")
        if os.path.isfile(cand) and os.path.dirname(os.path.abspath(cand)) != _HERE:
            return cand
    raise ImportError("overlay/models.py: no reference models.py found on sys.path after the overlay directory")


_spec = importlib.util.spec_from_file_location("_reference_models", _find_reference_models())
_reference_models = importlib.util.module_from_spec(_spec)
sys.modules["_reference_models"] = _reference_models
_spec.loader.exec_module(_reference_models)

globals().update({k: v for k, v in vars(_reference_models).items() if not k.startswith("__")})

from geometrics_amd.models import BatchMeshDeformationBlock, MeshEncoder, VertexBatchNorm  # noqa: E402,F401
