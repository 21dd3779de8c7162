"""Loader of the compiled pybind entry points (geometrics_amd/lib/geom_torch_shim.so, built by geometrics_amd.build from
csrc/torch_shim.cpp): `forward_cuda` functions with the reference's call shapes (chamfer_distance.cpp:36-38,
tri_distance.cpp:34-36) on top of the C ABI.  The package itself binds the C ABI with ctypes; this module is what the
reference's own python wrappers would import in place of their JIT-compiled `cd` / `tri` modules."""
import importlib.util
import os

from . import _lib

SHIM_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "geom_torch_shim.so")
_module = None


def module():
    global _module
    if _module is None:
        if not os.path.exists(SHIM_PATH):
            raise RuntimeError("geometrics_amd: %s is missing -- build it with `python -m geometrics_amd.build`" % SHIM_PATH)
        _lib.lib()      # libgeom_hip.so first (the shim links against it; same directory, $ORIGIN rpath)
        spec = importlib.util.spec_from_file_location("geom_torch_shim", SHIM_PATH)
        _module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_module)
        _module.set_reference_quirks(_lib.reference_quirks())      # follow the package-wide switch, not only the environment
    return _module


class _Cd:
    """`cd.forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2)` -- the reference's pybind module `cd`."""

    @staticmethod
    def forward_cuda(*args, **kw):
        return module().chamfer_forward_cuda(*args, **kw)


class _Tri:
    """`tri.forward_cuda(xyz1, tri1, tri2, tri3, dist, point, index)` -- the reference's pybind module `tri`."""

    @staticmethod
    def forward_cuda(*args, **kw):
        return module().tri_forward_cuda(*args, **kw)


cd, tri = _Cd(), _Tri()
