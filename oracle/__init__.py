"""CPU ORACLE for the GEOMetrics hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product path (``geometrics_amd``) never does.

* ``liboracle.so``  -- our plain-C restatement (``geom_oracle.c``; each function
  cites the reference file:line it follows), built by ``oracle/Makefile``.
* ``_ref/libref_nnsearch.so`` -- the reference's own CPU ``nnsearch``
  (old_GEOMetrics/chamfer_distance/src/my_lib.c:4-26) compiled verbatim from
  the reference checkout by ``build_ref.sh``; used to pin the restatement and
  as the ``cpu_baseline`` of kind "reference".
* ``ref_ops.py`` -- torch-CPU restatements of the differentiable python stages
  (sampling, losses, 0N-GCN), pinned by ``tests/golden`` fixtures emitted from
  the imported reference python.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FLAG_REF_TAIL_TRUNC = 1
FLAG_FIX_REGION6 = 2
FLAG_NN_FMA = 8

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile liboracle.so (always possible) and _ref (only if the reference is here)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "geom_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference"):
        ref = os.path.join(_HERE, "_ref", "libref_nnsearch_fma.so")     # the second product of build_ref.sh
        if force or not os.path.exists(ref):
            subprocess.check_call(["sh", os.path.join(_HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)


_lib = None
_ref = None
_ref_fma = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.oracle_nn_scan.argtypes = [ctypes.c_int] * 3 + [_f32p, _f32p, _f32p, _i32p]
        L.oracle_nn_scan_fma.argtypes = [ctypes.c_int] * 3 + [_f32p, _f32p, _f32p, _i32p]
        L.oracle_nn_tiled.argtypes = [ctypes.c_int] * 3 + [_f32p, _f32p, _f32p, _i32p, ctypes.c_uint]
        L.oracle_nn_grad.argtypes = [ctypes.c_int] * 3 + [_f32p] * 4 + [_i32p] * 2 + [_f32p] * 2
        L.oracle_tri_scan.argtypes = [ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_int,
                                      _f32p, _f32p, _f32p, _f32p, _i32p, _i32p, ctypes.c_uint]
        L.oracle_tri_scan_indexed.argtypes = [ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_int, _f32p,
                                              ctypes.c_int, _i64p, _f32p, _i32p, _i32p, ctypes.c_uint]
        L.oracle_tri_pair.argtypes = [_f32p] * 4 + [ctypes.c_uint, _i32p]
        L.oracle_tri_pair.restype = ctypes.c_float
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_nnsearch.so"))


def ref():
    """The reference's own nnsearch (None-safe: raises if it was never built)."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libref_nnsearch.so")
        if not os.path.exists(p):
            build()
        R = ctypes.CDLL(p)
        R.nnsearch.argtypes = [ctypes.c_int] * 3 + [_f32p, _f32p, _f32p, _i32p]
        _ref = R
    return _ref


def have_ref_fma():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_nnsearch_fma.so")) and _host_has_fma()


def _host_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            return " fma " in f.read().replace("\n", " ")
    except OSError:
        return False


def ref_fma():
    """The reference's nnsearch built with -mfma -ffp-contract=fast (needs an FMA-capable host CPU)."""
    global _ref_fma
    if _ref_fma is None:
        R = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_nnsearch_fma.so"))
        R.nnsearch.argtypes = [ctypes.c_int] * 3 + [_f32p, _f32p, _f32p, _i32p]
        _ref_fma = R
    return _ref_fma


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _one_way(fn, query, target, *extra):
    query, qp = _f(query)
    target, tp = _f(target)
    b, n, _ = query.shape
    m = target.shape[1]
    dist = np.zeros((b, n), np.float32)
    idx = np.zeros((b, n), np.int32)
    fn(b, n, m, qp, tp, dist.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p), *extra)
    return dist, idx


def nn_scan(query, target):
    """For each query[b,j] the arg-min target: (dist[b,n] f32, idx[b,n] i32)."""
    return _one_way(lib().oracle_nn_scan, query, target)


def nn_tiled(query, target, flags=0):
    return _one_way(lib().oracle_nn_tiled, query, target, flags)


def nn_scan_fma(query, target):
    """nn_scan in the FMA-contracted arithmetic (GEOM_FLAG_NN_FMA)."""
    return _one_way(lib().oracle_nn_scan_fma, query, target)


def ref_nnsearch(query, target):
    return _one_way(ref().nnsearch, query, target)


def ref_nnsearch_fma(query, target):
    return _one_way(ref_fma().nnsearch, query, target)


def chamfer_nn(xyz1, xyz2, flags=0, use_ref=False):
    """Both directions, argument order of ChamferDistance.forward: (dist1, idx1, dist2, idx2)."""
    if flags & FLAG_NN_FMA:
        if flags & ~FLAG_NN_FMA:
            raise ValueError("FLAG_NN_FMA does not combine with the tiled-kernel quirk flags")
        one = ref_nnsearch_fma if use_ref else nn_scan_fma
    else:
        one = ref_nnsearch if use_ref else (nn_scan if flags == 0 else (lambda q, t: nn_tiled(q, t, flags)))
    d1, i1 = one(xyz1, xyz2)
    d2, i2 = one(xyz2, xyz1)
    return d1, i1, d2, i2


def tri_scan(xyz, tri1, tri2, tri3, flags=0):
    """(dist[b,n] f32, point[b,n] i32 in 0..6, index[b,n] i32)."""
    xyz, xp = _f(xyz)
    tri1, p1 = _f(tri1)
    tri2, p2 = _f(tri2)
    tri3, p3 = _f(tri3)
    b, n, _ = xyz.shape
    m = tri1.shape[1]
    dist = np.zeros((b, n), np.float32)
    point = np.zeros((b, n), np.int32)
    index = np.zeros((b, n), np.int32)
    lib().oracle_tri_scan(b, n, xp, m, p1, p2, p3, dist.ctypes.data_as(_f32p),
                          point.ctypes.data_as(_i32p), index.ctypes.data_as(_i32p), flags)
    return dist, point, index


def tri_scan_indexed(xyz, verts, faces, flags=0):
    xyz, xp = _f(xyz)
    verts, vp = _f(verts)
    faces = np.ascontiguousarray(faces, dtype=np.int64)
    b, n, _ = xyz.shape
    nv = verts.shape[1]
    nf = faces.shape[0]
    dist = np.zeros((b, n), np.float32)
    point = np.zeros((b, n), np.int32)
    index = np.zeros((b, n), np.int32)
    lib().oracle_tri_scan_indexed(b, n, xp, nv, vp, nf, faces.ctypes.data_as(_i64p),
                                  dist.ctypes.data_as(_f32p), point.ctypes.data_as(_i32p),
                                  index.ctypes.data_as(_i32p), flags)
    return dist, point, index


def tri_pair(p, A, B, C, flags=0):
    arrs = [np.ascontiguousarray(x, dtype=np.float32) for x in (p, A, B, C)]
    opt = ctypes.c_int(0)
    d = lib().oracle_tri_pair(*[a.ctypes.data_as(_f32p) for a in arrs], flags, ctypes.byref(opt))
    return np.float32(d), opt.value


def nn_grad(xyz1, xyz2, graddist1, graddist2, idx1, idx2):
    xyz1, a = _f(xyz1)
    xyz2, b_ = _f(xyz2)
    graddist1, g1 = _f(graddist1)
    graddist2, g2 = _f(graddist2)
    idx1 = np.ascontiguousarray(idx1, np.int32)
    idx2 = np.ascontiguousarray(idx2, np.int32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    o1 = np.zeros_like(xyz1)
    o2 = np.zeros_like(xyz2)
    lib().oracle_nn_grad(b, n, m, a, b_, g1, g2, idx1.ctypes.data_as(_i32p), idx2.ctypes.data_as(_i32p),
                         o1.ctypes.data_as(_f32p), o2.ctypes.data_as(_f32p))
    return o1, o2
