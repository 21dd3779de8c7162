import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def fill_parameters(module, seed, gain=2.0):
    """Deterministic parameter values independent of torch's RNG stream (so a fixture only stores the seed):
    matrices ~ U(+-gain/sqrt(sum(shape))), vectors ~ U(+-0.1), in sorted-name order."""
    import zlib

    import torch
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
            bound = gain / np.sqrt(sum(p.shape)) if p.dim() >= 2 else 0.1
            p.copy_(torch.from_numpy(rng.uniform(-bound, bound, tuple(p.shape)).astype(np.float32)))
    return module


def tri_true_case(name):
    """Inputs of a tests/golden/tri_true_*.npz fixture, regenerated from their seeds (the fixture stores the expected
    per-point true squared distances + a checksum of the inputs): (verts [B,V,3], faces [F,3], points [B,N,3], true [B,N])."""
    from geometrics_amd import meshgen
    g = golden(name)
    V, F = meshgen.icosphere(int(g["level"]))
    first = int(g["first"]) if "first" in g else 0
    verts = meshgen.jittered_batch(V, int(g["batch"]), first=first)
    pts = meshgen.gt_cloud(int(g["batch"]), int(g["num"]), first=first, cube=bool(g["cube"]))
    assert float(verts.astype(np.float64).sum() + pts.astype(np.float64).sum()) == float(g["checksum"])
    return verts, F, pts, g["true_sqdist"]
