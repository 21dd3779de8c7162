"""Ahead-of-time build of libgeom_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m geometrics_amd.build [--force]

The library is built IN-TREE (geometrics_amd/lib/libgeom_hip.so) so that it travels with
the repo snapshot to the GPU box; it is git-ignored.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgeom_hip.so")
STAMP = os.path.join(LIBDIR, "libgeom_hip.digest")      # sha256 of the sources the library was built from (build artefact)
SHIM = os.path.join(LIBDIR, "geom_torch_shim.so")       # pybind `forward_cuda` entry points on top of the C ABI
SHIM_SRC = os.path.join(CSRC, "torch_shim.cpp")

# -ffp-contract=off: one canonical fp32 arithmetic shared with the CPU oracle (no FMA contraction).
# Division and sqrt stay correctly rounded (hipcc default; never -ffast-math).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


# per-file additions.  dense_gemm.hip: keep the MFMA accumulators in (unified) VGPRs -- with the default heuristic hipcc
# (ROCm 7.2) allocates the accumulators of the pipelined loop as untied AGPR tuples and repairs the rotation with ~85
# v_accvgpr moves per stage, in front of the first MFMA of every stage
EXTRA_FLAGS = {"dense_gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "zn_stack.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "deform_block.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_digest():
    """sha256 over every kernel source and header the library is built from (names + contents): what a committed
    profile / counter file is stamped with, so that stale counters are recognised (bench.py, tools/pmc_traffic_json.py)."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale():
    if not os.path.exists(LIB):
        return True
    if built_digest() not in (None, source_digest()):   # sources changed back and forth (a checkout): mtimes do not tell
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_shim(force=False, verbose=False):
    """The compiled pybind module with the reference's `forward_cuda` call shapes (chamfer_distance.cpp:36-38,
    tri_distance.cpp:34-36): host-only C++ (g++) against the torch headers, linked to libgeom_hip.so."""
    if not force and os.path.exists(SHIM) and os.path.getmtime(SHIM) >= max(os.path.getmtime(SHIM_SRC), os.path.getmtime(LIB)):
        return SHIM
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    tlib = ce.library_paths()[0]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=geom_torch_shim", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I", INCLUDE, "-I", sysconfig.get_paths()["include"], "-I", "/opt/rocm/include"]
    for inc in ce.include_paths():
        cmd += ["-I", inc]
    cmd += [SHIM_SRC, "-o", SHIM, "-L", LIBDIR, "-lgeom_hip", "-L", tlib, "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10",
            "-lc10_hip", "-ltorch_python", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SHIM


def build(force=False, verbose=False):
    lib = build_lib(force, verbose)
    build_shim(force, verbose)
    return lib


def build_lib(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc] + HIPCC_FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-I", INCLUDE, "-I", CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", LIB]
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:      # what the library was built from: the loader refuses a library older than its sources
        f.write(source_digest() + "\n")
    return LIB


def built_digest():
    """Digest of the sources the library on disk was built from (None: no stamp, e.g. a library built by other means)."""
    try:
        with open(STAMP) as f:
            return f.read().strip() or None
    except OSError:
        return None


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
