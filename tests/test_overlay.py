"""CPU: the drop-in overlay resolves the reference's imports to the HIP implementations and
exports every name the unmodified reference drivers take from their star-imports.
Needs the reference checkout (build container only); skipped on the GPU box."""
import ast
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")

PROBE = r'''
import os, sys, types
front = [OVERLAY, ROOT, REF]                        # what geometrics_amd.run builds before executing a driver
sys.path[:] = front + [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in front]
for name in ("torchvision", "torchvision.transforms", "torchvision.models"):       # absent from this image
    m = types.ModuleType(name); sys.modules[name] = m
class _T:
    def __init__(self, *a, **k): pass
    def __call__(self, x): return x
tv = sys.modules["torchvision.transforms"]
tv.Normalize = tv.Compose = tv.Resize = tv.ToTensor = _T
sys.modules["torchvision"].transforms = tv
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
import utils, layers, models, chamfer_distance, tri_distance
import geometrics_amd.utils as gu, geometrics_amd.layers as gl, geometrics_amd.models as gm
assert models.BatchMeshDeformationBlock is gm.BatchMeshDeformationBlock and models.__file__.startswith(OVERLAY)
assert models.VGG is models._reference_models.VGG and models.Decoder is models._reference_models.Decoder
assert utils.__file__.startswith(OVERLAY) and layers.__file__.startswith(OVERLAY)
assert utils._reference_utils.__file__.startswith(REF)
for n in ("batch_sample", "batch_point_to_point", "batch_point_to_surface", "calc_point_to_line", "adj_init",
          "calc_adj", "normalize_adj", "edge", "Plane", "chamfer_dist", "tri_dist"):
    assert getattr(utils, n) is getattr(gu, n), n
for n in ("batched_pooling", "batch_camera_info", "batch_calc_edge", "batch_get_lap_info"):
    assert getattr(utils, n) is getattr(gu, n), n
for n in ("load_initial", "ObjLoader", "Mesh_loader", "Voxel_loader", "render_mesh"):
    assert getattr(utils, n) is getattr(utils._reference_utils, n), n       # untouched reference code
for n in ("ZERON_GCN", "GCNMax", "Batch_Image_ZERON_GCNGCN", "BatchZERON_GCN", "BatchGCNMax"):
    assert getattr(layers, n) is getattr(gl, n), n
assert type(utils.chamfer_dist).__module__ == "geometrics_amd.chamfer_distance"
assert chamfer_distance.ChamferDistance is type(utils.chamfer_dist)
print("NAMES", " ".join(sorted(set(dir(utils)) | set(dir(layers)) | set(dir(models)))))
'''


def _star_names():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    code = "OVERLAY=%r; ROOT=%r; REF=%r\n" % (os.path.join(ROOT, "overlay"), ROOT, REF) + PROBE
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=REF, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("NAMES ")][0]
    return set(line.split()[1:])


def _free_names(path):
    """Names a driver loads that it never binds itself (so they must come from a star-import)."""
    tree = ast.parse(open(path).read())
    bound, used = set(dir(__builtins__) if not isinstance(__builtins__, dict) else __builtins__), set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Name):
            (used if isinstance(node.ctx, ast.Load) else bound).add(node.id)
        elif isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            bound.add(node.name)
            if isinstance(node, ast.FunctionDef):
                bound.update(a.arg for a in node.args.args + node.args.kwonlyargs)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            bound.update((a.asname or a.name).split(".")[0] for a in node.names if a.name != "*")
        elif isinstance(node, ast.ExceptHandler) and node.name:
            bound.add(node.name)
    return used - bound


def test_overlay_resolves_and_covers_the_drivers():
    names = _star_names()
    for driver in ("GEOMetrics.py", "auto_encoder.py"):
        missing = _free_names(os.path.join(REF, driver)) - names
        assert not missing, "%s needs names the overlay does not export: %s" % (driver, sorted(missing))


def test_launcher_puts_the_overlay_in_front(tmp_path):
    """geometrics_amd.run executes a driver as __main__ with [overlay, repo, script dir] leading sys.path."""
    script = tmp_path / "driver.py"
    script.write_text("import sys, layers\nprint('LAYERS', layers.__file__)\nprint('ARGS', sys.argv[1:])\n"
                      "assert __name__ == '__main__'\n")
    (tmp_path / "layers.py").write_text("raise RuntimeError('the driver-local module must be shadowed')\n")
    out = subprocess.run([sys.executable, "-m", "geometrics_amd.run", str(script), "--seed", "41"], cwd=str(tmp_path),
                         env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert os.path.join(ROOT, "overlay", "layers.py") in out.stdout and "['--seed', '41']" in out.stdout
