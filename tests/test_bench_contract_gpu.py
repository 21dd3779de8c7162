"""-m gpu: `python bench.py` end to end with a handful of steps -- the JSON line must carry the driver's contract
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
config.workload) plus the `roofline` and `cpu_baseline` objects, with a roofline fraction that is a fraction."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_line_carries_the_contract(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2"], cwd=ROOT,
                         capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["warmup"] == 2 and line["higher_is_better"] is True
    assert line["unit"] == "meshes/s" and line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 8 * 5 / (line["ms_per_step"] * 5e-3)) <= 0.01 * line["value"]     # whole-job meshes / wall time
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    if roof["frac"] is not None:        # counters present and valid for these kernel sources
        assert 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
        assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] == 1 and cpu["value"] > 0
    assert line["value"] > 10 * cpu["value"]                  # north_star: >= 10x the reference CPU Chamfer + tri path
    shape = line["reference_training_shape"]
    assert shape["step_us"] > 0 and "482" in shape["workload"]
