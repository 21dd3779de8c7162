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
    # the shipped library-GEMM selections LOADED on this box (a rejected file is a silent ~2x on a third of the step)
    assert line["config"]["gemm_selection"] == "tunableop file", line["config"]["gemm_selection"]
    # the line checks itself: the timed route against the CPU oracle, outside the timed region
    spot = line["parity_spot_check"]
    assert spot["idx_mismatches"] == 0 and spot["dist_bit_mismatches"] == 0 and spot["sampled_point_bit_mismatches"] == 0
    assert spot["loss_rel_err"] <= 1e-5 and spot["grad_pos_max_err_over_scale"] <= 1e-4
    drv = line["driver_step"]
    assert drv["ms_per_step"] > 0 and "GEOMetrics.py:110-174" in drv["workload"] and len(drv["stages_us"]) >= 5


def test_the_shipped_gemm_selections_load_here(gpu):
    from geometrics_amd import gemm_tuning
    try:
        assert gemm_tuning.enable() and gemm_tuning.status == "tunableop file"
    finally:
        gemm_tuning.disable()
