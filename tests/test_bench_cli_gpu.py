"""-m gpu: bench.py AS THE DRIVER TYPES IT.  `python bench.py --gpus N ...` with no launcher around it must start its N
ranks itself, and the single JSON line must say how many ranks the collective really spanned (`ranks_seen`).  On a 1-GPU
box the two ranks share cuda:0 and gloo carries the all-reduce (GEOM_DIST_BACKEND, the hook tests/test_dist_step_gpu.py
uses too); on an 8-GPU node the same command runs one rank per GPU over RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=540):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, "bench.py %s failed (%d):\n%s" % (" ".join(args), p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "expected exactly ONE JSON line on stdout, got %d:\n%s" % (len(lines), p.stdout[-2000:])
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_gpus_2_typed_without_a_launcher_spawns_its_ranks(gpu):
    line = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--steps-only", "--clock-warmup-ms", "0"],
                {"GEOM_DIST_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["config"]["global_batch"] == 16 and line["config"]["parallelism"] == "dp2"
    assert line["steps"] == 5 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["ms_per_step"] > 0
    r = line["ms_per_step_ranks"]
    assert r["min"] <= r["rank0"] <= r["max"] == line["ms_per_step"]
    assert "all-reduce over 2 ranks" in line["config"]["workload"]


@pytest.mark.timeout(900)
def test_bench_gpus_8_rehearsal_with_eight_ranks_on_one_gpu(gpu):
    """What the driver will type on an 8-GPU node, rehearsed on one: `python bench.py --gpus 8` spawns EIGHT ranks (free
    port, teardown by PID), each builds its 8-mesh shard of BASELINE config 5's 64 meshes, captures its step graphs and reads
    the TunableOp selections concurrently; the collective spans all eight (`ranks_seen`), the line reports the whole job.
    The ranks share cuda:0 and gloo carries the all-reduce (GEOM_DIST_BACKEND hook: RCCL needs a GPU per rank), so the
    numbers mean nothing -- the plumbing is what must not fail the first time eight GPUs are there."""
    line = _run(["--gpus", "8", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--steps-only", "--clock-warmup-ms", "0"],
                {"GEOM_DIST_BACKEND": "gloo"}, timeout=840)
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8
    assert line["config"]["global_batch"] == 64 and line["config"]["meshes_per_gpu"] == 8 and line["config"]["parallelism"] == "dp8"
    assert line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    r = line["ms_per_step_ranks"]
    assert r["min"] <= r["rank0"] <= r["max"] == line["ms_per_step"]
    assert "all-reduce over 8 ranks" in line["config"]["workload"]
    assert line["config"]["dp_sequence"].startswith("per step: graph A")      # gloo cannot be captured: the two-graph sequence


@pytest.mark.timeout(600)
def test_bench_single_gpu_line_has_the_contract_fields(gpu):
    line = _run(["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--steps-only", "--clock-warmup-ms", "0"])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1
    assert "all-reduce" not in line["config"]["workload"]          # no collective exists in the single-process step
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line


def test_more_ranks_than_devices_is_refused_loudly(gpu):
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GEOM_DIST_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       env=env, cwd=ROOT, timeout=300, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "HIP device" in p.stderr


@pytest.mark.timeout(600)
def test_bench_whole_n_gt_1_path_with_rccl_on_one_gpu(gpu):
    """GEOM_BENCH_FORCE_DP=1: bench.py's complete N > 1 path -- RCCL process group, gradient bucket, two-graph step with the
    asynchronous all-reduce, the timed loop AND the per-kernel probes that follow it with RCCL's watchdog thread alive -- in
    a 1-rank group on this GPU.  What an 8-GPU node will run, minus the other seven ranks; the line must come out complete
    (the probes capture thread-locally, so that the watchdog's event polling cannot invalidate their captures)."""
    line = _run(["--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--clock-warmup-ms", "0"], {"GEOM_BENCH_FORCE_DP": "1"})
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and "forced_dp" in line["config"]
    assert line["config"]["dp_sequence"].startswith("per step ONE graph")     # the collective was captured into the step graph
    assert line["config"]["launch"] == "hipgraph" and line["value"] > 0
    assert line.get("roofline_error") is None and line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1
    assert "surface_scan_kernel (both arg-min scans of the surface loss)" in line["other_kernels"]


@pytest.mark.timeout(600)
def test_a_rejected_selections_file_costs_less_than_three_per_cent(gpu):
    """The library products of the step are fast only with recorded solution selections, and TunableOp validates the shipped
    file against the PyTorch / hipBLASLt build: after a library update it is REFUSED.  bench.py then lets TunableOp pick this
    build's solutions at start-up (gemm_tuning.tune_products, under a second) instead of running on the default heuristic
    (0.506 ms per step against 0.448: measured).  GEOM_TUNING=reject simulates the refusal: the line must say so, and the step
    must be within 3 % of the step on the shipped file -- the headline does not hang on a version-locked file."""
    args = ["--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--steps-only"]
    shipped = _run(args)
    refused = _run(args, {"GEOM_TUNING": "reject"})
    untuned = _run(args, {"GEOM_TUNING": "reject", "GEOM_RETUNE": "0", "GEOM_OWN_PRODUCTS": "0"})
    assert shipped["config"]["gemm_selection"] == "tunableop file"
    assert refused["config"]["gemm_selection"].startswith("tuned at start-up")
    assert untuned["config"]["gemm_selection"] == "library default (tuning file rejected)"
    assert refused["ms_per_step"] <= 1.03 * shipped["ms_per_step"], (refused["ms_per_step"], shipped["ms_per_step"])
    assert untuned["ms_per_step"] > 1.05 * shipped["ms_per_step"]          # what the fallback is worth
    # same arithmetic whichever library kernel runs a product: fp32, the losses agree to summation order
    assert abs(refused["final_loss"] - shipped["final_loss"]) <= 1e-2 * abs(shipped["final_loss"])
