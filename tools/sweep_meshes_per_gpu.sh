#!/bin/bash
# ms per step and meshes/s against the shard size (graph replay, steps only): the per-step fixed cost that decides the
# strong-scaling reading of config 5 (64 meshes: 8 per GPU on 8 GPUs against all 64 on one).  GPU box, repo root:
#     bash tools/sweep_meshes_per_gpu.sh > gpurun_out/meshes_per_gpu_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "# python bench.py --meshes-per-gpu M --steps 200 --warmup 20 --steps-only --no-cpu-baseline (one MI355X, HIP-graph replay)"
echo "# meshes/GPU   ms/step   meshes/s   us per mesh"
for m in 1 2 4 8 16 32 64; do
    python $R/bench.py --meshes-per-gpu $m --steps 200 --warmup 20 --steps-only --no-cpu-baseline 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%8d   %9.4f   %9.1f   %8.2f' % ($m, d['ms_per_step'], d['value'], d['ms_per_step'] * 1e3 / $m))"
done
