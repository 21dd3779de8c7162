#!/bin/bash
# Round profiles, run on the GPU box from the repo root (bash tools/profile_round.sh r03):
#   gpurun_out/<tag>_step_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the step (graph replay, no timing loops)
#   gpurun_out/<tag>_step_timeline.txt       one step in launch order
#   gpurun_out/pmc/...                       three --pmc passes (eager step) -> tools/pmc_traffic_json.py
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_step -o s -- \
    python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --steps-only > $ROOT/gpurun_out/prof_step.log 2>&1
tail -1 $ROOT/gpurun_out/prof_step.log | cut -c1-160
cd $ROOT
python tools/shorten_stats.py $(ls gpurun_out/prof_step/*/s_kernel_stats.csv gpurun_out/prof_step/s_kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${TAG}_step_kernel_stats.csv
python tools/step_timeline.py gpurun_out/prof_step gpurun_out/${TAG}_step_timeline.txt | tail -24
bash tools/pmc_traffic.sh
python tools/pmc_traffic_json.py gpurun_out/pmc gpurun_out/${TAG}_pmc_counters.json
