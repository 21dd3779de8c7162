#!/bin/bash
# one step in launch order (rocprofv3 kernel trace of the graph replay), optionally with GEOM_FUSED_PLAN=off|fwd|all
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-tl}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_step -o s -- \
    python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --steps-only > $ROOT/gpurun_out/prof_step.log 2>&1
tail -1 $ROOT/gpurun_out/prof_step.log | cut -c1-120
cd $ROOT
python tools/step_timeline.py gpurun_out/prof_step gpurun_out/${TAG}_step_timeline.txt > /dev/null
cat gpurun_out/${TAG}_step_timeline.txt
