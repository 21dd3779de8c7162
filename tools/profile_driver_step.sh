#!/bin/bash
# rocprofv3 kernel trace of the captured driver step (bench.driver_step_times, 20 replays): kernel stats + one step aggregated
# per kernel -> gpurun_out/$TAG/ (TAG=r06 by default).   GPU box, repo root:   bash tools/profile_driver_step.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r06}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ds
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ds -o t -- python $ROOT/tools/profile_driver_step.py > $OUT/driver_step.log 2>&1
cd $ROOT
python tools/shorten_stats.py $(ls /tmp/ds/*/t_kernel_stats.csv /tmp/ds/t_kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_driver_step_kernel_stats.csv
python tools/driver_step_timeline.py /tmp/ds $OUT/${TAG}_driver_step_timeline.txt
python tools/driver_step_sequence.py /tmp/ds $OUT/${TAG}_driver_step_sequence.txt
