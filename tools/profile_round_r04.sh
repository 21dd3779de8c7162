#!/bin/bash
# Round-4 profile set, GPU box, repo root:   bash tools/profile_round_r04.sh    (everything lands in gpurun_out/r04/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
bash tools/profile_round.sh r04 > $OUT/profile_round.log 2>&1            # step kernel stats, timeline, three PMC passes -> json
mv gpurun_out/r04_step_kernel_stats.csv gpurun_out/r04_step_timeline.txt gpurun_out/r04_pmc_counters.json $OUT/ 2>/dev/null
bash tools/prof_dense.sh > /dev/null 2>&1
python tools/dense_pmc_json.py gpurun_out $OUT/r04_dense_pmc.json
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ts && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts -o t -- \
      python $ROOT/tools/profile_training_shape.py > $OUT/training_shape.log 2>&1 )
python tools/shorten_stats.py $(ls /tmp/ts/*/t_kernel_stats.csv /tmp/ts/t_kernel_stats.csv 2>/dev/null | head -1) $OUT/r04_training_shape_kernel_stats.csv
bash tools/probe/dp_trace.sh > $OUT/r04_dp_step_timeline.txt 2>&1
python tools/time_force_dp.py 2>&1 | grep force_dp > $OUT/time_force_dp.txt
python bench.py > $OUT/r04_bench_default_run.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2>/dev/null
ls -la $OUT
