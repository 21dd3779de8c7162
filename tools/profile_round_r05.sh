#!/bin/bash
# Round-5 profile set, GPU box, repo root:   bash tools/profile_round_r05.sh    (everything lands in gpurun_out/r05/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd $ROOT
bash tools/profile_round.sh r05 > $OUT/profile_round.log 2>&1            # step kernel stats, timeline, three PMC passes -> json
mv gpurun_out/r05_step_kernel_stats.csv gpurun_out/r05_step_timeline.txt gpurun_out/r05_pmc_counters.json $OUT/ 2>/dev/null
bash tools/prof_dense.sh > /dev/null 2>&1
python tools/dense_pmc_json.py gpurun_out $OUT/r05_dense_pmc.json
bash tools/profile_driver_step.sh > $OUT/profile_driver_step.log 2>&1    # the reference driver's whole step: kernel stats + one replay
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/enc && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc -o e -- \
      python $ROOT/tools/probe/encoder_only.py 10 > $OUT/encoder.log 2>&1 )
python tools/shorten_stats.py $(ls /tmp/enc/*/e_kernel_stats.csv /tmp/enc/e_kernel_stats.csv 2>/dev/null | head -1) $OUT/r05_encoder_kernel_stats.csv
python tools/time_encoder.py 2>&1 | grep -v "amdgpu.ids\|Warning\|run_backward" > $OUT/r05_encoder_step.txt
python tools/time_force_dp.py 2>&1 | grep -v amdgpu.ids > $OUT/time_force_dp.txt
cp $OUT/r05_pmc_counters.json $OUT/r05_step_kernel_stats.csv profiles/     # (on the box: the bench line reads its counter-derived fields from the profiles of THESE sources)
python bench.py > $OUT/r05_bench_default_run.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2>/dev/null
ls -la $OUT
