#!/bin/bash
# Round-6 profile set, GPU box, repo root:   bash tools/profile_round_r06.sh    (everything lands in gpurun_out/r06/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
bash tools/profile_round.sh r06 > $OUT/profile_round.log 2>&1            # step kernel stats, timeline, three PMC passes -> json
mv gpurun_out/r06_step_kernel_stats.csv gpurun_out/r06_step_timeline.txt gpurun_out/r06_pmc_counters.json $OUT/ 2>/dev/null
bash tools/prof_dense.sh > /dev/null 2>&1
python tools/dense_pmc_json.py gpurun_out $OUT/r06_dense_pmc.json
TAG=r06 bash tools/profile_driver_step.sh > $OUT/profile_driver_step.log 2>&1    # the reference driver's whole step: kernel stats + one replay + launch sequence
python tools/time_deform_layer.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_deform_layer_costs.txt     # what each piece of the one-launch layer costs
bash tools/probe/db_variants.sh run 2>&1 | grep -v "GEOM_LIB_OVERRIDE\|amdgpu.ids" >> $OUT/r06_deform_layer_costs.txt
bash tools/probe/db_stamps.sh run 2>&1 | grep -v "GEOM_LIB_OVERRIDE\|amdgpu.ids" > $OUT/r06_deform_layer_stamps.txt
python tools/time_driver_step.py --zero-edit 2>&1 | grep -v amdgpu.ids > $OUT/r06_driver_step_fused_vs_separate.txt
python tools/time_split_bf16.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_split_bf16_experiment.txt
python tools/time_force_dp.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_dp_fixed_cost.txt
cp $OUT/r06_pmc_counters.json $OUT/r06_step_kernel_stats.csv profiles/     # (on the box: the bench line reads its counter-derived fields from the profiles of THESE sources)
python bench.py > $OUT/r06_bench_default_run.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2>/dev/null
ls -la $OUT
