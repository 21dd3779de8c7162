cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_step
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_step -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --steps-only > $R/gpurun_out/prof_step.log 2>&1
tail -1 $R/gpurun_out/prof_step.log | cut -c1-200
cd $R && python tools/step_timeline.py gpurun_out/prof_step
