#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, two passes: TCC counters do not fit one) of the captured driver step's kernels:
# rocprofv3 --kernel-trace --pmc over tools/profile_driver_step.py, averaged per kernel name by tools/pmc_driver_step.py
#   GPU box, repo root:   TAG=r06 bash tools/pmc_driver_step.sh   ->  gpurun_out/$TAG/${TAG}_driver_step_pmc.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r06}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/dsp_$c
    rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/dsp_$c -o p -- python $ROOT/tools/profile_driver_step.py > $OUT/pmc_driver_step_$c.log 2>&1
done
cd $ROOT
python tools/pmc_driver_step.py /tmp/dsp_FETCH_SIZE /tmp/dsp_WRITE_SIZE $OUT/${TAG}_driver_step_pmc.txt
