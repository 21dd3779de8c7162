#!/bin/bash
# the ragged encoder step alone under rocprofv3 --kernel-trace --stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
python tools/probe/encoder_only.py 20 2>&1 | grep -v amdgpu.ids | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_enc
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_enc -o e -- python $ROOT/tools/probe/encoder_only.py 10 > /dev/null 2>&1
cd $ROOT
python tools/shorten_stats.py $(ls gpurun_out/prof_enc/*/e_kernel_stats.csv gpurun_out/prof_enc/e_kernel_stats.csv 2>/dev/null | head -1) gpurun_out/encoder_kernel_stats.csv
head -40 gpurun_out/encoder_kernel_stats.csv | cut -c1-150
