#!/bin/bash
# correctness + kernel times + counters of the culled Chamfer scan in one call
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c1; timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/c1 -o c1 --output-format csv -- python $R/tools/time_culled_nn.py > /tmp/c1.log 2>&1
grep -c identical /tmp/c1.log; grep -A3 DIFFERENT /tmp/c1.log | head -20; grep " us " /tmp/c1.log
grep -E "nn_|cull" /tmp/c1/c1_kernel_stats.csv | cut -c1-60,100-300
python $R/tools/probe/stats_tool.py 2>/dev/null | grep tiles
bash $R/tools/probe/cull_pmc.sh 2>/dev/null | grep -E "^brute |^culled "
bash $R/tools/probe/cull_knobs.sh
