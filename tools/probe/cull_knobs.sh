#!/bin/bash
# kernel time of the culled Chamfer scan under each ablation knob (tools/probe/libcullstats.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in 32 64 128 257 1 2 4 256 0; do
  rm -rf /tmp/ck; rocprofv3 --kernel-trace --stats -d /tmp/ck -o k --output-format csv -- python $R/tools/probe/stats_tool.py $k > /dev/null 2>&1
  echo "knob $k: $(grep culled_kernel /tmp/ck/k_kernel_stats.csv | cut -d, -f2-4 | tr '\n' ' ')"
done
