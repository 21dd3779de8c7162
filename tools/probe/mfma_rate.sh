#!/bin/bash
# GPU box: the MFMA-only probe with the reported engine clock sampled beside it (rocm-smi every 100 ms while it runs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/mfma_pipe_rate.txt}
( for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; sleep 0.1; done ) > /tmp/sclk.txt &
SMI=$!
$R/tools/probe/bin/mfma_rate > /tmp/mfma.txt 2>&1
wait $SMI
{ cat /tmp/mfma.txt; echo "# rocm-smi --showclocks (sclk) sampled every 0.1 s while the probe ran:"; sort /tmp/sclk.txt | uniq -c | sed 's/^/#   /'; } > $OUT
cat $OUT
