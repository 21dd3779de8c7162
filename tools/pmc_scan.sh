#!/bin/bash
# SQ instruction-mix counters of the two arg-min scans (own pass: --kernel-trace + --pmc only).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_scan
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT -o p -- python $GRAFT_REPO_ROOT/tools/time_scans.py 8 > $OUT/log.txt 2>&1
ls $OUT
