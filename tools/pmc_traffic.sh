#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE) and VALU instruction counts per kernel of one bench step.
# Three separate passes (TCC counters do not fit one pass; PMC runs carry --kernel-trace only), eager launches so
# that every kernel is an individual dispatch.  Run on the GPU box from the repo root:
#     bash tools/pmc_traffic.sh   ->  gpurun_out/pmc/{fetch,write,valu}/..., then
#     python tools/pmc_traffic_json.py gpurun_out/pmc profiles/r03_pmc_counters.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "valu SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
    set -- $pass
    name=$1; shift
    out=$ROOT/gpurun_out/pmc/$name
    rm -rf $out; mkdir -p $out
    rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $out -o p -- \
        python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --launch eager --steps-only > $out/log.txt 2>&1
    ls $out | tr '\n' ' '; echo
done
