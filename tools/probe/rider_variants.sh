#!/bin/bash
# The rider workgroups of dense_split_kernel_with_riders (csrc/dense_gemm.hip): how many, and how late they start.
# Builds the library with -DDG_RIDER_WGS / -DDG_RIDER_DELAY_US variants (in the build container), then on the GPU box
#   bash tools/probe/rider_variants.sh run
# prints the step time and the two launches' durations per variant (GEOM_LIB_OVERRIDE: never the product).
set -e
cd "$(dirname "$0")/../.."
VARIANTS="1024:0 1024:5 1024:0:3 512:0:3 256:0 128:0 128:5 64:5 16:5"
if [ "$1" != run ]; then
    python -m geometrics_amd.build > /dev/null
    mkdir -p tools/probe/bin
    for v in $VARIANTS; do
        IFS=: read w d pr <<< "$v"; extra=""; [ -n "$pr" ] && extra="-DDG_SPLIT_PRIO=$pr"; d=${d}${pr:+_p$pr}
        ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -fno-slp-vectorize -w \
            -mllvm -amdgpu-mfma-vgpr-form=1 -DDG_RIDER_WGS=$w -DDG_RIDER_DELAY_US=${d%%_*} $extra -I include -I geometrics_amd/csrc \
            -c geometrics_amd/csrc/dense_gemm.hip -o /tmp/dense_gemm_${w}_${d}.o
          objs=$(ls geometrics_amd/lib/*.o | grep -v dense_gemm.o)
          /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs /tmp/dense_gemm_${w}_${d}.o \
            -o tools/probe/bin/libgeom_riders_${w}_${d}.so ) &
    done
    wait
    ls -la tools/probe/bin/libgeom_riders_*
    exit 0
fi
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
one() {   # $1 = label, rest = env/flags
    label=$1; shift
    rm -rf /tmp/rp
    env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o r -- python $ROOT/bench.py --steps 100 --warmup 10 \
        --no-cpu-baseline --steps-only $FLAGS > /tmp/rp.log 2>&1
    python $ROOT/tools/step_timeline.py /tmp/rp | awk -v l="$label" '/dense_split/ {s=$NF} /dense_reduce/ {r=$NF} /launches/ {print l, "split", s, "reduce", r, $0}'
}
FLAGS="" one "no riders      " GEOM_ALLOW_STALE_LIB=1
FLAGS=--riders
for v in $VARIANTS; do
    IFS=: read w d pr <<< "$v"; d=${d}${pr:+_p$pr}
    one "wgs $w delay $d" GEOM_ALLOW_STALE_LIB=1 GEOM_LIB_OVERRIDE=$ROOT/tools/probe/bin/libgeom_riders_${w}_${d}.so
done
