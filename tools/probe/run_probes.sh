#!/bin/bash
# builds happen where hipcc is (the build container); the binaries travel in gpurun_out-less tools/probe/bin
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/probe/bin
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -I include -I geometrics_amd/csrc"
build() { name=$1; shift; /opt/rocm/bin/hipcc $F "$@" tools/probe/dense_probe.cpp -o tools/probe/bin/probe_$name & }
build base
build nofetch -DDG_PROBE_NO_FETCH
build noissue -DDG_PROBE_NO_ISSUE
build nostore -DDG_PROBE_NO_STORE
build nobar -DDG_PROBE_NO_BARRIER
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probe/mfma_rate.cpp -o tools/probe/bin/mfma_rate &
build mfmaonly -DDG_PROBE_NO_FETCH -DDG_PROBE_NO_ISSUE -DDG_PROBE_NO_STORE -DDG_PROBE_NO_BARRIER
build nomem -DDG_PROBE_NO_ISSUE -DDG_PROBE_NO_STORE
# the weight-gradient (split) kernel of the wide layer, piece by piece, one and two waves per SIMD (dense_probe ... prints "dW")
for kp in 1 2; do
    build split_kp${kp} -DDG_SPLIT_KP=$kp
    build split_kp${kp}_nofetch -DDG_SPLIT_KP=$kp -DDG_PROBE_NO_FETCH
    build split_kp${kp}_noissue -DDG_SPLIT_KP=$kp -DDG_PROBE_NO_ISSUE -DDG_PROBE_NO_STORE
    build split_kp${kp}_nostore -DDG_SPLIT_KP=$kp -DDG_PROBE_NO_STORE
    build split_kp${kp}_nog -DDG_SPLIT_KP=$kp -DDG_PROBE_NO_G
    build split_kp${kp}_nobar -DDG_SPLIT_KP=$kp -DDG_PROBE_NO_BARRIER
    build split_kp${kp}_mfmaonly -DDG_SPLIT_KP=$kp -DDG_PROBE_NO_FETCH -DDG_PROBE_NO_ISSUE -DDG_PROBE_NO_STORE -DDG_PROBE_NO_G -DDG_PROBE_NO_BARRIER
done
wait
ls -la tools/probe/bin

# counters + ablation knobs of the culled Chamfer scan (tools/probe/cull_report.sh, stats_tool.py, cull_knobs.sh, cull_pmc.sh)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -fno-slp-vectorize -DNN_CULL_STATS \
    -I include -I geometrics_amd/csrc -shared geometrics_amd/csrc/chamfer_nn.hip -o tools/probe/libcullstats.so

# the fused surface-scan launch with per-tile start / end stamps (tools/probe/scan_tile_stamps.py): the WHOLE library built with
# -DSCAN_TILE_STAMPS, so that the python operators run on it unchanged (GEOM_LIB_OVERRIDE=tools/probe/bin/libgeom_scan_stamps.so)
mkdir -p tools/probe/bin/stamps
for f in geometrics_amd/csrc/*.hip; do
    extra=""; case $f in *dense_gemm.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -fno-slp-vectorize -w -DSCAN_TILE_STAMPS $extra \
        -I include -I geometrics_amd/csrc -c $f -o tools/probe/bin/stamps/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc tools/probe/bin/stamps/*.o -o tools/probe/bin/libgeom_scan_stamps.so
rm -rf tools/probe/bin/stamps
ls -la tools/probe/bin/libgeom_scan_stamps.so
