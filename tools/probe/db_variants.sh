#!/bin/bash
# Ablation builds of csrc/deform_block.hip (probe macros DB_PROBE_*): what the weight-slice traffic and the MFMAs each cost.
# usage: db_variants.sh build | run
set -e
cd "$(dirname "$0")/../.."
VARIANTS=("base:" "hotslice:-DDB_PROBE_HOT_SLICE" "fewmfma:-DDB_PROBE_FEW_MFMA" "both:-DDB_PROBE_HOT_SLICE -DDB_PROBE_FEW_MFMA")
if [ "$1" = build ]; then
  mkdir -p tools/probe/bin
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -fno-slp-vectorize \
      -mllvm -amdgpu-mfma-vgpr-form=1 $flags -I include -I geometrics_amd/csrc -c geometrics_amd/csrc/deform_block.hip -o /tmp/db_$name.o &
  done
  wait
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    objs=$(ls geometrics_amd/lib/*.o | grep -v deform_block.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs /tmp/db_$name.o -o tools/probe/bin/libgeom_db_$name.so
  done
else
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    echo "== $name"
    GEOM_ALLOW_STALE_LIB=1 GEOM_LIB_OVERRIDE=$PWD/tools/probe/bin/libgeom_db_$name.so python tools/time_deform_layer.py 2>&1 | grep "full"
  done
fi
