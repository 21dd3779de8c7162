#!/bin/bash
# Ablation builds of csrc/zn_stack.hip (probe macros ZS_PROBE_*): one libgeom_hip variant per macro set, linked from the
# product's other objects, loaded through GEOM_LIB_OVERRIDE by tools/time_fused_layer.py.   usage: zs_variants.sh build|run
set -e
cd "$(dirname "$0")/../.."
VARIANTS=("base:" "nostore:-DZS_PROBE_NO_STORE" "nogather:-DZS_PROBE_NO_GATHER" "stamps:-DZS_PROBE_STAMPS")
if [ "$1" = build ]; then
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -fno-slp-vectorize \
      -mllvm -amdgpu-mfma-vgpr-form=1 $flags $ZS_EXTRA -I include -I geometrics_amd/csrc -c geometrics_amd/csrc/zn_stack.hip -o /tmp/zs_$name.o &
  done
  wait
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    objs=$(ls geometrics_amd/lib/*.o | grep -v zn_stack.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs /tmp/zs_$name.o -o tools/probe/bin/libgeom_zs_$name.so
  done
else
  shift
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    case $name in stamps*) continue;; esac
    echo "== $name"
    GEOM_ALLOW_STALE_LIB=1 GEOM_LIB_OVERRIDE=$PWD/tools/probe/bin/libgeom_zs_$name.so python tools/time_fused_layer.py "$@" 2>&1 | grep "fused"
  done
fi
