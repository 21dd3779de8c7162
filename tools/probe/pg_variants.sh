#!/bin/bash
# Ablation builds of csrc/pooling.hip (probe macros PG_PROBE_* / PL_PROBE_*): what the reads, the stores and the bare launch
# of the pooling's gather / forward / vertex-sum bodies each cost.   usage: pg_variants.sh build | run
set -e
cd "$(dirname "$0")/../.."
VARIANTS=("base:" "norows:-DPG_PROBE_NO_ROWS -DPL_PROBE_NO_LOADS" "nostore:-DPG_PROBE_NO_STORE -DPL_PROBE_NO_STORE" "bare:-DPG_PROBE_BARE -DPL_PROBE_NO_LOADS -DPL_PROBE_NO_STORE")
if [ "$1" = build ]; then
  mkdir -p tools/probe/bin
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -fno-slp-vectorize \
      $flags -I include -I geometrics_amd/csrc -c geometrics_amd/csrc/pooling.hip -o /tmp/pg_$name.o &
  done
  wait
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    objs=$(ls geometrics_amd/lib/*.o | grep -v pooling.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs /tmp/pg_$name.o -o tools/probe/bin/libgeom_pg_$name.so
  done
else
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    echo "== $name"
    GEOM_ALLOW_STALE_LIB=1 GEOM_LIB_OVERRIDE=$PWD/tools/probe/bin/libgeom_pg_$name.so python tools/time_pool_gather.py 2>&1 | grep "per call"
  done
fi
