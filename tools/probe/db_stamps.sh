#!/bin/bash
# Probe build of csrc/deform_block.hip with shader-clock stamps at the phase boundaries (-DDB_PROBE_STAMPS), linked from the
# product's other objects and loaded through GEOM_LIB_OVERRIDE by tools/probe/db_stamps.py.   usage: db_stamps.sh build|run
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  mkdir -p tools/probe/bin
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -fno-slp-vectorize \
    -mllvm -amdgpu-mfma-vgpr-form=1 -DDB_PROBE_STAMPS $DB_EXTRA -I include -I geometrics_amd/csrc -c geometrics_amd/csrc/deform_block.hip -o /tmp/db_stamps.o
  objs=$(ls geometrics_amd/lib/*.o | grep -v deform_block.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs /tmp/db_stamps.o -o tools/probe/bin/libgeom_db_stamps.so
else
  GEOM_ALLOW_STALE_LIB=1 GEOM_LIB_OVERRIDE=$PWD/tools/probe/bin/libgeom_db_stamps.so python tools/probe/db_stamps.py
fi
