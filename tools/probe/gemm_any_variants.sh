#!/bin/bash
# ablation builds of csrc/dense_any.hip (GPU box): the any-shape product at 18432 x 300 x 300 with the operand loads / the
# fragment reads compiled out -- what is left of the launch time when a phase is gone
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in base NO_GLOBAL NO_FRAG "NO_GLOBAL -DGA_PROBE_NO_FRAG"; do
  flags=""; [ "$v" != base ] && flags="-DGA_PROBE_$v"
  mkdir -p /tmp/ga && cp -r geometrics_amd/lib /tmp/ga/ 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -fno-slp-vectorize $flags -I include -I geometrics_amd/csrc \
      -c geometrics_amd/csrc/dense_any.hip -o /tmp/ga/dense_any.o || exit 1
  objs=$(ls geometrics_amd/lib/*.o | grep -v dense_any.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs /tmp/ga/dense_any.o -o /tmp/ga/libprobe.so || exit 1
  echo "== $v"
  GEOM_LIB_OVERRIDE=/tmp/ga/libprobe.so GEOM_ALLOW_STALE_LIB=1 SHAPES=300x300 python tools/time_gemm_any.py 2>&1 | grep "18432 x"
done
