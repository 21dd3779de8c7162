#!/bin/sh
# Compile the reference's own CPU nearest-neighbour search into oracle/_ref/.
#
# Source: $REF/old_GEOMetrics/chamfer_distance/src/my_lib.c, function `nnsearch`
# (lines 4-26).  The translation unit as a whole cannot be built here: its first
# line includes the torch-0.4 era <TH/TH.h>, which this image does not have, and
# we do not write stand-in headers.  `nnsearch` itself is plain C99 that touches
# no TH symbol, so this recipe streams exactly that function -- verbatim, from
# the reference file where it lies -- into gcc.  Nothing is copied into the repo;
# the only output is _ref/libref_nnsearch.so (git-ignored, travels with gpurun).
# The TH-typed glue (nnd_forward / nnd_backward, my_lib.c:28-111) is NOT built.
#
# Flags: -O2 -ffp-contract=off == what a default x86-64 `gcc -O2` build of the
# legacy FFI extension computes (baseline x86-64 has no FMA to contract into).
set -e
REF="${1:-/root/reference}"
SRC="$REF/old_GEOMetrics/chamfer_distance/src/my_lib.c"
HERE="$(cd "$(dirname "$0")" && pwd)"
if [ ! -f "$SRC" ]; then
    echo "build_ref.sh: reference checkout not present ($SRC); keeping any prebuilt _ref/" >&2
    exit 0
fi
mkdir -p "$HERE/_ref"
sed -n '/^void nnsearch(/,/^}/p' "$SRC" \
  | gcc -O2 -std=c99 -ffp-contract=off -fPIC -shared -x c - -o "$HERE/_ref/libref_nnsearch.so"
echo "built $HERE/_ref/libref_nnsearch.so"
# The same source text with FMA contraction allowed: gcc emits vmulss(dy,dy), vfmadd(dx,dx,.),
# vfmadd(dz,dz,.) = fma(dz,dz,fma(dx,dx,dy*dy)) -- the arithmetic of GEOM_FLAG_NN_FMA.
sed -n '/^void nnsearch(/,/^}/p' "$SRC" \
  | gcc -O2 -std=c99 -mfma -ffp-contract=fast -fPIC -shared -x c - -o "$HERE/_ref/libref_nnsearch_fma.so"
echo "built $HERE/_ref/libref_nnsearch_fma.so"
