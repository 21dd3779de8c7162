#!/bin/bash
# A/B on one box: the fused scan launch capped at 5 waves per SIMD (96 VGPRs, no spills; the shipped build) against 6 waves
# (80 VGPRs, 13 spilled; geometrics_amd/lib/alt_cap6.so built by hand from the same sources), alternating
for i in 1 2 3; do
  echo "cap 5:"; bash tools/probe/step_timeline_once.sh c5 | grep "surface_scan\|launches"
  echo "cap 6:"; GEOM_LIB_OVERRIDE=$PWD/geometrics_amd/lib/alt_cap6.so GEOM_ALLOW_STALE_LIB=1 bash tools/probe/step_timeline_once.sh c6 | grep "surface_scan\|launches"
done
