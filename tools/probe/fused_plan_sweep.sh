#!/bin/bash
# the step at several shard sizes with the boundary launches off / forward only / forward + backward (GEOM_FUSED_PLAN)
for m in ${@:-12 16 32 64}; do
  for plan in off fwd all; do
    GEOM_FUSED_PLAN=$plan python bench.py --meshes-per-gpu $m --steps 100 --warmup 10 --no-cpu-baseline --steps-only --clock-warmup-ms 0 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m meshes, plan $plan: %.1f us/step, %.0f meshes/s' % (1e3*d['ms_per_step'], d['value']))"
  done
done
