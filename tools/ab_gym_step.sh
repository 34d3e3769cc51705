#!/bin/bash
# Same-box A/B of two builds of libcrowdstep.so on the device-resident Gym step (bench.py gym_step_figures): tools/ab_gym_step.sh <other.so>
OTHER=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2 3; do
  for tag in tree other; do
    if [ $tag = other ]; then export CROWDSTEP_LIB=$OTHER; else unset CROWDSTEP_LIB; fi
    python3 -c "
import sys; sys.path.insert(0, '$R')
import bench
g = bench.gym_step_figures(4096, 25)
print('$tag', {k: round(v, 1) for k, v in g.items() if k in ('no_reset', 'same_step', 'next_step')})" 2>&1 | grep -v amdgpu.ids
  done
done
