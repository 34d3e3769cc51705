#!/bin/bash
# Same-box A/B of two builds of libcrowdstep.so: tools/ab_lib.sh <other.so> [bench.py arguments]  (alternates base / other three times)
# (reads the FULL result -- gpurun_out/bench_full.json --: the compact stdout line keeps the mean kernel time only)
OTHER=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2 3; do
  for tag in tree other; do
    if [ $tag = other ]; then export CROWDSTEP_LIB=$OTHER; else unset CROWDSTEP_LIB; fi
    python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 100 --warmup 20 --repeats 20 --full-json $R/gpurun_out/ab_full.json "$@" > /dev/null 2>&1
    python3 -c "
import json; d=json.load(open('$R/gpurun_out/ab_full.json')); r=d['roofline']; g=d.get('gym_step') or {}; print('$tag', round(r['kernel_avg_ms']*1e3,2), round(r['kernel_median_ms']*1e3,2), 'us', r['variant'][:60], '| gym_step', {k: round(v,1) for k,v in g.items() if k in ('no_reset','same_step','next_step')})"
  done
done
