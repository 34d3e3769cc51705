#!/bin/bash
# Same-box A/B of two builds of libcrowdstep.so: tools/ab_lib.sh <other.so> [bench.py arguments]  (alternates base / other three times)
OTHER=$1; shift
for rep in 1 2 3; do
  for tag in tree other; do
    if [ $tag = other ]; then export CROWDSTEP_LIB=$OTHER; else unset CROWDSTEP_LIB; fi
    python3 bench.py --no-cpu-baseline --no-other-configs --steps 100 --warmup 20 --repeats 20 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', round(r['kernel_avg_ms']*1e3,2), round(r['kernel_median_ms']*1e3,2), 'us', r['variant'][:60])"
  done
done
