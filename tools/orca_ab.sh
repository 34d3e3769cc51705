#!/bin/bash
# usage (GPU box): tools/orca_ab.sh  -- cfg4 kernel time, lp3_rows vs the static walk, r1 protocol (dense phase) and cycle average
for mode in rows static; do
  for proto in "--steps 20 --warmup 3 --repeats 1" "--steps 20 --warmup 3 --repeats 7"; do
    CROWDSTEP_ORCA_LP3=$mode python3 bench.py --model orca --scenario circle --no-other-configs --no-cpu-baseline $proto 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$mode | $proto | kernel_ms avg %.4f min %.4f max %.4f' % (r['kernel_avg_ms'], r['kernel_min_ms'], r['kernel_max_ms']))"
  done
done
