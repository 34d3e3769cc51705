#!/bin/bash
# cfg5's per-GPU shard (8192 x 50 + walls) and the 65536-world job with the wall pairs (default) and without (CROWDSTEP_WALL_PAIRS=0),
# per reach of the wall force (CROWDSTEP_WALL_EFOLDS; usage: tools/wall_pairs_ab.sh "36 20")
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/wall_pairs_ab.txt
: > $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-gym-step --full-json $R/gpurun_out/ab_full.json --agents 50 --scenario circle --static 3 --walls --device-generator"
for ef in ${1:-36}; do
  for wp in 1 0; do
    export CROWDSTEP_WALL_PAIRS=$wp CROWDSTEP_WALL_EFOLDS=$ef
    for cfg in "shard20-70|--worlds 8192 --steps 50 --warmup 20 --repeats 10" "shard0-20|--worlds 8192 --steps 20 --warmup 0 --repeats 10" "job65536_20-70|--worlds 65536 --steps 50 --warmup 20 --repeats 5"; do
      $B ${cfg#*|} > /dev/null 2>&1
      python3 -c "
import json; d=json.load(open('$R/gpurun_out/ab_full.json')); r=d['roofline']; print('efolds=$ef wall_pairs=$wp ${cfg%%|*}', round(r['kernel_avg_ms']*1e3,2), 'us frac', round(r['frac'],3))" >> $OUT
    done
  done
done
cat $OUT
