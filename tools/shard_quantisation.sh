#!/bin/bash
# Kernel time of the cfg5 shard build against the number of worlds (= wavefronts) of the launch: does the time move in steps of
# one "round" of resident wavefronts (3 per SIMD x 1024 SIMDs = 3072)?   tools/shard_quantisation.sh [extra bench.py arguments]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for w in ${WORLDS:-1024 2048 3072 4096 6144 7168 8192 9216 10240 12288}; do
  python3 $R/bench.py --agents 50 --scenario circle --walls --static 3 --device-generator --worlds $w --warmup 20 --steps 50 --repeats 10 \
      --no-cpu-baseline --no-other-configs --no-gym-step --full-json $R/gpurun_out/sq_full.json "$@" > /dev/null 2>&1 || exit 1
  python3 -c "
import json; d=json.load(open('$R/gpurun_out/sq_full.json')); r=d['roofline']
print('worlds %6d | kernel_us %8.2f | us per 1024 worlds %7.2f | frac %.3f | %s' % ($w, r['kernel_avg_ms']*1e3, r['kernel_avg_ms']*1e3/($w/1024), r['frac'], r['variant'][:90]))"
done
