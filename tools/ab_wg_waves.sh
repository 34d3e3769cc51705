#!/bin/bash
# The one-wavefront step kernels launched as workgroups of 1 / 2 / 4 independent wavefronts (CROWDSTEP_WG_WAVES): what the dispatcher's
# ramp over many small workgroups costs.  usage (GPU box): tools/ab_wg_waves.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ab_wg_waves.txt
: > $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-gym-step --full-json $R/gpurun_out/ab_full.json --steps 50 --warmup 20 --repeats 20"
show() { python3 -c "
import json; d=json.load(open('$R/gpurun_out/ab_full.json')); r=d['roofline']; print('$1', round(r['kernel_avg_ms']*1e3,2), round(r['kernel_median_ms']*1e3,2), 'us finite', d.get('finite_fraction'))" >> $OUT; }
for rep in 1 2; do
  for wg in 1 2 4; do
    export CROWDSTEP_WG_WAVES=$wg
    $B > /dev/null 2>&1; show "cfg3 wg=$wg"
    $B --substeps 1 > /dev/null 2>&1; show "cfg3_1substep wg=$wg"
    $B --worlds 8192 --agents 50 --scenario circle --static 3 --walls --device-generator > /dev/null 2>&1; show "cfg5shard wg=$wg"
    $B --worlds 32768 --repeats 5 > /dev/null 2>&1; show "cfg3_32768 wg=$wg"
    $B --per-agent-params > /dev/null 2>&1; show "peragent wg=$wg"
    $B --worlds 1000 --agents 17 > /dev/null 2>&1; show "1000x17 wg=$wg"
  done
done
cat $OUT
