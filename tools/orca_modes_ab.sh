#!/bin/bash
# cfg4 kernel time per arithmetic mode (CROWDSTEP_ORCA_MATH), per linearProgram2 form (CROWDSTEP_ORCA_LP2=walk: the round-4 form) and per
# linearProgram3 form (CROWDSTEP_ORCA_LP3=iter: linearProgram1 inside the lane groups' walk, the round-2..4 form)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/orca_modes_ab.txt
: > $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-gym-step --full-json $R/gpurun_out/ab_full.json --model orca --scenario circle --steps 20"
for cfg in "${@:-table:rows}"; do
  lp2=${cfg%%:*}; lp3=${cfg#*:}
  for m in ${AB_ORCA_MATH:-exact fast fma}; do
    export CROWDSTEP_ORCA_MATH=$m CROWDSTEP_ORCA_LP2=$lp2 CROWDSTEP_ORCA_LP3=$lp3
    for ph in 25 0; do
      $B --warmup $ph > /dev/null 2>&1
      python3 -c "
import json; d=json.load(open('$R/gpurun_out/ab_full.json')); r=d['roofline']; print('lp2=$lp2 lp3=$lp3 math=$m warmup=$ph', round(r['kernel_avg_ms']*1e3,2), 'us', r['variant'][:90])" >> $OUT
    done
  done
done
cat $OUT
