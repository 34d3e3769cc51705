#!/bin/bash
# Same-box A/B of variant builds (csrc/build.py build(variant=<name>) -> libcrowdstep_<name>.so): bash tools/ab_libs.sh name1 name2 ...   ("tree" = the product build)
# (the one generic A/B script: the round-specific ones -- ab_fold, ab_round5, ab_wg_waves, wall_pairs_ab, orca_modes_ab ... -- were deleted in
#  round 6; their results are in HISTORY.md / profiles/archive.)  A run that fails prints FAILED under its label instead of the previous run's figures.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ab_libs.txt
: > $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-gym-step --full-json $R/gpurun_out/ab_full.json"
run() {   # label, bench flags...: the figures of THIS run or "FAILED" (never a stale ab_full.json)
  local label=$1; shift
  rm -f $R/gpurun_out/ab_full.json
  if $B "$@" > /dev/null 2>&1 && [ -f $R/gpurun_out/ab_full.json ]; then python3 -c "
import json; d=json.load(open('$R/gpurun_out/ab_full.json')); r=d['roofline']; print('$label', round(r['kernel_avg_ms']*1e3,2), round(r['kernel_median_ms']*1e3,2), 'us', r['variant'][:90], d.get('build_id','')[:8])" >> $OUT
  else echo "$label FAILED" >> $OUT; fi; }
for rep in 1 2; do
  for tag in "$@"; do
    if [ $tag = tree ]; then unset CROWDSTEP_LIB; else export CROWDSTEP_LIB=$R/social_navigation_pyenvs_amd/libcrowdstep_$tag.so; fi
    run "cfg3 $tag" --steps 100 --warmup 20 --repeats 20
    if [ -z "$AB_ORCA_ONLY" ]; then
      run "cfg5shard $tag" --steps 50 --warmup 20 --repeats 10 --worlds 8192 --agents 50 --scenario circle --static 3 --walls --device-generator
      run "cfg2 $tag" --steps 100 --warmup 20 --repeats 20 --agents 10 --model sfm_helbing --scenario circle
    fi
    for m in ${AB_ORCA_MATH:-exact fma}; do
      run "cfg4_dense $tag $m" --model orca --scenario circle --steps 20 --warmup 25 --orca-math $m
      run "cfg4_first20 $tag $m" --model orca --scenario circle --steps 20 --warmup 0 --orca-math $m
    done
  done
done
cat $OUT
