#!/bin/bash
# Same-box A/B of variant builds (csrc/build.py --variant=<name> -> libcrowdstep_<name>.so): tools/ab_libs.sh name1 name2 ...   ("tree" = the product build)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ab_libs.txt
: > $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-gym-step --full-json $R/gpurun_out/ab_full.json"
show() { python3 -c "
import json; d=json.load(open('$R/gpurun_out/ab_full.json')); r=d['roofline']; print('$1', round(r['kernel_avg_ms']*1e3,2), round(r['kernel_median_ms']*1e3,2), 'us', r['variant'][:90])" >> $OUT; }
for rep in 1 2; do
  for tag in "$@"; do
    if [ $tag = tree ]; then unset CROWDSTEP_LIB; else export CROWDSTEP_LIB=$R/social_navigation_pyenvs_amd/libcrowdstep_$tag.so; fi
    $B --steps 100 --warmup 20 --repeats 20 > /dev/null 2>&1; show "cfg3 $tag"
    if [ -z "$AB_ORCA_ONLY" ]; then
      $B --steps 50 --warmup 20 --repeats 10 --worlds 8192 --agents 50 --scenario circle --static 3 --walls --device-generator > /dev/null 2>&1; show "cfg5shard $tag"
      $B --steps 100 --warmup 20 --repeats 20 --agents 10 --model sfm_helbing --scenario circle > /dev/null 2>&1; show "cfg2 $tag"
    fi
    for m in ${AB_ORCA_MATH:-exact fma}; do
      export CROWDSTEP_ORCA_MATH=$m
      $B --model orca --scenario circle --steps 20 --warmup 25 > /dev/null 2>&1; show "cfg4_dense $tag $m"
      $B --model orca --scenario circle --steps 20 --warmup 0 > /dev/null 2>&1; show "cfg4_first20 $tag $m"
    done
    unset CROWDSTEP_ORCA_MATH
  done
done
cat $OUT
