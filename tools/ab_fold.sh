#!/bin/bash
# What the take-over code of cs_gym_step_staged (FOLD, compiled into the step kernels without walls) costs the plain cs_step launches:
# the product build against a variant built with -DCS_NO_FOLD.  usage (GPU box): tools/ab_fold.sh   (build the variant here first:
# python3 -c "from social_navigation_pyenvs_amd.csrc import build as b; b.build(variant='nofold', extra_flags=['-DCS_NO_FOLD'])")
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ab_fold.txt
: > $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-gym-step --full-json $R/gpurun_out/ab_full.json --steps 50 --warmup 20 --repeats 20"
show() { python3 -c "
import json; d=json.load(open('$R/gpurun_out/ab_full.json')); r=d['roofline']; print('$1', round(r['kernel_avg_ms']*1e3,2), round(r['kernel_median_ms']*1e3,2), 'us', r['variant'][:80])" >> $OUT; }
for rep in 1 2; do
  for tag in tree nofold; do
    if [ $tag = tree ]; then unset CROWDSTEP_LIB; else export CROWDSTEP_LIB=$R/social_navigation_pyenvs_amd/libcrowdstep_$tag.so; fi
    $B > /dev/null 2>&1; show "cfg3 $tag"
    $B --robot > /dev/null 2>&1; show "robot26 $tag"
    $B --per-agent-params > /dev/null 2>&1; show "peragent $tag"
    $B --agents 30 > /dev/null 2>&1; show "n30 $tag"
    $B --model hsfm_new_guo > /dev/null 2>&1; show "new_guo $tag"
    $B --model hsfm_new_moussaid > /dev/null 2>&1; show "moussaid $tag"
  done
done
cat $OUT
