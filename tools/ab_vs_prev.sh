#!/bin/bash
# same-box A/B of this tree (and of its variant libraries: AB_VARIANTS="rx0 ...", built by csrc/build.py build(variant=...)) against a checkout of an
# earlier commit under _ab_r5/ (git worktree add -f _ab_r5 <commit>; build there): headline and cfg2, interleaved twice.
#   bash tools/ab_vs_prev.sh > gpurun_out/ab_vs_prev.txt
run() {  # tree, label, lib, bench flags
  ( cd $1 && CROWDSTEP_LIB=$3 python bench.py --gpus 1 --steps 50 --warmup 20 --no-other-configs --no-cpu-baseline --no-gym-step $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', '${4:-cfg3}', 'kernel_us', d['kernel_us'], 'ms_per_step', d['ms_per_step'], d['roofline']['variant'], d['build_id'][:8])" )
}
for rep in 1 2; do
  for cfg in "" "--model sfm_helbing --agents 10 --scenario circle" ${AB_EXTRA_CFG:+"$AB_EXTRA_CFG"}; do
    run . this "" "$cfg"
    for v in $AB_VARIANTS; do run . $v $PWD/social_navigation_pyenvs_amd/libcrowdstep_$v.so "$cfg"; done
    [ -d _ab_r5 ] && run _ab_r5 prev "" "$cfg"
  done
done
