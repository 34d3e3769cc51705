#!/bin/bash
# Kernel time of one launch against the number of substeps it fuses: the intercept is what a launch costs before the first substep
# (descriptor reads, parameter rows, the wall table, the state load) and after the last (commit).   tools/launch_floor.sh <bench.py arguments>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for s in ${SUBSTEPS:-1 2 5 10 20 40}; do
  python3 $R/bench.py --substeps $s --warmup 20 --steps 50 --repeats 10 --no-cpu-baseline --no-other-configs --no-gym-step --full-json $R/gpurun_out/lf_full.json "$@" > /dev/null 2>&1 || exit 1
  python3 -c "
import json; d=json.load(open('$R/gpurun_out/lf_full.json')); r=d['roofline']
print('substeps %3d | kernel_us %8.2f | %s' % ($s, r['kernel_avg_ms']*1e3, r['variant'][:90]))"
done
