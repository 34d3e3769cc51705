#!/bin/bash
# usage (GPU box): tools/pmc_pass.sh TAG "COUNTER1 COUNTER2 ..." [bench args]  -- one rocprofv3 PMC pass, per-kernel means
TAG=$1; CTRS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "k_sfm_step" not in k and "k_orca" not in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean/dispatch {sum(v)/len(v):14.1f}  (n={len(v)})")
PY
