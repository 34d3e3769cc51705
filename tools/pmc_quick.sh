#!/bin/bash
# one PMC pass (SQ issue counters) of the default bench workload: VALU / LDS instructions per wavefront-substep of the step kernel.
# usage: tools/pmc_quick.sh TAG [bench.py arguments]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/pmcq_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/run -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 20 --repeats 1 "$@" > /dev/null 2> $O/log.txt || { tail -5 $O/log.txt; exit 1; }
python3 - "$(find $O/run -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_sfm_step" not in k and "k_orca" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, c in acc.items():
    w = c["SQ_WAVES"]
    print(k[:70], "| dispatches", n[k], "| VALU/wave-substep %.1f LDS %.1f | wave quad-cycles/substep %.1f | VALU active %.3f wait_any %.3f" % (
        c["SQ_INSTS_VALU"] / w / 20, c["SQ_INSTS_LDS"] / w / 20, c["SQ_WAVE_CYCLES"] / w / 20, c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
