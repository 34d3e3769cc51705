#!/bin/bash
# usage (GPU box): tools/pmc_stall.sh TAG [bench.py arguments]
# Two SQ passes (8 slots each) of one bench workload with the LDS / wait split of the step kernel: where a wavefront's cycles go
# (VALU issue, LDS issue stall, parked on s_waitcnt) -- the question behind HISTORY.md Part II 4.1d (cfg5's wall pass is not issue-bound).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/stall_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
A="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY"
B="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
for p in A B; do
  eval C=\$$p
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/run$p -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 20 --repeats 1 "$@" > /dev/null 2> $O/log$p.txt || { echo "pass $p failed"; tail -5 $O/log$p.txt; exit 1; }
  cp "$(find $O/run$p -name '*counter_collection.csv' | head -1)" $O/pass$p.csv; rm -rf $O/run$p
done
python3 - $O/passA.csv $O/passB.csv <<'PY' | tee $O/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in sys.argv[1:]:
    seen = collections.defaultdict(float)
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "k_sfm_step" not in k and "k_orca" not in k and "k_big" not in k: continue
        acc[k][(fn[-5], r["Counter_Name"])] += float(r["Counter_Value"])
for k, c in acc.items():
    print(k[:90])
    for p in "AB":
        w, cyc = c[(p, "SQ_WAVES")], c[(p, "SQ_WAVE_CYCLES")]
        if not w: continue
        for (pp, name), v in sorted(c.items()):
            if pp != p or name in ("SQ_WAVES",): continue
            if name.startswith("SQ_INSTS"): print(f"   {name:24s} {v / w / 20:10.1f} per wave-substep")
            else: print(f"   {name:24s} {v / w / 20:10.1f} quad-cycles per wave-substep = {v / cyc:6.3f} of the wave's cycles")
PY
