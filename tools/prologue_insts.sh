#!/bin/bash
# Vector / scalar / LDS instructions and wave cycles of a k_sfm_step launch with 1 and with 20 fused substeps (cfg3): the difference / 19 is
# a substep, the rest is what a launch executes around its substeps (load phase, prologue, epilogue).  usage (GPU box): tools/prologue_insts.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prologue_insts
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ns in 1 20; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/ns$ns -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-gym-step --steps 20 --warmup 5 --repeats 1 --substeps $ns > /dev/null 2> $O/ns$ns.log || { tail -5 $O/ns$ns.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
res = {}
for ns in (1, 20):
    f = glob.glob("$O/ns%d/**/*counter_collection.csv" % ns, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_sfm_step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[ns] = {k: sum(v) / len(v) for k, v in acc.items()}
w = res[20]["SQ_WAVES"]
print("per wavefront (%d wavefronts per launch):" % w)
for k in sorted(res[20]):
    if k == "SQ_WAVES": continue
    a, b = res[1][k] / w, res[20][k] / w
    sub = (b - a) / 19
    print("  %-18s 1 substep %9.1f | 20 substeps %9.1f | per substep %8.1f | around the substeps %8.1f" % (k, a, b, sub, a - sub))
PY
