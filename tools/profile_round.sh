#!/bin/bash
# usage (GPU box): tools/profile_round.sh TAG   -- bench line, rocprofv3 kernel stats, PMC traffic passes, other configs
TAG=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err && echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv && head -5 $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/pmc_write.log 2>&1
python3 - <<PY > $O/pmc_traffic.txt
import csv, glob
for name in ("fetch", "write"):
    for fn in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fn)) if "k_sfm_step" in r["Kernel_Name"]]
        print(name, "mean per dispatch", sum(v) / len(v), "n", len(v))
PY
cat $O/pmc_traffic.txt
for cfg in "--agents 10 --model sfm_helbing --scenario circle" "--agents 10 --model sfm_guo --scenario circle" \
           "--model hsfm_new_guo" "--model hsfm_new_moussaid" "--worlds 8192 --agents 50 --model hsfm_farina --scenario circle --walls" \
           "--worlds 16384" "--model orca --scenario circle --steps 20 --warmup 3"; do
  python3 bench.py --no-cpu-baseline $cfg 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', '| kernel_ms %.4f frac %.3f value %.3e' % (d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['value']))" | tee -a $O/other_configs.txt
done
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
