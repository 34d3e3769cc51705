#!/bin/bash
# usage (here, after a gpurun of tools/profile_round2.sh TAG): tools/collect_profiles.sh TAG
# copies the judged summaries from gpurun_out/TAG (scratch) into profiles/ (tracked): kernel stats, PMC counter CSVs, the bench
# lines of the profiled commands, and pmc_summary.json (which bench.py reads back)
TAG=$1; R=$(cd "$(dirname "$0")/.." && pwd); S=$R/gpurun_out/$TAG
for f in $S/*_kernel_stats.csv $S/*_pmc_*.csv $S/*_bench.json; do cp $f $R/profiles/${TAG}_$(basename $f); done
cp $S/pmc_summary.json $R/profiles/pmc_summary.json
[ -f $R/gpurun_out/${TAG}_bench.json ] && cp $R/gpurun_out/${TAG}_bench.json $R/profiles/${TAG}_bench.json
[ -f $R/gpurun_out/parity_report.json ] && cp $R/gpurun_out/parity_report.json $R/profiles/${TAG}_parity_report.json
ls $R/profiles | grep "^${TAG}_" | wc -l
