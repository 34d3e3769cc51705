#!/usr/bin/env python3
"""usage (here, after the gpurun calls of tools/profile_round6.sh TAG have merged their files into gpurun_out/TAG): tools/pmc_summary.py gpurun_out/TAG gpurun_out/TAG/pmc_summary.json; tools/collect_profiles.py TAG
Copies the judged summaries from gpurun_out/TAG (scratch) into profiles/ (tracked): kernel stats, the bench lines of the profiled
commands, the PMC counter CSVs REDUCED to the step kernels' rows (dispatch, kernel, counter, value: 1 MB instead of 7), and
pmc_summary.json (which bench.py reads back: roofline.traffic / valu / valu_frac / pmc_build_matches)."""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, D = os.path.join(R, "gpurun_out", tag), os.path.join(R, "profiles")
for f in glob.glob(os.path.join(D, tag + "_*")):
    os.remove(f)
n = 0
for f in sorted(glob.glob(S + "/*_pmc_*.csv")):
    with open(os.path.join(D, f"{tag}_{os.path.basename(f)}"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_sfm_step" in k or "k_orca_step" in k:
                w.writerow([r["Dispatch_Id"], k[:72], r["Counter_Name"], r["Counter_Value"]])
    n += 1
for f in glob.glob(S + "/*_kernel_stats.csv") + glob.glob(S + "/*_bench.json") + glob.glob(S + "/*_bench_full.json"):
    shutil.copy(f, os.path.join(D, f"{tag}_{os.path.basename(f)}"))
summary = json.load(open(os.path.join(S, "pmc_summary.json")))
json.dump(summary, open(os.path.join(D, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(n, "counter files;", round(sum(os.path.getsize(x) for x in glob.glob(os.path.join(D, tag + "_*"))) / 1e6, 2), "MB; build",
      sorted({v.get("build_id") for v in summary.values() if isinstance(v, dict)}))
