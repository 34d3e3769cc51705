#!/bin/bash
# cfg4 (4096 x 25 ORCA, circle crossing) kernel time in its two named phases; usage: tools/orca_ab.sh <tag>
B="python3 bench.py --no-cpu-baseline --no-other-configs --model orca --scenario circle --steps 20"
$B --warmup 0 > gpurun_out/orca_first20_$1.json 2>gpurun_out/orca_ab.err
$B --warmup 25 > gpurun_out/orca_dense_$1.json 2>>gpurun_out/orca_ab.err
python3 - "$1" <<'PY'
import json,sys
for ph in ("first20","dense"):
    d=json.loads(open(f"gpurun_out/orca_{ph}_{sys.argv[1]}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print(ph, sys.argv[1], round(r["kernel_avg_ms"]*1e3,1), "us", r["variant"])
PY
