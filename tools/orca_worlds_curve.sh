#!/bin/bash
# ORCA (cfg4's crossing, dense phase: Gym steps 25..45) kernel time against worlds per GPU, one GPU.  usage: tools/orca_worlds_curve.sh
for w in 1024 2048 4096 8192 16384 32768; do
  python3 bench.py --no-cpu-baseline --no-other-configs --no-gym-step --model orca --scenario circle --steps 20 --warmup 25 --repeats 5 --worlds $w "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('orca worlds %6d | kernel_us %9.2f | us per 4096 worlds %8.2f | frac %.4f' % ($w, r['kernel_avg_ms']*1e3, r['kernel_avg_ms']*1e3*4096/$w, r['frac']))"
done
