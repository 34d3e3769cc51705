#!/bin/bash
# Kernel time against worlds per GPU (one GPU), then -- what the driver's SCALE run measures -- BASELINE.json configs[4] as bench.py
# states it: `--total-worlds 65536` split over the ranks (the cfg5 entry of other_configs in every bench line; on ONE GPU the
# whole job).  usage: tools/scaling_curve.sh [extra bench.py args]
for w in 512 1024 2048 3072 4096 6144 8192 12288 16384 32768; do
  python3 bench.py --no-cpu-baseline --no-other-configs --steps 50 --repeats 7 --worlds $w "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('worlds %6d | kernel_us %8.2f frac %.3f' % ($w, d['roofline']['kernel_avg_ms']*1e3, d['roofline']['frac']))"
done
python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --repeats 5 --total-worlds 65536 --agents 50 --model hsfm_farina --scenario circle --walls --static 3 --device-generator "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5: 65536 worlds x 50 on %d GPU(s) | kernel_us %8.2f frac %.3f scaling %s' % (d['n_gpus'], d['roofline']['kernel_avg_ms']*1e3, d['roofline']['frac'], d['scaling']))"
