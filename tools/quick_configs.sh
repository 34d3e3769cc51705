#!/bin/bash
# usage (GPU box): tools/quick_configs.sh  -- kernel time of the main configurations (no CPU baseline)
for cfg in "" "--worlds 16384" "--worlds 8192 --agents 50 --model hsfm_farina --scenario circle --walls" "--worlds 8192 --agents 50 --model hsfm_farina --scenario circle" \
           "--agents 10 --model sfm_helbing --scenario circle" "--model hsfm_new_guo" "--model hsfm_new_moussaid" "$@"; do
  python3 bench.py --no-cpu-baseline $cfg 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-90s | kernel_us %8.2f frac %.3f' % ('$cfg', d['roofline']['kernel_avg_ms']*1e3, d['roofline']['frac']))"
done
