#!/bin/bash
for w in 512 1024 2048 3072 4096 6144 8192 12288 16384 32768; do
  python3 bench.py --no-cpu-baseline --worlds $w "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('worlds %6d | kernel_us %8.2f frac %.3f' % ($w, d['roofline']['kernel_avg_ms']*1e3, d['roofline']['frac']))"
done
