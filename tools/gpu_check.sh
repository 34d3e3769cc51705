#!/bin/bash
# usage (on the GPU box): tools/gpu_check.sh TAG [bench args...]   -- gpu tests, one bench line, section stamps
TAG=${1:-x}; shift
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; tail -4 gpurun_out/${TAG}_tests.log
python bench.py --no-cpu-baseline "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python -c "
import json,sys; d=json.load(open('gpurun_out/${TAG}_bench.json')); print('kernel_ms', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'], 'finite', d['finite_fraction'])"
python tools/stamp_probe.py 25 hsfm_farina 2>&1 | tail -10
