#!/bin/bash
# usage (GPU box): bash tools/profile_round6_extra.sh TAG -- the side figures of the round: the ORCA stage table of the stamps build (exact arithmetic: the
# library default), worlds-per-GPU curves, the facade's latency split, the Gym step figures, the parity report of the GPU suite.
TAG=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for ph in "0 20" "25 20"; do set -- $ph
  STAMP_WARMUP=$1 STAMP_STEPS=$2 python3 tools/stamp_probe.py 25 orca > $O/orca_stage_table_w$1.txt 2>&1; tail -12 $O/orca_stage_table_w$1.txt
done
python3 tools/facade_latency.py 300 $O/facade_latency.json > $O/facade_latency.txt 2>&1; tail -4 $O/facade_latency.txt
for W in 512 1024 2048 4096 8192 16384 32768; do
  python3 bench.py --worlds $W --steps 50 --warmup 20 --no-other-configs --no-cpu-baseline --no-gym-step 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('worlds %6d | kernel_us %8.2f frac %.3f' % ($W, d['kernel_us'], d['roofline']['frac']))" | tee -a $O/worlds_per_gpu_curve.txt
done
for W in 1024 2048 4096 8192 16384 32768; do
  python3 bench.py --worlds $W --model orca --scenario circle --steps 20 --warmup 25 --no-other-configs --no-cpu-baseline --no-gym-step 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('orca dense (exact) worlds %6d | kernel_us %8.2f per 4096 worlds %8.2f' % ($W, d['kernel_us'], d['kernel_us']*4096.0/$W))" | tee -a $O/orca_worlds_per_gpu_curve.txt
done
