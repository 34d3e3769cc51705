#!/bin/bash
# usage (GPU box): bash tools/profile_round6.sh TAG [config ...]      (supersedes profile_round{,2,3,4,5}.sh)
# Per configuration: rocprofv3 kernel stats of bench.py (stationary protocol: every replay restored to the post-warm-up worlds) and
# four PMC passes (FETCH_SIZE, WRITE_SIZE, SQ issue counters, SQ instruction classes -- separate passes, never combined with a trace domain).
# tools/pmc_summary.py turns the counter CSVs into pmc_summary.json (stamped with the library build id), which bench.py reads back
# from profiles/ (roofline.traffic / valu / valu_frac / pmc_build_matches).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
declare -A CFG
CFG[cfg3]=""
CFG[cfg2]="--agents 10 --model sfm_helbing --scenario circle"
CFG[cfg4_first20]="--model orca --scenario circle --warmup 0 --steps 20"
CFG[cfg4_dense]="--model orca --scenario circle --warmup 25 --steps 20"
CFG[cfg4_first20_fma]="--model orca --scenario circle --warmup 0 --steps 20 --orca-math fma"
CFG[cfg4_dense_fma]="--model orca --scenario circle --warmup 25 --steps 20 --orca-math fma"
CFG[n15]="--agents 15"
CFG[n40]="--agents 40"
CFG[cfg5]="--worlds 8192 --agents 50 --model hsfm_farina --scenario circle --walls --static 3 --device-generator"
CFG[cfg3x4]="--worlds 16384"
CFG[moussaid]="--model hsfm_new_moussaid"
CFG[cfg3_new_guo]="--model hsfm_new_guo"
CFG[robot26]="--robot"
CFG[n30]="--agents 30"
CFG[peragent]="--per-agent-params"
CFG[cfg5_nowalls]="--worlds 8192 --agents 50 --model hsfm_farina --scenario circle --static 3 --device-generator"
NAMES=${@:-cfg3 cfg2 cfg4_first20 cfg4_dense cfg4_first20_fma cfg4_dense_fma cfg5 moussaid cfg3_new_guo robot26 n30 peragent n15 n40}
# the DRIVER's command (BENCH_rNN.json): python3 bench.py --gpus 1 --steps 20 --warmup 5 -- its line is the one quoted first in README / DESIGN
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_protocol_bench.json 2> $O/driver_protocol_bench.log || { echo "driver-protocol bench failed"; tail -5 $O/driver_protocol_bench.log; exit 1; }
cp gpurun_out/bench_full.json $O/driver_protocol_bench_full.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/driver_stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gym-step > /dev/null 2> $O/driver_stats.log || { echo "driver stats run failed"; tail -5 $O/driver_stats.log; exit 1; }
cp $(find $O/driver_stats -name "*kernel_stats.csv" | head -1) $O/driver_protocol_kernel_stats.csv && head -4 $O/driver_protocol_kernel_stats.csv | cut -c1-160
rm -rf $O/driver_stats
# the default bench command itself under the kernel trace: its k_sfm_step<..., 25, 1> row is the kernel behind `value`
rocprofv3 --kernel-trace --stats --output-format csv -d $O/main_stats -- python3 bench.py --no-gym-step > $O/main_bench.json 2> $O/main_stats.log || { echo "main stats run failed"; tail -5 $O/main_stats.log; exit 1; }
cp $(find $O/main_stats -name "*kernel_stats.csv" | head -1) $O/main_kernel_stats.csv && head -4 $O/main_kernel_stats.csv | cut -c1-160
rm -rf $O/main_stats
for name in $NAMES; do
  A="${CFG[$name]} --no-cpu-baseline --no-other-configs"
  S="--steps 50 --warmup 20"
  case "$A" in *--steps*) S="";; esac
  echo "== $name: $A"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_stats -- python3 bench.py $A $S --repeats 4 > $O/${name}_bench.json 2> $O/${name}_stats.log || { echo "stats run failed"; tail -5 $O/${name}_stats.log; exit 1; }
  cp $(find $O/${name}_stats -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv && head -3 $O/${name}_kernel_stats.csv | cut -c1-160
  # (the counters are taken over the SAME window as the kernel time above: round 3 paired the instruction count of Gym steps 20-30 with
  #  the time of steps 20-70, which made cfg5's wall pass look stall-bound -- profiles/r4k_cfg5_wall_pass_counters_*.txt)
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
              "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
    p=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/${name}_pmc_$p -- python3 bench.py $A $S --repeats 1 > /dev/null 2> $O/${name}_pmc_$p.log || { echo "pmc $p failed"; tail -5 $O/${name}_pmc_$p.log; exit 1; }
    cp $(find $O/${name}_pmc_$p -name "*counter_collection.csv" | head -1) $O/${name}_pmc_$p.csv
    rm -rf $O/${name}_pmc_$p
  done
  rm -rf $O/${name}_stats
done
# the SQ counters once more over the driver protocol's window of the HEADLINE (instruction counts follow the crowd's state; the other
# configurations run over their own windows whatever --steps says)
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
MIX="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"
name=cfg3_w5s20; A="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
for pass in "$SQ" "$MIX"; do
  p=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/${name}_pmc_$p -- python3 bench.py $A --repeats 1 > /dev/null 2> $O/${name}_pmc_$p.log || { echo "pmc $name failed"; tail -5 $O/${name}_pmc_$p.log; exit 1; }
  cp $(find $O/${name}_pmc_$p -name "*counter_collection.csv" | head -1) $O/${name}_pmc_$p.csv
  rm -rf $O/${name}_pmc_$p
done
python3 tools/pmc_summary.py $O $O/pmc_summary.json && python3 -c "
import json; d=json.load(open('$O/pmc_summary.json'))
for k,v in d.items(): print(k, v.get('build_id'), 'B/agent/launch', v.get('hbm_bytes_per_agent_launch'), 'VALU/wave-substep', (v.get('valu') or {}).get('valu_insts_per_wave_substep'))"
