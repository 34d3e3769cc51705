run() { python3 bench.py --no-cpu-baseline --no-other-configs --steps 100 --warmup 20 --repeats 20 $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(r['kernel_avg_ms']*1e3,2), round(r['kernel_median_ms']*1e3,2), 'us')"; }
for rep in 1 2; do
export CROWDSTEP_LIB=$PWD/social_navigation_pyenvs_amd/libcrowdstep_base.so; run base
unset CROWDSTEP_LIB; run tree
done
for v in base tree; do if [ $v = base ]; then export CROWDSTEP_LIB=$PWD/social_navigation_pyenvs_amd/libcrowdstep_base.so; else unset CROWDSTEP_LIB; fi
run cfg5nw_$v "--worlds 8192 --agents 50 --scenario circle --static 3 --device-generator --steps 50"; run cfg5_$v "--worlds 8192 --agents 50 --scenario circle --static 3 --walls --device-generator --steps 50"; run n30_$v "--agents 30"; run guo_$v "--model hsfm_new_guo"; run mou_$v "--model hsfm_new_moussaid"; run robot_$v "--robot"; run x4_$v "--worlds 16384"; done
