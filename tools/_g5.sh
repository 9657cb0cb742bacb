set -u
ZEDO_TEST_MATH=f32 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-mode"
run() { # tag, env, args
  env $2 $B $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/pass', j['ms_per_step'], 'wall/oilstep', j['wall_ms_per_oil_step'], j['kernel_time_ms_sampled_raw'], 'frac', (j['roofline'] or {}).get('frac'), j['selection_sha16'], j['mpjpe_best_of_H_m'], j['pa_mpjpe_best_of_H_m'])
"
}
run cfg1 "X=1" "--poses 886 --hypo 1"
run cfg0 "X=1" "--poses 64 --hypo 1 --oil 100"
run cfg1_h36m "X=1" "--workload 3 --poses 886 --hypo 1"
run full "X=1" ""
export TMPDIR=/tmp; R=$(pwd); cd /tmp; rm -rf /tmp/zp5
rocprofv3 --kernel-trace --stats -d /tmp/zp5 -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode > /dev/null 2>&1
python3 $R/tools/rocpd_summary.py $(find /tmp/zp5 -name '*_results.db' | head -1) | grep -i "ipo\|post_reduce\|calls" | cut -c1-160
rm -rf /tmp/zp6
rocprofv3 --kernel-trace --stats -d /tmp/zp6 -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --poses 886 --hypo 1 --no-cpu-baseline --no-alt-mode > /dev/null 2>&1
python3 $R/tools/rocpd_summary.py $(find /tmp/zp6 -name '*_results.db' | head -1) | head -12 | cut -c1-160
