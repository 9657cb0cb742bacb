set -u
ZEDO_TEST_MATH=f32 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-mode"
run() { # tag, env, args
  env $2 $B $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/pass', j['ms_per_step'], 'wall/oilstep', j['wall_ms_per_oil_step'], j['kernel_time_ms_sampled_raw'], 'frac', (j['roofline'] or {}).get('frac'), j['selection_sha16'])
"
}
run cfg1_new "X=1" "--poses 886 --hypo 1"
run cfg1_nosplit "ZEDO_POST_NO_SPLIT=1" "--poses 886 --hypo 1"
run cfg0_new "X=1" "--poses 64 --hypo 1 --oil 100"
run cfg0_nosplit "ZEDO_POST_NO_SPLIT=1" "--poses 64 --hypo 1 --oil 100"
run s8_new "X=1" "--poses 127"
run s8_nomix "ZEDO_NO_MID_MIX=1" "--poses 127"
run s8_nomix_nosplit "ZEDO_NO_MID_MIX=1 ZEDO_POST_NO_SPLIT=1" "--poses 127"
run s4_new "X=1" "--poses 254"
run s4_nomix "ZEDO_NO_MID_MIX=1" "--poses 254"
run r4096 "X=1" "--poses 82 --oil 300"
run r4096_nomix "ZEDO_NO_MID_MIX=1" "--poses 82 --oil 300"
run r9000 "X=1" "--poses 180 --oil 300"
run r9000_nomix "ZEDO_NO_MID_MIX=1" "--poses 180 --oil 300"
run full "X=1" ""
