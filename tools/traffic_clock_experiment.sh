#!/bin/bash
# Does the fabric-side traffic of the hidden-layer launch move its shader clock / its duration?  (round-3 review item 4)
# For every build variant: two un-profiled bench runs (poses/s, avg launch ms and the in-kernel shader clock, live) and
# one FETCH_SIZE + one WRITE_SIZE counter pass (bytes per launch).  The variant list is walked TWICE (A/B/.../A/B/...) so
# that box drift shows up as a difference between the two rounds of the same variant.
#   bash tools/traffic_clock_experiment.sh r03 > gpurun_out/traffic_clock_r03.txt
# The product library is rebuilt without any knob at the end.
set -u
TAG=${1:-rXX}
ROOT=$(pwd)
export TMPDIR=/tmp
VARIANTS=("" "-DZEDO_EXP_XSC1=1" "-DZEDO_EXP_XSC1=2" "-DZEDO_EXP_XSC1=3" "-DZEDO_EXP_MAP=1")
NAMES=("product" "X tiles sc1" "X tiles sc0 sc1" "X tiles nt" "naive map: one column tile of W per XCD")
restore() { (cd $ROOT/zedo-release_amd/csrc && touch zedo_gemm.hip && make EXTRA= 2>&1 | grep -E "error"); echo "== product library rebuilt"; }
trap restore EXIT
P='import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]; print("poses/s %.2f  hidden launch %.4f ms  %.2f TFLOP/s  kernel clock %s GHz  frac@clock %s  box probe %.3f GHz" % (d["value"], r["avg_launch_ms"], r["achieved"], r["kernel_shader_clock_ghz"], r["frac_at_kernel_clock"], r["box_shader_clock_ghz"]))'
for round in 1 2; do
  for i in "${!VARIANTS[@]}"; do
    v="${VARIANTS[$i]}"
    (cd $ROOT/zedo-release_amd/csrc && touch zedo_gemm.hip && make EXTRA="$v" 2>&1 | grep -E "error")
    echo "== round $round  [${NAMES[$i]}]  EXTRA='$v'"
    for rep in 1 2; do
      python3 $ROOT/bench.py --no-cpu-baseline --no-alt-mode --steps 2 --warmup 1 2>&1 | python3 -c "$P"
    done
    if [ $round = 1 ]; then
      cd /tmp; rm -rf /tmp/ztc_f /tmp/ztc_w
      rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/ztc_f -o f -- python3 $ROOT/bench.py --steps 1 --warmup 0 --oil 40 --no-cpu-baseline --no-alt-mode > /dev/null 2>&1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/ztc_w -o w -- python3 $ROOT/bench.py --steps 1 --warmup 0 --oil 40 --no-cpu-baseline --no-alt-mode > /dev/null 2>&1
      cd $ROOT
      python3 tools/pmc_summary.py 50750 /tmp/ztc_${i}.json $(find /tmp/ztc_f /tmp/ztc_w -name '*_results.db') 2>&1 | tail -1
    fi
  done
done
