# A/B of build variants of the library on one GPU box:  bash tools/ab_sched.sh "" "-DMY_EXPERIMENT=1" ...  (the closed -D knobs of
# rounds 1-5 - ZEDO_SCHED_*, ZEDO_PAIR_*, ZEDO_EXP_* - left the product sources in round 6; an experiment brings its own #ifdef)
# The knobs are APPENDED to the Makefile's flags (EXTRA=...; -ffp-contract=off stays), and the product library is
# rebuilt without any knob when the script ends, so that no differently-built .so is left in the tree.
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["kernel_shader_clock_ghz"], d["kernel_time_ms_sampled_avg"])'
restore() { (cd zedo-release_amd/csrc && touch zedo_gemm.hip && make EXTRA= 2>&1 | grep -E "error"); echo "== product library rebuilt"; }
trap restore EXIT
for v in "$@"; do
  (cd zedo-release_amd/csrc && touch zedo_gemm.hip && make EXTRA="$v" 2>&1 | grep -E "error" )
  echo "== build [$v]"
  python bench.py --no-cpu-baseline --no-alt-mode --steps ${AB_STEPS:-2} --warmup 1 2>&1 | tail -1 | python -c "$P"
done
