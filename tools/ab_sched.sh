# A/B of compile-time knobs of zedo_gemm.hip on one GPU box:  bash tools/ab_sched.sh "" "-DZEDO_PAIR_W8_RES=0" ...
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["kernel_time_ms_sampled_avg"])'
for v in "$@"; do
  (cd zedo-release_amd/csrc && touch zedo_gemm.hip && make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $v" 2>&1 | grep -E "error" )
  echo "== build [$v]"
  python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | tail -1 | python -c "$P"
done
