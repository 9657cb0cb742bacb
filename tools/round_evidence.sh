#!/bin/bash
# One round's bench evidence on the GPU box (run through gpurun from the repo root):  bash tools/round_evidence.sh r04
# -> gpurun_out/bench_<tag>.json (default command), bench_<tag>_driver_form.json (the driver's command form),
#    config_shards_<tag>.jsonl (every BASELINE configuration as the row shard one GPU sees), then tools/profile_round.sh <tag>.
set -u
TAG=${1:-rXX}
mkdir -p gpurun_out
last() { grep '^{' | tail -1; }
python bench.py 2>gpurun_out/bench_${TAG}.err | last > gpurun_out/bench_${TAG}.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>>gpurun_out/bench_${TAG}.err | last > gpurun_out/bench_${TAG}_driver_form.json
: > gpurun_out/config_shards_${TAG}.jsonl
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
$B --poses 64 --hypo 1 --oil 100 2>/dev/null | last >> gpurun_out/config_shards_${TAG}.jsonl                      # configs[0]
$B --poses 886 --hypo 1 2>/dev/null | last >> gpurun_out/config_shards_${TAG}.jsonl                              # configs[1] shape, 3DPW settings
$B --workload 3 --scaling strong --poses 886 --hypo 1 2>/dev/null | last >> gpurun_out/config_shards_${TAG}.jsonl  # configs[1], H36M settings
$B --poses 127 2>/dev/null | last >> gpurun_out/config_shards_${TAG}.jsonl                                       # configs[2] strong-scaled over 8
$B --poses 254 2>/dev/null | last >> gpurun_out/config_shards_${TAG}.jsonl                                       # ... over 4
$B --workload 3 --scaling weak --oil 20 --no-alt-mode 2>/dev/null | last >> gpurun_out/config_shards_${TAG}.jsonl   # configs[3]: 3 544 000-row shard
$B --workload 4 --scaling weak --oil 100 --no-alt-mode 2>/dev/null | last >> gpurun_out/config_shards_${TAG}.jsonl  # configs[4]: 625 000-row shard + all-gather
python - <<PY
import json
for f in ("gpurun_out/bench_${TAG}.json", "gpurun_out/bench_${TAG}_driver_form.json"):
    j = json.load(open(f)); r = j["roofline"]; a = j.get("alt_mode") or {}
    print(f, j["value"], j["ms_per_step"], "frac", r["frac"], "at clock", r["frac_at_kernel_clock"], r["kernel_shader_clock_ghz"], "e2e", j["end_to_end_frac"],
          "sum/wall", j["sum_kernel_ms_per_oil_step"], j["wall_ms_per_oil_step"], "alt", a.get("value"), (a.get("roofline") or {}).get("traffic"), "cpu", (j.get("cpu_baseline") or {}).get("value"))
for l in open("gpurun_out/config_shards_${TAG}.jsonl"):
    j = json.loads(l); print(j["config"]["rows_per_gpu"], j["ms_per_step"], j["value"], j["end_to_end_tflops"], (j["roofline"] or {}).get("frac"), (j.get("alt_mode") or {}).get("ms_per_step"))
PY
bash tools/profile_round.sh ${TAG} > gpurun_out/profile_round_${TAG}.log 2>&1
# kernel traces of the small-batch configurations (configs[1]'s shape, configs[0])
export TMPDIR=/tmp; R=$(pwd); cd /tmp; rm -rf /tmp/zpe1 /tmp/zpe0
rocprofv3 --kernel-trace --stats -d /tmp/zpe1 -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --poses 886 --hypo 1 --no-cpu-baseline --no-alt-mode > /dev/null 2>&1
python3 $R/tools/rocpd_summary.py $(find /tmp/zpe1 -name '*_results.db' | head -1) > $R/gpurun_out/kernel_stats_${TAG}_cfg1.txt
rocprofv3 --kernel-trace --stats -d /tmp/zpe0 -o kt -- python3 $R/bench.py --steps 4 --warmup 1 --poses 64 --hypo 1 --oil 100 --no-cpu-baseline --no-alt-mode > /dev/null 2>&1
python3 $R/tools/rocpd_summary.py $(find /tmp/zpe0 -name '*_results.db' | head -1) > $R/gpurun_out/kernel_stats_${TAG}_cfg0.txt
cd $R
# the HIP-side ensembles of the three configs[2] captures (tests/test_ensemble_gpu.py asserts on shorter ones)
python tools/ensemble_gpu.py --members 32 --out gpurun_out/ensemble_${TAG} > gpurun_out/ensemble_${TAG}.log 2>&1
tail -3 gpurun_out/ensemble_${TAG}.log | cut -c1-400
head -12 gpurun_out/kernel_stats_${TAG}.txt | cut -c1-170
