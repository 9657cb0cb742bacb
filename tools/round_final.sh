set -u
ROOT=$(pwd)
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r06_gputest_final.log
python bench.py > gpurun_out/bench_r06_final.json 2> gpurun_out/bench_r06_final.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r06_driver_form.json 2> gpurun_out/bench_r06_driver_form.err
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/zprof_kt /tmp/zprof_kt16
rocprofv3 --kernel-trace --stats -d /tmp/zprof_kt -o kt -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode > $ROOT/gpurun_out/prof_r06b_bench.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $(find /tmp/zprof_kt -name '*_results.db' | head -1) > $ROOT/gpurun_out/kernel_stats_r06.txt
rocprofv3 --kernel-trace --stats -d /tmp/zprof_kt16 -o kt -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode --math f16x3 > $ROOT/gpurun_out/prof_r06b_bench_f16x3.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $(find /tmp/zprof_kt16 -name '*_results.db' | head -1) > $ROOT/gpurun_out/kernel_stats_r06_f16x3.txt
cd $ROOT
cat gpurun_out/r06_gputest_final.log
head -16 gpurun_out/kernel_stats_r06.txt | cut -c1-160
