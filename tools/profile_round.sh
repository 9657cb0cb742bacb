#!/bin/bash
# One round's profile evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02
# -> gpurun_out/kernel_stats_<tag>.txt  (rocprofv3 --kernel-trace of the default bench command)
#    gpurun_out/hbm_traffic_<tag>.json   (separate --pmc passes: FETCH_SIZE, WRITE_SIZE, SQ counters)
#    gpurun_out/hbm_traffic_f16x3_<tag>.json   (the same passes in the split-fp16 mode)
# Copy both into profiles/ afterwards.  Counter passes use --kernel-trace only (no other trace domain).
set -u
TAG=${1:-rXX}
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rm -rf /tmp/zprof_*
rocprofv3 --kernel-trace --stats -d /tmp/zprof_kt -o kt -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode > $ROOT/gpurun_out/prof_${TAG}_bench.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $(find /tmp/zprof_kt -name '*_results.db' | head -1) > $ROOT/gpurun_out/kernel_stats_${TAG}.txt
# the opt-in split-fp16 mode of the hidden layers (alt_mode of the bench line), profiled on its own
rocprofv3 --kernel-trace --stats -d /tmp/zprof_kt16 -o kt -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode --math f16x3 > $ROOT/gpurun_out/prof_${TAG}_bench_f16x3.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $(find /tmp/zprof_kt16 -name '*_results.db' | head -1) > $ROOT/gpurun_out/kernel_stats_${TAG}_f16x3.txt
ARGS="--steps 1 --warmup 0 --oil 40 --no-cpu-baseline --no-alt-mode --no-geometry"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/zprof_f -o f -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/pmc_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/zprof_w -o w -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/pmc_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/zprof_s -o s -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/pmc_${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d /tmp/zprof_v -o v -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/pmc_${TAG}_valu.log 2>&1
ZEDO_PMC_TAG="rocprofv3 --pmc passes of 'bench.py $ARGS', round $TAG" python3 $ROOT/tools/pmc_summary.py 50750 $ROOT/gpurun_out/hbm_traffic_${TAG}.json $(find /tmp/zprof_f /tmp/zprof_w /tmp/zprof_s /tmp/zprof_v -name '*_results.db') > $ROOT/gpurun_out/pmc_${TAG}_summary.log 2>&1
# the same FETCH_SIZE / WRITE_SIZE / SQ passes for the split-fp16 mode (alt_mode.roofline.traffic of the bench line)
ARGS16="$ARGS --math f16x3"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/zprof_f16 -o f -- python3 $ROOT/bench.py $ARGS16 > $ROOT/gpurun_out/pmc_${TAG}_fetch16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/zprof_w16 -o w -- python3 $ROOT/bench.py $ARGS16 > $ROOT/gpurun_out/pmc_${TAG}_write16.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/zprof_s16 -o s -- python3 $ROOT/bench.py $ARGS16 > $ROOT/gpurun_out/pmc_${TAG}_sq16.log 2>&1
ZEDO_PMC_HIDDEN=layer16_pair_kernel ZEDO_PMC_TAG="rocprofv3 --pmc passes of 'bench.py $ARGS16', round $TAG" python3 $ROOT/tools/pmc_summary.py 50750 $ROOT/gpurun_out/hbm_traffic_f16x3_${TAG}.json $(find /tmp/zprof_f16 /tmp/zprof_w16 /tmp/zprof_s16 -name '*_results.db') > $ROOT/gpurun_out/pmc_${TAG}_summary16.log 2>&1
tail -3 $ROOT/gpurun_out/prof_${TAG}_bench.log | cut -c1-600
head -14 $ROOT/gpurun_out/kernel_stats_${TAG}.txt | cut -c1-200
head -12 $ROOT/gpurun_out/kernel_stats_${TAG}_f16x3.txt | cut -c1-200
cat $ROOT/gpurun_out/pmc_${TAG}_summary.log | cut -c1-400
cat $ROOT/gpurun_out/pmc_${TAG}_summary16.log | cut -c1-400
