#!/bin/bash
# SQ counters of the split-fp16 hidden-layer launch (run through gpurun from the repo root): where do a SIMD's cycles go?
#   bash tools/pmc_f16x3.sh > gpurun_out/pmc_f16x3_r03.txt
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 0 --oil 40 --no-cpu-baseline --no-alt-mode --math f16x3"
rm -rf /tmp/zp16_a /tmp/zp16_b
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/zp16_a -o a -- python3 $ROOT/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -d /tmp/zp16_b -o b -- python3 $ROOT/bench.py $ARGS > /dev/null 2>&1
cd $ROOT
python3 - <<'PY'
import sqlite3, glob, re
out = {}
for d in ("/tmp/zp16_a", "/tmp/zp16_b"):
    for p in glob.glob(d + "/**/*_results.db", recursive=True):
        db = sqlite3.connect(p)
        q = "select name, dispatch_id, counter_name, sum(counter_value) from pmc_events group by name, dispatch_id, counter_name"
        for name, disp, c, v in db.execute(q):
            m = re.search(r"(layer16_pair_kernel<\d>|layer16_pre_kernel<\d>|layer16_post_kernel<\d>)", name)
            if m:
                out.setdefault(m.group(1), {}).setdefault(c, []).append(v)
for k, d in sorted(out.items()):
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    simd_cycles = avg.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    line = {c: round(v) for c, v in avg.items()}
    if simd_cycles:
        line["mfma_busy_frac"] = round(avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles, 3)
    print(k, line)
PY
