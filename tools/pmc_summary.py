#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes (rocpd sqlite) into profiles/hbm_traffic.json.

Each pass is its own rocprofv3 run of the same command (one counter group per run, --kernel-trace only):

    rocprofv3 --pmc FETCH_SIZE  --kernel-trace -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --oil 40 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE  --kernel-trace -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT \
              --kernel-trace -d gpurun_out/pmc_sq -o sq -- python3 bench.py ...

usage: python tools/pmc_summary.py ROWS out.json db1 [db2 ...]

HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: both counters are in KiB and FETCH_SIZE reports
half of a 16 B/lane coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Counter values are
summed over the dimension instances rocprofv3 stores per dispatch, then averaged over the dispatches of a kernel.
"""
import json
import re
import sqlite3
import sys


def per_kernel(path):
    db = sqlite3.connect(path)
    out = {}
    q = ("select name, dispatch_id, counter_name, sum(counter_value) from pmc_events "
         "group by name, dispatch_id, counter_name")
    for name, _disp, cname, val in db.execute(q):
        out.setdefault(name, {}).setdefault(cname, []).append(val)
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in out.items()}


def short(name):
    m = re.search(r"(layer16_pair_kernel|layer16_small_kernel|layer16_pre_kernel|layer16_post_kernel|layer_pair_kernel|layer_kernel|"
                  r"post_reduce_kernel|reproj_step_kernel|ipo_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    rows, outp, dbs = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
    merged = {}
    for p in dbs:
        for k, d in per_kernel(p).items():
            e = merged.setdefault(short(k), {})
            for c, (avg, n) in d.items():
                e[c] = avg
                e["launches"] = n
    for e in merged.values():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
            e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 1024)
    # ZEDO_PMC_HIDDEN: name prefix of the hidden-layer kernel of the profiled arithmetic mode (f16x3: layer16_pair_kernel)
    import os
    hid = os.environ.get("ZEDO_PMC_HIDDEN", "layer_pair_kernel")
    hidden = [e for k, e in merged.items() if k.startswith(hid) and "hbm_bytes_per_launch" in e]
    doc = {"note": __doc__.split("usage:")[1].split("\n", 2)[2].strip(), "rows": rows,
           "rows_launch": rows, "hidden_kernel": hid,
           "collected": os.environ.get("ZEDO_PMC_TAG", "unlabelled counter pass"), "kernels": merged}
    if hidden:
        alg = rows * 1024 * 4 * 2.5 + 1024 * 1024 * 4      # read X, write Y, residual on every other layer, + W
        b = sum(e["hbm_bytes_per_launch"] for e in hidden) / len(hidden)
        doc.update(hidden_dense_bytes_per_launch=int(b), hidden_dense_algorithmic_bytes_per_launch=int(alg),
                   ratio=round(b / alg, 3))
    json.dump(doc, open(outp, "w"), indent=1)
    print(json.dumps({k: v for k, v in doc.items() if k not in ("kernels", "note")}))


if __name__ == "__main__":
    main()
