#!/usr/bin/env python3
"""Per-workgroup timeline of a dense-layer ubench variant (tools/ubench/ubench_gemm M VARIANT out.bin):
prologue / K-loop / epilogue durations and how long each CU has 0 / 1 / 2 workgroups inside the K loop."""
import collections
import sys

import numpy as np

for path in sys.argv[1:]:
    a = np.fromfile(path, dtype=np.int64).reshape(-1, 8)
    a = a[(a[:, 0] > 0) & (a[:, 3] > 0)]
    if not len(a):
        print(path, "no marks")
        continue
    t0, t1, t2, t3, hw, xcc = (a[:, i] for i in range(6))
    us = 10e-3                                     # wall_clock64 ticks at 100 MHz
    pro, loop, epi = (t1 - t0) * us, (t2 - t1) * us, (t3 - t2) * us
    print(f"{path}: {len(a)} workgroups, span {(t3.max() - t0.min()) * us:.1f} us; mean prologue {pro.mean():.2f} "
          f"loop {loop.mean():.2f} epilogue {epi.mean():.2f} us; loop p10/p90 {np.percentile(loop, 10):.1f}/{np.percentile(loop, 90):.1f}; "
          f"epilogue p10/p50/p90 {np.percentile(epi, 10):.2f}/{np.percentile(epi, 50):.2f}/{np.percentile(epi, 90):.2f}")
    cu = ((xcc & 0xF) << 16) | (hw & 0xFF00)       # HW_ID: cu_id[11:8], sh_id[12], se_id[15:13]
    per = collections.defaultdict(list)
    for i in range(len(a)):
        per[int(cu[i])].append((t0[i], t1[i], t2[i], t3[i]))
    res = np.zeros(4)
    for v in per.values():
        ev = sorted([(x[1], 1) for x in v] + [(x[2], -1) for x in v])
        cur, last = 0, min(x[0] for x in v)
        for t, d in ev:
            res[min(cur, 3)] += t - last
            last, cur = t, cur + d
        res[0] += max(x[3] for x in v) - last
    print(f"   {len(per)} CUs, {np.mean([len(v) for v in per.values()]):.1f} workgroups each; CU time with 0/1/2/3+ workgroups "
          f"inside the K loop: {np.round(res / res.sum(), 3)}")
