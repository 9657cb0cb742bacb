#!/usr/bin/env python3
"""Micro-benchmark of the score-network step (zedo_sde_step) at the BASELINE row count, with the
library's sampled HIP-event timing.  usage: python3 tools/bench_hidden.py [rows] [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zedo-release_amd"))
import numpy as np
import torch
import zedo_hip as zh
from lib.dataset import synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50750
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W = zh.Weights(syn.make_weights(0))
s = zh.Schedule(W, np.array([0.05], np.float32))
x = torch.randn(B, 17, 3, device="cuda") * 0.3
for _ in range(2):
    zh.sde_step(W, s, 0, x)
torch.cuda.synchronize()
zh.profile_start(1, 4096)
t0 = time.perf_counter()
for _ in range(iters):
    zh.sde_step(W, s, 0, x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
p = zh.profile_stop()
fl = 2.0 * B * 1024 * 1024
print(f"rows {B}: step {dt * 1e3:.3f} ms  ({B * 8597504 / dt / 1e12:.1f} TF end-to-end)")
for k, v in p.items():
    if v["avg_ms"]:
        extra = f"  {fl / v['avg_ms'] / 1e9:.1f} TF" if k == "hidden_dense" else ""
        print(f"  {k:16s} avg {v['avg_ms'] * 1e3:9.1f} us  n={v['samples']}{extra}")
