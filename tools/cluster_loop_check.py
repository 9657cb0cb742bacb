#!/usr/bin/env python3
"""The one-launch OIL loop of small batches (ZEDO_CLUSTER_LOOP=1) against the per-layer launches: bitwise equality of
(x, T) and time per iteration.  usage: python tools/cluster_loop_check.py   (GPU box; spawns one child per mode)"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, time, hashlib
import numpy as np, torch
root = %r
sys.path.insert(0, os.path.join(root, "zedo-release_amd")); sys.path.insert(0, os.path.join(root, "oracle"))
import zedo_hip as zh, zedo_oracle as O
from lib.dataset import synthetic as syn
out = {}
W = zh.Weights(syn.make_weights(0))
dev = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
for (H, N, S) in ((1, 886, 1000), (1, 64, 100), (3, 300, 60), (2, 500, 60), (1, 1, 30), (7, 37, 25)):
    d = syn.make_poses(N, seed=N, conf_mode="uniform")
    rng = np.random.default_rng(N)
    x0 = (0.25 * rng.standard_normal((H * N, 17, 3))).astype(np.float32)
    T0 = np.tile(d["db_3d"][:, 0, :], (H, 1)).astype(np.float32)
    sched = zh.Schedule(W, O.oil_timestamps(S))
    geom = zh.reproj_prepare(dev(d["db_2d"][:, :, :2]), dev(d["camera_param"]), dev(d["db_2d"][:, :, 2]))
    best = 1e9
    for rep in range(3):
        x, T = dev(x0), dev(T0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        zh.oil_run(W, sched, x, geom, T, 0, S, S // 5)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / S)
    # a second call continuing from a mid-loop state (step_begin > 0) must work too
    x2, T2 = dev(x0), dev(T0)
    zh.oil_run(W, sched, x2, geom, T2, 0, S // 2, S // 5)
    zh.oil_run(W, sched, x2, geom, T2, S // 2, S, S // 5)
    assert torch.equal(x2, x) and torch.equal(T2, T)
    out[f"{H}x{N}x{S}"] = dict(sha=hashlib.sha256(x.cpu().numpy().tobytes() + T.cpu().numpy().tobytes()).hexdigest()[:16],
                               us_per_step=round(best * 1e6, 2), finite=bool(torch.isfinite(x).all()))
print("RESULT " + json.dumps(out))
''' % ROOT


def run(mode):
    env = dict(os.environ)
    env.pop("ZEDO_CLUSTER_LOOP", None)
    if mode:
        env["ZEDO_CLUSTER_LOOP"] = "1"
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-3000:])
        raise SystemExit(1)
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


if __name__ == "__main__":
    a, b = run(False), run(True)
    ok = True
    for k in a:
        same = a[k]["sha"] == b[k]["sha"]
        ok &= same and b[k]["finite"]
        print(f"{k:>14}: per-layer launches {a[k]['us_per_step']:8.2f} us/step   one launch {b[k]['us_per_step']:8.2f} us/step   "
              f"bitwise equal: {same}")
    sys.exit(0 if ok else 1)
