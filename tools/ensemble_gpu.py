#!/usr/bin/env python3
"""HIP-side null distribution of the configs[2] end-to-end numbers (GPU box).

The 500-iteration IPO (Adam, lr 0.1, L1 loss; reference run/opt_main.py:180-195) does not converge, its last iterate is
chaotic: detections that differ in the last bit (lib/dataset/synthetic.py::perturb_ulp) re-draw it - in the reference
(tests/golden/driver_pw3d_full_env*.npz, tools/gen_golden.py::gen_driver_pw3d_full_env) and here alike.  This tool runs M
such members of every configs[2] capture through the fused pipeline (3 s each) and writes, per member: the dataset-mean
MPJPE / PA-MPJPE, the per-pose best errors and the IPO end state as quantile functions (tests/_ipo_summary.py).

    python tools/ensemble_gpu.py --members 32 --out gpurun_out/ensemble_r04
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "zedo-release_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import _ipo_summary as ips                                                    # noqa: E402
import zedo_hip                                                               # noqa: E402
from lib.dataset import synthetic as syn                                      # noqa: E402
from zedo_hip.pipeline import Pipeline, ZeDOConfig                            # noqa: E402

HIP_SEED0 = 100        # the reference's members use streams 1, 2, ...; the HIP ensemble 101, 102, ... (+ 0 = unperturbed)


def member(W, g, seed):
    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    d = syn.make_poses(N, seed=int(g["seed_pose"]), conf_mode=str(g["conf_mode"]))
    cl = syn.make_clusters(H, seed=int(g["seed_cl"]))
    db2 = d["db_2d"].copy()
    db2[:, :, :2] = syn.perturb_ulp(db2[:, :, :2], seed)
    cfg = ZeDOConfig(IPO_keylist=[int(k) for k in g["keylist"]], IPO_T=float(g["ipo_T"]), IPO_minScaleT=float(g["minT"]),
                     OIL_iterations=S)
    pipe = Pipeline(W, cfg, "cuda").load(cl, db2, d["camera_param"])
    R, T = zedo_hip.ipo_fit(pipe.x0, pipe.uv, pipe.K, cfg.IPO_keylist, cfg.RotAxes, cfg.IPO_T, cfg.IPO_minScaleT,
                            cfg.IPO_maxScaleT, cfg.IPO_iterations, N * len(cfg.IPO_keylist) * 2, H * N)
    cs = torch.stack([R[:, 0, 0], R[:, 1, 0]], -1).reshape(H, N, 2).cpu().numpy()
    sm = ips.summary(cs, T.reshape(H, N, 3).cpu().numpy(), (cl - cl[:, 0:1])[:, None], db2[:, :, :2], d["camera_param"],
                     cfg.IPO_keylist, cfg.IPO_T)
    x, _ = pipe.run()
    gt = torch.as_tensor((d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float64), device="cuda")
    out = dict(sm)
    for key, proto in (("p1", False), ("p2", True)):
        _, best, idx = zedo_hip.min_mpjpe(x, gt, N, procrustes=proto)
        out["best_" + key] = best.cpu().numpy()
        out["argmin_" + key] = idx.cpu().numpy()
    out["mpjpe"], out["pa_mpjpe"] = float(out["best_p1"].mean()), float(out["best_p2"].mean())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=32)
    ap.add_argument("--captures", default="driver_pw3d_full,driver_pw3d_full_b,driver_pw3d_full_c")
    ap.add_argument("--out", default="gpurun_out/ensemble")
    a = ap.parse_args()
    W = zedo_hip.Weights(syn.make_weights(seed=0))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    rep = {}
    for name in a.captures.split(","):
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        t0 = time.time()
        seeds = [0] + [HIP_SEED0 + i for i in range(1, a.members + 1)]
        ms = [member(W, g, s) for s in seeds]
        arr = {k: np.stack([np.asarray(m[k]) for m in ms]) for k in ms[0]}
        np.savez_compressed(f"{a.out}_{name}.npz", seeds=np.array(seeds), ref_mpjpe=g["mpjpe"], ref_pa=g["pa_mpjpe"],
                            ref_best_p1=g["best_p1"], ref_best_p2=g["best_p2"], **arr)
        e1, e2 = arr["mpjpe"] * 1e3, arr["pa_mpjpe"] * 1e3
        rep[name] = dict(members=len(seeds), seconds=round(time.time() - t0, 1),
                         mpjpe_mm=dict(ref=float(g["mpjpe"]) * 1e3, unperturbed=float(e1[0]), mean=float(e1.mean()), std=float(e1.std(ddof=1)),
                                       min=float(e1.min()), max=float(e1.max())),
                         pa_mm=dict(ref=float(g["pa_mpjpe"]) * 1e3, unperturbed=float(e2[0]), mean=float(e2.mean()), std=float(e2.std(ddof=1)),
                                    min=float(e2.min()), max=float(e2.max())),
                         mean_loss=dict(mean=float(arr["mean_loss"].mean()), std=float(arr["mean_loss"].std(ddof=1))))
        print(json.dumps({name: rep[name]}), flush=True)
    with open(a.out + ".json", "w") as f:
        json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
