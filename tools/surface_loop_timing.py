#!/usr/bin/env python3
"""Time the reference's loop (run/opt_main.py:202-220) driven unchanged through the drop-in sampling_fn /
gradient_field_gen surface against the fused zedo_oil_run, at BASELINE configs[1]'s size (N = 886, H = 1).

    python tools/surface_loop_timing.py [--poses 886] [--steps 1000]  ->  one JSON line (GPU box only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zedo-release_amd"))

import numpy as np
import torch


def generic(a):
    """a 1000-step run of a non-shipped sampler configuration at BASELINE configs[1]'s 886 rows, H = 1, before / after."""
    from lib.algorithms.advanced.model import ScoreModelFC_Adv
    from lib.dataset import synthetic as syn
    from run import _driver
    cfg = _driver.load_config(os.path.join(ROOT, "zedo-release_amd", "configs", "optim", "concat_pose_optimization_h36m.py"))
    cfg.training.sde = a.generic
    cfg.sampling.probability_flow = True
    assert _driver.not_fused_because(cfg) is not None
    N, S = a.poses, a.steps
    cfg.ZeDO.OIL_iterations = S
    dev = torch.device("cuda")
    model = ScoreModelFC_Adv(cfg, 17, 3, 1024, 512, 3)
    sd = {k: torch.tensor(v) for k, v in syn.make_weights(seed=0).items()}
    sd["sigmas"] = torch.tensor(syn.sigmas_buffer())
    model.load_state_dict(sd)
    model.eval()
    d = syn.make_poses(N, seed=101, conf_mode="uniform")
    cl = syn.make_clusters(1, seed=17)
    sde = _driver.make_sde(cfg)
    out, res = {}, {}
    for tag, rt in (("host_round_trip", True), ("device_resident", False), ("host_round_trip", True), ("device_resident", False)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = _driver.stepwise_loop(cfg, model, sde, cl, d["db_2d"].copy(), d["camera_param"], S, dev, host_round_trip=rt)
        torch.cuda.synchronize()
        out.setdefault(tag, []).append(round((time.perf_counter() - t0) * 1e3, 1))
        res[tag] = x
    print(json.dumps(dict(config=f"training.sde = {a.generic}, euler_maruyama / none, probability flow: generic per-step route "
                                 "(score network on the HIP path, update rule in torch element-wise operators)",
                          poses=N, steps=S, host_threads=torch.get_num_threads(), ms_per_pass_incl_ipo=out,
                          ms_per_step={k: round(min(v) / S, 4) for k, v in out.items()},
                          speedup=round(min(out["host_round_trip"]) / min(out["device_resident"]), 3),
                          bitwise_equal=bool(torch.equal(res["host_round_trip"], res["device_resident"])))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=886)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--generic", default=None, metavar="SDE",
                    help="time run/_driver.py::stepwise_loop for a configuration OUTSIDE the fused pipeline (e.g. vpsde): the "
                         "device-resident loop (sampling_fn.step_device) against the reference's per-step D2H/H2D round trip")
    a = ap.parse_args()
    if a.generic:
        return generic(a)
    from lib.algorithms.advanced import sampling
    from lib.algorithms.advanced.model import ScoreModelFC_Adv
    from lib.algorithms.advanced.simple_zeroshot_opt import gradient_field_gen
    from lib.dataset import synthetic as syn
    from run import _driver
    from zedo_hip.pipeline import Pipeline, ZeDOConfig
    import zedo_hip as zh
    cfg = _driver.load_config(os.path.join(ROOT, "zedo-release_amd", "configs", "optim", "concat_pose_optimization_h36m.py"))
    cfg.sampling.probability_flow = True
    N, S = a.poses, a.steps
    cfg.ZeDO.OIL_iterations = S
    dev = torch.device("cuda")
    w = syn.make_weights(seed=0)
    model = ScoreModelFC_Adv(cfg, 17, 3, 1024, 512, 3)
    sd = {k: torch.tensor(v) for k, v in w.items()}
    sd["sigmas"] = torch.tensor(syn.sigmas_buffer())
    model.load_state_dict(sd)
    model.eval()
    d = syn.make_poses(N, seed=101, conf_mode="uniform")
    cl = syn.make_clusters(1, seed=17)
    sde = _driver.make_sde(cfg)
    pipe = Pipeline(model.hip_weights(), ZeDOConfig.h36m(OIL_iterations=S), dev).load(cl, d["db_2d"], d["camera_param"])
    x0, T0 = pipe.run(oil_steps=0)                       # IPO + rotate only
    # fused loop
    times = {}
    for rep in range(2):
        x, T = x0.clone(), T0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        zh.oil_run(pipe.weights, pipe.sched, x, pipe.geom, T, 0, S, S // 5)
        torch.cuda.synchronize()
        times["fused"] = (time.perf_counter() - t0) / S
    x_fused = x
    # the reference's statements, per step, through the surface
    sampling_fn = sampling.get_sampling_fn(cfg, sde, (N, 17, 3), lambda v: v, cfg.ZeDO.sampling_eps, device=dev)
    condition = torch.tensor(d["db_2d"][:, :, :2], device=dev).float()
    conf = torch.tensor(d["db_2d"][:, :, 2], device=dev).float()
    K = torch.tensor(d["camera_param"], device=dev).float()
    timestamp = torch.linspace(sde.T, cfg.ZeDO.sampling_eps, S, device=dev)
    sect = {}

    def surface_loop(tag):
        nonlocal_T = T0.clone().reshape(N, 1, 3)
        denoise_x, T = x0.clone(), nonlocal_T
        acc = dict(gradient_field_gen=0.0, add=0.0, sampling_fn=0.0, retensor=0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(S):
                ta = time.perf_counter()
                if i < S // 5:
                    g = gradient_field_gen(condition, denoise_x, K, t=T, conf=conf, returnT=False)
                else:
                    g, T = gradient_field_gen(condition, denoise_x, K, conf=conf, returnT=True)
                tb = time.perf_counter()
                denoise_x += g
                tc = time.perf_counter()
                trajs, results = sampling_fn(model, condition=condition * 0, gradient=g, denoise_x=denoise_x,
                                             t=timestamp[i], t_step=i, args=None)
                td = time.perf_counter()
                denoise_x = torch.tensor(results).to(dev)            # run/opt_main.py:220, as written
                te = time.perf_counter()
                acc["gradient_field_gen"] += tb - ta; acc["add"] += tc - tb; acc["sampling_fn"] += td - tc
                acc["retensor"] += te - td
        torch.cuda.synchronize()
        times[tag] = (time.perf_counter() - t0) / S
        sect[tag] = {k: round(v / S * 1e3, 4) for k, v in acc.items()}
        return denoise_x

    nthreads = torch.get_num_threads()
    surface_loop("surface")
    denoise_x = surface_loop("surface")
    # the same statements with the host's intra-op pool limited to 8 threads: on a 256-thread host the caller's
    # torch.tensor(results) (a 180 KB CPU copy) wakes the whole pool and costs more than the GPU step
    torch.set_num_threads(8)
    surface_loop("surface_8_threads")
    surface_loop("surface_8_threads")
    torch.set_num_threads(nthreads)
    ls = sampling_fn.loop_schedule
    print(json.dumps(dict(poses=N, steps=S, fused_ms_per_step=round(times["fused"] * 1e3, 4),
                          surface_ms_per_step=round(times["surface"] * 1e3, 4),
                          ratio=round(times["surface"] / times["fused"], 2),
                          surface_8_threads_ms_per_step=round(times["surface_8_threads"] * 1e3, 4),
                          ratio_8_threads=round(times["surface_8_threads"] / times["fused"], 2),
                          host_threads=nthreads, host_ms_per_step_by_statement=sect,
                          schedule_hits=ls.hits, schedule_misses=ls.misses,
                          bitwise_equal=bool(torch.equal(denoise_x, x_fused)))))


if __name__ == "__main__":
    main()
