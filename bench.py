#!/usr/bin/env python3
"""bench.py - poses/sec of the ZeDO optimisation-in-the-loop sampling path on MI355X.

One "step" = one full pass of the hot path over one synthetic batch shaped like BASELINE configs[2]
(3DPW: N=1015 poses x H=50 hypotheses, IPO 500 iterations, S=1000 OIL steps, P1+P2 selection), inputs
resident in HBM when the timed region starts.  `value` = poses fully processed per second over all
ranks (N * n_gpus * steps / time); for --gpus > 1 (weak scaling: 1015 poses per GPU) the H*N_total rows are
sharded contiguously over the ranks and the per-pose minimum is combined with one RCCL MIN all-reduce.

Launch forms (both give ONE JSON line from rank 0 with n_gpus = rccl_ranks = number of ranks):
  python bench.py --gpus N ...                          the parent - before ANY GPU call - starts N fresh child processes
                                                        (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), one per GPU, and
                                                        exits non-zero if any of them fails;
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (ranks from the environment).
--scaling weak (default): --poses per GPU;  --scaling strong: --poses in total, rows sharded over the ranks.
--workload 2 (default) BASELINE configs[2]: 3DPW settings, 1015 poses x H=50, P1+P2 selection (MIN all-reduce);
           3 configs[3]: H36M settings, the full test set (567 040 poses) x H=50, action-wise selection - strong by
             definition (weak: 70 880 poses per GPU);
           4 configs[4]: run.inference without --eval on 100 000 detections x H=50: no selection, the one exchange is
             the all-gather of every hypothesis of every pose (weak: 12 500 poses per GPU).

Every run with a process group (--gpus N > 1, or ZEDO_FORCE_DIST=1) additionally
  * checks itself before anything is timed (multi_rank_selfcheck): a fixed small problem sharded over the live ranks through BOTH exchange
    steps of the path - the P1/P2 MIN selection and the all-gather of run.inference - against the unsharded run on rank 0, whatever the
    workload; a mismatch of either ends every rank with exit code 4;
  * (workload 2) adds a `strong` object beside the weak-scaling headline: BASELINE configs[2]'s 1015 poses x H=50 as ONE fixed problem,
    on rank 0 alone and split over all N ranks, timed back to back (--strong-steps passes each, both math modes): speed-up, efficiency,
    the projection from one-GPU shard measurements (profiles/strong_shards_r*.jsonl) and bit-identity of the two selections.

The JSON line also carries
  roofline     : the dominant kernel (the four 1024x1024 fp32-MFMA dense layers): algorithmic FLOP per launch
                 / its average launch duration, measured live with sampled HIP events on the launch stream;
  cpu_baseline : the CPU port under oracle/ (numpy IPO + torch-CPU-operator OIL steps on every host core, i.e. the
                 operators the reference itself runs on a CPU) timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "zedo-release_amd"),):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

N_POSES, N_HYPO, S_OIL = 1015, 50, 1000
FLOP_PER_ROW_STEP = 2 * (51 * 1024 + 4 * 1024 * 1024 + 1024 * 51)   # 8 597 504
PEAK_FP32_MFMA_TFLOPS = 157.3                                        # MI355X_MICROARCH.md


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(weights, cfg_kw, seed):
    """CPU port of the reference path (run/opt_main.py:166-228) on the host cores: one hypothesis slice of the reference
    batch (1015 poses): IPO 500 iterations (numpy oracle) + a slice of the OIL loop that crosses the switch to the
    least-squares T at the loop's own 1/5 point (torch CPU operators = what the reference itself runs on a CPU) + the
    P1 / P2 metric, extrapolated per pose-hypothesis to S = 1000 and H = 50.  Timed at the fastest thread count the box
    offers AND on one thread.  Test infrastructure used only as a baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import zedo_oracle as O
    import zedo_oracle_mt as M
    from lib.dataset import synthetic as syn
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    n, steps, steps_1t = N_POSES, 120, 15
    d = syn.make_poses(n, seed=seed)
    cl = syn.make_clusters(1, seed=seed)
    cond, K = d["db_2d"][:, :, :2], d["camera_param"]
    x0 = np.broadcast_to((cl - cl[:, 0:1])[0][None], (n, 17, 3)).astype(np.float32)
    kl = list(range(17))
    t0 = time.perf_counter()
    T0 = O.ipo_init_T(cond, K, 8.0)
    R, T, _, _, _ = O.ipo_fit(x0[:, kl], T0, K, cond[:, kl], "z", 0.2, 2.0, 500)
    t_ipo = time.perf_counter() - t0
    port = M.StepPort(weights, cond, K, d["db_2d"][:, :, 2].copy())
    x_init = torch.tensor(np.einsum("bij,bkj->bki", R, x0).astype(np.float32))
    T_init = torch.tensor(T.astype(np.float32))
    ts = O.oil_timestamps(S_OIL)
    # thread count: more is not faster for [1015 x 1024] operands, and the GPU boxes' hosts are SHARED (load average 30-70 of 256 threads from other
    # tenants: at 64 threads a step takes 13 ms or 80 ms from one call to the next, at 16-32 threads 12-15 ms with a 25 ms mean; 128: 200 ms):
    # every candidate runs a 12-step slice and its MEAN step (what the reported slice will be made of) decides
    x, Tt = x_init, T_init
    cores, best, probes = 1, float("inf"), {}
    for th in [c for c in (8, 16, 32, 64, 128) if c <= avail] or [avail]:
        torch.set_num_threads(th)
        for i in range(2):
            port.step(x, Tt, ts[0], False)              # warm the pool at this size
        dts = []
        for i in range(12):
            t0 = time.perf_counter()
            port.step(x, Tt, ts[i], i >= 3)
            dts.append(time.perf_counter() - t0)
        dt = float(np.mean(dts))
        probes[th] = round(dt * 1e3, 1)
        if dt < best:
            cores, best = th, dt
        if dt > 2.0 * best:
            break

    def oil_slice(nsteps):       # the first 1/5 with the IPO's T, the rest with the least-squares T, like the loop itself
        x, Tt = x_init, T_init
        port.step(x, Tt, ts[0], False)
        t0 = time.perf_counter()
        for i in range(nsteps):
            x, Tt = port.step(x, Tt, ts[i * (S_OIL // nsteps)], i >= nsteps // 5)
        return (time.perf_counter() - t0) / nsteps, x

    torch.set_num_threads(cores)
    t_step, x = oil_slice(steps)
    torch.set_num_threads(1)
    t_step_1t, _ = oil_slice(steps_1t)
    torch.set_num_threads(cores)
    gt = d["db_3d"] - d["db_3d"][:, 0:1]
    t0 = time.perf_counter()
    O.hypothesis_errors(x.numpy()[:, None], gt, False)
    O.hypothesis_errors(x.numpy()[:, None], gt, True)
    t_eval = time.perf_counter() - t0
    per_pose_hyp = (t_ipo + S_OIL * t_step + t_eval) / n
    per_pose_hyp_1t = (t_ipo + S_OIL * t_step_1t + t_eval) / n
    return dict(value=1.0 / (N_HYPO * per_pose_hyp), unit="poses/s", cores=int(cores), kind="port",
                cpu_model=cpu_model_name(), threads_available=int(avail), thread_count_probe_ms_per_step=probes,
                host_load_average=[round(v, 1) for v in os.getloadavg()],      # the hosts are shared: this baseline moves with the neighbours' load
                one_thread=dict(value=1.0 / (N_HYPO * per_pose_hyp_1t), unit="poses/s", cores=1,
                                ms_per_step=round(t_step_1t * 1e3, 1), steps_timed=steps_1t),
                ms_per_step=round(t_step * 1e3, 2), steps_timed=steps, ipo_s=round(t_ipo, 2), metric_s=round(t_eval, 2),
                # the reference itself (its own modules imported on CPU) could only be timed in the build container
                # (8-core Xeon @ 2.10 GHz): 0.40-0.43 poses/s at H=50, S=1000 on 8 threads,
                # 5 862 pose-steps/s = 0.117 poses/s on one thread; this port measured 0.41 there
                reference_in_build_container=dict(value=[0.40, 0.43], unit="poses/s", cores=8, one_thread_value=0.117,
                                                  cpu_model="Intel Xeon @ 2.10 GHz (8 cores)", source="BASELINE.md section 2"),
                sample=f"CPU port (oracle/), one hypothesis slice of {n} poses: IPO 500 it, numpy ({t_ipo:.1f} s) + {steps} OIL steps "
                       f"spread over the {S_OIL}-step schedule, the first {steps // 5} with the IPO's T and the rest with the "
                       f"least-squares T (torch CPU operators on {cores} of {avail} usable threads - the fastest count: "
                       f"{t_step * 1e3:.1f} ms/step; one thread: {t_step_1t * 1e3:.0f} ms/step over {steps_1t} steps) + P1/P2 metric, "
                       f"extrapolated to H={N_HYPO}, S={S_OIL}")


def f16x3_tile_split(rows_launch, cus=256):
    """Rows per tile shape of one split-fp16 hidden-layer launch: the branch logic of zedo_gemm16.hip::launch_layer16,
    restated (rows padded to 64).  -> [(BM, BN, rows), ...]"""
    mp = -(-int(rows_launch) // 64) * 64
    if mp <= 2048:
        return [(64, 64, mp)]
    per_round = cus * 2 * 128 // (1024 // 256)
    whole = (mp // per_round) * per_round
    rest = mp - whole
    if whole == 0 and rest < 8192:
        mid = (mp // 128) * 128
        return [(128, 128, mid), (64, 64, mp - mid)]
    big = whole + (rest // 128) * 128 if rest >= 5120 else whole
    return [(128, 256, big), (64, 64, mp - big)]


def roofline_f16x3(h2, rows_launch, box16=None):
    """Roofline object of the split-fp16 hidden layer: three fp16 MFMAs (al.bh + ah.bl + ah.bh) per fp32 product block:
    ISSUED flop against the dense fp16 peak, plus the bytes the tiles pull through LDS-DMA against what that path
    sustains (the binding resource).  The tile model follows the launch's own shape choice (f16x3_tile_split)."""
    issued = 3 * 2.0 * rows_launch * 1024 * 1024
    ghz = h2.get("shader_clock_ghz") or 0.0
    split = [t for t in f16x3_tile_split(rows_launch) if t[2] > 0]
    # per tile and 16-deep k block both operands' planes: (BM + BN) rows x 64 bytes; 64 k blocks; 1024 / BN column tiles
    dma_bytes = sum((r / bm) * (1024 / bn) * (bm + bn) * 64 * 64 for bm, bn, r in split)
    shape = " + ".join(f"{bm}x{bn} tiles on {r} rows" for bm, bn, r in split)
    t = h2["avg_ms"] * 1e-3
    return dict(bound="mfma", kernel=f"zedo::layer16_pair_kernel / layer16_small_kernel ({shape}; split-fp16 operands as k-block-major planes, 3 x v_mfma_f32_32x32x16_f16 per 16-k block)",
                achieved=round(issued / t / 1e12, 1), peak=2500.0, unit="TFLOP/s", frac=round(issued / t / 1e12 / 2500.0, 4),
                traffic=f16x3_traffic(rows_launch), fp32_equivalent_tflops=round(2.0 * rows_launch * 1024 * 1024 / t / 1e12, 1),
                avg_launch_ms=round(h2["avg_ms"], 4), sampled_launches=h2["samples"], launches=h2["launches"],
                kernel_shader_clock_ghz=round(ghz, 3) if ghz else None,
                frac_at_kernel_clock=(round(issued / t / 1e12 / (2500.0 * ghz / 2.4), 4) if ghz else None),
                # the attainable ceiling: a bare v_mfma_f32_32x32x16_f16 stream on THIS box, measured right after the timed region
                # (zedo_probe_mfma_peak_f16: two waves per SIMD, no LDS / memory traffic) and the clock power management granted it
                box_f16_mfma_peak_measured=(round(box16[0], 1) if box16 else None),
                box_f16_shader_clock_ghz=(round(box16[1], 3) if box16 else None),
                frac_of_box_peak=(round(issued / t / 1e12 / box16[0], 4) if box16 else None),
                lds_dma_bytes_per_launch=int(dma_bytes), lds_dma_tb_per_s=round(dma_bytes / t / 1e12, 2),
                lds_dma_sustained_tb_per_s=(round(16.0 * 256 * ghz * 1e9 / 1e12, 2) if ghz else None),
                vmem_model_ms=round(((rows_launch / 32) * 32 * 64 * 3 * 32 + dma_bytes / 1024 * 65 + rows_launch * 4096 / 1024 * 187)
                                    / 1024 / (ghz * 1e9) * 1e3, 4) if ghz else None,
                note="power management holds the shader clock at 1.5-1.85 GHz under the dense fp16 MFMA stream (lower the busier the pipe). "
                     "vmem_model_ms = (MFMA + LDS-DMA + store issue cycles) / 1024 SIMDs at the kernel's clock: round 3's additive model, kept "
                     "for continuity - round 5 showed that an LDS-DMA stalls only the wave that issues it (profiles/vmem_issue_r05.txt); what "
                     "the layer sits on is the MFMA floor at that clock (165-198 us), the DMA stream of the 128 x 256 tiles (91 us alone with "
                     "k-block-major planes, 154 us before) and the epilogue, which overlap only partly (profiles/f16x3_designs_r05.txt)")


def f16x3_traffic(rows_launch):
    """HBM-side bytes per split-fp16 hidden launch from the committed counter pass (profiles/hbm_traffic_f16x3.json,
    tools/pmc_f16x3.sh), for the launch shape it was collected on; None otherwise."""
    tp = os.path.join(ROOT, "profiles", "hbm_traffic_f16x3.json")
    if not os.path.exists(tp):
        return None
    tj = json.load(open(tp))
    return tj.get("hidden_dense_bytes_per_launch") if int(tj.get("rows_launch", -1)) == int(rows_launch) else None


HBM_ACHIEVABLE_GBS = 6300.0        # MI355X_MICROARCH.md: what a streaming kernel sustains of the 8 TB/s HBM3E peak


def geometry_kernels(zh, dev, sizes=((N_POSES, N_HYPO), (70880, N_HYPO)), reps=5):
    """the HBM / latency-bound kernels of the path - IPO, the stand-alone reprojection correction (the same reproj_row the
    post_dense epilogue fuses), rotate_init, row_error + pose_min (P1, and P2 with its fp64 Jacobi SVD) - timed on their own with events
    on the launch stream, at BASELINE configs[2]'s 50 750 rows and at configs[3]'s per-GPU shard of 3 544 000 rows: microseconds,
    algorithmic bytes (every operand the kernel addresses, counted once: per-row state + per-pose constants), GB/s and the fraction of
    the 6.3 TB/s a streaming kernel sustains.  IPO and P2 are arithmetic-bound (500 Adam iterations in registers; an fp64 3x3 SVD per
    row): their GB/s say how far from the HBM roof the arithmetic keeps them, not how well they stream."""
    from lib.dataset import synthetic as syn
    out = {}
    for N, H in sizes:
        B = N * H
        d = syn.make_poses(N, seed=11)
        cl = syn.make_clusters(H, seed=11)
        t = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        x0 = t(cl - cl[:, 0:1])
        uv, K, conf = t(d["db_2d"][:, :, :2]), t(d["camera_param"]), t(d["db_2d"][:, :, 2])
        geom = zh.reproj_prepare(uv, K, conf)
        gt = t((d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float64), torch.float64)
        kl17, kl3 = list(range(17)), [0, 1, 4]

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            ms = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn()
                e1.record()
                e1.synchronize()
                ms.append(e0.elapsed_time(e1))
            return float(np.median(ms)), r

        rec = {}

        def put(name, ms, nbytes, bound):
            gbs = nbytes / (ms * 1e-3) / 1e9
            rec[name] = dict(us=round(ms * 1e3, 1), bytes=int(nbytes), gb_per_s=round(gbs, 1), frac_of_6p3_tb_s=round(gbs / HBM_ACHIEVABLE_GBS, 4), bound=bound)

        ms, (R, T) = timed(lambda: zh.ipo_fit(x0, uv, K, kl17, "z", 8.0, 0.2, 2.0, 500, N * 17 * 2, B))
        put("ipo_17_joints_500_it", ms, B * (17 * 20 + 44 + 48), "VALU issue / latency: 500 Adam iterations x 17 joints in registers, ~0.25 MFLOP per row")
        ms, _ = timed(lambda: zh.ipo_fit(x0, uv, K, kl3, "z", 3.0, 0.5, 2.0, 500, N * 3 * 2, B))
        put("ipo_3_joints_500_it", ms, B * (3 * 20 + 44 + 48), "VALU issue / latency")
        ms, x = timed(lambda: zh.rotate_init(x0, R, N))
        put("rotate_init", ms, B * (36 + 204) + H * 204, "HBM (reads R, writes the pose rows)")
        Tc = T.clone()
        ms, _ = timed(lambda: zh.reproj_grad(x, geom, Tc, True))
        put("reprojection_least_squares_T", ms, B * (204 + 12 + 204 + 12) + N * 544, "HBM (pose rows in and out; per-pose rays from the L2)")
        ms, _ = timed(lambda: zh.min_mpjpe(x, gt, N, False))
        put("row_error_pose_min_P1", ms, B * (204 + 8 + 8) + N * (408 + 12), "HBM")
        ms, _ = timed(lambda: zh.min_mpjpe(x, gt, N, True))
        put("row_error_pose_min_P2_procrustes", ms, B * (204 + 8 + 8) + N * (408 + 12), "fp64 arithmetic: one 3x3 Jacobi SVD per row")
        out[f"{B}_rows"] = rec
        del x, R, T, Tc, geom
    return out


def selection_digest(out):
    """sha256 over the per-pose best errors and winning hypothesis indices of both protocols (bit-exact
    comparison of two runs, e.g. with and without the RCCL exchange step)."""
    import hashlib
    h = hashlib.sha256()
    for k in ("p1", "p2"):
        h.update(out[k][0].cpu().numpy().tobytes())
        h.update(out[k][1].cpu().numpy().astype(np.int32).tobytes())
    return h.hexdigest()[:16]


WORKLOADS = {
    # poses = the configuration's TOTAL pose count; weak scaling gives every GPU `weak` poses instead
    2: dict(name="BASELINE configs[2] (3DPW pw3d_test, H=50, 1000 steps)", settings="pw3d", poses=1015, weak=1015,
            select="p1p2", default_scaling="weak"),
    3: dict(name="BASELINE configs[3] (H36M full test set = ZeDO.sample 1: 567 040 poses, H=50)", settings="h36m", poses=567040,
            weak=70880, select="h36m", default_scaling="strong"),
    4: dict(name="BASELINE configs[4] (run.inference, 100 000 synthetic 2D detections, H=50, no --eval)", settings="h36m",
            poses=100000, weak=12500, select="gather", default_scaling="strong"),
}


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def rank_environments(n, port, base=None):
    """Environment of each of the n ranks the launcher starts (one process per GPU, RCCL rendezvous on 127.0.0.1)."""
    envs = []
    for r in range(n):
        e = dict(os.environ if base is None else base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC
        envs.append(e)
    return envs


def supervise(procs, poll_s=0.2):
    """Wait for the rank processes: the first rank that exits non-zero ends the others (terminate, kill after 20 s) and
    its code (1 for a signal) is returned; 0 when every rank exits 0."""
    import subprocess
    rc = 0
    try:
        live = set(range(len(procs)))
        while live and rc == 0:
            time.sleep(poll_s)
            for r in sorted(live):
                c = procs[r].poll()
                if c is not None:
                    live.discard(r)
                    if c != 0:
                        print(f"bench.py: rank {r} exited with code {c}", file=sys.stderr)
                        rc = c if c > 0 else 1
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=20)
            except subprocess.TimeoutExpired:
                pr.kill()
    return rc


def launch_ranks(n, argv, dry=False, cmd=None, build=True):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh processes, one per GPU.  The parent makes
    no HIP call of its own (it only counts devices; the ranks are fresh child processes either way, never an exec of
    this one), builds / verifies libzedo_hip.so ONCE before it starts them and exports ZEDO_NO_BUILD=1, so that the
    ranks fail loudly instead of racing N `make`s; a failing rank ends the others and the parent exits with its code.
    ZEDO_SHARE_DEVICE=1 + ZEDO_DIST_BACKEND=gloo: all N ranks on device 0 (rehearsal on a one-GPU box).
    build=False (tests of the launcher logic with stand-in rank commands): the parent neither builds nor looks for the library."""
    import subprocess
    import zedo_build
    share = os.environ.get("ZEDO_SHARE_DEVICE") == "1"
    visible = torch.cuda.device_count()
    port = free_port()
    envs = rank_environments(n, port)
    for e in envs:
        e["ZEDO_NO_BUILD"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__)] + argv if cmd is None else cmd
    if dry:
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "ZEDO_NO_BUILD")
        print(json.dumps({"dry_launch": True, "gpus": n, "visible_gpus": visible, "cmd": cmd,
                          "library": zedo_build.LIB_PATH, "library_present": os.path.exists(zedo_build.LIB_PATH),
                          "ranks": [{k: e[k] for k in keys} for e in envs]}), flush=True)
        return 0
    if visible < (1 if share else n):
        print(f"bench.py: --gpus {n} but only {visible} GPU(s) are visible", file=sys.stderr)
        return 2
    if share and os.environ.get("ZEDO_DIST_BACKEND", "nccl").lower() != "gloo":
        print("bench.py: ZEDO_SHARE_DEVICE=1 needs ZEDO_DIST_BACKEND=gloo (RCCL refuses two ranks on one device)", file=sys.stderr)
        return 2
    if build:
        try:
            # once, here - never in the ranks; a parent that was itself told not to build (ZEDO_NO_BUILD=1) only verifies
            zedo_build.ensure_library(allow_build=os.environ.get("ZEDO_NO_BUILD") != "1")
        except ImportError as e:
            print(f"bench.py: {e}", file=sys.stderr)
            return 3
    return supervise([subprocess.Popen(cmd, env=e) for e in envs])

SELFCHECK = dict(poses=64, hypo=5, oil=20)     # the fixed small problem of multi_rank_selfcheck


def multi_rank_selfcheck(zp, dist, weights, wl, dev, rank, world):
    """Before anything is timed: does the N-rank result equal the one-rank result, bit for bit, on THIS transport - for BOTH
    exchange steps of the path, whatever the workload: the P1 / P2 MIN selection (run/opt_main.py:224-228: two all-reduces per
    protocol) and the all-gather of every hypothesis of every pose (run/inference.py:233-236)?
    The reference is one GPU running its hypotheses one after the other (run/opt_main.py:166); sharding rows over ranks is this
    repository's addition, and `bench.py --gpus N` under the driver is the only place a process group with more than one RCCL
    member ever runs - so the run checks itself.  A fixed small problem (64 poses x 5 hypotheses x 20 steps, the workload's own
    settings) is processed (a) sharded over the ranks of the live process group, through the very exchange steps, and (b) unsharded
    on rank 0 with no collective at all; the digests of both kinds travel in the JSON line and a mismatch of EITHER makes every rank
    exit with code 4 before the timed region.
    ZEDO_BENCH_CORRUPT_RANK=k (test hook): rank k adds 1e-3 m to its shard before the exchange; ZEDO_BENCH_CORRUPT_KIND=selection|gather
    restricts the corruption to one of the two checks (default: both)."""
    import hashlib
    from zedo_hip.pipeline import Pipeline, ZeDOConfig, gather_row_shards, reduce_min_over_ranks, shard_rows
    from lib.dataset import synthetic as syn
    N, H, S = SELFCHECK["poses"], SELFCHECK["hypo"], SELFCHECK["oil"]
    h36m = wl["settings"] == "h36m"
    mm_gt = h36m and wl["select"] == "h36m"
    d = syn.make_poses(N, seed=77, dtype3d=np.float64 if mm_gt else np.float32)
    cfg = (ZeDOConfig.h36m if h36m else ZeDOConfig.pw3d)(OIL_iterations=S)
    pipe = Pipeline(weights, cfg, dev).load(syn.make_clusters(H, seed=77), d["db_2d"], d["camera_param"])
    if mm_gt:
        mm = d["db_3d"] * 1000.0
        gt = (mm - mm[:, 0:1]) / 1000.0
    else:
        gt = (d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float64)
    gt_dev = torch.tensor(gt, dtype=torch.float64, device=dev)
    corrupt = os.environ.get("ZEDO_BENCH_CORRUPT_RANK")
    corrupt_kind = os.environ.get("ZEDO_BENCH_CORRUPT_KIND")

    def digest(kind, x, lo, exchange):
        h = hashlib.sha256()
        if kind == "gather":
            res = gather_row_shards(x, H * N) if exchange else x
            h.update(res.cpu().numpy().tobytes())
        else:
            sel = pipe.select(x, gt_dev, row_offset=lo)
            for k in ("p1", "p2"):
                best, idx = reduce_min_over_ranks(*sel[k]) if exchange else sel[k]
                h.update(best.cpu().numpy().tobytes())
                h.update(idx.cpu().numpy().astype(np.int32).tobytes())
        return h.hexdigest()[:16]

    lo, rows = shard_rows(H * N, rank, world)
    x, _ = pipe.run(row_offset=lo, rows=rows)
    xw = pipe.run()[0] if rank == 0 else None
    out, bad = {}, 0
    what = {"selection": "P1/P2 MIN exchange (two all-reduces per protocol)", "gather": "all-gather of every hypothesis of every pose"}
    for kind in ("selection", "gather"):
        xs = x + 1e-3 if (corrupt is not None and rank == int(corrupt) and corrupt_kind in (None, "", kind)) else x
        sharded = digest(kind, xs, lo, True)
        whole = digest(kind, xw, 0, False) if rank == 0 else None
        if rank == 0 and whole != sharded:
            bad |= 1 if kind == "selection" else 2
        out[kind] = dict(sha=sharded, sha_unsharded=whole, exchange=what[kind])
    flag = torch.tensor([bad], dtype=torch.int64, device=dev)
    zp.all_reduce(flag, dist.ReduceOp.MAX)
    bad = int(flag.item())
    out["selection"]["ok"], out["gather"]["ok"] = not (bad & 1), not (bad & 2)
    return dict(ok=bad == 0, backend=dist.get_backend(), ranks=world, **out,
                problem=f"{N} poses x {H} hypotheses x {S} steps, {wl['settings']} settings; rows sharded {world}-way vs unsharded on rank 0")


STRONG_SHARD_FILES = ("strong_shards_r06.jsonl", "strong_shards_r04.jsonl")      # newest first


def strong_projection(rows_per_rank, math):
    """What ONE GPU measured for a shard of this many rows of workload 2 (profiles/strong_shards_r*.jsonl: bench.py --poses ...
    on one MI355X) -> (rows, pass time in ms, file), or None when no shard within 2 % of this size was measured."""
    tp = next((q for q in (os.path.join(ROOT, "profiles", f) for f in STRONG_SHARD_FILES) if os.path.exists(q)), None)
    if tp is None:
        return None
    best = None
    for l in open(tp):
        try:
            j = json.loads(l)
        except ValueError:
            continue
        r = j.get("config", {}).get("rows_per_gpu")
        if not r or abs(r - rows_per_rank) > 0.02 * rows_per_rank:
            continue
        ms = j["ms_per_step"] if j.get("math") == math else (j.get("alt_mode") or {}).get("ms_per_step")
        if ms and (best is None or abs(r - rows_per_rank) < abs(best[0] - rows_per_rank)):
            best = (r, float(ms), os.path.relpath(tp, ROOT))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", type=int, default=2, choices=sorted(WORKLOADS), help="BASELINE.json configs[i]")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="weak: --poses per GPU (default for workload 2); strong: --poses in total (default for 3, 4)")
    ap.add_argument("--poses", type=int, default=None, help="override the workload's pose count (per GPU if weak, total if strong)")
    ap.add_argument("--hypo", type=int, default=N_HYPO)
    ap.add_argument("--oil", type=int, default=S_OIL)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--math", choices=("f32", "f16x3"), default="f32",
                    help="arithmetic of the hidden layers for the HEADLINE numbers (default: exact fp32 MFMA)")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the second timed run in the other math mode (alt_mode object)")
    ap.add_argument("--dry-launch", action="store_true", help="print the rank environments the launcher would start, and exit")
    ap.add_argument("--strong-poses", type=int, default=N_POSES,
                    help="poses of the fixed problem of the `strong` object (multi-rank runs of workload 2; default BASELINE configs[2]'s 1015)")
    ap.add_argument("--strong-steps", type=int, default=3, help="timed passes of the `strong` object (each also run by rank 0 alone)")
    ap.add_argument("--no-strong", action="store_true", help="multi-rank runs: skip the `strong` object")
    ap.add_argument("--no-geometry", action="store_true", help="skip the geometry_kernels object (one-GPU runs: IPO / reprojection / selection "
                                                                 "kernels timed on their own at 50 750 and 3 544 000 rows)")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # ZEDO_BENCH_FORCE_LAUNCH=1: go through the launcher with one rank too (how the launcher itself is exercised on a one-GPU box)
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.dry_launch or os.environ.get("ZEDO_BENCH_FORCE_LAUNCH") == "1"):
        sys.exit(launch_ranks(a.gpus, [x for x in sys.argv[1:] if x != "--dry-launch"], dry=a.dry_launch))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    import torch.distributed as dist
    from zedo_hip import pipeline as zp
    local = zp.local_device_index()        # LOCAL_RANK; device 0 for every rank with ZEDO_SHARE_DEVICE=1 (rehearsal)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or zp.force_dist()   # ZEDO_FORCE_DIST=1: 1-rank test of the RCCL path
    if use_dist:
        zp.init_dist(rank, world, dev, 29511)   # backend nccl == RCCL on ROCm (ZEDO_DIST_BACKEND=gloo: rehearsal transport)

    import zedo_hip as zh
    from zedo_hip.pipeline import Pipeline, ZeDOConfig, gather_row_shards, reduce_min_over_ranks, shard_rows
    from lib.dataset import synthetic as syn

    # ---- synthetic problem: every rank builds the same global problem, processes its own row shard
    wl = WORKLOADS[a.workload]
    scaling = a.scaling or wl["default_scaling"]
    per = a.poses if a.poses is not None else (wl["weak"] if scaling == "weak" else wl["poses"])
    N_total = per * world if scaling == "weak" else per
    H, S = a.hypo, a.oil
    stated = (a.poses is None and H == N_HYPO and S == S_OIL)
    weights = syn.make_weights(seed=0)
    h36m = wl["settings"] == "h36m"
    d = syn.make_poses(N_total, seed=2024, dtype3d=np.float64 if (h36m and wl["select"] == "h36m") else np.float32)
    clusters = syn.make_clusters(H, seed=2024)
    cfg = (ZeDOConfig.h36m if h36m else ZeDOConfig.pw3d)(OIL_iterations=S)
    os.environ["ZEDO_MATH"] = a.math
    pipe = Pipeline(weights, cfg, dev).load(clusters, d["db_2d"], d["camera_param"])
    if wl["select"] == "h36m":          # millimetre float64 ground truth, centred the way h36m.py:400-401 does
        mm = d["db_3d"] * 1000.0
        gt = (mm - mm[:, 0:1]) / 1000.0
        actions = 2 + (np.arange(N_total) % 15)
    else:
        gt = (d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float64)
    gt_dev = torch.tensor(gt, dtype=torch.float64, device=dev) if wl["select"] != "gather" else None
    lo, rows = shard_rows(H * N_total, rank, world)

    fail_rank = os.environ.get("ZEDO_BENCH_FAIL_RANK")      # test hook: that rank dies before the exchange step

    selfcheck = None
    if use_dist:     # every run that has a process group checks N ranks == 1 rank before it times anything
        selfcheck = multi_rank_selfcheck(zp, dist, weights, wl, dev, rank, world)
        if not selfcheck["ok"]:
            if rank == 0:
                for kind in ("selection", "gather"):
                    if not selfcheck[kind]["ok"]:
                        print(f"bench.py: multi_rank_selfcheck FAILED ({kind}) - {world} ranks over {selfcheck['backend']} give "
                              f"{selfcheck[kind]['sha']}, one rank gives {selfcheck[kind]['sha_unsharded']} ({selfcheck['problem']})",
                              file=sys.stderr, flush=True)
            sys.exit(4)

    def one_pass():
        x, T = pipe.run(row_offset=lo, rows=rows)
        if fail_rank is not None and rank == int(fail_rank):
            torch.cuda.synchronize()
            os._exit(9)
        if wl["select"] == "gather":     # run/inference.py:233-236: every hypothesis of every pose, on every rank
            return x, {"results": gather_row_shards(x, H * N_total)}
        sel = pipe.select(x, gt_dev, row_offset=lo)
        out = {}
        for k, (best, idx) in sel.items():
            out[k] = reduce_min_over_ranks(best, idx)
        if wl["select"] == "h36m":       # action-wise mean of means (h36m.py:424-433) on the host, like eval_multi
            for k in ("p1", "p2"):
                b = out[k][0].cpu().numpy()
                out[k + "_actionwise"] = float(np.mean([b[actions == act].mean() for act in range(2, 17)]))
        return x, out

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            zp.barrier()
            torch.cuda.synchronize()

    def timed_run():
        for _ in range(a.warmup):
            one_pass()
        fence()
        zh.profile_start(sample_every=37, max_samples=4096)   # prime stride: samples all four hidden layers evenly
        t0 = time.perf_counter()
        for _ in range(a.steps):
            x, out = one_pass()
        fence()
        dt = time.perf_counter() - t0
        prof = zh.profile_stop()
        per_rank = [dt]
        if use_dist:
            # every rank's own time of the K passes (between the two fences): value uses the MAX, the JSON shows all of them
            mine = torch.tensor([dt], dtype=torch.float64, device=dev)
            allt = torch.empty((world,), dtype=torch.float64, device=dev)
            zp.all_gather_into_tensor(allt, mine)
            per_rank = [float(v) for v in allt.cpu().tolist()]
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            zp.all_reduce(tt, dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, prof, x, out, per_rank

    def strong_run(math):
        """The `strong` object of a multi-rank run: ONE fixed problem - BASELINE configs[2]'s 1015 poses x H hypotheses - (a) on rank 0
        alone, no collective (what the reference's single GPU does, run/opt_main.py:166), and (b) with its H*N rows split over all ranks
        + the MIN exchange; both timed here, back to back, so that speed-up and efficiency are measured inside one run, next to the
        projection from one-GPU shard measurements.  The two selections must agree bit for bit (exit code 4 otherwise)."""
        Ns, K = a.strong_poses, max(1, a.strong_steps)
        ds = syn.make_poses(Ns, seed=2024)
        cfg_s = ZeDOConfig.pw3d(OIL_iterations=S)
        pipe.weights.set_math(math)
        ps = Pipeline(pipe.weights, cfg_s, dev).load(clusters, ds["db_2d"], ds["camera_param"])
        gts = torch.tensor((ds["db_3d"] - ds["db_3d"][:, 0:1]).astype(np.float64), dtype=torch.float64, device=dev)
        lo_s, rows_s = shard_rows(H * Ns, rank, world)

        def sharded():
            xs, _ = ps.run(row_offset=lo_s, rows=rows_s)
            return {k: reduce_min_over_ranks(b, i) for k, (b, i) in ps.select(xs, gts, row_offset=lo_s).items()}

        def alone():
            xs, _ = ps.run()
            return ps.select(xs, gts)

        t_one, sha_one = 0.0, None
        fence()
        if rank == 0:                      # the other ranks wait in the fence below
            alone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                o1 = alone()
            torch.cuda.synchronize()
            t_one = (time.perf_counter() - t0) / K
            sha_one = selection_digest(o1)
        sharded()
        fence()
        t0 = time.perf_counter()
        for _ in range(K):
            oN = sharded()
        fence()
        mine = time.perf_counter() - t0
        allt = torch.empty((world,), dtype=torch.float64, device=dev)
        zp.all_gather_into_tensor(allt, torch.tensor([mine], dtype=torch.float64, device=dev))
        per = [float(v) / K for v in allt.cpu().tolist()]
        t_n = max(per)
        sha_n = selection_digest(oN)
        bad = torch.tensor([int(rank == 0 and sha_one != sha_n)], dtype=torch.int64, device=dev)
        zp.all_reduce(bad, dist.ReduceOp.MAX)
        pipe.weights.set_math(a.math)
        res = dict(math=math, scaling="strong", n_gpus=world, steps=K,
                   workload=f"BASELINE configs[2]'s shape, {Ns} poses x H={H} x {S} OIL steps in total: rows split over {world} rank(s) + P1/P2 MIN exchange, "
                            "against the same problem on rank 0 alone (no collective), timed back to back in this run",
                   rows_per_gpu=rows_s, value=round(Ns / t_n, 3), unit="poses/s", ms_per_step=round(t_n * 1e3, 2),
                   one_rank=dict(value=round(Ns / t_one, 3) if t_one else None, ms_per_step=round(t_one * 1e3, 2)),
                   speedup_vs_one_rank=round(t_one / t_n, 3) if t_one else None,
                   efficiency=round(t_one / t_n / world, 4) if t_one else None,
                   selection_sha16=sha_n, one_rank_selection_sha16=sha_one, matches_one_rank=int(bad.item()) == 0,
                   rank_pass_s=dict(min=round(min(per), 5), max=round(max(per), 5), all=[round(v, 5) for v in per]))
        proj = strong_projection(rows_s, math) if (Ns == N_POSES and S == S_OIL and H == N_HYPO) else None
        res["efficiency_vs_projection"] = (dict(projected_ms_per_step=proj[1], projected_rows_per_rank=proj[0], measured_ms_per_step=round(t_n * 1e3, 2),
                                                efficiency=round(proj[1] / (t_n * 1e3), 4),
                                                source=f"profiles/{os.path.basename(proj[2])} (one MI355X running the shard of one rank, no communication)")
                                           if proj else None)
        return res

    dt, prof, x, out, per_rank = timed_run()
    # the other arithmetic mode of the hidden layers on the same problem (reported as alt_mode, never as `value`)
    alt = None
    if not a.no_alt_mode and wl["select"] != "gather":
        alt_math = "f16x3" if a.math == "f32" else "f32"
        pipe.weights.set_math(alt_math)
        dt2, prof2, _, out2, _ = timed_run()
        pipe.weights.set_math(a.math)
        alt = dict(math=alt_math, dt=dt2, prof=prof2, out=out2)

    # multi-rank runs of workload 2: the fixed-problem (strong-scaling) object beside the weak-scaling headline
    strong = None
    if use_dist and a.workload == 2 and not a.no_strong:
        strong = strong_run(a.math)
        if not a.no_alt_mode:
            strong["alt_mode"] = strong_run("f16x3" if a.math == "f32" else "f32")
        if not (strong["matches_one_rank"] and (strong.get("alt_mode") or strong)["matches_one_rank"]):
            if rank == 0:
                print(f"bench.py: strong object - {world} ranks give {strong['selection_sha16']}, rank 0 alone gives "
                      f"{strong['one_rank_selection_sha16']} on the same problem", file=sys.stderr, flush=True)
            sys.exit(4)

    box_tf = box_ghz = box16 = None
    if rank == 0:
        box_tf, box_ghz = zh.probe_mfma_peak(200000)      # ~0.2 s, outside the timed region
        box16 = zh.probe_mfma_peak_f16(1000000)           # ~0.5 s of bare fp16 MFMAs: long enough for power management to settle
    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        poses_per_s = N_total * a.steps / dt
        row_steps_per_s = poses_per_s * H * S
        hid = prof["hidden_dense"]
        roof = None
        if hid["avg_ms"]:
            # zedo_oil_run walks the rows in chunks of at most ZEDO_CHUNK_ROWS (default 2^20) per launch
            cap = int(os.environ.get("ZEDO_CHUNK_ROWS", 1 << 20))
            rows_launch = rows / -(-rows // cap)
            flop_launch = 2.0 * rows_launch * 1024 * 1024
            ach = flop_launch / (hid["avg_ms"] * 1e-3) / 1e12
            traffic = None
            tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # PMC pass, collected separately
            mfma_busy = None
            tj = {}
            if os.path.exists(tp) and rows == N_POSES * N_HYPO:   # the counter pass was collected on this launch shape
                tj = json.load(open(tp))
                traffic = tj.get("hidden_dense_bytes_per_launch")
                mfma_busy = {k: round(v["mfma_util"], 4) for k, v in tj.get("kernels", {}).items()
                             if k.startswith("layer_pair_kernel") and "mfma_util" in v} or None
            if a.math == "f16x3":
                roof = roofline_f16x3(hid, rows_launch, box16)
            else:
              roof = dict(bound="mfma", kernel="zedo::layer_pair_kernel<{GN_SILU|GN_SILU_RES}> (128x128 tiles on 16-deep K tiles, 3 workgroups per CU, + 64x64 / 64x128 remainder tiles in the same launch): "
                                             "one 1024x1024 dense layer + GroupNorm + SiLU [+ residual] over all rows",
                        achieved=round(ach, 2), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / PEAK_FP32_MFMA_TFLOPS, 4), traffic=traffic,
                        # the same kernel against what THIS box's matrix pipe sustains right now (bare MFMA loop)
                        box_mfma_peak_measured=round(box_tf, 2), box_shader_clock_ghz=round(box_ghz, 3),
                        # the shader clock the sampled launches of THIS kernel really ran at (power management lowers it
                        # under the full MFMA + LDS + HBM load, by a box-dependent amount) and the fraction of the
                        # matrix peak AT that clock (157.3 TFLOP/s is quoted at 2.4 GHz)
                        kernel_shader_clock_ghz=(round(hid["shader_clock_ghz"], 3) if hid.get("shader_clock_ghz") else None),
                        frac_at_kernel_clock=(round(ach / (PEAK_FP32_MFMA_TFLOPS * hid["shader_clock_ghz"] / 2.4), 4)
                                              if hid.get("shader_clock_ghz") else None),
                        frac_of_box_peak=round(ach / box_tf, 4),
                        flop_per_launch=flop_launch, avg_launch_ms=round(hid["avg_ms"], 4),
                        avg_launch_ms_bracketed=round(hid["avg_ms_raw"], 4), event_bracket_ms=round(hid.get("bracket_ms") or 0.0, 5),
                        sampled_launches=hid["samples"], launches=hid["launches"],
                        mfma_busy_pmc=mfma_busy,   # SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles
                        # achieved / avg_launch_ms are measured in THIS run; traffic and mfma_busy_pmc are not:
                        # they are replayed from the committed counter pass of the same command
                        measured_live=["achieved", "frac", "avg_launch_ms", "sampled_launches", "launches"],
                        replayed={"fields": ["traffic", "mfma_busy_pmc"],
                                  "source": "profiles/hbm_traffic.json" if traffic is not None else None,
                                  "collected": tj.get("collected") if traffic is not None else None})
        if wl["select"] == "gather":
            res = out["results"]
            quality = {"results_shape": [int(v) for v in res.shape], "results_finite": bool(torch.isfinite(res).all().item()),
                       "results_sha16": __import__("hashlib").sha256(res[:: max(1, res.shape[0] // 4096)].cpu().numpy().tobytes()).hexdigest()[:16]}
        else:
            quality = {"mpjpe_best_of_H_m": round(float(out["p1"][0].mean().item()), 6),
                       "pa_mpjpe_best_of_H_m": round(float(out["p2"][0].mean().item()), 6),
                       "selection_sha16": selection_digest(out)}
            if wl["select"] == "h36m":
                quality["mpjpe_actionwise_m"] = round(out["p1_actionwise"], 6)
                quality["pa_mpjpe_actionwise_m"] = round(out["p2_actionwise"], 6)
        what = {"p1p2": "P1/P2 min-over-hypotheses selection", "h36m": "action-wise P1/P2 min-over-hypotheses selection",
                "gather": "all-gather of every hypothesis of every pose (results resident on every rank; the D2H copy + np.save "
                          "of run/inference.py:236 are file I/O outside the path)"}[wl["select"]]
        line = {
            "metric": "poses/sec (1000-step sampler, H=50)", "value": round(poses_per_s, 3), "unit": "poses/s",
            "n_gpus": world, "rccl_ranks": (dist.get_world_size() if use_dist else 1),
            "dist_backend": (dist.get_backend() if use_dist else None), "devices_shared": os.environ.get("ZEDO_SHARE_DEVICE") == "1",
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            # exact fp32 throughout by default; --math f16x3: fp32 state / epilogues / thin layers, hidden layers as split-fp16 products
            "dtype": "f32" if a.math == "f32" else "f32 (hidden layers: split fp16 x3, fp32 accumulate)", "math": a.math, "data": "synthetic",
            "config": {"workload": (wl["name"] if stated else wl["name"] + " - NON-STATED size") +
                                   f": N={per} poses{'/GPU' if scaling == 'weak' else ' in total'} x H={H} hypotheses, "
                                   f"IPO 500 it ({len(cfg.IPO_keylist)} joints) + {S} OIL steps + {what}; "
                                   "random-init ScoreModelFC_Adv weights, synthetic detections",
                       "baseline_config": a.workload, "settings": wl["settings"],
                       "rows_per_gpu": rows, "poses_total": N_total, "sharding": f"rows over {world} rank(s)"},
            "pose_hyp_steps_per_s": round(row_steps_per_s, 1),
            "end_to_end_tflops": round(row_steps_per_s * FLOP_PER_ROW_STEP / 1e12, 2),
            # all six layers' algorithmic FLOP over the wall time of the whole pass (IPO, reprojection, selection,
            # launch gaps included) against the same fp32-MFMA peak
            "end_to_end_frac": (round(row_steps_per_s * FLOP_PER_ROW_STEP / 1e12 / PEAK_FP32_MFMA_TFLOPS / world, 4)
                                if a.math == "f32" else None),     # algorithmic fp32 flop against the fp32 peak: exact-fp32 mode only
            **quality,
            "roofline": roof,
            "kernel_time_ms_sampled_avg": {k: (round(v["avg_ms"], 4) if v["avg_ms"] else None) for k, v in prof.items()},
            # self-check (tests/test_rccl_gpu.py asserts it): the sampled per-class kernel times, each less the cost of an
            # empty event bracket on the same stream (event_bracket_ms; kernel_time_ms_sampled_raw keeps the bracketed
            # values), scaled to ALL launches of the timed region, per OIL iteration - against the wall time per OIL
            # iteration, which also contains the IPO, the selection and the gaps between launches: sum <= wall
            "kernel_time_ms_sampled_raw": {k: (round(v["avg_ms_raw"], 4) if v["avg_ms_raw"] else None) for k, v in prof.items()},
            "event_bracket_ms": round(prof["hidden_dense"].get("bracket_ms") or 0.0, 5),
            "sum_kernel_ms_per_oil_step": round(sum(v["launches"] * v["avg_ms"] for v in prof.values() if v["avg_ms"]) / (a.steps * S), 5),
            "wall_ms_per_oil_step": round(dt * 1e3 / (a.steps * S), 5),
            # N ranks == 1 rank, bit for bit, on the live transport, checked before the timed region (None without a process group)
            "multi_rank_selfcheck": selfcheck,
            # multi-rank runs of workload 2: one fixed problem (configs[2]) on rank 0 alone vs split over all ranks, timed back to back
            "strong": strong,
            # each rank's own seconds per pass between the fences (value / ms_per_step use the maximum)
            "rank_pass_s": {"min": round(min(per_rank) / a.steps, 5), "max": round(max(per_rank) / a.steps, 5),
                            "argmin": int(np.argmin(per_rank)), "argmax": int(np.argmax(per_rank)),
                            "all": [round(v / a.steps, 5) for v in per_rank]},
        }
        if scaling == "strong" and a.workload == 2 and S == S_OIL and H == N_HYPO:
            # strong scaling of configs[2]: what one GPU measured for a shard of this size (no communication) against this run
            proj = strong_projection(rows, a.math)
            line["efficiency_vs_projection"] = (dict(projected_ms_per_step=proj[1], projected_rows_per_rank=proj[0],
                                                     measured_ms_per_step=round(ms_per_step, 2), efficiency=round(proj[1] / ms_per_step, 4),
                                                     source=f"{proj[2]} (one MI355X running the shard of one rank)")
                                                if proj else None)
        if alt is not None:
            h2 = alt["prof"]["hidden_dense"]
            pps2 = N_total * a.steps / alt["dt"]
            cap = int(os.environ.get("ZEDO_CHUNK_ROWS", 1 << 20))
            rows_launch = rows / -(-rows // cap)
            am = {"math": alt["math"], "value": round(pps2, 3), "unit": "poses/s", "ms_per_step": round(alt["dt"] / a.steps * 1e3, 2),
                  "speedup_vs_headline": round(pps2 / poses_per_s, 3),
                  "mpjpe_best_of_H_m": round(float(alt["out"]["p1"][0].mean().item()), 6),
                  "pa_mpjpe_best_of_H_m": round(float(alt["out"]["p2"][0].mean().item()), 6),
                  "d_mpjpe_vs_headline_mm": round((float(alt["out"]["p1"][0].mean().item()) - float(out["p1"][0].mean().item())) * 1e3, 4),
                  "d_pa_mpjpe_vs_headline_mm": round((float(alt["out"]["p2"][0].mean().item()) - float(out["p2"][0].mean().item())) * 1e3, 4),
                  "selection_sha16": selection_digest(alt["out"]),
                  "kernel_time_ms_sampled_avg": {k: (round(v["avg_ms"], 4) if v["avg_ms"] else None) for k, v in alt["prof"].items()}}
            if h2["avg_ms"] and alt["math"] == "f16x3":
                am["roofline"] = roofline_f16x3(h2, rows_launch, box16)
            elif h2["avg_ms"]:
                am["roofline"] = dict(bound="mfma", achieved=round(2.0 * rows_launch * 1024 * 1024 / (h2["avg_ms"] * 1e-3) / 1e12, 2),
                                      peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", avg_launch_ms=round(h2["avg_ms"], 4),
                                      frac=round(2.0 * rows_launch * 1024 * 1024 / (h2["avg_ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4))
            line["alt_mode"] = am
        else:
            line["alt_mode"] = None
        # GB/s of the HBM / latency-bound kernels, at configs[2]'s rows and at configs[3]'s per-GPU shard
        line["geometry_kernels"] = geometry_kernels(zh, dev) if (world == 1 and not a.no_geometry) else None
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights, None, 2024)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if use_dist:
        zp.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
