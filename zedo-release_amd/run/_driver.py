"""Common body of run/opt_main.py and run/inference.py: the reference's evaluation driver
(run/opt_main.py:55-228, run/inference.py:55-241) on the fused HIP pipeline.

Differences to the reference, all inside the hot path: the H hypotheses are batched as rows (h, n) instead
of a sequential Python loop, IPO / OIL / selection run in libzedo_hip.so with the state resident on the GPU
(no per-step host round trip), and with WORLD_SIZE > 1 the rows are sharded contiguously over the ranks and
the per-pose minimum is combined with one RCCL MIN all-reduce.
"""
import argparse
import importlib.util
import os

import numpy as np
import torch

N_JOINTS, JOINT_DIM, HIDDEN_DIM, EMBED_DIM, CONDITION_DIM = 17, 3, 1024, 512, 3


def build_parser(description, inference=False):
    p = argparse.ArgumentParser(description=description)
    p.add_argument("--config", type=str, required=True, help="python file with get_config() (configs/optim/*.py)")
    p.add_argument("--ckpt_dir", type=str)
    p.add_argument("--ckpt_name", type=str)
    p.add_argument("--gt", action="store_true", default=False, help="use gt2d as condition")
    p.add_argument("--hypo", type=int, default=1, help="number of hypotheses")
    p.add_argument("--synthetic", type=int, default=0, metavar="N",
                   help="no dataset / cluster / checkpoint files: N seeded synthetic poses, random-init weights")
    p.add_argument("--oil_iterations", type=int, default=None, help="override config.ZeDO.OIL_iterations")
    p.add_argument("--math", choices=("f32", "f16x3"), default=None,
                   help="arithmetic of the dense layers: f32 = exact fp32 MFMA (default), f16x3 = split-fp16 operands on the fp16 "
                        "matrix pipe at fp32-level accuracy, ~2x faster (same as the ZEDO_MATH environment variable)")
    if inference:
        p.add_argument("--eval", action="store_true", default=None, help="evaluation mode")
        p.add_argument("--data", type=str, default=None, help="npz with db_2d, camera_param[, db_3d] ('wild' dataset)")
        p.add_argument("--out", type=str, default="results.npy")
    return p


def load_config(path):
    spec = importlib.util.spec_from_file_location("zedo_user_config", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.get_config()


def cluster_file(dataset, hypo):
    """reference run/opt_main.py:58-65, run/inference.py:68-69"""
    name = {"h36m": "h36m", "3dhp": "3dhp", "3dpw": "h36m", "ski": "h36m_sitting", "wild": "h36m"}[dataset]
    return f"clusters/{name}_cluster{hypo}.npy"


def make_dataset(config, args, inference):
    from pathlib import Path
    ds = config.data.dataset
    if args.synthetic:
        from lib.dataset import synthetic as syn
        from lib.dataset.h36m import H36MDataset3D
        from lib.dataset.pw3d import PW3D
        d = syn.make_poses(args.synthetic, seed=config.seed)
        if ds == "h36m":
            act = 2 + (np.arange(args.synthetic) % 15)
            return H36MDataset3D.from_arrays(d["db_2d"], d["db_3d"].astype(np.float64) * 1000.0, d["camera_param"], act)
        if ds == "3dhp":
            from lib.dataset.mpii3dHP import ACTIONS, MPII3DHP
            act = np.array(ACTIONS)[np.arange(args.synthetic) % len(ACTIONS)]
            return MPII3DHP.from_arrays(d["db_2d"], d["db_3d"].astype(np.float64) * 1000.0, d["camera_param"], act)
        if ds == "ski":
            from lib.dataset.skiPose import skiPose
            return skiPose.from_arrays(d["db_2d"], d["db_3d"], d["camera_param"])
        return PW3D.from_arrays(d["db_2d"], d["db_3d"], d["camera_param"])
    if ds == "h36m":
        from lib.dataset.h36m import H36MDataset3D
        return H36MDataset3D(Path("data", "h36m"), "test", gt2d=args.gt, abs_coord=True,
                             sample_interval=config.ZeDO.sample, flip=False)
    if ds == "3dpw":
        from lib.dataset.pw3d import PW3D
        return PW3D(Path("data", "3dpw"), "test", gt2d=args.gt, abs_coord=True, sample_interval=config.ZeDO.sample,
                    flip=False)
    if ds == "3dhp":
        from lib.dataset.mpii3dHP import MPII3DHP
        return MPII3DHP(Path("data", "3dhp"), "test", gt2d=args.gt, abs_coord=True, sample_interval=config.ZeDO.sample,
                        flip=False)
    if ds == "ski":
        from lib.dataset.skiPose import skiPose
        return skiPose(Path("data", "ski"), "test", gt2d=args.gt, abs_coord=True, sample_interval=config.ZeDO.sample,
                       flip=False)
    if ds == "wild" and inference:
        from lib.dataset.custom import CustomDataset
        if not args.data:
            raise SystemExit("--data <file.npz> is required for the 'wild' dataset")
        return CustomDataset.from_npz(args.data)
    raise NotImplementedError(f"dataset '{ds}' is outside the ported path (SURVEY.md 2, rows 14-15: infant pipeline)")


def make_sde(config):
    """run/opt_main.py:139-150"""
    from lib.algorithms.advanced import sde_lib
    name, m = config.training.sde.lower(), config.model
    if name == "vpsde":
        return sde_lib.VPSDE(beta_min=m.beta_min, beta_max=m.beta_max, N=m.num_scales, T=m.t)
    if name == "subvpsde":
        return sde_lib.subVPSDE(beta_min=m.beta_min, beta_max=m.beta_max, N=m.num_scales, T=m.t)
    if name == "vesde":
        return sde_lib.VESDE(sigma_min=m.sigma_min, sigma_max=m.sigma_max, N=m.num_scales, T=m.t)
    raise NotImplementedError(f"SDE {config.training.sde} unknown.")


def not_fused_because(config):
    """The fused pipeline (zedo_oil_run) is the closed form x' = a_i x + c_i eps(x, 999 t_i) of ONE sampler
    configuration - the one every shipped configs/optim/*.py selects (sampling.py:80-127 + run/opt_main.py:157).
    Returns None when `config` is that configuration, else the first field that differs: the driver then steps
    the reference's loop through get_sampling_fn, which implements the other SDEs / update rules, instead of
    silently running a different algorithm."""
    s, t, m = config.sampling, config.training, config.model
    checks = (("training.sde", t.sde.lower(), "subvpsde"), ("sampling.method", s.method.lower(), "pc"),
              ("sampling.predictor", s.predictor.lower(), "euler_maruyama"),
              ("sampling.corrector", s.corrector.lower(), "none"), ("training.continuous", bool(t.continuous), True),
              ("model.scale_by_sigma", bool(m.scale_by_sigma), False),
              ("sampling.noise_removal", bool(s.noise_removal), True))
    for name, got, want in checks:
        if got != want:
            return f"{name} = {got!r}, fused path: {want!r}"
    return None


def stepwise_loop(config, model, sde, sample_poses, gt_2d, K, S, device, hypotheses=None, host_round_trip=False):
    """run/opt_main.py:166-222 as written there - one hypothesis at a time, IPO through RotOpt (one zedo_ipo_fit
    launch), then S iterations of gradient_field_gen + one sampler step - for configurations outside the fused
    pipeline.  hypotheses = (first, count): only that contiguous range of the hypothesis loop (one rank's share).
    Returns rows [count*N,17,3] (h-major).
    The loop is device-resident (round 6): the sampler's `step_device` twin hands the updated rows back as a device
    tensor and the time stamps are host floats, so no iteration synchronises or copies - the reference's pc_sampler
    returns numpy (sampling.py:515-527) and its driver re-uploads it (run/opt_main.py:220): 2 x S blocking copies that
    move the values and change no bit.  host_round_trip=True steps the public numpy-returning `sampling_fn` exactly as
    the reference's driver does (the parity reference of the device-resident loop, tests/test_surface_gpu.py)."""
    from lib.algorithms.advanced import sampling
    from lib.algorithms.advanced.simple_zeroshot_opt import RotOpt, gradient_field_gen
    z = config.ZeDO
    N = len(gt_2d)
    sampling_fn = sampling.get_sampling_fn(config, sde, (N, N_JOINTS, JOINT_DIM), lambda v: v, z.sampling_eps,
                                           device=device)
    condition = torch.tensor(gt_2d[:, :, :2], device=device).float()
    conf = torch.tensor(gt_2d[:, :, 2], device=device).float()
    zero_condition = condition * 0
    Kd = torch.tensor(K, device=device).float()
    centred = torch.tensor(sample_poses - sample_poses[:, 0:1, :], device=device).float()
    timestamp = torch.linspace(sde.T, z.sampling_eps, S, device=device)
    t_host = timestamp.cpu().tolist()          # the same fp32 values the reference reads one by one with float(t)
    step_device = None if host_round_trip else getattr(sampling_fn, "step_device", None)
    out = []
    h_lo, h_cnt = (0, len(sample_poses)) if hypotheses is None else hypotheses
    for sid in range(h_lo, h_lo + h_cnt):
        x0 = centred[sid:sid + 1]
        rot_opt = RotOpt(N, axis=z.RotAxes, minT=z.IPO_minScaleT, maxT=z.IPO_maxScaleT).to(device)
        R, T = rot_opt.fit(x0, condition, Kd, z.IPO_keylist, z.IPO_T, z.IPO_iterations)
        with torch.no_grad():
            import zedo_hip
            denoise_x = zedo_hip.rotate_init(x0.contiguous(), R.contiguous(), N)      # rot_mat.bmm(x^T)^T, :201
            for i in range(S):
                if i < S // 5:
                    g = gradient_field_gen(condition, denoise_x, Kd, t=T, conf=conf, returnT=False)
                else:
                    g, T = gradient_field_gen(condition, denoise_x, Kd, conf=conf, returnT=True)
                denoise_x += g
                if step_device is not None:
                    denoise_x = step_device(model, condition=zero_condition, gradient=g, denoise_x=denoise_x, t=t_host[i], t_step=i, args=None)
                    continue
                _, results = sampling_fn(model, condition=condition * 0, gradient=g, denoise_x=denoise_x,
                                         t=timestamp[i], t_step=i, args=None)
                denoise_x = torch.as_tensor(results).to(device)
        out.append(denoise_x)
    if not out:
        return torch.empty((0, N_JOINTS, JOINT_DIM), dtype=torch.float32, device=device)
    return torch.cat(out, 0).contiguous()


def run(args, inference=False):
    from lib.algorithms.advanced import sde_lib
    from lib.algorithms.advanced.model import ScoreModelFC_Adv
    from lib.algorithms.ema import ExponentialMovingAverage
    from zedo_hip.pipeline import (Pipeline, ZeDOConfig, barrier, force_dist, gather_row_shards, init_dist,
                                   local_device_index, shard_hypotheses, shard_rows)

    config = load_config(args.config)
    if getattr(args, "math", None):
        os.environ["ZEDO_MATH"] = args.math          # read by zedo_hip.Weights when the model's device copy is built
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("this driver needs an MI355X: the sampling path has no CPU fallback")
    torch.cuda.set_device(local_device_index())      # LOCAL_RANK (device 0 for every rank with ZEDO_SHARE_DEVICE=1)
    device = torch.device("cuda", torch.cuda.current_device())
    use_dist = world > 1 or force_dist()      # ZEDO_FORCE_DIST=1: the RCCL path with one rank (tests)
    if use_dist:
        init_dist(rank, world, device, 29512)     # backend nccl == RCCL on ROCm (ZEDO_DIST_BACKEND=gloo: rehearsal transport)

    if args.synthetic:
        from lib.dataset import synthetic as syn
        sample_poses = syn.make_clusters(args.hypo, seed=config.seed)
    else:
        sample_poses = np.load(cluster_file(config.data.dataset, args.hypo)).astype(np.float32)

    model = ScoreModelFC_Adv(config, n_joints=N_JOINTS, joint_dim=JOINT_DIM, hidden_dim=HIDDEN_DIM,
                             embed_dim=EMBED_DIM, cond_dim=CONDITION_DIM)
    ema = ExponentialMovingAverage(model.parameters(), decay=config.model.ema_rate)
    test_dataset = make_dataset(config, args, inference)
    gt_3d, K, gt_2d = test_dataset.db_3d, test_dataset.camera_param, test_dataset.db_2d

    if args.synthetic:
        from lib.dataset import synthetic as syn
        sd = {k: torch.tensor(v) for k, v in syn.make_weights(seed=config.seed).items()}
        sd["sigmas"] = torch.tensor(syn.sigmas_buffer(config.model.sigma_max, config.model.sigma_min, config.model.num_scales))
        model.load_state_dict(sd)
        config.ZeDO.batch = len(gt_3d)
    else:
        ckpt_path = os.path.join(args.ckpt_dir, args.ckpt_name)
        print(f"loading model from {ckpt_path}")
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        model.load_state_dict({k[7:]: v for k, v in ckpt["model_state_dict"].items()})   # strip DataParallel's 'module.'
        ema.load_state_dict(ckpt["ema"])     # loaded, never applied - exactly like the reference
        print(f"=> loaded checkpoint '{ckpt_path}' (step {ckpt['step']})")
    model.eval()

    config.sampling.probability_flow = True                       # run/opt_main.py:157
    assert config.ZeDO.batch == len(gt_3d), f"batch: {config.ZeDO.batch}, dataset len: {len(gt_3d)}"
    sde = make_sde(config)
    z = config.ZeDO
    S = args.oil_iterations or z.OIL_iterations
    H, N = len(sample_poses), len(gt_3d)
    lo, rows = shard_rows(H * N, rank, world)
    why = not_fused_because(config)
    if why is None:
        cfg = ZeDOConfig(z.IPO_iterations, z.IPO_keylist, z.RotAxes, z.IPO_T, z.IPO_minScaleT, z.IPO_maxScaleT, S,
                         z.sampling_eps, sde.T, sde.N, sde.beta_0, sde.beta_1)
        pipe = Pipeline(model.hip_weights(), cfg, device).load(sample_poses, gt_2d, K)
        x, T = pipe.run(row_offset=lo, rows=rows)
    else:
        # the per-step surface runs one hypothesis of ALL poses at a time (a batch of N rows with the global loss
        # normaliser of IPO), so ranks share the hypothesis loop: whole hypotheses, contiguous, unpadded
        h_lo, h_cnt = shard_hypotheses(H, rank, world)
        lo, rows = h_lo * N, h_cnt * N
        if rank == 0:
            print(f"configuration outside the fused pipeline ({why}): stepping the loop of run/opt_main.py:166-222 "
                  f"through the per-step sampling_fn surface, hypotheses split over {world} rank(s)")
        x = stepwise_loop(config, model, sde, sample_poses, gt_2d, K, S, device, hypotheses=(h_lo, h_cnt))

    batch_results = None
    if inference:          # results.npy holds every hypothesis: [N, H, 17, 3] (run/inference.py:233-236)
        full = gather_row_shards(x, H * N, lo=None if why is None else lo)     # one RCCL all-gather of the row shards
        batch_results = full.reshape(H, N, N_JOINTS, JOINT_DIM).permute(1, 0, 2, 3).cpu().numpy()
        if rank == 0:
            np.save(args.out, batch_results)
    errs = None
    if not inference or args.eval:
        print("eval...")
        p1 = test_dataset.eval_multi(("rows", x), protocol2=False, print_verbose=rank == 0, row_offset=lo)
        p2 = test_dataset.eval_multi(("rows", x), protocol2=True, print_verbose=rank == 0, row_offset=lo)
        errs = (p1, p2)
    if use_dist:
        import torch.distributed as dist
        barrier()
        dist.destroy_process_group()
    return batch_results, errs
