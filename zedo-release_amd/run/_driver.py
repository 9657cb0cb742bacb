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


def run(args, inference=False):
    from lib.algorithms.advanced import sde_lib
    from lib.algorithms.advanced.model import ScoreModelFC_Adv
    from lib.algorithms.ema import ExponentialMovingAverage
    from zedo_hip.pipeline import Pipeline, ZeDOConfig, shard_rows

    config = load_config(args.config)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("this driver needs an MI355X: the sampling path has no CPU fallback")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    device = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

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

    name = config.training.sde.lower()
    if name != "subvpsde":
        raise NotImplementedError(f"SDE '{name}': the fused pipeline implements the sub-VP SDE of the shipped configs")
    sde = sde_lib.subVPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max,
                           N=config.model.num_scales, T=config.model.t)
    config.sampling.probability_flow = True
    assert config.ZeDO.batch == len(gt_3d), f"batch: {config.ZeDO.batch}, dataset len: {len(gt_3d)}"

    z = config.ZeDO
    S = args.oil_iterations or z.OIL_iterations
    cfg = ZeDOConfig(z.IPO_iterations, z.IPO_keylist, z.RotAxes, z.IPO_T, z.IPO_minScaleT, z.IPO_maxScaleT, S,
                     z.sampling_eps, sde.T, sde.N, sde.beta_0, sde.beta_1)
    pipe = Pipeline(model.hip_weights(), cfg, device).load(sample_poses, gt_2d, K)
    H, N = pipe.H, pipe.N
    lo, rows = shard_rows(H * N, rank, world)
    x, T = pipe.run(row_offset=lo, rows=rows)

    batch_results = None
    if inference:          # results.npy holds every hypothesis: [N, H, 17, 3] (run/inference.py:233-236)
        full = torch.zeros((H * N, N_JOINTS, JOINT_DIM), dtype=torch.float32, device=device)
        full[lo:lo + rows] = x
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(full)          # disjoint shards: a sum is a gather
        batch_results = full.reshape(H, N, N_JOINTS, JOINT_DIM).permute(1, 0, 2, 3).cpu().numpy()
        if rank == 0:
            np.save(args.out, batch_results)
    errs = None
    if not inference or args.eval:
        print("eval...")
        p1 = test_dataset.eval_multi(("rows", x), protocol2=False, print_verbose=rank == 0, row_offset=lo)
        p2 = test_dataset.eval_multi(("rows", x), protocol2=True, print_verbose=rank == 0, row_offset=lo)
        errs = (p1, p2)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return batch_results, errs
