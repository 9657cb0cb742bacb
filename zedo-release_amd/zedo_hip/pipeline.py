"""Device-resident ZeDO pipeline: the loop of reference run/opt_main.py:166-224 with all H hypotheses
batched as rows (h, n) and every stage running in libzedo_hip.so.

    ISO  x0[h]   = cluster_h - cluster_h[0]                      (opt_main.py:167-168,173)
    IPO  (R, T)  = zedo_ipo_fit                                   (opt_main.py:177-195)
         x       = R x0                                           (opt_main.py:201)
    OIL  S steps of {reprojection correction, score-network probability-flow step}   (opt_main.py:202-220)
    selection    = per-pose min over hypotheses of (PA-)MPJPE     (eval_multi)

Rows may be a contiguous shard of the H*N global rows (one shard per GPU); the only exchange is the final
MIN over ranks, done by the caller (run/opt_main.py, bench.py) with torch.distributed.
"""
import os

import numpy as np
import torch

from . import (SINGULAR_MSG, Schedule, Weights, ZedoError, ipo_fit, min_mpjpe, oil_run, reproj_degenerate, reproj_prepare,
               rotate_init)


def linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps) in fp32 without touching a device (opt_main.py:198)."""
    return torch.linspace(float(start), float(end), int(steps), dtype=torch.float32).numpy()


class ZeDOConfig:
    """The values of configs/optim/concat_pose_optimization_*.py that the path reads."""

    def __init__(self, IPO_iterations=500, IPO_keylist=(0, 1, 4), RotAxes="z", IPO_T=3.0, IPO_minScaleT=0.5,
                 IPO_maxScaleT=2.0, OIL_iterations=1000, sampling_eps=0.01, sde_T=0.1, num_scales=1000,
                 beta_min=0.1, beta_max=20.0):
        self.IPO_iterations, self.IPO_keylist, self.RotAxes = int(IPO_iterations), list(IPO_keylist), RotAxes
        self.IPO_T, self.IPO_minScaleT, self.IPO_maxScaleT = float(IPO_T), float(IPO_minScaleT), float(IPO_maxScaleT)
        self.OIL_iterations, self.sampling_eps, self.sde_T = int(OIL_iterations), float(sampling_eps), float(sde_T)
        self.num_scales, self.beta_min, self.beta_max = int(num_scales), float(beta_min), float(beta_max)

    @classmethod
    def h36m(cls, **kw):
        return cls(**kw)

    @classmethod
    def pw3d(cls, **kw):
        d = dict(IPO_keylist=tuple(range(17)), IPO_T=8.0, IPO_minScaleT=0.2)
        d.update(kw)
        return cls(**d)


class Pipeline:
    def __init__(self, state_dict, cfg, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            self.weights = state_dict if isinstance(state_dict, Weights) else Weights(state_dict)
            ts = linspace_f32(cfg.sde_T, cfg.sampling_eps, cfg.OIL_iterations)
            self.sched = Schedule(self.weights, ts, cfg.beta_min, cfg.beta_max, cfg.num_scales)

    def _dev(self, a, dtype=torch.float32):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=self.device)

    def load(self, sample_poses, db_2d, camera_param):
        """Upload one problem: clusters [H,17,3], detections [N,17,3]=(u,v,conf), intrinsics [N,3,3]."""
        sp = self._dev(sample_poses)
        self.x0 = (sp - sp[:, 0:1, :]).contiguous()             # ISO initial poses, centred on joint 0
        d2 = self._dev(db_2d)
        self.uv = d2[:, :, :2].contiguous()
        self.conf = d2[:, :, 2].contiguous()
        self.K = self._dev(camera_param)
        self.H, self.N = self.x0.shape[0], self.uv.shape[0]
        with torch.cuda.device(self.device):
            # the reference clamps conf in place inside gradient_field_gen; keep that observable
            self.geom = reproj_prepare(self.uv, self.K, self.conf, self.conf)
            self.singular_poses = reproj_degenerate(self.geom)     # once per problem (the rays do not change)
        return self

    def run(self, row_offset=0, rows=None, oil_steps=None):
        """IPO + OIL for global rows [row_offset, row_offset+rows) -> (x [rows,17,3], T [rows,3]) on device."""
        c = self.cfg
        row_offset = int(row_offset)
        B = self.H * self.N - row_offset if rows is None else int(rows)
        S = c.OIL_iterations if oil_steps is None else int(oil_steps)
        if row_offset < 0 or B < 0 or row_offset + B > self.H * self.N:
            raise ValueError(f"rows [{row_offset}, {row_offset + B}) are outside the {self.H} x {self.N} problem")
        if B == 0:          # more ranks than rows: this rank holds an empty shard
            return (torch.empty((0, self.x0.shape[1], 3), dtype=torch.float32, device=self.device),
                    torch.empty((0, 3), dtype=torch.float32, device=self.device))
        with torch.cuda.device(self.device):
            if S > c.OIL_iterations // 5 and self.singular_poses:     # the loop reaches the least-squares T
                raise ZedoError(SINGULAR_MSG.format(n=self.singular_poses))
            R, T = ipo_fit(self.x0, self.uv, self.K, c.IPO_keylist, c.RotAxes, c.IPO_T, c.IPO_minScaleT,
                           c.IPO_maxScaleT, c.IPO_iterations, self.N * len(c.IPO_keylist) * 2, B, row_offset)
            x = rotate_init(self.x0, R, self.N, row_offset)
            oil_run(self.weights, self.sched, x, self.geom, T, 0, S, c.OIL_iterations // 5, row_offset)
        return x, T

    def select(self, x, gt_centred, row_offset=0):
        """-> dict(p1=(best[N], idx[N]), p2=(best[N], idx[N])) for the local rows (fp64 / int32 tensors)."""
        gt = self._dev(gt_centred, torch.float64)
        if x.shape[0] == 0:
            return {k: empty_selection(self.N, self.device) for k in ("p1", "p2")}
        with torch.cuda.device(self.device):
            _, b1, i1 = min_mpjpe(x, gt, self.N, False, row_offset)
            _, b2, i2 = min_mpjpe(x, gt, self.N, True, row_offset)
        return dict(p1=(b1, i1), p2=(b2, i2))


def empty_selection(N, device):
    """What zedo_min_mpjpe reports for poses without a local row: (+inf, -1)."""
    return (torch.full((N,), float("inf"), dtype=torch.float64, device=device),
            torch.full((N,), -1, dtype=torch.int32, device=device))


def force_dist():
    """ZEDO_FORCE_DIST=1 (alias ZEDO_BENCH_FORCE_DIST): take the multi-rank code path (process group, RCCL
    collectives) with WORLD_SIZE = 1 too - how the exchange step is exercised on a one-GPU box."""
    return os.environ.get("ZEDO_FORCE_DIST") == "1" or os.environ.get("ZEDO_BENCH_FORCE_DIST") == "1"


def dist_backend():
    """Transport of the exchange step.  Default `nccl` (= RCCL on ROCm, one rank per GPU, over xGMI).
    ZEDO_DIST_BACKEND=gloo is a REHEARSAL transport: the same collectives on host copies of the (<= 8 KB per pose vector,
    or the gathered rows) tensors, so that N real ranks - N processes, N process-group members, every rank on its own
    row shard - can run where RCCL cannot: RCCL refuses two ranks on one device, and a one-GPU box is all a test has."""
    b = os.environ.get("ZEDO_DIST_BACKEND", "nccl").strip().lower() or "nccl"
    if b not in ("nccl", "gloo"):
        raise ValueError(f"ZEDO_DIST_BACKEND={b!r}: expected nccl or gloo")
    return b


def local_device_index():
    """The GPU of this rank: LOCAL_RANK - or device 0 for every rank with ZEDO_SHARE_DEVICE=1 (N ranks rehearsed on one
    MI355X; only with ZEDO_DIST_BACKEND=gloo)."""
    if os.environ.get("ZEDO_SHARE_DEVICE") == "1":
        if dist_backend() != "gloo":
            raise ValueError("ZEDO_SHARE_DEVICE=1 needs ZEDO_DIST_BACKEND=gloo: RCCL does not accept two ranks on one device")
        return 0
    return int(os.environ.get("LOCAL_RANK", "0"))


def init_dist(rank, world, device, default_port):
    """One process group per run: backend nccl (RCCL) bound to this rank's device, or the gloo rehearsal transport."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(default_port))
    if dist_backend() == "gloo":
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # nccl == RCCL on ROCm
    return dist


def _host_transport(t):
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce(t, op):
    """dist.all_reduce in place; over the gloo rehearsal transport a device tensor travels as a host copy."""
    import torch.distributed as dist
    if _host_transport(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def all_gather_into_tensor(out, t):
    import torch.distributed as dist
    if _host_transport(t):
        ho = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(ho, t.cpu())
        out.copy_(ho)
    else:
        dist.all_gather_into_tensor(out, t)
    return out


def barrier():
    import torch.distributed as dist
    dist.barrier()


def dist_active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_dist())


def reduce_min_over_ranks(best, idx):
    """The one exchange step of the sharded path: MIN over ranks of the per-pose error (RCCL all-reduce),
    then the lowest hypothesis index among the ranks that hold that minimum.  NaN follows np.amin / np.argmin
    (lib/dataset/h36m.py:411-412): a NaN on any rank wins and the lowest NaN hypothesis is reported; it travels
    as -inf (errors are >= 0) because a collective MIN does not define its NaN behaviour."""
    import torch.distributed as dist
    if not dist_active():
        return best, idx
    big = 2 ** 31 - 1
    key = torch.where(torch.isnan(best), torch.full_like(best, float("-inf")), best)
    g = key.clone()
    all_reduce(g, dist.ReduceOp.MIN)
    cand = torch.where((key == g) & (idx >= 0), idx.to(torch.int64), torch.full_like(idx, big, dtype=torch.int64))
    all_reduce(cand, dist.ReduceOp.MIN)
    cand = torch.where(cand == big, torch.full_like(cand, -1), cand)        # a pose no rank holds
    g = torch.where(g == float("-inf"), torch.full_like(g, float("nan")), g)
    return g, cand.to(torch.int32)


def gather_row_shards(x_local, total_rows, lo=None):
    """All ranks' contiguous row shards -> the full [total_rows, ...] tensor on every rank: one all-gather of equally
    sized (padded) shards.  run.inference writes every hypothesis of every pose (run/inference.py:233-236), so its
    result cannot stay sharded.  lo = None: the shards are the split of shard_rows(total_rows, rank, world); lo = this
    rank's first global row: any contiguous, ordered, gap-free split (e.g. whole hypotheses per rank,
    shard_hypotheses) - the shard sizes then travel in one small all-gather first."""
    import torch.distributed as dist
    if not dist_active():
        assert x_local.shape[0] == total_rows
        return x_local
    world = dist.get_world_size()
    tail = tuple(x_local.shape[1:])
    if lo is None:
        per = -(-total_rows // world)
        counts = None
    else:
        mine = torch.tensor([int(lo), x_local.shape[0]], dtype=torch.int64, device=x_local.device)
        allc = torch.empty((world * 2,), dtype=torch.int64, device=x_local.device)
        all_gather_into_tensor(allc, mine)
        counts = allc.reshape(world, 2).cpu().tolist()
        per = max(1, max(c for _, c in counts))
    pad = torch.zeros((per,) + tail, dtype=x_local.dtype, device=x_local.device)
    pad[:x_local.shape[0]] = x_local
    out = torch.empty((world * per,) + tail, dtype=x_local.dtype, device=x_local.device)
    all_gather_into_tensor(out, pad)
    if counts is None:
        return out[:total_rows]
    full = torch.empty((total_rows,) + tail, dtype=x_local.dtype, device=x_local.device)
    covered = 0
    for r, (rlo, cnt) in enumerate(counts):
        if cnt:
            assert rlo == covered, "row shards must be contiguous, ordered and gap-free"
            full[rlo:rlo + cnt] = out[r * per:r * per + cnt]
            covered += cnt
    assert covered == total_rows
    return full


def shard_hypotheses(H, rank, world):
    """Contiguous, unpadded split of the H hypotheses (EvaSampler.py:78-107 applied to hypotheses): for loops that
    cannot split inside a hypothesis (the step-wise sampler surface runs one hypothesis of all N poses at a time)."""
    per = -(-H // world)
    lo = min(rank * per, H)
    return lo, min(lo + per, H) - lo


def shard_rows(total_rows, rank, world):
    """Contiguous, unpadded split of the flattened (h, n) rows (the rule of reference
    lib/dataset/EvaSampler.py:78-107 applied to rows instead of samples)."""
    per = -(-total_rows // world)
    lo = min(rank * per, total_rows)
    hi = min(lo + per, total_rows)
    return lo, hi - lo
