"""N > 1 path on CPU: two processes over gloo exercise the exact exchange step of the sharded pipeline
(zedo_hip.pipeline.reduce_min_over_ranks) with shards produced by shard_rows.  The per-shard
(error, first-arg-min) inputs are computed with the oracle, emulating zedo_min_mpjpe's contract
(+inf / -1 for poses without a local row); the result must equal the unsharded eval_multi."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_min(err_rows, N, lo):
    """what zedo_min_mpjpe returns for local rows [lo, lo+len): per-pose min and lowest arg-min hypothesis"""
    best = np.full(N, np.inf)
    idx = np.full(N, -1, np.int32)
    for b, e in enumerate(err_rows):
        g = lo + b
        n, h = g % N, g // N
        if e < best[n]:
            best[n], idx[n] = e, h
    return best, idx


def _worker(rank, world, port, N, H, protocol2, out_dir):
    for p in (os.path.join(ROOT, "zedo-release_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import zedo_oracle as O
    from zedo_hip.pipeline import reduce_min_over_ranks, shard_rows
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(0)                       # same data on every rank
    gt = rng.standard_normal((N, 17, 3)) * 0.25
    gt -= gt[:, 0:1]
    preds = (gt[:, None] + 0.05 * rng.standard_normal((N, H, 17, 3))).astype(np.float32)
    preds[::3, 1] = preds[::3, 0]                        # exact ties across hypotheses (and across ranks)
    err = O.hypothesis_errors(preds, gt, protocol2)      # [N,H]
    rows = err.T.reshape(-1)                             # row = h*N + n
    lo, n = shard_rows(H * N, rank, world)
    best, idx = _local_min(rows[lo:lo + n], N, lo)
    gb, gi = reduce_min_over_ranks(torch.tensor(best), torch.tensor(idx))
    if rank == 0:
        np.savez(os.path.join(out_dir, f"res_{int(protocol2)}.npz"), best=gb.numpy(), idx=gi.numpy(),
                 ref_best=err.min(1), ref_idx=err.argmin(1))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("protocol2", [False, True])
def test_two_rank_min_reduction_equals_unsharded(tmp_path, protocol2):
    N, H, world = 23, 5, 2
    mp.spawn(_worker, args=(world, _free_port(), N, H, protocol2, str(tmp_path)), nprocs=world, join=True)
    r = np.load(tmp_path / f"res_{int(protocol2)}.npz")
    assert np.array_equal(r["best"], r["ref_best"])
    assert np.array_equal(r["idx"], r["ref_idx"])          # lowest hypothesis index wins ties, like np.argmin


def _worker_gather_nan(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "zedo-release_amd"))
    import torch.distributed as dist
    from zedo_hip.pipeline import gather_row_shards, reduce_min_over_ranks, shard_hypotheses, shard_rows
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    # all-gather of uneven contiguous row shards (run.inference): 3 ranks x 5 rows cover 13 rows as 5 + 5 + 3
    total = 13
    full = torch.arange(total * 17 * 3, dtype=torch.float32).reshape(total, 17, 3)
    lo, n = shard_rows(total, rank, world)
    got = gather_row_shards(full[lo:lo + n].clone(), total)
    ok_gather = bool(torch.equal(got, full))
    # whole hypotheses per rank (the step-wise loop of non-fused configurations): H = 4 over 3 ranks = 2 + 2 + 0
    # hypotheses of N = 3 poses, shard sizes that are NOT the shard_rows split (12 rows: 6 + 6 + 0 vs 4 + 4 + 4)
    H4, N3 = 4, 3
    h_lo, h_cnt = shard_hypotheses(H4, rank, world)
    assert (h_lo, h_cnt) == [(0, 2), (2, 2), (4, 0)][rank]
    got_h = gather_row_shards(full[h_lo * N3:(h_lo + h_cnt) * N3].clone(), H4 * N3, lo=h_lo * N3)
    ok_gather = ok_gather and bool(torch.equal(got_h, full[:H4 * N3]))
    # more ranks than rows: the last rank holds an empty shard
    lo2, n2 = shard_rows(2, rank, world)
    got2 = gather_row_shards(full[lo2:lo2 + n2].clone(), 2)
    ok_empty = bool(torch.equal(got2, full[:2])) and (n2 == 0) == (rank == 2)
    # NaN follows np.amin / np.argmin across ranks; a pose nobody holds stays (+inf, -1)
    inf, nan = float("inf"), float("nan")
    best = [torch.tensor([0.5, nan, 0.3, inf, nan], dtype=torch.float64),
            torch.tensor([0.4, 0.1, nan, inf, nan], dtype=torch.float64),
            torch.tensor([0.4, 0.2, 0.3, inf, 0.0], dtype=torch.float64)][rank]
    idx = [torch.tensor([0, 1, 0, -1, 1], dtype=torch.int32), torch.tensor([2, 3, 3, -1, 2], dtype=torch.int32),
           torch.tensor([4, 5, 4, -1, 5], dtype=torch.int32)][rank]
    gb, gi = reduce_min_over_ranks(best, idx)
    if rank == 0:
        np.savez(os.path.join(out_dir, "gn.npz"), ok_gather=ok_gather, ok_empty=ok_empty, best=gb.numpy(), idx=gi.numpy())
    ok = torch.tensor([float(ok_gather and ok_empty)])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    assert ok.item() == 1.0
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_gather_of_uneven_shards_and_nan_minimum(tmp_path):
    mp.spawn(_worker_gather_nan, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    r = np.load(tmp_path / "gn.npz")
    assert bool(r["ok_gather"]) and bool(r["ok_empty"])
    np.testing.assert_array_equal(r["best"], np.array([0.4, np.nan, np.nan, np.inf, np.nan]))
    assert list(r["idx"]) == [2, 1, 3, -1, 1]       # lowest index at the minimum; lowest NaN index; nobody: -1


def test_single_process_is_identity():
    from zedo_hip.pipeline import reduce_min_over_ranks
    b, i = torch.tensor([1.0, 2.0]), torch.tensor([3, 4], dtype=torch.int32)
    gb, gi = reduce_min_over_ranks(b, i)
    assert gb is b and gi is i
