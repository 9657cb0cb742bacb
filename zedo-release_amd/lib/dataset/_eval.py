"""Shared body of `eval_multi` (reference lib/dataset/h36m.py:365-442, pw3d.py:286-345, custom.py:62-108):
per (pose, hypothesis) MPJPE / Procrustes-MPJPE and the per-pose minimum, on the device
(zedo_metric.hip); the action-wise / plain means over poses stay on the host (a few hundred floats).

When torch.distributed is initialised and `rows` is a shard, the per-pose minimum is combined with one
RCCL MIN all-reduce (zedo_hip.pipeline.reduce_min_over_ranks).
"""
import numpy as np
import torch


def hypothesis_min(preds, gt_centred, protocol2, valid_ind=None, row_offset=0, device=None):
    """preds: np/torch [N,H,J,3] (reference layout) or a torch CUDA tensor of rows [B,J,3] with row = h*N + n
    (pass it as a tuple ("rows", tensor)); the rows may be a contiguous shard starting at global row `row_offset`.
    gt_centred: [N,J,3] float64 metres, root-centred.  valid_ind: per pose the hypotheses that count (reference
    h36m.py:396-397 skips the others).  Returns (best [N] float64 np, idx [N] int np)."""
    import zedo_hip
    from zedo_hip.pipeline import empty_selection, reduce_min_over_ranks
    N = gt_centred.shape[0]
    if isinstance(preds, tuple) and preds[0] == "rows":
        rows = preds[1]
        device = rows.device
        if rows.shape[0] == 0:                      # empty shard (more ranks than rows)
            best, idx = reduce_min_over_ranks(*empty_selection(N, device))
            return best.cpu().numpy(), idx.cpu().numpy()
    else:
        device = torch.device("cuda") if device is None else device
        p = preds if isinstance(preds, torch.Tensor) else torch.as_tensor(np.asarray(preds))
        assert p.shape[0] == N
        rows = p.to(device=device, dtype=torch.float32).permute(1, 0, 2, 3).reshape(-1, p.shape[2], 3).contiguous()
    gt = torch.as_tensor(np.asarray(gt_centred, dtype=np.float64), device=device)
    err, best, idx = zedo_hip.min_mpjpe(rows, gt, N, procrustes=protocol2, row_offset=row_offset)
    if valid_ind is not None:
        # hypotheses not listed for a pose do not take part: their error becomes +inf ON THE DEVICE, for whatever
        # shard of the rows this rank holds, and the per-pose minimum is taken again (zedo_pose_min)
        ok_rows = valid_rows_mask(valid_ind, N, int(row_offset), rows.shape[0], device)
        err = torch.where(ok_rows, err, torch.full_like(err, float("inf")))
        best, idx = zedo_hip.pose_min(err, N, row_offset)
        idx = torch.where(torch.isinf(best), torch.full_like(idx, -1), idx)     # nothing listed on this rank for the pose
    best, idx = reduce_min_over_ranks(best, idx)
    return best.cpu().numpy(), idx.cpu().numpy()


def valid_rows_mask(valid_ind, N, row_offset, B, device):
    """bool [B] on `device`: row g = row_offset + i (g = h*N + n) is True when hypothesis h is listed in valid_ind[n].
    One flat index list built on the host (one fetch of valid_ind[n] per pose - none at all for a rectangular integer array: the
    full H36M test set has 567 040 poses) and ONE scatter on the device.  The container contract is the reference's own - `valid_ind[idx]` supports `in` / iteration
    (h36m.py:400 `sec_idx not in valid_ind[idx]`) - so a list, a tuple, an array of lists or a mapping {pose index: [...]}
    all work: the entries are fetched BY INDEX 0 .. N-1, never by iterating the container itself."""
    import itertools
    if isinstance(valid_ind, np.ndarray) and valid_ind.dtype != object and valid_ind.ndim == 2 and valid_ind.shape[0] >= N:
        # fast path: a rectangular array [N][m] of hypothesis indices - no Python statement per pose at all
        h = valid_ind[:N].astype(np.int64).reshape(-1)
        lens = np.full((N,), valid_ind.shape[1], dtype=np.int64)
    else:
        def entry(n):                       # each valid_ind[n] is fetched exactly once
            v = valid_ind[n]
            return v if isinstance(v, (list, tuple, np.ndarray)) else list(v)
        per_pose = [entry(n) for n in range(N)]
        lens = np.fromiter((len(v) for v in per_pose), dtype=np.int64, count=N)
        h = np.fromiter(itertools.chain.from_iterable(per_pose), dtype=np.int64, count=int(lens.sum()))
    g = h * N + np.repeat(np.arange(N, dtype=np.int64), lens) - int(row_offset)
    g = g[(h >= 0) & (g >= 0) & (g < B)]
    ok = torch.zeros((B,), dtype=torch.bool, device=device)
    if g.size:
        ok[torch.as_tensor(g, device=device)] = True
    return ok


def subsample(preds, gt_centred, sample_interval, row_offset=0):
    """`sample_interval` of the reference's eval_multi (h36m.py:386-387, pw3d.py:297-298): every k-th prediction is
    kept and prediction i of the kept ones is scored against ground-truth item i - the reference indexes the
    ground truth with the position in the subsampled list, not with the original index; mirrored as is.
    -> (preds, gt, row_offset) of the subsampled problem.  Device rows ("rows", tensor [B,J,3], row = h*N + n, possibly
    a shard starting at row_offset) are indexed on the device: the kept rows of a contiguous shard are again a
    contiguous shard of the subsampled problem's rows h*N' + n/k, N' = ceil(N/k)."""
    if sample_interval is None:
        return preds, gt_centred, row_offset
    k = int(sample_interval)
    N = gt_centred.shape[0]
    Nk = -(-N // k)
    if isinstance(preds, tuple) and preds[0] == "rows":
        rows = preds[1]
        g = torch.arange(int(row_offset), int(row_offset) + rows.shape[0], device=rows.device)
        keep = (g % N) % k == 0
        lo = int(row_offset)
        new_off = (lo // N) * Nk + -(-(lo % N) // k)       # kept rows in front of this shard
        return ("rows", rows[keep].contiguous()), gt_centred[:Nk], new_off
    preds = preds[::k]
    return preds, gt_centred[:len(preds)], row_offset


def print_table(title, cols, values, fmt="%.5f"):
    """Plain-text stand-in for prettytable (not installed offline)."""
    cells = [title] + [fmt % v for v in values]      # cols[0] heads the label column
    heads = [str(c) for c in cols]
    w = [max(len(a), len(b)) for a, b in zip(cells, heads)]
    line = "+" + "+".join("-" * (x + 2) for x in w) + "+"
    print(line)
    print("|" + "|".join(" " + h.ljust(x) + " " for h, x in zip(heads, w)) + "|")
    print(line)
    print("|" + "|".join(" " + c.ljust(x) + " " for c, x in zip(cells, w)) + "|")
    print(line)
