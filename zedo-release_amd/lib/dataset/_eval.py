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
    (pass it as a tuple ("rows", tensor)).  gt_centred: [N,J,3] float64 metres, root-centred.
    Returns (best [N] float64 np, idx [N] int np)."""
    import zedo_hip
    from zedo_hip.pipeline import dist_active, empty_selection, reduce_min_over_ranks
    N = gt_centred.shape[0]
    if isinstance(preds, tuple) and preds[0] == "rows":
        rows = preds[1]
        device = rows.device
        if rows.shape[0] == 0:                      # empty shard (more ranks than rows)
            assert valid_ind is None
            best, idx = reduce_min_over_ranks(*empty_selection(N, device))
            return best.cpu().numpy(), idx.cpu().numpy()
    else:
        device = torch.device("cuda") if device is None else device
        p = preds if isinstance(preds, torch.Tensor) else torch.as_tensor(np.asarray(preds))
        assert p.shape[0] == N
        rows = p.to(device=device, dtype=torch.float32).permute(1, 0, 2, 3).reshape(-1, p.shape[2], 3).contiguous()
    gt = torch.as_tensor(np.asarray(gt_centred, dtype=np.float64), device=device)
    err, best, idx = zedo_hip.min_mpjpe(rows, gt, N, procrustes=protocol2, row_offset=row_offset)
    if valid_ind is not None:                       # reference: skip hypotheses not listed for a pose
        if row_offset != 0 or rows.shape[0] % N or dist_active():
            raise NotImplementedError("valid_ind needs every hypothesis of every pose on one rank (unsharded rows)")
        H = rows.shape[0] // N
        e = err.reshape(H, N).T.cpu().numpy()
        mask = np.full_like(e, np.inf)
        for n in range(N):
            mask[n, list(valid_ind[n])] = 0
        e = e + mask
        return e.min(1), e.argmin(1)
    best, idx = reduce_min_over_ranks(best, idx)
    return best.cpu().numpy(), idx.cpu().numpy()


def subsample(preds, gt_centred, sample_interval):
    """`sample_interval` of the reference's eval_multi (h36m.py:386-387, pw3d.py:297-298): every k-th prediction is
    kept and prediction i of the kept ones is scored against ground-truth item i - the reference indexes the
    ground truth with the position in the subsampled list, not with the original index; mirrored as is."""
    if sample_interval is None:
        return preds, gt_centred
    if isinstance(preds, tuple):
        raise NotImplementedError("sample_interval needs predictions in the [N,H,J,3] layout")
    preds = preds[::sample_interval]
    return preds, gt_centred[:len(preds)]


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
