"""Geometry helpers of the metric (reference lib/utils/transforms.py).

`procrustes` / `align_to_gt` are small numpy functions kept for callers that align single poses; the
evaluation path (eval_multi over N x H poses) runs the same arithmetic in zedo_metric.hip.
"""
import numpy as np


def procrustes(A, B, scaling=True, reflection="best"):
    """Similarity alignment of B onto A (MATLAB `procrustes` semantics, reference :42-127).
    Returns (d, Z, tform) with Z the transformed B."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    assert A.shape[0] == B.shape[0]
    mu_a, mu_b = A.mean(0), B.mean(0)
    A0, B0 = A - mu_a, B - mu_b
    ss_a, ss_b = (A0 ** 2).sum(), (B0 ** 2).sum()
    na, nb = np.sqrt(ss_a), np.sqrt(ss_b)
    A0, B0 = A0 / na, B0 / nb
    if B0.shape[1] < A0.shape[1]:
        B0 = np.concatenate((B0, np.zeros((B0.shape[0], A0.shape[1] - B0.shape[1]))), 1)
    U, s, Vt = np.linalg.svd(A0.T @ B0)
    V = Vt.T
    R = V @ U.T
    if reflection != "best" and (np.linalg.det(R) < 0) != bool(reflection):
        V[:, -1] *= -1
        s[-1] *= -1
        R = V @ U.T
    tr = s.sum()
    if scaling:
        scale = tr * na / nb
        d = 1 - tr ** 2
        Z = na * tr * (B0 @ R) + mu_a
    else:
        scale = 1
        d = 1 + ss_b / ss_a - 2 * tr * nb / na
        Z = nb * (B0 @ R) + mu_a
    R = R[:B.shape[1], :]
    return d, Z, dict(rotation=R, scale=scale, translation=mu_a - scale * (mu_b @ R))


def align_to_gt(pose, pose_gt):
    """reference :143-148"""
    return procrustes(pose_gt, pose)[1]


def camera_to_world_frame(P, R, T):
    """reference :20-38"""
    return (R.T.dot(P.T) + T).T
