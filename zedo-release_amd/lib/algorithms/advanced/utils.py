"""Model-function helpers of the drop-in surface (reference lib/algorithms/advanced/utils.py).

Only what the sampling path uses is here: `quaternion_to_matrix` (:59-88), `get_model_fn` (:703-732),
`get_score_fn` (:736-800) and the two flatten helpers (:803-810).  The vendored PyTorch3D rotation
library, the model registry and the PCK/AUC helpers of the reference are not on the path.
"""
import numpy as np
import torch

from . import sde_lib


def quaternion_to_matrix(quaternions):
    """Rotation matrices from real-first quaternions, two_s = 2 / |q|^2 (reference :59-88)."""
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    m = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return m.reshape(quaternions.shape[:-1] + (3, 3))


def get_model_fn(model, train=False):
    """model_fn(x, labels, condition, mask) -> model output, switching train/eval like the reference."""

    def model_fn(x, labels, condition, mask):
        model.train() if train else model.eval()
        return model(x, labels, condition, mask)

    return model_fn


def get_score_fn(sde, model, train=False, continuous=False):
    """score_fn(x[B,j,3], t[B], condition, mask) -> score[B,j,3] (reference :736-800)."""
    model_fn = get_model_fn(model, train=train)

    if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
        def score_fn(x, t, condition, mask):
            if continuous or isinstance(sde, sde_lib.subVPSDE):
                labels = t * 999                     # time embedding assumes labels in [0, 999]
                out = model_fn(x, labels, condition, mask)
                std = sde.marginal_prob(torch.zeros_like(x), t)[1]
            else:
                labels = t * (sde.N - 1)
                out = model_fn(x, labels, condition, mask)
                std = sde.sqrt_1m_alphas_cumprod.to(labels.device)[labels.long()]
            return -out / std[:, None, None]
    elif isinstance(sde, sde_lib.VESDE):
        def score_fn(x, t, condition, mask):
            if continuous:
                labels = sde.marginal_prob(torch.zeros_like(x), t)[1]
            else:
                labels = torch.round((sde.T - t) * (sde.N - 1)).long()
            return model_fn(x, labels, condition, mask)
    else:
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
    return score_fn


def to_flattened_numpy(x):
    return x.detach().cpu().numpy().reshape((-1,))


def from_flattened_numpy(x, shape):
    return torch.from_numpy(x.reshape(shape))


def compute_PCK(gts, preds, scales=1000, eval_joints=None, threshold=150):
    """Percentage of joints closer than `threshold` mm to the label (reference :814-838).  gts / preds [N,J,3] in
    metres; `scales` is ignored exactly as in the reference (a fixed factor of 1000 converts to mm)."""
    gts, preds = np.asarray(gts), np.asarray(preds)
    joints = list(range(gts.shape[1])) if eval_joints is None else eval_joints
    err = np.sqrt(np.sum(np.power(preds - gts, 2), axis=2)) * 1000          # [N, J], dtype of the inputs
    err = np.take(err, joints, axis=1)
    return float((err < threshold).sum() / err.size) * 100


def compute_AUC(gts, preds, scales=1000, eval_joints=None):
    """Mean PCK over the thresholds 0, 5, ..., 150 mm of `mpii_compute_3d_pck.m` (reference :841-849)."""
    return np.mean([compute_PCK(gts, preds, scales, eval_joints, th) for th in np.linspace(0, 150, 31)])
