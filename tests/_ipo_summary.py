"""Summaries of an IPO end state (numpy, float64) shared by tools/gen_golden.py (reference side) and the GPU tests (HIP side).

Not part of the reference's algorithm: an EVALUATION of what the 500 Adam iterations of run/opt_main.py:180-195 left
behind - rotation about z as (cos, sin) and T = T0 * clamp(scale) per (hypothesis, pose) - so that two implementations
whose last iterates differ pose by pose (the loop is chaotic) can be compared as DISTRIBUTIONS."""
import numpy as np

QGRID = np.linspace(0.0, 1.0, 201)


def end_state_loss(cs, T, x0, uv, K, keylist):
    """Per-row L1 reprojection loss of the end state: mean over keylist x 2 of |proj(K (R x0 + T)) - uv|
    (the statement of simple_zeroshot_opt.py:26-31 + opt_main.py:190-191 evaluated at the final parameters).
    cs [..., N, 2], T [..., N, 3], x0 [..., 17, 3] (broadcast over N), uv [N, 17, 2], K [N, 3, 3]."""
    cs = np.asarray(cs, np.float64)
    T = np.asarray(T, np.float64)
    x = np.asarray(x0, np.float64)[..., keylist, :]
    c, s = cs[..., 0:1], cs[..., 1:2]
    xr = np.stack([c * x[..., 0] - s * x[..., 1], s * x[..., 0] + c * x[..., 1], np.broadcast_to(x[..., 2], (c * x[..., 0]).shape)], -1)
    p = xr + T[..., None, :]
    Kd = np.asarray(K, np.float64)
    q = np.einsum("nij,...nkj->...nki", Kd, p)
    proj = q[..., :2] / q[..., 2:3]
    return np.abs(proj - np.asarray(uv, np.float64)[:, keylist, :]).mean(axis=(-1, -2))


def quantiles(a):
    return np.quantile(np.asarray(a, np.float64).reshape(-1), QGRID)


def t0_z(uv, K, ipo_T):
    """z component of T0 = IPO_T * normalise(K^-1 [u_pelvis, v_pelvis, 1]) (opt_main.py:177-179), float64."""
    uv = np.asarray(uv, np.float64)
    pel = np.concatenate([uv[:, 0, :], np.ones((uv.shape[0], 1))], -1)
    r = np.linalg.solve(np.asarray(K, np.float64), pel[:, :, None])[:, :, 0]
    return ipo_T * r[:, 2] / np.linalg.norm(r, axis=-1)


def summary(cs, T, x0, uv, K, keylist, ipo_T):
    """201-point quantile functions of the rotation angle, the depth scale T_z / T0_z and the end-state loss.
    cs [H, N, 2], T [H, N, 3], x0 [H, 1, 17, 3] centred cluster poses."""
    ang = np.arctan2(np.asarray(cs, np.float64)[..., 1], np.asarray(cs, np.float64)[..., 0])
    scale = np.asarray(T, np.float64)[..., 2] / t0_z(uv, K, ipo_T)
    loss = end_state_loss(cs, T, x0, uv, K, keylist)
    return dict(q_angle=quantiles(ang), q_scale=quantiles(scale), q_loss=quantiles(loss),
                mean_loss=np.float64(loss.mean()), frac_scale_clamped=np.float64(np.mean((scale <= 0.2000001) | (scale >= 1.9999999))))
