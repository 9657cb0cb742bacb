"""CPU oracle for the ZeDO optimisation-in-the-loop sampling path (numpy only).

TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's ``cpu_baseline`` leg may import this module; the product package
(zedo-release_amd/) never does and fails loudly when its HIP library is missing.

This is a restatement, function by function, of the reference's algorithm for
the path named in BASELINE.json (citations are relative to /root/reference).
Parity is PINNED: tests/test_oracle_golden.py checks every function below
against golden vectors captured by importing the reference itself in the build
container (tools/gen_golden.py -> tests/golden/*.npz).  The reference has no
tests or fixtures of its own, so those captures are the pin.

Every function takes ``dtype`` (np.float32 mirrors the reference's arithmetic;
np.float64 is the arbiter used to size tolerances).
"""
import math

import numpy as np

# ----------------------------------------------------------------------------
# score network: lib/algorithms/advanced/model.py
# ----------------------------------------------------------------------------

def timestep_embedding(timesteps, embedding_dim, max_positions=10000, dtype=np.float32):
    """model.py:81-95 get_timestep_embedding (sin | cos halves, fp32 frequencies)."""
    timesteps = np.asarray(timesteps, dtype=dtype).reshape(-1)
    half = embedding_dim // 2
    emb = math.log(max_positions) / (half - 1)
    freq = np.exp(np.arange(half, dtype=np.float32).astype(dtype) * dtype(-emb))
    arg = timesteps[:, None] * freq[None, :]
    out = np.concatenate([np.sin(arg), np.cos(arg)], axis=1)
    if embedding_dim % 2 == 1:
        out = np.pad(out, ((0, 0), (0, 1)))
    return out.astype(dtype)


def silu(x):
    """torch.nn.SiLU: x * sigmoid(x) (model.py:111)."""
    return x / (1.0 + np.exp(-x))


def group_norm(x, gamma, beta, groups=32, eps=1e-5):
    """torch.nn.GroupNorm(32, C) on a [B, C] input (model.py:116,145,150):
    per row and per block of C/groups consecutive channels, biased variance."""
    B, C = x.shape
    xg = x.reshape(B, groups, C // groups)
    mu = xg.mean(axis=2, keepdims=True)
    var = ((xg - mu) ** 2).mean(axis=2, keepdims=True)
    y = (xg - mu) / np.sqrt(var + x.dtype.type(eps))
    return y.reshape(B, C) * gamma[None, :] + beta[None, :]


def _lin(w, name, x):
    return x @ w[name + ".weight"].T + w[name + ".bias"][None, :]


def cast_weights(weights, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}


def time_embed(w, labels, embed_dim=512, dtype=np.float32):
    """model.py:251-259: temb = SiLU(Linear(posemb(labels))).  labels: [S] -> [S, embed]."""
    pe = timestep_embedding(labels, embed_dim, dtype=dtype)
    return silu(_lin(w, "shared_time_embed.0", pe))


def score_model_forward(w, x, labels, n_blocks=2, dtype=np.float32):
    """model.py:215-298 ScoreModelFC_Adv.forward in eval mode (dropout = identity,
    condition/mask ignored, scale_by_sigma False).

    x: [B, J, 3]; labels: scalar or [B] (the reference always passes one value
    repeated, advanced/sampling.py:497) -> eps [B, J, 3].
    """
    x = np.asarray(x, dtype=dtype)
    B = x.shape[0]
    labels = np.asarray(labels, dtype=dtype).reshape(-1)
    temb = time_embed(w, labels, w["shared_time_embed.0.weight"].shape[0], dtype)  # [1 or B, E]
    h = _lin(w, "pre_dense", x.reshape(B, -1)) + _lin(w, "pre_dense_t", temb)
    h = silu(group_norm(h, w["pre_gnorm.weight"], w["pre_gnorm.bias"]))
    for b in range(1, n_blocks + 1):
        h1 = _lin(w, f"b{b}_dense1", h) + _lin(w, f"b{b}_dense1_t", temb)
        h1 = silu(group_norm(h1, w[f"b{b}_gnorm1.weight"], w[f"b{b}_gnorm1.bias"]))
        h2 = _lin(w, f"b{b}_dense2", h1) + _lin(w, f"b{b}_dense2_t", temb)
        h2 = silu(group_norm(h2, w[f"b{b}_gnorm2.weight"], w[f"b{b}_gnorm2.bias"]))
        h = h + h2
    return _lin(w, "post_dense", h).reshape(x.shape)


def time_bias_table(w, labels, n_blocks=2, dtype=np.float32):
    """Row-invariant part of every hidden layer (model.py:265,273,281):
    tbias[s, l, :] = W_l_t @ temb(labels[s]) + b_l_t + b_l for the 1+2*n_blocks
    hidden layers.  Pins the table the HIP library builds once per schedule."""
    temb = time_embed(w, labels, w["shared_time_embed.0.weight"].shape[0], dtype)
    names = ["pre_dense"] + [f"b{b}_dense{k}" for b in range(1, n_blocks + 1) for k in (1, 2)]
    return np.stack([_lin(w, n + "_t", temb) + w[n + ".bias"][None, :] for n in names], axis=1)


# ----------------------------------------------------------------------------
# sub-VP SDE + predictor: lib/algorithms/advanced/sde_lib.py, sampling.py, utils.py
# ----------------------------------------------------------------------------

def subvp_sde(x, t, beta_0=0.1, beta_1=20.0):
    """sde_lib.py:187-192 subVPSDE.sde -> (drift, diffusion) for scalar t."""
    dt = x.dtype.type
    t = dt(t)
    beta_t = dt(beta_0) + t * dt(beta_1 - beta_0)
    drift = dt(-0.5) * beta_t * x
    discount = dt(1.0) - np.exp(dt(-2 * beta_0) * t - dt(beta_1 - beta_0) * t ** 2)
    return drift, np.sqrt(beta_t * discount)


def subvp_marginal_std(t, beta_0=0.1, beta_1=20.0, dtype=np.float32):
    """sde_lib.py:194-198 subVPSDE.marginal_prob std = 1 - exp(2*log_mean_coeff)."""
    t = dtype(t)
    lmc = dtype(-0.25) * t ** 2 * dtype(beta_1 - beta_0) - dtype(0.5) * t * dtype(beta_0)
    return dtype(1.0) - np.exp(dtype(2.0) * lmc)


def score_fn(w, x, t, beta_0=0.1, beta_1=20.0, dtype=np.float32):
    """utils.py:751-777 get_score_fn (subVP branch): labels = 999 t, score = -eps/std."""
    eps = score_model_forward(w, x, dtype(t) * dtype(999), dtype=dtype)
    return -eps / subvp_marginal_std(t, beta_0, beta_1, dtype)


def pc_step(w, x, t, n_sde=1000, beta_0=0.1, beta_1=20.0, dtype=np.float32):
    """One pc_sampler call (sampling.py:450-527) in the shipped configuration:
    NoneCorrector (:327-335), EulerMaruyamaPredictor (:180-191) on the
    probability-flow reverse SDE (sde_lib.py:93-100), noise_removal -> x_mean.
    dt = -1/n_sde regardless of the number of OIL iterations (sampling.py:186)."""
    x = np.asarray(x, dtype=dtype)
    drift, diffusion = subvp_sde(x, t, beta_0, beta_1)
    score = score_fn(w, x, t, beta_0, beta_1, dtype)
    drift = drift - diffusion ** 2 * score
    return x + drift * dtype(-1.0 / n_sde)


def step_coeffs(ts, n_sde=1000, beta_0=0.1, beta_1=20.0):
    """Closed form of pc_step: x' = a*x + c*eps(x, 999 t).
    float64 in, float64 out; the library rounds to fp32."""
    t = np.asarray(ts, dtype=np.float64)
    beta = beta_0 + t * (beta_1 - beta_0)
    disc = 1.0 - np.exp(-2 * beta_0 * t - (beta_1 - beta_0) * t ** 2)
    std = 1.0 - np.exp(2.0 * (-0.25 * t ** 2 * (beta_1 - beta_0) - 0.5 * t * beta_0))
    a = 1.0 + 0.5 * beta / n_sde
    c = -(beta * disc) / (n_sde * std)
    return a, c


def oil_timestamps(S, T=0.1, eps=0.01, dtype=np.float32):
    """torch.linspace(sde.T, sampling_eps, S) (opt_main.py:198), fp32 semantics:
    start + i*step for the first half, end - (S-1-i)*step for the second."""
    f32 = np.float32
    step = f32((f32(eps) - f32(T)) / f32(S - 1)) if S > 1 else f32(0)
    i = np.arange(S)
    # torch's CPU kernel fuses the multiply-add (one rounding): emulate with an exact fp64 product
    lo = (np.float64(f32(T)) + np.float64(step) * i).astype(f32)
    hi = (np.float64(f32(eps)) - np.float64(step) * (S - 1 - i)).astype(f32)
    return np.where(i < S // 2, lo, hi).astype(dtype)


# ----------------------------------------------------------------------------
# reprojection optimiser: lib/algorithms/advanced/simple_zeroshot_opt.py
# ----------------------------------------------------------------------------

def clamp_conf(conf):
    """simple_zeroshot_opt.py:64-66 (the reference does this in place on the caller's tensor)."""
    return np.clip(conf, conf.dtype.type(1e-4), conf.dtype.type(1.0))


def rays_from_keypoints(key2d, K, dtype=np.float32):
    """simple_zeroshot_opt.py:61-71: ray = Kinv [u v 1]^T, divided by its z."""
    key2d = np.asarray(key2d, dtype=dtype)
    Kinv = np.linalg.inv(np.asarray(K, dtype=dtype))
    hom = np.concatenate([key2d, np.ones(key2d.shape[:2] + (1,), dtype=dtype)], axis=-1)
    ray = np.einsum("bij,bkj->bki", Kinv, hom)
    return ray / ray[:, :, 2:]


def gradient_field_gen(key2d, key3d, K, t=None, conf=None, dtype=np.float32):
    """simple_zeroshot_opt.py:46-125 (noise_type None).  Returns (gradient, T).

    t given -> T = t (steps < S//5); t None -> weighted least-squares T through
    the 3x3 normal equations (:73-93) with the sign fix T_z < 0 -> -T (:93).
    gradient = ((x+T).r^) r^ - (x+T)  (:99,109 + perpendicular_distance :33-36).
    """
    key3d = np.asarray(key3d, dtype=dtype)
    B, J, _ = key3d.shape
    ray = rays_from_keypoints(key2d, K, dtype)
    if t is None:
        A = np.zeros((B, 2 * J, 3), dtype=dtype)
        b = np.zeros((B, 2 * J, 1), dtype=dtype)
        b[:, 0::2, 0] = key3d[:, :, 0] - key3d[:, :, 2] * ray[:, :, 0]
        b[:, 1::2, 0] = key3d[:, :, 1] - key3d[:, :, 2] * ray[:, :, 1]
        A[:, 0::2, 0] = -1
        A[:, 0::2, 2] = ray[:, :, 0]
        A[:, 1::2, 1] = -1
        A[:, 1::2, 2] = ray[:, :, 1]
        if conf is not None:
            c2 = clamp_conf(np.asarray(conf, dtype=dtype))
            c2 = (c2 * c2)[:, :, None]
            A[:, 0::2, :] *= c2
            A[:, 1::2, :] *= c2
            b[:, 0::2, :] *= c2
            b[:, 1::2, :] *= c2
        At = np.transpose(A, (0, 2, 1))
        T = np.transpose(np.linalg.inv(At @ A) @ (At @ b), (0, 2, 1))  # [B,1,3]
        neg = T[:, 0, 2] < 0
        T[neg] = -T[neg]
    else:
        T = np.asarray(t, dtype=dtype)
    rn = ray / np.linalg.norm(ray, axis=-1, keepdims=True)
    p = key3d + T
    grad = np.sum(p * rn, axis=-1, keepdims=True) * rn - p
    return grad.astype(dtype), T.astype(dtype)


# ----------------------------------------------------------------------------
# IPO: RotOpt + Adam (simple_zeroshot_opt.py:8-31, utils.py:59-88, opt_main.py:177-195)
# ----------------------------------------------------------------------------

def quaternion_to_matrix(q):
    """utils.py:59-88 (real part first, two_s = 2/|q|^2)."""
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    two_s = q.dtype.type(2.0) / (q * q).sum(-1)
    o = np.stack([
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def ipo_init_T(cond, K, ipo_T, dtype=np.float32):
    """opt_main.py:177-179: T0 = IPO_T * normalise(Kinv [u0 v0 1])."""
    cond = np.asarray(cond, dtype=dtype)
    pel = np.concatenate([cond[:, 0, :], np.ones((cond.shape[0], 1), dtype=dtype)], axis=-1)
    T = np.einsum("bij,bj->bi", np.linalg.inv(np.asarray(K, dtype=dtype)), pel)[:, None, :]
    return (T / np.linalg.norm(T, axis=-1, keepdims=True) * dtype(ipo_T)).astype(dtype)


_AXIS_SLOT = {"x": 1, "y": 2, "z": 3}


def _dq_tables():
    """dQ/dq_c for R = I + two_s * Q(q) (Q_ab bilinear in q); returns [4][3][3] coefficient
    lambdas as index/sign lists: entry (a,b) of dQ/dq_c = sum sign*q[idx]."""
    # encoded as dict c -> list of (a, b, sign, idx) with idx in 0..3 (r,i,j,k); factor 2 folded in sign
    R_, I_, J_, K_ = 0, 1, 2, 3
    return {
        R_: [(0, 1, -1, K_), (0, 2, 1, J_), (1, 0, 1, K_), (1, 2, -1, I_), (2, 0, -1, J_), (2, 1, 1, I_)],
        I_: [(0, 1, 1, J_), (0, 2, 1, K_), (1, 0, 1, J_), (1, 1, -2, I_), (1, 2, -1, R_), (2, 0, 1, K_),
             (2, 1, 1, R_), (2, 2, -2, I_)],
        J_: [(0, 0, -2, J_), (0, 1, 1, I_), (0, 2, 1, R_), (1, 0, 1, I_), (1, 2, 1, K_), (2, 0, -1, R_),
             (2, 1, 1, K_), (2, 2, -2, J_)],
        K_: [(0, 0, -2, K_), (0, 1, -1, R_), (0, 2, 1, I_), (1, 0, 1, R_), (1, 1, -2, K_), (1, 2, 1, J_),
             (2, 0, 1, I_), (2, 1, 1, J_)],
    }


def ipo_loss_and_grads(q, scale, x, T0, K, cond, axes, minT, maxT, normaliser):
    """Forward of RotOpt (simple_zeroshot_opt.py:20-25) + L1 mean loss (opt_main.py:185-191)
    and its hand-derived gradient w.r.t. (q[:, slots], scale).

    q [B,4] (absent axes stay 0), scale [B], x [B,k,3], T0 [B,3], K [B,3,3], cond [B,k,2].
    normaliser = N*k*2 of the WHOLE reference batch (torch.mean over all poses).
    """
    dt = q.dtype.type
    Rm = quaternion_to_matrix(q)                              # [B,3,3]
    sc = np.clip(scale, dt(minT), dt(maxT))
    p = np.einsum("bij,bkj->bki", Rm, x) + (T0 * sc[:, None])[:, None, :]
    w = np.einsum("bij,bkj->bki", K, p)
    uv = w[..., :2] / w[..., 2:]
    e = uv - cond
    loss = np.abs(e).sum() / dt(normaliser)
    g_uv = np.sign(e) / dt(normaliser)
    g_w = np.empty_like(w)
    g_w[..., :2] = g_uv / w[..., 2:]
    g_w[..., 2] = -(g_uv * w[..., :2]).sum(-1) / (w[..., 2] ** 2)
    g_p = np.einsum("bji,bkj->bki", K, g_w)                   # K^T g_w
    g_sc = (g_p.sum(1) * T0).sum(-1)
    g_scale = np.where((scale >= dt(minT)) & (scale <= dt(maxT)), g_sc, dt(0))
    G = np.einsum("nka,nkc->nac", g_p, x)                     # dL/dR
    n2 = (q * q).sum(-1)
    two_s = dt(2.0) / n2
    Q = (Rm - np.eye(3, dtype=q.dtype)[None]) / two_s[:, None, None]
    g_two_s = (G * Q).sum((1, 2))
    g_q = np.zeros_like(q)
    tabs = _dq_tables()
    for c in range(4):
        acc = np.zeros_like(n2)
        for (a, b, s, idx) in tabs[c]:
            acc = acc + dt(s) * G[:, a, b] * q[:, idx]
        g_q[:, c] = two_s * acc - g_two_s * two_s * two_s * q[:, c]
    slots = [0] + [_AXIS_SLOT[a] for a in axes]
    mask = np.zeros(4, dtype=bool)
    mask[slots] = True
    g_q[:, ~mask] = 0
    return loss, g_q, g_scale, uv


def ipo_fit(x0k, T0, K, condk, axes="z", minT=0.5, maxT=2.0, iters=500, normaliser=None,
            lr=0.1, b1=0.9, b2=0.999, adam_eps=1e-8, dtype=np.float32, trace=None):
    """opt_main.py:180-195: 500 Adam(lr=0.1) iterations on (rot_vect, rot_vect_<axes>, scale).

    x0k [B,k,3] = cluster pose restricted to IPO_keylist, condk [B,k,2], T0 [B,1,3].
    Returns (R [B,3,3], T [B,1,3] = T0*clamp(scale), q [B,4], scale [B], last loss).
    trace: optional list receiving (q, scale, loss, exp_avg_q, exp_avg_sq_q, exp_avg_scale, exp_avg_sq_scale,
    |residual| [B,k,2] of the forward that produced this iteration's gradient) copies after each iteration.
    """
    x0k = np.asarray(x0k, dtype=dtype)
    condk = np.asarray(condk, dtype=dtype)
    K = np.asarray(K, dtype=dtype)
    T0 = np.asarray(T0, dtype=dtype).reshape(-1, 3)
    B, k, _ = x0k.shape
    if normaliser is None:
        normaliser = B * k * 2
    q = np.zeros((B, 4), dtype=dtype)
    q[:, 0] = 1
    scale = np.ones(B, dtype=dtype)
    mq, vq = np.zeros_like(q), np.zeros_like(q)
    ms, vs = np.zeros_like(scale), np.zeros_like(scale)
    loss = dtype(0)
    for it in range(1, iters + 1):
        loss, gq, gs, uv_ = ipo_loss_and_grads(q, scale, x0k, T0, K, condk, axes, minT, maxT, normaliser)
        # torch.optim.Adam single-tensor update (no weight decay / amsgrad)
        step_size = dtype(lr / (1 - b1 ** it))
        bc2_sqrt = dtype(math.sqrt(1 - b2 ** it))
        for p, g, m, v in ((q, gq, mq, vq), (scale, gs, ms, vs)):
            m += (g - m) * dtype(1 - b1)
            v *= dtype(b2)
            v += dtype(1 - b2) * g * g
            denom = np.sqrt(v) / bc2_sqrt + dtype(adam_eps)
            p -= step_size * (m / denom)
        if trace is not None:
            trace.append((q.copy(), scale.copy(), dtype(loss), mq.copy(), vq.copy(), ms.copy(), vs.copy(),
                          np.abs(uv_ - condk)))
    R = quaternion_to_matrix(q)
    T = (T0 * np.clip(scale, dtype(minT), dtype(maxT))[:, None])[:, None, :]
    return R.astype(dtype), T.astype(dtype), q, scale, loss


# ----------------------------------------------------------------------------
# the OIL loop and the whole pipeline: run/opt_main.py:166-224
# ----------------------------------------------------------------------------

def oil_loop(w, x, cond, K, conf, T, S=1000, sde_T=0.1, sampling_eps=0.01, n_sde=1000,
             beta_0=0.1, beta_1=20.0, dtype=np.float32, snapshots=None):
    """opt_main.py:197-220: S iterations of {gradient_field_gen, x += g, pc_step}.
    T is used for i < S//5 and re-solved afterwards.  Returns (x, T).
    snapshots: optional dict step(1-based) -> x copy is filled for the requested keys."""
    x = np.array(x, dtype=dtype)
    T = np.array(T, dtype=dtype)
    ts = oil_timestamps(S, sde_T, sampling_eps, np.float32)
    for i in range(S):
        if i < S // 5:
            g, _ = gradient_field_gen(cond, x, K, t=T, conf=conf, dtype=dtype)
        else:
            g, T = gradient_field_gen(cond, x, K, t=None, conf=conf, dtype=dtype)
        x = x + g
        x = pc_step(w, x, dtype(ts[i]), n_sde, beta_0, beta_1, dtype)
        if snapshots is not None and (i + 1) in snapshots:
            snapshots[i + 1] = x.copy()
    return x, T


def zedo_pipeline(w, sample_poses, db_2d, camera_param, cfg, dtype=np.float32):
    """run/opt_main.py:166-224 for all hypotheses -> batch_results [N, H, 17, 3].

    cfg: dict with IPO_iterations, IPO_keylist, RotAxes, IPO_T, IPO_minScaleT,
    IPO_maxScaleT, OIL_iterations, sampling_eps, sde_T, num_scales, beta_min, beta_max."""
    sample_poses = np.asarray(sample_poses, dtype=dtype)
    centred = sample_poses - sample_poses[:, 0:1, :]
    cond = np.asarray(db_2d[:, :, :2], dtype=dtype)
    conf = clamp_conf(np.asarray(db_2d[:, :, 2], dtype=dtype))
    K = np.asarray(camera_param, dtype=dtype)
    N = cond.shape[0]
    kl = list(cfg["IPO_keylist"])
    out = []
    for h in range(sample_poses.shape[0]):
        x0 = np.broadcast_to(centred[h][None], (N, 17, 3)).astype(dtype)
        T0 = ipo_init_T(cond, K, cfg["IPO_T"], dtype)
        R, T, _, _, _ = ipo_fit(x0[:, kl, :], T0, K, cond[:, kl, :], cfg["RotAxes"], cfg["IPO_minScaleT"],
                                cfg["IPO_maxScaleT"], cfg["IPO_iterations"], dtype=dtype)
        x = np.einsum("bij,bkj->bki", R, x0)
        x, _ = oil_loop(w, x, cond, K, conf, T, cfg["OIL_iterations"], cfg["sde_T"], cfg["sampling_eps"],
                        cfg["num_scales"], cfg["beta_min"], cfg["beta_max"], dtype)
        out.append(x)
    return np.swapaxes(np.array(out), 0, 1)


# ----------------------------------------------------------------------------
# metric: lib/utils/transforms.py:42-127,143-148 and eval_multi
# ----------------------------------------------------------------------------

def procrustes_align(gt, pred):
    """transforms.py:42-127 procrustes(A=gt, B=pred, scaling=True, reflection='best').Z,
    batched over leading dims; float64."""
    A = np.asarray(gt, dtype=np.float64)
    Bm = np.asarray(pred, dtype=np.float64)
    A_bar = A.mean(-2, keepdims=True)
    B_bar = Bm.mean(-2, keepdims=True)
    A0 = A - A_bar
    B0 = Bm - B_bar
    A_norm = np.sqrt((A0 ** 2).sum((-1, -2), keepdims=True))
    B_norm = np.sqrt((B0 ** 2).sum((-1, -2), keepdims=True))
    A0 = A0 / A_norm
    B0 = B0 / B_norm
    M = np.swapaxes(A0, -1, -2) @ B0
    U, s, Vt = np.linalg.svd(M)
    R = np.swapaxes(Vt, -1, -2) @ np.swapaxes(U, -1, -2)
    return A_norm * s.sum(-1)[..., None, None] * (B0 @ R) + A_bar


def hypothesis_errors(preds, gt_centred, protocol2=False):
    """Inner double loop of eval_multi (h36m.py:394-412, pw3d.py:302-328):
    err[n,h] = mean_j ||pred[n,h,j] - gt_c[n,j]||, after Procrustes alignment for P2.
    gt_centred = gt - gt[:, 0:1] in metres (H36M: mm/1000, float64)."""
    gt = np.asarray(gt_centred, dtype=np.float64)[:, None]
    p = np.asarray(preds, dtype=np.float64)
    if protocol2:
        p = procrustes_align(np.broadcast_to(gt, p.shape), p)
    return np.sqrt(((p - gt) ** 2).sum(-1)).mean(-1)


def eval_multi(preds, gt_centred, protocol2=False, actions=None):
    """eval_multi: min over hypotheses per pose, then H36M action-wise mean of means over
    action ids 2..16 (h36m.py:419-433) when ``actions`` is given, else the plain mean
    (pw3d.py:336).  Returns (scalar, per-pose min [N], per-pose argmin [N])."""
    err = hypothesis_errors(preds, gt_centred, protocol2)
    best = err.min(1)
    idx = err.argmin(1)
    if actions is None:
        return float(best.mean()), best, idx
    actions = np.asarray(actions)
    per_action = [best[actions == a].mean() for a in range(2, 17)]
    return float(np.mean(per_action)), best, idx
