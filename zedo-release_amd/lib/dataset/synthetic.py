"""Seeded synthetic inputs for the ZeDO hot path (numpy only, torch-independent).

No dataset, cluster file or checkpoint of the reference exists offline, so every
parity fixture, test and benchmark of this repo is driven from the generators
below (recipe: SURVEY.md section 8d).  They stand in for

* ``H36MDataset3D`` / ``PW3D``           (reference lib/dataset/h36m.py:206-263,
                                          lib/dataset/pw3d.py:177-227)
* ``clusters/<ds>_cluster{H}.npy``       (reference run/opt_main.py:58-65)
* ``checkpoint_1500.pth``                (reference run/opt_main.py:120-137)

and keep the same array contract: ``db_2d [N,17,3]=(u,v,conf)``,
``db_3d [N,17,3]``, ``camera_param [N,3,3]``, ``sample_poses [H,17,3]`` and a
state dict with the 34 parameter tensors + ``sigmas`` of ``ScoreModelFC_Adv``.

Random streams come from ``numpy.random.Philox`` so that the values do not
depend on the torch version; ``weights_checksum`` pins them.
"""
import hashlib

import numpy as np

N_JOINTS = 17
JOINT_DIM = 3
HIDDEN_DIM = 1024
EMBED_DIM = 512

# (name, shape) in the order of ScoreModelFC_Adv.state_dict() (reference
# lib/algorithms/advanced/model.py:113-152), without the float64 `sigmas` buffer.
def state_dict_layout(n_joints=N_JOINTS, joint_dim=JOINT_DIM, hidden=HIDDEN_DIM,
                      embed=EMBED_DIM, n_blocks=2):
    d = n_joints * joint_dim
    lay = [
        ("pre_dense.weight", (hidden, d)), ("pre_dense.bias", (hidden,)),
        ("pre_dense_t.weight", (hidden, embed)), ("pre_dense_t.bias", (hidden,)),
        ("pre_gnorm.weight", (hidden,)), ("pre_gnorm.bias", (hidden,)),
        ("shared_time_embed.0.weight", (embed, embed)), ("shared_time_embed.0.bias", (embed,)),
    ]
    for b in range(1, n_blocks + 1):
        for k in (1, 2):
            lay += [
                (f"b{b}_dense{k}.weight", (hidden, hidden)), (f"b{b}_dense{k}.bias", (hidden,)),
                (f"b{b}_dense{k}_t.weight", (hidden, embed)), (f"b{b}_dense{k}_t.bias", (hidden,)),
                (f"b{b}_gnorm{k}.weight", (hidden,)), (f"b{b}_gnorm{k}.bias", (hidden,)),
            ]
    lay += [("post_dense.weight", (d, hidden)), ("post_dense.bias", (d,))]
    return lay


def _rng(seed, stream):
    return np.random.Generator(np.random.Philox(key=[int(seed), int(stream)]))


def make_weights(seed=0, post_gain=0.1, hidden=HIDDEN_DIM, embed=EMBED_DIM, n_blocks=2,
                 n_joints=N_JOINTS, joint_dim=JOINT_DIM, prior="random"):
    """Random-init weights of the ScoreModelFC_Adv architecture.

    Linear layers: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias (the
    torch.nn.Linear default); GroupNorm gamma U(0.5,1.5), beta U(-0.2,0.2) (not
    the identity default, so that a swapped gamma/beta is caught); post_dense is
    scaled by ``post_gain`` so that the 1000-step loop is not expansive
    ("Rounding drift in OIL").
    Returns an ordered dict name -> float32 array (state-dict order, no sigmas).

    prior="tied": a CONTRACTIVE stand-in for a trained denoiser, from the same random draw: post_dense.weight =
    pre_dense.weight^T (bias 0), so that the Jacobian of eps(x) = W_pre^T D(x) W_pre-like is positive semi-definite and
    the step x' = a x + c eps(x), c < 0, pulls every row towards one attractor, and the second GroupNorm of both
    residual blocks scaled by 0.1 (the random blocks perturb that pull instead of dominating it).  With the default
    draw the 1000-step loop is expansive (two fp32 runs end 2.5e-4 m apart, a start rotated by 0.01 rad ends 1.9e-4 m
    away); with this one it forgets its start (1e-16 m) and fp32 and fp64 runs end 9e-7 m apart.
    """
    if prior not in ("random", "tied"):
        raise ValueError(prior)
    out = {}
    for i, (name, shape) in enumerate(state_dict_layout(n_joints, joint_dim, hidden, embed, n_blocks)):
        g = _rng(seed, 1000 + i)
        if "gnorm" in name:
            if name.endswith("weight"):
                a = g.uniform(0.5, 1.5, size=shape)
            else:
                a = g.uniform(-0.2, 0.2, size=shape)
        else:
            base = name.rsplit(".", 1)[0]
            fan_in = dict(state_dict_layout(n_joints, joint_dim, hidden, embed, n_blocks))[base + ".weight"][1]
            bound = 1.0 / np.sqrt(fan_in)
            a = g.uniform(-bound, bound, size=shape)
            if name.startswith("post_dense"):
                a = a * post_gain
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    if prior == "tied":
        out["post_dense.weight"] = np.ascontiguousarray(out["pre_dense.weight"].T)
        out["post_dense.bias"] = np.zeros_like(out["post_dense.bias"])
        for b in range(1, n_blocks + 1):
            out[f"b{b}_gnorm2.weight"] = (out[f"b{b}_gnorm2.weight"] * np.float32(0.1)).astype(np.float32)
            out[f"b{b}_gnorm2.bias"] = (out[f"b{b}_gnorm2.bias"] * np.float32(0.1)).astype(np.float32)
    return out


def sigmas_buffer(sigma_max=50.0, sigma_min=0.01, num_scales=1000):
    """The float64 ``sigmas`` buffer of the state dict (reference model.py:68-78,132)."""
    return np.exp(np.linspace(np.log(sigma_max), np.log(sigma_min), num_scales))


def weights_checksum(weights):
    h = hashlib.sha256()
    for k, v in weights.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v, dtype="<f4").tobytes())
    return h.hexdigest()


def make_clusters(H, seed=0):
    """Stand-in for clusters/h36m_cluster{H}.npy: float32 [H,17,3], 0.25*N(0,1)."""
    return (0.25 * _rng(seed, 10).standard_normal((H, N_JOINTS, JOINT_DIM))).astype(np.float32)


def make_poses(N, seed=0, conf_mode="ones", rot_z=True, dtype3d=np.float32):
    """Synthetic camera-frame poses + detections.

    gt3d = 0.25*N(0,1) with joint 0 zeroed, rotated about z by a per-pose angle,
    placed at root (0.1,-0.2,5.0) m (+ small per-pose jitter); uv = K*(gt3d+root)
    perspective-projected; conf = 1, or U(0.2,1) ("uniform"), or with values
    outside [1e-4,1] ("wild") to exercise the clamp of gradient_field_gen
    (reference simple_zeroshot_opt.py:64-66).
    Returns dict(db_2d [N,17,3] f32, db_3d [N,17,3] (absolute camera coords, metres),
                 camera_param [N,3,3] f32).
    """
    g = _rng(seed, 20)
    p = 0.25 * g.standard_normal((N, N_JOINTS, 3))
    p[:, 0, :] = 0.0
    if rot_z:
        ang = g.uniform(-np.pi, np.pi, size=N)
        c, s = np.cos(ang), np.sin(ang)
        R = np.zeros((N, 3, 3))
        R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = c, -s, s, c, 1.0
        p = np.einsum("nij,nkj->nki", R, p)
    root = np.array([0.1, -0.2, 5.0]) + 0.3 * g.standard_normal((N, 1, 3)) * np.array([1.0, 1.0, 0.5])
    cam = p + root
    K = np.tile(np.array([[1145.0, 0, 512.0], [0, 1145.0, 512.0], [0, 0, 1.0]]), (N, 1, 1))
    K[:, 0, 0] += g.uniform(-5, 5, size=N)
    K[:, 1, 1] += g.uniform(-5, 5, size=N)
    uvw = np.einsum("nij,nkj->nki", K, cam)
    uv = uvw[..., :2] / uvw[..., 2:]
    if conf_mode == "ones":
        conf = np.ones((N, N_JOINTS))
    elif conf_mode == "uniform":
        conf = g.uniform(0.2, 1.0, size=(N, N_JOINTS))
    elif conf_mode == "wild":
        conf = g.uniform(0.2, 1.0, size=(N, N_JOINTS))
        conf[:, 3] = 1.7
        conf[:, 5] = 1e-6
    else:
        raise ValueError(conf_mode)
    db_2d = np.concatenate([uv, conf[..., None]], -1).astype(np.float32)
    return dict(db_2d=db_2d, db_3d=cam.astype(dtype3d), camera_param=K.astype(np.float32))


def perturb_ulp(a, seed):
    """A copy of the float32 array ``a`` with every element moved by -1, 0 or +1 unit in the last place
    (equiprobable, Philox stream of ``seed``).  The end-to-end loop is chaotic in the IPO's last iterate (Adam on an
    L1 loss, reference run/opt_main.py:180-195): inputs that differ in the last bit re-draw it, in the reference and
    here alike - the ensembles of tests/golden/driver_pw3d_full_env*.npz (reference) and of
    tests/test_ensemble_gpu.py (HIP) are built from detections perturbed this way; ``seed`` 0 returns ``a`` itself."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if int(seed) == 0:
        return a.copy()
    s = _rng(seed, 30).integers(-1, 2, size=a.shape)
    up = np.nextafter(a, np.float32(np.inf), dtype=np.float32)
    dn = np.nextafter(a, np.float32(-np.inf), dtype=np.float32)
    return np.where(s > 0, up, np.where(s < 0, dn, a)).astype(np.float32)
