"""GPU tests of the corners of the drop-in surface that used to raise instead of working (round-2 review):
`valid_ind` and `sample_interval` of eval_multi on device-resident row shards, the step-wise loop of non-fused
sampler configurations split over ranks, `scale_by_sigma` through the per-step sampler, and the singular
least-squares system failing the way the reference's torch.inverse does."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfg_path(name):
    return os.path.join(ROOT, "zedo-release_amd", "configs", "optim", f"concat_pose_optimization_{name}.py")


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


@pytest.fixture(scope="module")
def zh():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import zedo_hip
    return zedo_hip


@pytest.fixture(scope="module")
def model(weights0, zh):
    from lib.algorithms.advanced.model import ScoreModelFC_Adv
    from lib.dataset import synthetic as syn
    from run._driver import load_config
    m = ScoreModelFC_Adv(load_config(cfg_path("h36m")), 17, 3, 1024, 512, 3)
    sd = {k: torch.tensor(v) for k, v in weights0.items()}
    sd["sigmas"] = torch.tensor(syn.sigmas_buffer())
    m.load_state_dict(sd)
    return m.eval()


def _eval_problem(N=37, H=6, seed=3):
    rng = np.random.default_rng(seed)
    gt = rng.standard_normal((N, 17, 3)) * 0.25
    preds = (gt[:, None] + 0.05 * rng.standard_normal((N, H, 17, 3))).astype(np.float32)
    return gt.astype(np.float32), preds


@pytest.mark.parametrize("protocol2", [False, True])
def test_valid_ind_on_row_shards_equals_the_dense_layout(zh, golden, protocol2):
    """eval_multi(valid_ind=...) (reference h36m.py:396-397) with the predictions as device rows, whole and cut into
    shards (per-shard minima combined like reduce_min_over_ranks does), against the oracle's per-(pose, hypothesis)
    errors restricted to the listed hypotheses."""
    import zedo_oracle as O
    from lib.dataset._eval import hypothesis_min
    gt, preds = _eval_problem()
    N, H = preds.shape[:2]
    gtc = (gt - gt[:, 0:1]).astype(np.float64)
    rng = np.random.default_rng(11)
    valid = [sorted(rng.choice(H, size=rng.integers(1, H + 1), replace=False).tolist()) for _ in range(N)]
    err = O.hypothesis_errors(preds, gtc, protocol2)                      # [N,H]
    masked = np.full_like(err, np.inf)
    for n in range(N):
        masked[n, valid[n]] = err[n, valid[n]]
    ref_best, ref_idx = masked.min(1), masked.argmin(1)
    rows = dev(np.swapaxes(preds, 0, 1).reshape(H * N, 17, 3))
    best, idx = hypothesis_min(("rows", rows), gtc, protocol2, valid)
    np.testing.assert_allclose(best, ref_best, atol=1e-7 if protocol2 else 1e-12, rtol=0)
    assert np.array_equal(idx, ref_idx)
    # the dense layout takes the same route
    b2, i2 = hypothesis_min(preds, gtc, protocol2, valid)
    assert np.array_equal(b2, best) and np.array_equal(i2, idx)
    # three uneven shards, minima combined on the host the way the MIN all-reduce does
    cuts = [0, 50, 51, H * N]
    parts = [hypothesis_min(("rows", rows[a:b].contiguous()), gtc, protocol2, valid, row_offset=a) for a, b in zip(cuts[:-1], cuts[1:])]
    pb = np.stack([p[0] for p in parts])
    pi = np.stack([np.where(p[1] >= 0, p[1], 1 << 30) for p in parts])
    comb = pb.min(0)
    comb_i = np.where(pb == comb[None], pi, 1 << 30).min(0)
    assert np.array_equal(comb, best) and np.array_equal(comb_i, idx)


def test_sample_interval_on_device_rows(zh):
    """eval_multi(sample_interval=k) with ("rows", ...) - whole and as a shard - equals the [N,H,J,3] layout
    (reference pw3d.py:297-298: every k-th prediction, scored against ground-truth item i of the kept list)."""
    from lib.dataset.pw3d import PW3D
    gt, preds = _eval_problem(N=41, H=5, seed=5)
    N, H = preds.shape[:2]
    ds = PW3D.from_arrays(np.zeros((N, 17, 3), np.float32), gt, np.tile(np.eye(3, dtype=np.float32), (N, 1, 1)))
    rows = dev(np.swapaxes(preds, 0, 1).reshape(H * N, 17, 3))
    for k in (2, 3, 7):
        for p2 in (False, True):
            want = ds.eval_multi(preds, protocol2=p2, sample_interval=k)
            got = ds.eval_multi(("rows", rows), protocol2=p2, sample_interval=k)
            assert got == want, (k, p2, got, want)
    # a shard: rows [60, 160) hold hypotheses 1..3 partially; per-pose minima over what the shard holds
    from lib.dataset._eval import hypothesis_min, subsample
    gtc = ds.gt_centred()
    (tag, sub), g2, off = subsample(("rows", rows[60:160].contiguous()), gtc, 3, 60)
    b_sh, i_sh = hypothesis_min((tag, sub), g2, False, row_offset=off)
    dense, g3, _ = subsample(preds, gtc, 3)
    full_rows = dev(np.swapaxes(dense, 0, 1).reshape(-1, 17, 3))
    Nk = len(g3)
    b_ref, i_ref = hypothesis_min(("rows", full_rows[off:off + sub.shape[0]].contiguous()), g3, False, row_offset=off)
    assert sub.shape[0] == sum(1 for g in range(60, 160) if (g % N) % 3 == 0) and Nk == 14
    assert np.array_equal(b_sh, b_ref) and np.array_equal(i_sh, i_ref)


def test_singular_least_squares_system_raises_like_torch_inverse(zh, weights0):
    """An EXACTLY singular least-squares system: every joint of pose 2 detected at one pixel whose ray has exactly
    representable coordinates (K with power-of-two focal length and centre, pixel (1024, 768): r = (0.5, 0.25)), no
    confidences.  AtA of simple_zeroshot_opt.py:89 then has an exact zero pivot and the reference's torch.inverse raises
    (captured: "linalg.inv: (Batch element 2): The diagonal element 3 is zero ... singular"); the kernels' closed form
    would divide 0 by 0.  The surface and the fused pipeline raise as well - but only when a least-squares T is actually
    requested.  (Rays that coincide only up to rounding give the reference a finite garbage T, and the kernels too.)"""
    from lib.algorithms.advanced.simple_zeroshot_opt import gradient_field_gen, invalidate_ray_cache
    from lib.dataset import synthetic as syn
    from zedo_hip.pipeline import Pipeline, ZeDOConfig
    d = syn.make_poses(6, seed=4)
    uv = d["db_2d"][:, :, :2].copy()
    Kn = d["camera_param"].copy()
    Kn[2] = np.array([[1024, 0, 512], [0, 1024, 512], [0, 0, 1]], np.float32)
    uv[2] = np.array([1024.0, 768.0], np.float32)
    x = dev(0.1 * np.random.default_rng(0).standard_normal((6, 17, 3)))
    K, T = dev(Kn), dev(d["db_3d"][:, 0:1, :])
    geom = zh.reproj_prepare(dev(uv), K)
    assert zh.reproj_degenerate(geom) == 1
    assert zh.reproj_degenerate(zh.reproj_prepare(dev(d["db_2d"][:, :, :2]), dev(d["camera_param"]))) == 0
    uv_d = dev(uv)
    g = gradient_field_gen(uv_d, x, K, t=T)               # T given: no inverse in the reference either
    assert bool(torch.isfinite(g).all())
    with pytest.raises(torch.linalg.LinAlgError, match="singular"):
        gradient_field_gen(uv_d, x, K, t=None, returnT=True)
    invalidate_ray_cache()
    db2 = d["db_2d"].copy()
    db2[:, :, :2] = uv
    db2[:, :, 2] = 1.0
    pipe = Pipeline(weights0, ZeDOConfig.h36m(OIL_iterations=10, IPO_iterations=5), "cuda").load(syn.make_clusters(2, seed=1), db2, Kn)
    assert pipe.singular_poses == 1
    with pytest.raises(zh.ZedoError, match="singular"):
        pipe.run()
    x_ok, _ = pipe.run(oil_steps=2)                       # stays in the first fifth of the loop: the given T, no solve
    assert bool(torch.isfinite(x_ok).all())


def test_ray_cache_hits_for_confidences_that_need_conversion(zh):
    """with a float64 / non-contiguous `conf` the converted copy was a new tensor on every call, so the ray cache
    never hit and every gradient_field_gen step paid a device allocation + a stream sync.  The cache now keys on the
    CALLER's tensors: the second call with the same objects is a hit, the clamp of conf (reference :64-66) is visible on
    the caller's tensor, and a write to it (version counter) is a miss again."""
    from lib.algorithms.advanced import simple_zeroshot_opt as szo
    from lib.dataset import synthetic as syn
    d = syn.make_poses(5, seed=8, conf_mode="wild")
    uv, K = dev(d["db_2d"][:, :, :2]), dev(d["camera_param"])
    x = dev(0.1 * np.random.default_rng(1).standard_normal((5, 17, 3)))
    conf64 = torch.tensor(d["db_2d"][:, :, 2], dtype=torch.float64, device="cuda")
    wide = torch.zeros((5, 34), dtype=torch.float32, device="cuda")
    wide[:, ::2] = conf64.float()
    strided = wide[:, ::2]                                   # fp32 but not contiguous
    ref = szo.gradient_field_gen(uv, x, K, conf=dev(d["db_2d"][:, :, 2]), returnT=True)
    for conf in (conf64, strided):
        szo.invalidate_ray_cache()
        calls = []
        real = zh.reproj_prepare
        zh.reproj_prepare = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            g1, T1 = szo.gradient_field_gen(uv, x, K, conf=conf, returnT=True)
            assert float(conf.max()) == 1.0 and abs(float(conf.min()) - 1e-4) < 1e-9           # clamped in place
            g2, T2 = szo.gradient_field_gen(uv, x, K, conf=conf, returnT=True)
            assert len(calls) == 1, "second call with the same tensors must hit the cache"
            conf[0, 0] = 0.5                                                                    # version counter moves
            szo.gradient_field_gen(uv, x, K, conf=conf, returnT=True)
            assert len(calls) == 2
        finally:
            zh.reproj_prepare = real
        assert torch.equal(g1, g2) and torch.equal(T1, T2) and torch.equal(g1, ref[0]) and torch.equal(T1, ref[1])
    szo.invalidate_ray_cache()


def test_scale_by_sigma_takes_the_generic_route_and_divides_by_sigma(zh, model, weights0):
    """A subVPSDE configuration with model.scale_by_sigma = True (reference model.py:294: eps / sigmas[t]) must not be
    served by the closed-form zedo_sde_step (which knows no sigma): pc_sampler has to go through model.forward, and the
    result must equal the reference's update written out with torch operators around the HIP score network."""
    import copy
    from lib.algorithms.advanced import sampling
    from run import _driver
    cfg = _driver.load_config(cfg_path("h36m"))
    cfg.sampling.probability_flow = True
    cfg.model.scale_by_sigma = True
    assert "scale_by_sigma" in _driver.not_fused_because(cfg)
    m2 = copy.copy(model)
    m2.config = cfg
    sde = _driver.make_sde(cfg)
    fn = sampling.get_sampling_fn(cfg, sde, (9, 17, 3), lambda v: v, 0.01, device=torch.device("cuda"))
    x = dev(0.3 * np.random.default_rng(2).standard_normal((9, 17, 3)))
    t = 0.0637
    _, xm = fn(m2, condition=None, denoise_x=x.clone(), t=torch.tensor(t), t_step=7)
    assert fn.loop_schedule.hits == 0                      # the whole-loop (fused) schedule was not used
    # the same step from its parts: eps from the HIP network (scale_by_sigma off), divided by sigmas[int(999 t)]
    labels = torch.ones(9, device="cuda") * t * 999
    eps = model(x, labels, None, None) / model.sigmas.to("cuda")[labels.long()].reshape(-1, 1, 1).float()
    vt = torch.ones(9, device="cuda") * t
    std = sde.marginal_prob(torch.zeros_like(x), vt)[1]
    score = -eps / std[:, None, None]
    drift, diffusion = sde.sde(x, vt)
    drift = drift - diffusion[:, None, None] ** 2 * score * 1.0          # sde_lib.py:96: the factor is 1.0 for the ODE too
    want = (x + drift * (-1.0 / sde.N)).cpu().numpy()
    np.testing.assert_allclose(xm, want, rtol=2e-6, atol=2e-7)
    # and the unscaled model on the same sampler does take the fused route, with a different result
    fn_f = sampling.get_sampling_fn(cfg, sde, (9, 17, 3), lambda v: v, 0.01, device=torch.device("cuda"))
    _, xf = fn_f(model, condition=None, denoise_x=x.clone(), t=torch.tensor(t), t_step=7)
    assert np.abs(xf - xm).max() > 1e-6


def test_stepwise_loop_split_over_ranks_by_hypothesis(zh, model):
    """Non-fused sampler configurations on more than one rank: every rank runs whole hypotheses of the reference's loop
    (run/opt_main.py:166-222); the concatenation of the ranks' rows is the single-rank result, bit for bit."""
    from lib.dataset import synthetic as syn
    from run import _driver
    from zedo_hip.pipeline import shard_hypotheses
    cfg = _driver.load_config(cfg_path("h36m"))
    cfg.sampling.probability_flow = True
    cfg.sampling.predictor = "reverse_diffusion"           # leaves the fused pipeline
    assert _driver.not_fused_because(cfg) is not None
    N, H, S = 20, 5, 6
    d = syn.make_poses(N, seed=8, conf_mode="uniform")
    cl = syn.make_clusters(H, seed=8)
    sde = _driver.make_sde(cfg)
    torch.manual_seed(0)
    full = _driver.stepwise_loop(cfg, model, sde, cl, d["db_2d"].copy(), d["camera_param"], S, torch.device("cuda"))
    parts = []
    torch.manual_seed(0)
    for r in range(3):
        parts.append(_driver.stepwise_loop(cfg, model, sde, cl, d["db_2d"].copy(), d["camera_param"], S, torch.device("cuda"),
                                           hypotheses=shard_hypotheses(H, r, 3)))
    assert [p.shape[0] for p in parts] == [2 * N, 2 * N, N]
    assert torch.equal(torch.cat(parts), full)
