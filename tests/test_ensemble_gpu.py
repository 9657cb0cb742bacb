"""Calibrated end-to-end parity at BASELINE configs[2] (-m gpu): the reference's and the kernels' own REPRODUCIBILITY.

The IPO (500 Adam iterations, lr 0.1, L1 loss; reference run/opt_main.py:180-195) does not converge, its last iterate is
chaotic, and the 1000-step loop of the random-init fixture weights does not contract it away: two fp32 implementations
whose every single iteration agrees to 2.4e-7 (test_hip_parity.py) end poses apart and their dataset-mean MPJPE differs by
tenths of a millimetre - as does the reference against ITSELF.  Round 3 let that through on a 3-standard-error clause;
this file replaces it with measured null distributions:

  * detections moved by -1 / 0 / +1 ulp (lib/dataset/synthetic.py::perturb_ulp) re-draw the chaos without changing the
    algorithm.  (Thread count does NOT: the reference's IPO and loop are bit-identical on 1, 4 and 8 threads.)
  * reference side, captured in the build container (tools/gen_golden.py): the IPO end state of 16 / 8 / 8 such members of the
    three configs[2] draws as quantile functions (driver_pw3d_full*_ipoens.npz), and the FULL run (IPO + 1000 steps +
    selection) of four members of EACH draw (driver_pw3d_full{,_b,_c}_env{1..4}.npz, 2.4 CPU-hours each);
  * HIP side, here: M members through the fused pipeline (3 s each).

Asserted: every reference run's dataset-mean MPJPE lies inside the central 95 % of the HIP ensemble; PA-MPJPE within
0.05 mm outright; the means of the two ensembles within max(0.05 mm, what the two ensembles' own spread can resolve);
the IPO end-state distributions (rotation angle, depth scale, end-state loss) quantile by quantile within 8 single-member
standard deviations of the ensembles.  Numbers go to gpurun_out/parity_report.jsonl."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
HIP_SEED0 = 100            # reference members use perturbation streams 1, 2, ...; the HIP members 101, 102, ...


def tq(p, dof):
    """Student-t quantile (scipy ships with the image)."""
    from scipy import stats
    return float(stats.t.ppf(p, dof))


def _report(rec):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.jsonl", "a") as f:
        f.write(json.dumps(rec) + "\n")


class Draw:
    """One configs[2] capture: inputs regenerated from the fixture's seeds, checked against its hash."""

    def __init__(self, name):
        import hashlib
        from lib.dataset import synthetic as syn
        self.name = name
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.g = g
        self.N, self.H, self.S = int(g["N"]), int(g["H"]), int(g["S"])
        self.d = syn.make_poses(self.N, seed=int(g["seed_pose"]), conf_mode=str(g["conf_mode"]))
        self.cl = syn.make_clusters(self.H, seed=int(g["seed_cl"]))
        h = hashlib.sha256()
        for a in (self.d["db_2d"], self.d["camera_param"], self.cl):
            h.update(np.ascontiguousarray(a).tobytes())
        assert h.hexdigest() == str(g["inputs_sha"]), "inputs differ from the captured run"
        self.keylist = [int(k) for k in g["keylist"]]
        self.ipo_T, self.minT = float(g["ipo_T"]), float(g["minT"])
        self.gt = (self.d["db_3d"] - self.d["db_3d"][:, 0:1]).astype(np.float64)

    def detections(self, seed):
        from lib.dataset import synthetic as syn
        db2 = self.d["db_2d"].copy()
        db2[:, :, :2] = syn.perturb_ulp(db2[:, :, :2], seed)
        return db2

    def pipeline(self, W, seed):
        from zedo_hip.pipeline import Pipeline, ZeDOConfig
        cfg = ZeDOConfig(IPO_keylist=self.keylist, IPO_T=self.ipo_T, IPO_minScaleT=self.minT, OIL_iterations=self.S)
        return Pipeline(W, cfg, "cuda").load(self.cl, self.detections(seed), self.d["camera_param"]), cfg

    def ipo_summary(self, W, seed):
        import zedo_hip
        import _ipo_summary as ips
        pipe, cfg = self.pipeline(W, seed)
        R, T = zedo_hip.ipo_fit(pipe.x0, pipe.uv, pipe.K, cfg.IPO_keylist, cfg.RotAxes, cfg.IPO_T, cfg.IPO_minScaleT, cfg.IPO_maxScaleT,
                                cfg.IPO_iterations, self.N * len(cfg.IPO_keylist) * 2, self.H * self.N)
        cs = torch.stack([R[:, 0, 0], R[:, 1, 0]], -1).reshape(self.H, self.N, 2).cpu().numpy()
        return ips.summary(cs, T.reshape(self.H, self.N, 3).cpu().numpy(), (self.cl - self.cl[:, 0:1])[:, None],
                           self.detections(seed)[:, :, :2], self.d["camera_param"], cfg.IPO_keylist, cfg.IPO_T)

    def end_to_end(self, W, seed):
        import zedo_hip
        pipe, _ = self.pipeline(W, seed)
        x, _ = pipe.run()
        gt = torch.as_tensor(self.gt, device="cuda")
        _, b1, _ = zedo_hip.min_mpjpe(x, gt, self.N, procrustes=False)
        _, b2, _ = zedo_hip.min_mpjpe(x, gt, self.N, procrustes=True)
        return float(b1.mean().item()) * 1e3, float(b2.mean().item()) * 1e3            # mm


@pytest.fixture(scope="module")
def W(weights0, math_mode):
    import zedo_hip
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return zedo_hip.Weights(weights0)


DRAWS = ["driver_pw3d_full", "driver_pw3d_full_b", "driver_pw3d_full_c"]


@pytest.mark.parametrize("name", DRAWS)
def test_ipo_end_state_distribution_matches_the_references(W, math_mode, name):
    """SURVEY 7 stage (B) "distributional checks at 500 (final L1 loss, scale histogram)", asserted (round 3 only logged the
    comparison): 201-point quantile functions of the rotation angle about z, the depth scale T_z / T0_z and the end-state
    reprojection loss over the 50 750 (hypothesis, pose) fits - ensemble mean of 12 HIP members against the ensemble mean of
    the reference's members, tolerance per quantile = 8 x the pooled single-member standard deviation of the two ensembles
    (floors 2e-4 rad / 1e-5 / 2e-4 px where a quantile does not move at all), i.e. TAKEN FROM THE ENSEMBLES.  Measured:
    <= 5.5 (angle), 1.6 (scale), 6.2 (loss) such deviations; in absolute terms 5.1e-3 rad, 1.5e-3, 1.3e-2 px of 43."""
    if math_mode != "f32":
        pytest.skip("the IPO kernel does not depend on the arithmetic mode of the dense layers: covered by the f32 session")
    dr = Draw(name)
    ref = np.load(os.path.join(GOLDEN, name + "_ipoens.npz"))
    assert str(ref["inputs_sha"]) == str(dr.g["inputs_sha"])
    ms = [dr.ipo_summary(W, HIP_SEED0 + i) for i in range(1, 13)]
    hip = {k: np.stack([np.asarray(m[k]) for m in ms]) for k in ms[0]}
    rec = {"test": "ipo_end_state_distribution", "capture": name, "members_hip": len(ms), "members_ref": int(ref["q_loss"].shape[0])}
    sl = slice(2, -2)          # the outermost 1 % on either side are single extreme rows
    for k, floor, abs_tol in (("q_angle", 2e-4, 1.2e-2), ("q_scale", 1e-5, 3e-3), ("q_loss", 2e-4, 3e-2)):
        a, b = ref[k], hip[k]
        d = np.abs(a.mean(0) - b.mean(0))[sl]
        sd = np.maximum(np.sqrt((a.var(0, ddof=1) + b.var(0, ddof=1)) / 2)[sl], floor)
        rec[k] = dict(max_abs_diff=float(d.max()), max_in_member_sd=float((d / sd).max()), median_in_member_sd=float(np.median(d / sd)))
        assert (d <= 8.0 * sd).all(), (k, rec[k])
        assert d.max() <= abs_tol, (k, rec[k])
    dm = float(ref["mean_loss"].mean() - hip["mean_loss"].mean())
    rec["mean_loss_px"] = dict(ref=float(ref["mean_loss"].mean()), hip=float(hip["mean_loss"].mean()), diff=dm,
                               member_sd_ref=float(ref["mean_loss"].std(ddof=1)), member_sd_hip=float(hip["mean_loss"].std(ddof=1)))
    _report(rec)
    assert abs(dm) <= 3e-3, rec["mean_loss_px"]                  # of ~43 px: 7e-5 relative (measured <= 1.5e-3)


def _members(name, math_mode):
    if math_mode != "f32":
        return 6                # the split-fp16 mode shares the IPO kernel bit for bit; a short ensemble shows its loop agrees
    return 32 if name == "driver_pw3d_full" else 10


_ENSEMBLES = {}


@pytest.mark.parametrize("name", DRAWS)
def test_reference_runs_lie_inside_the_hip_ensemble(W, math_mode, name):
    """The north-star number at configs[2] with a calibrated yardstick.  HIP ensemble: M ulp-perturbed members of the capture
    through the fused pipeline (IPO + 1000 steps + selection).  Reference: the captured run and four more ulp-perturbed
    members of the REFERENCE's own run per draw (2.4 CPU-hours each: its fp32 reproducibility envelope).
      (a) every reference run's dataset-mean MPJPE inside the central 95 % of the HIP ensemble (t prediction interval of the
          M members; family-wise over the K reference runs);
      (b) PA-MPJPE: every reference run within 0.05 mm of the HIP ensemble mean - the bar, outright;
      (c) MPJPE: HIP ensemble mean within max(0.05 mm, E) of the reference mean, E = what the two ensembles can resolve:
          t(0.975) x pooled member sd x sqrt(1/M + 1/K) - with the member sd of the REFERENCE where it has members."""
    dr = Draw(name)
    M = _members(name, math_mode)
    e = np.array([dr.end_to_end(W, HIP_SEED0 + i) for i in range(1, M + 1)])
    _ENSEMBLES[(name, math_mode)] = e
    refs = [(float(dr.g["mpjpe"]) * 1e3, float(dr.g["pa_mpjpe"]) * 1e3)]
    k = 1
    while os.path.exists(os.path.join(GOLDEN, f"{name}_env{k}.npz")):
        z = np.load(os.path.join(GOLDEN, f"{name}_env{k}.npz"))
        assert int(z["perturb"]) == k
        refs.append((float(z["mpjpe"]) * 1e3, float(z["pa_mpjpe"]) * 1e3))
        k += 1
    refs = np.array(refs)
    m1, s1 = e[:, 0].mean(), e[:, 0].std(ddof=1)
    K = len(refs)
    # (a) K reference runs against one 95 % band: family-wise (Bonferroni), each run inside the central 1 - 0.05 / K
    half = tq(1 - 0.025 / K, M - 1) * s1 * np.sqrt(1 + 1 / M)
    s_ref = refs[:, 0].std(ddof=1) if K >= 3 else None
    sp = np.sqrt(((M - 1) * s1 ** 2 + (K - 1) * s_ref ** 2) / (M + K - 2)) if s_ref is not None else s1
    E = tq(0.975, M + K - 2) * sp * np.sqrt(1 / M + 1 / K)
    rec = {"test": "end_to_end_ensemble", "capture": name, "math": math_mode, "members_hip": M, "reference_runs": K,
           "mpjpe_mm": dict(hip_mean=float(m1), hip_member_sd=float(s1), hip_min=float(e[:, 0].min()), hip_max=float(e[:, 0].max()),
                            central95_half_width=float(half), reference=[float(v) for v in refs[:, 0]],
                            reference_member_sd=(float(s_ref) if s_ref is not None else None),
                            reference_max_pairwise=float(refs[:, 0].max() - refs[:, 0].min()),
                            mean_diff=float(m1 - refs[:, 0].mean()), resolvable=float(E)),
           "pa_mpjpe_mm": dict(hip_mean=float(e[:, 1].mean()), hip_member_sd=float(e[:, 1].std(ddof=1)), reference=[float(v) for v in refs[:, 1]],
                               max_diff=float(np.abs(refs[:, 1] - e[:, 1].mean()).max()))}
    _report(rec)
    print(json.dumps(rec))
    assert (np.abs(refs[:, 0] - m1) <= half).all(), rec["mpjpe_mm"]                       # (a)
    assert rec["pa_mpjpe_mm"]["max_diff"] <= 0.05, rec["pa_mpjpe_mm"]                     # (b)
    assert abs(m1 - refs[:, 0].mean()) <= max(0.05, E), rec["mpjpe_mm"]                   # (c)


def test_pooled_over_the_three_draws(math_mode):
    """The three draws together: mean over draws of (HIP ensemble mean - reference mean), against what three such
    differences can resolve: asserted against max(0.05 mm, 2 standard errors from the measured member spreads).  With the 15
    reference runs of round 4 the pooled difference reads -0.026 +- 0.059 mm (exact fp32): inside the bare 0.05 mm."""
    if any((n, math_mode) not in _ENSEMBLES for n in DRAWS):
        pytest.skip("needs the ensembles of test_reference_runs_lie_inside_the_hip_ensemble from this session")
    diffs, var = [], 0.0
    for n in DRAWS:
        e = _ENSEMBLES[(n, math_mode)][:, 0]
        g = np.load(os.path.join(GOLDEN, n + ".npz"))
        refs = [float(g["mpjpe"]) * 1e3]
        k = 1
        while os.path.exists(os.path.join(GOLDEN, f"{n}_env{k}.npz")):
            refs.append(float(np.load(os.path.join(GOLDEN, f"{n}_env{k}.npz"))["mpjpe"]) * 1e3)
            k += 1
        diffs.append(e.mean() - np.mean(refs))
        s = e.std(ddof=1)
        s_ref = float(np.std(refs, ddof=1)) if len(refs) >= 3 else s      # its own member spread where it has members, else the kernels' (measured equal)
        var += s ** 2 / len(e) + s_ref ** 2 / len(refs)
    pooled, se = float(np.mean(diffs)), float(np.sqrt(var) / len(DRAWS))
    _report({"test": "end_to_end_ensemble_pooled", "math": math_mode, "per_draw_mean_diff_mm": [float(d) for d in diffs], "pooled_mm": pooled,
             "standard_error_mm": se})
    assert abs(pooled) <= max(0.05, 2.0 * se), (pooled, se, diffs)


def test_contractive_prior_collapses_the_spread(math_mode):
    """DESIGN 4 predicted that the 0.2-0.3 mm between two fp32 runs of configs[2] is the IPO's chaotic last iterate carried
    through an EXPANSIVE loop (random-init weights), and that a denoiser which pulls towards an attractor - what a trained
    one does - would contract it away.  Measured: the reference's full configs[2] run with the
    contractive "tied" prior (lib/dataset/synthetic.py::make_weights(prior="tied"); tools/gen_golden.py::
    gen_driver_pw3d_full_tied, 2.6 CPU-hours) against the fused pipeline on the same inputs and weights - both dataset
    means within the north-star's 0.05 mm OUTRIGHT (a tenth of it, in fact), the median pose's best error within 0.001 mm and
    99 % of the poses within 0.05 mm (a single pose may still sit on another branch: 0.37 mm between two HIP members was
    seen for one of 1015), and four ulp-perturbed HIP members scatter by less than 0.005 mm where the random-init weights
    scatter by 0.13-0.19 mm (measured: 0.0002 mm)."""
    import zedo_hip
    from lib.dataset import synthetic as syn
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dr = Draw("driver_pw3d_full")
    g = np.load(os.path.join(GOLDEN, "driver_pw3d_full_tied.npz"))
    assert str(g["inputs_sha"]) == str(dr.g["inputs_sha"])
    w = syn.make_weights(seed=0, prior="tied")
    assert syn.weights_checksum(w) == str(g["weights_sha"])
    Wt = zedo_hip.Weights(w)
    pipe, _ = dr.pipeline(Wt, 0)
    x, _ = pipe.run()
    assert bool(torch.isfinite(x).all())
    gt = torch.as_tensor(dr.gt, device="cuda")
    rec = {"test": "contractive_prior", "math": math_mode}
    for key, proto, name in (("p1", False, "mpjpe"), ("p2", True, "pa_mpjpe")):
        _, best, _ = zedo_hip.min_mpjpe(x, gt, dr.N, procrustes=proto)
        b = best.cpu().numpy()
        rec[name] = dict(hip_mm=float(b.mean()) * 1e3, ref_mm=float(g[name]) * 1e3, d_mm=float(b.mean() - float(g[name])) * 1e3,
                         per_pose_abs_max_mm=float(np.abs(b - g[f"best_{key}"]).max()) * 1e3,
                         per_pose_abs_p99_mm=float(np.percentile(np.abs(b - g[f"best_{key}"]), 99)) * 1e3,
                         per_pose_abs_median_mm=float(np.median(np.abs(b - g[f"best_{key}"]))) * 1e3)
    ms = np.array([dr.end_to_end(Wt, HIP_SEED0 + i) for i in range(1, 5)])
    rec["hip_members_sd_mm"] = [float(ms[:, 0].std(ddof=1)), float(ms[:, 1].std(ddof=1))]
    _report(rec)
    print(json.dumps(rec))
    for name in ("mpjpe", "pa_mpjpe"):
        assert abs(rec[name]["d_mm"]) <= 0.005, rec[name]                       # a tenth of the bar
        assert rec[name]["per_pose_abs_median_mm"] <= 0.001 and rec[name]["per_pose_abs_p99_mm"] <= 0.05, rec[name]
    assert max(rec["hip_members_sd_mm"]) <= 0.005, rec["hip_members_sd_mm"]


def test_configs1_reference_runs_lie_inside_the_hip_ensemble(W, math_mode):
    """The same calibration for BASELINE configs[1] (H36M settings, 886 poses, ONE hypothesis, action-wise means): the reference's
    run on eight ulp-perturbed copies of the detections (tests/golden/driver_h36m_full_env{1..8}.npz, 3 CPU-minutes each) scatters by
    0.006 mm in MPJPE and by 0.022 mm (sd), 0.062 mm (extremes) in PA-MPJPE - there is no best-of-50 to average the chaotic IPO away,
    so with one hypothesis it is the ALIGNED error the last iterate moves.  A single HIP run sits 0.046 mm from the unperturbed
    reference run in PA-MPJPE: inside the reference's own range.  32 HIP members (0.1 s each): every reference run inside the
    central 95 % (family-wise) in both metrics, ensemble means within max(0.05 mm, resolvable)."""
    import hashlib
    from lib.dataset import synthetic as syn
    from lib.dataset.h36m import H36MDataset3D
    from zedo_hip.pipeline import Pipeline, ZeDOConfig
    g = np.load(os.path.join(GOLDEN, "driver_h36m_full.npz"))
    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    d = syn.make_poses(N, seed=int(g["seed_pose"]), conf_mode=str(g["conf_mode"]), dtype3d=np.float64)
    cl = syn.make_clusters(H, seed=int(g["seed_cl"]))
    h = hashlib.sha256()
    for a in (d["db_2d"], d["camera_param"], cl):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(g["inputs_sha"])
    cfg = ZeDOConfig(IPO_keylist=[int(k) for k in g["keylist"]], IPO_T=float(g["ipo_T"]), IPO_minScaleT=float(g["minT"]), OIL_iterations=S)
    ds = H36MDataset3D.from_arrays(d["db_2d"], d["db_3d"] * 1000.0, d["camera_param"], 2 + (np.arange(N) % 15))
    M = 32 if math_mode == "f32" else 8
    e = []
    for i in range(1, M + 1):
        db2 = d["db_2d"].copy()
        db2[:, :, :2] = syn.perturb_ulp(db2[:, :, :2], HIP_SEED0 + i)
        x, _ = Pipeline(W, cfg, "cuda").load(cl, db2, d["camera_param"]).run()
        e.append((ds.eval_multi(("rows", x), protocol2=False) * 1e3, ds.eval_multi(("rows", x), protocol2=True) * 1e3))
    e = np.array(e)
    refs = [(float(g["mpjpe"]) * 1e3, float(g["pa_mpjpe"]) * 1e3)]
    k = 1
    while os.path.exists(os.path.join(GOLDEN, f"driver_h36m_full_env{k}.npz")):
        z = np.load(os.path.join(GOLDEN, f"driver_h36m_full_env{k}.npz"))
        refs.append((float(z["mpjpe"]) * 1e3, float(z["pa_mpjpe"]) * 1e3))
        k += 1
    refs = np.array(refs)
    K = len(refs)
    assert K >= 9
    rec = {"test": "configs1_ensemble", "math": math_mode, "members_hip": M, "reference_runs": K}
    for c, name in ((0, "mpjpe_mm"), (1, "pa_mpjpe_mm")):
        m1, s1, s_ref = e[:, c].mean(), e[:, c].std(ddof=1), refs[:, c].std(ddof=1)
        half = tq(1 - 0.025 / K, M - 1) * s1 * np.sqrt(1 + 1 / M)
        sp = np.sqrt(((M - 1) * s1 ** 2 + (K - 1) * s_ref ** 2) / (M + K - 2))
        E = tq(0.975, M + K - 2) * sp * np.sqrt(1 / M + 1 / K)
        rec[name] = dict(hip_mean=float(m1), hip_member_sd=float(s1), reference=[float(v) for v in refs[:, c]], reference_member_sd=float(s_ref),
                         reference_max_pairwise=float(refs[:, c].max() - refs[:, c].min()), central95_half_width=float(half),
                         mean_diff=float(m1 - refs[:, c].mean()), resolvable=float(E))
    _report(rec)
    print(json.dumps(rec))
    for name in ("mpjpe_mm", "pa_mpjpe_mm"):
        r = rec[name]
        assert (np.abs(np.array(r["reference"]) - r["hip_mean"]) <= r["central95_half_width"]).all(), (name, r)
        assert abs(r["mean_diff"]) <= max(0.05, r["resolvable"]), (name, r)

