"""GPU parity tests: the HIP path (through the C ABI / ctypes binding) against the oracle and the
golden vectors captured from the reference.  Run with `-m gpu` on an MI355X.

Tolerances (metres unless stated) are fp32 round-off for single calls; long loops are bounded by the
reference's own fp32-vs-fp64 drift recorded in the fixtures.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def zh():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import zedo_hip
    return zedo_hip


@pytest.fixture(scope="module")
def W(zh, weights0, math_mode):
    w = zh.Weights(weights0)              # in the arithmetic mode of this part of the run (conftest.py::math_mode)
    assert w.math == math_mode
    return w


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def test_abi_version(zh):
    assert zh.abi_version() == 5


def test_schedule_tables(zh, W, weights0, golden):
    import zedo_oracle as O
    for S in (1000, 100):
        ts = O.oil_timestamps(S)
        s = zh.Schedule(W, ts)
        tb, a, c = s.read()
        ref = O.time_bias_table(weights0, ts * np.float32(999))
        np.testing.assert_allclose(tb, ref, atol=5e-6, rtol=0)
        a64, c64 = O.step_coeffs(ts.astype(np.float64))
        np.testing.assert_allclose(a, a64.astype(np.float32), rtol=2e-7, atol=0)   # libm vs numpy exp: 1 ulp
        np.testing.assert_allclose(c, c64.astype(np.float32), rtol=2e-7, atol=0)
    g = golden("model_forward")
    s = zh.Schedule(W, g["ts"])
    np.testing.assert_allclose(s.read()[0], g["tbias"], atol=5e-6, rtol=0)


def test_score_network_golden(zh, W, golden):
    g = golden("model_forward")
    s = zh.Schedule(W, g["ts"])
    for i in range(len(g["ts"])):
        eps = zh.score_eps(W, s, i, dev(g["x"])).cpu().numpy()
        np.testing.assert_allclose(eps, g["eps"][i], atol=2e-6, rtol=0)


@pytest.mark.parametrize("B", [1, 300, 1031])
def test_score_network_vs_oracle_ragged_batches(zh, W, weights0, B):
    import zedo_oracle as O
    rng = np.random.default_rng(B)
    x = (0.3 * rng.standard_normal((B, 17, 3))).astype(np.float32)
    ts = np.array([0.1, 0.03], np.float32)
    s = zh.Schedule(W, ts)
    for i, t in enumerate(ts):
        eps = zh.score_eps(W, s, i, dev(x)).cpu().numpy()
        ref = O.score_model_forward(weights0, x, np.float32(t) * np.float32(999))
        np.testing.assert_allclose(eps, ref, atol=3e-6, rtol=0)


def test_pc_step_golden(zh, W, golden):
    p = golden("pc_step")
    for S in (1000, 100):
        s = zh.Schedule(W, p[f"ts_{S}"])
        for k, i in enumerate(p[f"idx_{S}"]):
            x = dev(p["x"])
            zh.sde_step(W, s, int(i), x)
            np.testing.assert_allclose(x.cpu().numpy(), p[f"xmean_{S}"][k], atol=5e-7, rtol=0)


def test_gradient_field_gen_golden(zh, golden):
    r = golden("reproj")
    ones = np.ones((16, 17), np.float32)
    for tag, conf in (("wild", r["conf_wild"]), ("ones", ones), ("none", None)):
        cc = torch.empty(16, 17, device="cuda") if conf is not None else None
        geom = zh.reproj_prepare(dev(r["uv"]), dev(r["K"]), None if conf is None else dev(conf), cc)
        if tag == "wild":
            assert np.array_equal(cc.cpu().numpy(), r["conf_after_wild"])
        T = dev(r["T_given"].reshape(16, 3))
        g = zh.reproj_grad(dev(r["x"]), geom, T, False).cpu().numpy()
        np.testing.assert_allclose(g, r[f"g_given_{tag}"], atol=3e-6, rtol=0)
        assert np.array_equal(T.cpu().numpy(), r["T_given"].reshape(16, 3))
        T = torch.zeros(16, 3, device="cuda")
        g = zh.reproj_grad(dev(r["x"]), geom, T, True).cpu().numpy()
        np.testing.assert_allclose(T.cpu().numpy(), r[f"T_solve_{tag}"].reshape(16, 3), atol=2e-5, rtol=0)
        np.testing.assert_allclose(g, r[f"g_solve_{tag}"], atol=5e-6, rtol=0)
    # sign fix (T_z < 0 -> -T)
    geom = zh.reproj_prepare(dev(r["uv_neg"]), dev(r["K"]), dev(ones))
    T = torch.zeros(16, 3, device="cuda")
    g = zh.reproj_grad(dev(r["x_rel"]), geom, T, True).cpu().numpy()
    np.testing.assert_allclose(T.cpu().numpy(), r["T_neg"].reshape(16, 3), atol=3e-5, rtol=0)
    np.testing.assert_allclose(g, r["g_neg"], atol=5e-6, rtol=0)
    geom = zh.reproj_prepare(dev(r["uv"]), dev(r["K"]), dev(ones))
    T = torch.zeros(16, 3, device="cuda")
    g = zh.reproj_grad(dev(r["x_far"]), geom, T, True).cpu().numpy()
    np.testing.assert_allclose(T.cpu().numpy(), r["T_far"].reshape(16, 3), atol=2e-4, rtol=0)
    np.testing.assert_allclose(g, r["g_far"], atol=3e-5, rtol=0)


def _report(name, rows):
    """Append parity numbers to gpurun_out/parity_report.jsonl (copied into profiles/ by hand)."""
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.jsonl", "a") as f:
        f.write(json.dumps({"test": name, "rows": rows}) + "\n")


@pytest.mark.parametrize("S", [100, 1000])
def test_oil_loop_golden_snapshots(zh, W, golden, S):
    """Fused loop from pinned (R,T,x0) against the reference's snapshots.

    Two valid fp32 implementations of this loop cannot agree to round-off: every step rounds
    x+T (about 5 m, ulp 4.8e-7) before projecting onto the ray, and the loop does not contract those
    errors.  The reference run in fp32 is itself `gap` = |ref_f32 - ref_f64| away from exact arithmetic
    (2e-5 m after 20 steps, 1.6e-3 m after 1000 with the random fixture weights).  Criterion: the HIP
    path is at least as close to the fp64 arbiter as the reference's own fp32 run (x1.5 margin), and
    within the sum of both gaps of the fp32 run.
    """
    import zedo_oracle as O
    g = golden("oil")
    steps = [int(s) for s in g[f"snap_steps_{S}"]]
    sched = zh.Schedule(W, O.oil_timestamps(S))   # bit-identical to torch.linspace (test_oracle_golden)
    N = g["x_init"].shape[0]
    geom = zh.reproj_prepare(dev(g["db2d"][:, :, :2]), dev(g["K"]), dev(g["db2d"][:, :, 2]))
    x = dev(g["x_init"])
    T = dev(g["T_init"].reshape(N, 3))
    gap = np.abs(g[f"snaps_{S}_f32"] - g[f"snaps_{S}_f64"]).reshape(len(steps), -1).max(1)
    prev, rows = 0, []
    for i, s in enumerate(steps):
        zh.oil_run(W, sched, x, geom, T, prev, s, S // 5)
        prev = s
        xn = x.cpu().numpy()
        d64 = float(np.abs(xn - g[f"snaps_{S}_f64"][i]).max())
        d32 = float(np.abs(xn - g[f"snaps_{S}_f32"][i]).max())
        rows.append(dict(step=s, hip_vs_ref64=d64, hip_vs_ref32=d32, ref32_vs_ref64=float(gap[i])))
    _report(f"oil_golden_S{S}", rows)
    for r in rows:
        assert r["hip_vs_ref64"] <= 1.5 * r["ref32_vs_ref64"] + 2e-6, r
        assert r["hip_vs_ref32"] <= 2.5 * r["ref32_vs_ref64"] + 2e-6, r
    assert rows[0]["hip_vs_ref32"] <= 2e-6          # one step: round-off


ALT_WEIGHTS = {"w1": dict(seed=1), "tied": dict(seed=0, prior="tied")}


@pytest.mark.parametrize("tag", sorted(ALT_WEIGHTS))
def test_single_call_goldens_on_other_weights(zh, golden, math_mode, tag):
    """every parity number was on ONE synthetic weight set.  The single-call goldens again on a second
    random draw (seed 1) and on the contractive "tied" prior (lib/dataset/synthetic.py::make_weights), captured from the
    reference by tools/gen_golden.py::gen_weights_alt - the same tolerances as the seed-0 tests above: time-bias rows
    5e-6, network output 2e-6, pc_sampler's x_mean 5e-7, the 100-step loop by the fp64-arbiter criterion."""
    import zedo_oracle as O
    from lib.dataset import synthetic as syn
    g = golden("weights_alt")
    w = syn.make_weights(**ALT_WEIGHTS[tag])
    assert syn.weights_checksum(w) == str(g[f"sha_{tag}"])
    Wt = zh.Weights(w)
    s = zh.Schedule(Wt, g["ts"])
    np.testing.assert_allclose(s.read()[0], g[f"tbias_{tag}"], atol=5e-6, rtol=0)
    for i in range(len(g["ts"])):
        eps = zh.score_eps(Wt, s, i, dev(g["x"])).cpu().numpy()
        # per unit of output: the tied prior's eps is 60x larger than the random draws' (max |eps| 13.7)
        np.testing.assert_allclose(eps, g[f"eps_{tag}"][i], atol=2e-6 * max(1.0, float(np.abs(g[f"eps_{tag}"][i]).max())), rtol=0)
    s1000 = zh.Schedule(Wt, O.oil_timestamps(1000))
    for k, i in enumerate(g["idx_1000"]):
        x = dev(g["x"])
        zh.sde_step(Wt, s1000, int(i), x)
        np.testing.assert_allclose(x.cpu().numpy(), g[f"xmean_{tag}"][k], atol=5e-7, rtol=0)
    S, steps = 100, [int(v) for v in g["snap_steps"]]
    sched = zh.Schedule(Wt, O.oil_timestamps(S))
    N = g["x_init"].shape[0]
    geom = zh.reproj_prepare(dev(g["db2d"][:, :, :2]), dev(g["K"]), dev(g["db2d"][:, :, 2]))
    x, T = dev(g["x_init"]), dev(g["T_init"].reshape(N, 3))
    gap = np.abs(g[f"snaps_{tag}_f32"] - g[f"snaps_{tag}_f64"]).reshape(len(steps), -1).max(1)
    prev, rows = 0, []
    for i, st in enumerate(steps):
        zh.oil_run(Wt, sched, x, geom, T, prev, st, S // 5)
        prev = st
        xn = x.cpu().numpy()
        rows.append(dict(step=st, hip_vs_ref64=float(np.abs(xn - g[f"snaps_{tag}_f64"][i]).max()),
                         hip_vs_ref32=float(np.abs(xn - g[f"snaps_{tag}_f32"][i]).max()), ref32_vs_ref64=float(gap[i])))
    _report(f"oil_alt_weights_{tag}", rows)
    for r in rows:
        assert r["hip_vs_ref64"] <= 1.5 * r["ref32_vs_ref64"] + 2e-6, r
        assert r["hip_vs_ref32"] <= 2.5 * r["ref32_vs_ref64"] + 2e-6, r
    np.testing.assert_allclose(T.cpu().numpy(), g[f"T_final_{tag}_f64"].reshape(N, 3), atol=1.5 * float(np.abs(g[f"T_final_{tag}_f32"] - g[f"T_final_{tag}_f64"]).max()) + 2e-5, rtol=0)


def test_oil_loop_vs_oracle_ragged(zh, W, weights0):
    """B = H*N rows with H=3, N=37 (B not a multiple of any tile), 30 steps, switch at 6; same criterion
    with the oracle run in fp64 as arbiter and in fp32 as the reference-precision run."""
    import zedo_oracle as O
    from lib.dataset import synthetic as syn
    H, N, S = 3, 37, 30
    d = syn.make_poses(N, seed=77, conf_mode="uniform")
    rng = np.random.default_rng(5)
    x0 = (d["db_3d"] - d["db_3d"][:, 0:1])[None] + 0.05 * rng.standard_normal((H, N, 17, 3))
    x0 = x0.reshape(H * N, 17, 3).astype(np.float32)
    T0 = np.tile(d["db_3d"][:, 0, :], (H, 1)).astype(np.float32)
    ts = O.oil_timestamps(S)
    sched = zh.Schedule(W, ts)
    geom = zh.reproj_prepare(dev(d["db_2d"][:, :, :2]), dev(d["camera_param"]), dev(d["db_2d"][:, :, 2]))
    x, T = dev(x0), dev(T0)
    zh.oil_run(W, sched, x, geom, T, 0, S, S // 5)
    w64 = O.cast_weights(weights0, np.float64)
    for h in range(H):
        sl = slice(h * N, (h + 1) * N)
        args = (d["db_2d"][:, :, :2], d["camera_param"], d["db_2d"][:, :, 2], T0[sl, None, :], S)
        x32, T32 = O.oil_loop(weights0, x0[sl], *args)
        x64, T64 = O.oil_loop(w64, x0[sl], *args, dtype=np.float64)
        gap = np.abs(x32 - x64).max()
        dx = np.abs(x[sl].cpu().numpy() - x64).max()
        assert dx <= 1.5 * gap + 2e-6, (h, dx, gap)
        gapT = np.abs(T32 - T64).max()
        dT = np.abs(T[sl].cpu().numpy() - T64[:, 0, :]).max()
        assert dT <= 1.5 * gapT + 2e-5, (h, dT, gapT)


IPO_CASES = [(N, axes, kname) for N in (8, 64) for axes in ("z", "xyz") for kname in ("h36m", "pw3d")]


@pytest.mark.parametrize("N,axes,kname", IPO_CASES)
def test_ipo_trajectory_golden(zh, golden, N, axes, kname):
    """IPO against the reference's own parameter trajectories (tools/gen_golden.py::gen_ipo, opt_main.py:180-195).

    Adam on an L1 loss is chaotic: a residual whose sign differs between two implementations moves a parameter by
    2 x lr x (1 - beta1) = 0.02 in one iteration.  The reference run in fp32 is itself gap(it) = |ref32 - ref64| away
    from its own fp64 run (RotOpt().double(): 2.4e-8 after 1 iteration, 1e-7..6e-7 after 5, 1e-5..1e-3 after 30, up
    to 5e-2 after 50 - by then single poses have taken a different sign somewhere, in ref32 as in any other fp32
    implementation; the numpy oracle shows the same events at the same iterations).  Criterion, with the fp64 run as
    arbiter like the OIL loop's: iteration 1 within 1e-7, iteration 5 within 2e-6, iterations 1..30 the worst pose
    within 4 x gap + 1e-7 and the median pose within 1.5 x the reference's own median gap; iterations 31..50, where the worst pose measures a sign event and not arithmetic, the
    MEDIAN pose within 2 x the median gap + 1e-7 and every pose within 0.1.  What the kernel does at every state on
    the way is pinned separately by test_ipo_single_iterations_from_reference_state.  All 50 achieved deviations
    go to the parity report."""
    import zedo_oracle as O
    g = golden("ipo")
    kl, ipoT, minT = ([0, 1, 4], 3.0, 0.5) if kname == "h36m" else (list(range(17)), 8.0, 0.2)
    tag = f"{N}_{axes}_{kname}"
    uv = dev(g[f"db2d_{N}"][:, :, :2])
    K = dev(g[f"K_{N}"])
    x0 = dev(g["cluster0"][None])
    norm = N * len(kl) * 2
    q32, s32 = g[f"trace_q_{tag}"], g[f"trace_scale_{tag}"]
    q64, s64 = g[f"trace_q64_{tag}"], g[f"trace_scale64_{tag}"]
    rows = []
    for it in range(1, q64.shape[0] + 1):
        R, T, q, sc = zh.ipo_fit(x0, uv, K, kl, axes, ipoT, minT, 2.0, it, norm, N, return_params=True)
        qn, sn = q.cpu().numpy().astype(np.float64), sc.cpu().numpy().astype(np.float64)
        s64i, s32i = s64[it - 1].reshape(-1), s32[it - 1].reshape(-1).astype(np.float64)
        dp = np.maximum(np.abs(qn - q64[it - 1]).max(1), np.abs(sn - s64i))                     # per pose
        gp = np.maximum(np.abs(q32[it - 1] - q64[it - 1]).max(1), np.abs(s32i - s64i))
        d32 = max(np.abs(qn - q32[it - 1]).max(), np.abs(sn - s32i).max())
        rows.append(dict(it=it, hip_vs_ref64=float(dp.max()), hip_vs_ref32=float(d32), ref32_vs_ref64=float(gp.max()),
                         hip_vs_ref64_median=float(np.median(dp)), ref32_vs_ref64_median=float(np.median(gp)),
                         hip_vs_ref64_p90=float(np.percentile(dp, 90)), ref32_vs_ref64_p90=float(np.percentile(gp, 90)),
                         poses_beyond_2_gaps=int((dp > 2.0 * gp + 1e-6).sum()), poses=int(len(dp)),
                         # the pose the REFERENCE's own fp32 run is farthest from its fp64 run on, and everybody else
                         ref_worst_pose=int(np.argmax(gp)), hip_at_ref_worst_pose=float(dp[int(np.argmax(gp))]),
                         hip_other_poses=float(np.delete(dp, int(np.argmax(gp))).max()) if len(dp) > 1 else 0.0))
    _report(f"ipo_{tag}", rows)
    for r in rows:
        if r["it"] <= 30:
            # The worst of N poses is ONE ill-conditioned pose whose deviation grows x1.5 per iteration from iteration ~3
            # on, in the reference's fp32 run and here alike (64 poses, xyz, 17 joints: reference 0.25 -> 0.54 -> 1.5 ->
            # 3.9 e-6 at iterations 2 / 5 / 8 / 10, kernel 0.27 -> 0.70 -> 3.4 -> 8.0): the same unstable mode excited by
            # two different first roundings.  A ratio of two such amplitudes says nothing at a factor of 2 (round 3's
            # sequential joint sum drew 1.7, round 4's half-wave butterfly 2.3), so THAT pose - identified by the reference
            # itself: the pose its own fp32 run is farthest from its fp64 run on - is held to 4 gaps, every OTHER pose to the
            # 2 gaps of rounds 1-3 (round 4 had relaxed all poses to 4), and the BULK gets a bound of its own: the
            # median pose within 1.5 x the reference's median gap (measured over the eight cases: worst pose 0.57 ... 2.25
            # gaps, median pose 0.85 ... 1.35 x the reference's median).
            assert r["hip_at_ref_worst_pose"] <= 4.0 * r["ref32_vs_ref64"] + 1e-7, {k: r[k] for k in ("it", "ref_worst_pose", "hip_at_ref_worst_pose", "ref32_vs_ref64")}
            assert r["hip_other_poses"] <= 2.0 * r["ref32_vs_ref64"] + 1e-7, {k: r[k] for k in ("it", "ref_worst_pose", "hip_other_poses", "ref32_vs_ref64")}
            assert r["hip_vs_ref64_median"] <= 1.5 * r["ref32_vs_ref64_median"] + 2e-8, {k: r[k] for k in ("it", "hip_vs_ref64_median", "ref32_vs_ref64_median")}
        else:
            assert r["hip_vs_ref64_median"] <= 2.0 * r["ref32_vs_ref64_median"] + 1e-7 and r["hip_vs_ref64"] <= 0.1, r
            # a regression that hits a minority of the poses late in the fit must not hide behind the median.  Past
            # iteration 30 the per-pose deviations grow exponentially (x1.25 per iteration) from wherever a pose's first
            # rounding difference happened to fall, so which poses carry the 90th percentile differs between any two fp32
            # runs; measured over the 8 cases: p90(hip) / p90(reference fp32) = 0.5 ... 8.3 (= the drift of 9 more
            # iterations), 0 ... 13 of 64 poses beyond 2 x their own gap.  Held to 12 x the reference's own 90th-percentile
            # gap (+ one sign event's worth where 8 poses make the 90th percentile the worst pose) and to a quarter of
            # the poses (three of eight) beyond two gaps: an error that moves 10 % of the poses by more than that is caught here, every
            # smaller one by the per-iteration test below, which has no chaos to hide behind.
            assert r["hip_vs_ref64_p90"] <= 12.0 * r["ref32_vs_ref64_p90"] + (1e-7 if N >= 64 else 0.03), r
            assert r["poses_beyond_2_gaps"] <= max(3, r["poses"] // 4), {k: r[k] for k in ("it", "poses_beyond_2_gaps", "poses", "hip_vs_ref64_p90", "ref32_vs_ref64_p90")}
    assert rows[0]["hip_vs_ref64"] <= 1e-7 and rows[4]["hip_vs_ref64"] <= 2e-6, (rows[0], rows[4])
    # T0 (0 iterations): scale = 1
    R, T = zh.ipo_fit(x0, uv, K, kl, axes, ipoT, minT, 2.0, 0, norm, N)
    np.testing.assert_allclose(T.cpu().numpy(), g[f"T0_{tag}"].reshape(N, 3), atol=1e-6, rtol=0)
    assert np.allclose(R.cpu().numpy(), np.eye(3)[None], atol=0)
    # 500 iterations: the trajectories have decorrelated per pose by then, so compare what the fit is for - the
    # achieved loss (oracle forward on the HIP parameters) against the reference's final loss
    R, T, q, sc = zh.ipo_fit(x0, uv, K, kl, axes, ipoT, minT, 2.0, 500, norm, N, return_params=True)
    cond = g[f"db2d_{N}"][:, :, :2]
    x0n = np.broadcast_to(g["cluster0"][None], (N, 17, 3)).astype(np.float32)
    T0 = O.ipo_init_T(cond, g[f"K_{N}"], ipoT).reshape(N, 3)
    loss, _, _, _ = O.ipo_loss_and_grads(q.cpu().numpy(), sc.cpu().numpy(), x0n[:, kl], T0, g[f"K_{N}"], cond[:, kl],
                                         axes, minT, 2.0, norm)
    _report(f"ipo_{tag}_final", [dict(loss_hip=float(loss), loss_ref=float(g[f"loss_{tag}"]))])
    assert abs(loss - g[f"loss_{tag}"]) <= 0.05 * g[f"loss_{tag}"]
    Rn = R.cpu().numpy()
    assert np.allclose(np.einsum("bij,bkj->bik", Rn, Rn), np.eye(3)[None], atol=1e-5)
    np.testing.assert_allclose(Rn, O.quaternion_to_matrix(q.cpu().numpy()), atol=1e-6, rtol=0)


IPO_TWINS = r"""
import hashlib, json, os, sys
import numpy as np
root = %r
sys.path.insert(0, os.path.join(root, "zedo-release_amd"))
import torch
import zedo_hip as zh
from lib.dataset import synthetic as syn
out = {}
# the shipped key lists (17 joints: 3DPW, [0, 1, 4]: H36M) and custom ZeDO.IPO_keylist values of other lengths (round 6: every length
# 1 .. 17 has its lane-per-row instantiation)
for tag, kl, ipoT, minT, N, H in (("pw3d", list(range(17)), 8.0, 0.2, 301, 3), ("h36m", [0, 1, 4], 3.0, 0.5, 257, 2),
                                  ("k1", [0], 3.0, 0.5, 129, 2), ("k5", [0, 2, 5, 11, 14], 3.0, 0.5, 131, 2),
                                  ("k8", [0, 1, 4, 7, 8, 11, 14, 16], 8.0, 0.2, 130, 2), ("k12", list(range(0, 12)), 8.0, 0.2, 67, 2),
                                  ("k16", list(range(1, 17)), 8.0, 0.2, 65, 2)):
    d = syn.make_poses(N, seed=5, conf_mode="uniform")
    cl = syn.make_clusters(H, seed=5)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    x0 = dev(cl - cl[:, 0:1])
    for axes in ("z", "xyz"):
        R, T, q, sc = zh.ipo_fit(x0, dev(d["db_2d"][:, :, :2]), dev(d["camera_param"]), kl, axes, ipoT, minT, 2.0, 500, N * len(kl) * 2, H * N, return_params=True)
        h = hashlib.sha256()
        for t in (R, T, q, sc):
            h.update(t.cpu().numpy().tobytes())
        out[tag + "_" + axes] = h.hexdigest()
        assert bool(torch.isfinite(R).all())
print("RESULT " + json.dumps(out))
"""


def test_ipo_kernels_are_bitwise_twins():
    """The IPO has two kernels - one row per half-wave with a joint per lane (small batches: latency), one lane per row
    (batches that fill the chip: throughput) - chosen by the LOCAL row count.  That is only legitimate if a row's fit does
    not depend on the choice: both are pinned (ZEDO_IPO_KERNEL=half|row, read once per process) on the same problems - the
    17-joint and the 3-joint key list and custom lists of 1, 5, 8, 12 and 16 joints, z and xyz axes, 500 iterations - and must
    agree BIT FOR BIT in R, T, q and scale."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for pin in ("half", "row"):
        e = dict(os.environ)
        e["ZEDO_IPO_KERNEL"] = pin
        r = subprocess.run([sys.executable, "-c", IPO_TWINS % root], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        res[pin] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["half"] == res["row"], (res["half"], res["row"])



IPO_CUSTOM = [(N, axes, kname) for N in (8, 64) for axes in ("z", "xyz") for kname in ("k1", "k5", "k8", "k12")]


@pytest.mark.parametrize("N,axes,kname", IPO_CASES + IPO_CUSTOM)
def test_ipo_single_iterations_from_reference_state(zh, golden, N, axes, kname):
    """Every one of the first 50 Adam iterations, taken on its own from the REFERENCE's optimiser state.

    Trajectories of this fit decorrelate (a residual whose sign two implementations disagree on moves a parameter by
    2 x lr x (1 - beta1)), so a trajectory comparison alone cannot tell a wrong gradient term from chaos.  Here each
    iteration starts from the state (parameters, exp_avg, exp_avg_sq) of the reference's fp64 run - reproduced by the
    fp64 oracle, which is first checked against the reference's own fp64 trace (tools/gen_golden.py::gen_ipo,
    RotOpt().double(), opt_main.py:180-195) to 1e-8 - runs ONE iteration of ipo_kernel (zedo_ipo_fit_resume) and must
    land on the reference's next state: |delta parameter| <= 5e-7 for every pose whose residuals are all
    sign-unambiguous (|e| >= 1e-3 px; the root joint sits on the ray through the origin and contributes no gradient,
    it is not counted).  Ambiguous (pose, iteration) pairs are counted and reported, not compared."""
    import zedo_oracle as O
    # round 6: also for key lists the shipped configurations do not use (1, 5, 8, 12 joints: tests/golden/ipo_custom.npz,
    # tools/gen_golden.py::gen_ipo_custom) - every key-list length has its own lane-per-row instantiation since this round
    if kname in ("h36m", "pw3d"):
        g = golden("ipo")
        kl, ipoT, minT = ([0, 1, 4], 3.0, 0.5) if kname == "h36m" else (list(range(17)), 8.0, 0.2)
        step_tol = 5e-7                   # measured over the eight shipped cases: <= 2.4e-7
    else:
        g = golden("ipo_custom")
        kl = [int(k) for k in g[f"keylist_{kname}"]]
        ipoT, minT = (3.0, 0.5) if kname in ("k1", "k5") else (8.0, 0.2)
        # one Adam step from a float64 state, in fp32: a few ulp of a unit-size parameter (1.2e-7 each).  The sixteen custom cases measure up
        # to 5.8e-7 (64 poses, xyz, 5 joints, iteration 2: one pose); held to 1e-6 = 8 ulp, the shipped cases stay at 5e-7
        step_tol = 1e-6
    tag = f"{N}_{axes}_{kname}"
    cond, Kn = g[f"db2d_{N}"][:, :, :2], g[f"K_{N}"]
    uv, K, x0 = dev(cond), dev(Kn), dev(g["cluster0"][None])
    norm = N * len(kl) * 2
    c64, K64 = cond.astype(np.float64), Kn.astype(np.float64)
    x64 = np.broadcast_to(g["cluster0"][None], (N, 17, 3)).astype(np.float64)
    tr = []
    O.ipo_fit(x64[:, kl], O.ipo_init_T(c64, K64, ipoT, dtype=np.float64), K64, c64[:, kl], axes, minT, 2.0, 50,
              normaliser=norm, dtype=np.float64, trace=tr)
    for it in range(50):                                   # the arbiter itself is the reference's fp64 run
        assert np.abs(tr[it][0] - g[f"trace_q64_{tag}"][it]).max() <= 1e-8
        assert np.abs(tr[it][1] - g[f"trace_scale64_{tag}"][it].reshape(-1)).max() <= 1e-8

    def pack(q, sc, mq, vq, ms, vs):
        return np.concatenate([q, sc[:, None], mq, ms[:, None], vq, vs[:, None]], axis=1)
    z4, z1 = np.zeros((N, 4)), np.zeros(N)
    q0 = z4.copy(); q0[:, 0] = 1
    states = [pack(q0, np.ones(N), z4, z4, z1, z1)] + [pack(t[0], t[1], t[3], t[4], t[5], t[6]) for t in tr]
    moving = np.abs(g["cluster0"][kl]).max(-1) > 0          # joints off the root
    worst, n_amb, worst_m, worst_v = 0.0, 0, 0.0, 0.0
    for it in range(50):
        st = dev(states[it].astype(np.float32))
        zh.ipo_fit(x0, uv, K, kl, axes, ipoT, minT, 2.0, 1, norm, N, state=st, it_begin=it)
        out = st.cpu().numpy().astype(np.float64)
        clear = tr[it][7][:, moving, :].reshape(N, -1).min(1) >= 1e-3
        n_amb += int((~clear).sum())
        d = np.abs(out - states[it + 1])
        worst = max(worst, float(d[clear, :5].max()))
        worst_m = max(worst_m, float(d[clear, 5:10].max()))
        worst_v = max(worst_v, float((d[clear, 10:] / (np.abs(states[it + 1][clear, 10:]) + 1e-12)).max()))
        assert d[clear, :5].max() <= step_tol, (it, d[clear, :5].max())
    _report(f"ipo_resync_{tag}", [dict(iterations=50, poses=N, ambiguous_pose_iterations=n_amb,
                                        max_param_delta=worst, max_exp_avg_delta=worst_m,
                                        max_exp_avg_sq_rel_delta=worst_v)])
    assert n_amb <= 0.05 * 50 * N, n_amb


def test_rotate_init(zh):
    rng = np.random.default_rng(3)
    H, N = 4, 9
    x0 = rng.standard_normal((H, 17, 3)).astype(np.float32)
    R = rng.standard_normal((H * N, 3, 3)).astype(np.float32)
    x = zh.rotate_init(dev(x0), dev(R), N).cpu().numpy()
    ref = np.einsum("bij,bkj->bki", R, np.repeat(x0, N, axis=0))
    np.testing.assert_allclose(x, ref, atol=1e-6, rtol=0)


def test_min_mpjpe_golden(zh, golden):
    g = golden("eval_multi")
    preds = g["preds"]                       # [N,H,17,3] (reference layout)
    N, H = preds.shape[:2]
    rows = np.ascontiguousarray(np.swapaxes(preds, 0, 1).reshape(H * N, 17, 3))   # row = h*N + n
    gt = (g["gt_mm_h36m"] - g["gt_mm_h36m"][:, 0:1]) / 1000.0
    for p2, key, tol in ((False, "err_p1", 1e-12), (True, "err_p2", 3e-7)):
        err, best, best_h = zh.min_mpjpe(dev(rows), dev(gt, torch.float64), N, procrustes=p2)
        e = err.cpu().numpy().reshape(H, N).T
        np.testing.assert_allclose(e, g[key], atol=tol, rtol=0)
        np.testing.assert_allclose(best.cpu().numpy(), g[key].min(1), atol=tol, rtol=0)
        assert np.array_equal(best_h.cpu().numpy(), g[key].argmin(1))
    # shard: rows [N+5, 3N+2) only
    lo, hi = N + 5, 3 * N + 2
    err, best, best_h = zh.min_mpjpe(dev(rows[lo:hi]), dev(gt, torch.float64), N, procrustes=False, row_offset=lo)
    full = g["err_p1"]
    for n in range(N):
        hs = [h for h in range(H) if lo <= h * N + n < hi]
        ref = min(full[n, h] for h in hs)
        assert abs(best[n].item() - ref) < 1e-12 and best_h[n].item() == min(hs, key=lambda h: full[n, h])


@pytest.mark.parametrize("tag", ["planar_pred", "planar_gt"])
def test_min_mpjpe_rank2_alignment_matches_the_reference_value(zh, golden, tag):
    """Planar prediction / planar ground truth (exactly in z = 0, and in a random plane to float32 rounding): the
    SVD of the alignment has a zero singular value whose direction sign is arbitrary in numpy/LAPACK, but the
    ERROR is the same for either sign (tests/test_oracle_golden.py shows why); the kernel's choice must therefore
    reproduce the reference's error values (tools/gen_golden.py::gen_eval, transforms.py:42-127)."""
    g = golden("eval_multi")
    G, P, ref = g[f"deg_{tag}_gt"], g[f"deg_{tag}_pred"], g[f"deg_{tag}_err_p2"]
    err, best, idx = zh.min_mpjpe(dev(P.astype(np.float32)), dev(G - 0.0, torch.float64), len(G), procrustes=True)
    _report(f"procrustes_rank2_{tag}", [dict(max_abs_err_delta=float(np.abs(err.cpu().numpy() - ref).max()))])
    np.testing.assert_allclose(err.cpu().numpy(), ref, atol=3e-7, rtol=0)


def test_min_mpjpe_nan_hypothesis_poisons_the_pose_like_numpy(zh, golden):
    """np.amin / np.argmin (lib/dataset/h36m.py:411-412) return NaN and the first NaN index when any hypothesis of
    a pose is NaN; a diverged sample must not be masked by a finite minimum."""
    g = golden("eval_multi")
    preds = g["preds"].copy()
    N, H = preds.shape[:2]
    preds[3, 2, 5, 1] = np.nan            # pose 3: hypothesis 2
    preds[7, 4] = np.nan                  # pose 7: hypotheses 1 and 4 -> first NaN index 1
    preds[7, 1, 0, 0] = np.nan
    rows = np.ascontiguousarray(np.swapaxes(preds, 0, 1).reshape(H * N, 17, 3))
    gt = (g["gt_mm_h36m"] - g["gt_mm_h36m"][:, 0:1]) / 1000.0
    for p2 in (False, True):
        err, best, best_h = zh.min_mpjpe(dev(rows), dev(gt, torch.float64), N, procrustes=p2)
        e = err.cpu().numpy().reshape(H, N).T
        assert np.isnan(e[3, 2]) and np.isnan(e[7, 1]) and np.isnan(e[7, 4]) and np.isnan(e).sum() == 3
        np.testing.assert_array_equal(best.cpu().numpy(), np.amin(e, axis=1))     # NaN == NaN under assert_array_equal
        assert np.array_equal(best_h.cpu().numpy(), np.argmin(e, axis=1))
        assert best_h[3].item() == 2 and best_h[7].item() == 1


def test_selection_staged_kernel_is_bitwise_the_lane_per_row_kernel(zh):
    """Round 6: for J = 17 the per-row error kernel stages the 64 rows of a wave (one contiguous piece of the pose tensor, coalesced
    16-byte loads) and their ground-truth poses through the LDS; the arithmetic per row is the old statement sequence.  A pose tensor
    whose base is not 16-byte aligned takes the old one-lane-per-row kernel: the same rows through both must agree BIT FOR BIT - errors
    of every row, per-pose minimum, arg-min - for P1 and P2, on a shard (row_offset) whose tiles wrap around the N poses several
    times, with a ragged last tile and NaN hypotheses."""
    rng = np.random.default_rng(66)
    N, H, off = 23, 211, 17
    B = N * H - off - 5
    x = (0.3 * rng.standard_normal((B, 17, 3))).astype(np.float32)
    x[5, 3, 1] = np.nan
    x[700:703] = np.nan
    gt = dev(0.3 * rng.standard_normal((N, 17, 3)), torch.float64)
    xa = dev(x)
    buf = torch.empty(B * 51 + 1, dtype=torch.float32, device="cuda")
    xb = buf[1:].view(B, 17, 3)
    xb.copy_(xa)
    assert xa.data_ptr() % 16 == 0 and xb.data_ptr() % 16 == 4 and xb.is_contiguous()
    bits = lambda t: t.view(torch.int64) if t.dtype == torch.float64 else t
    for p2 in (False, True):
        a = zh.min_mpjpe(xa, gt, N, procrustes=p2, row_offset=off)
        b = zh.min_mpjpe(xb, gt, N, procrustes=p2, row_offset=off)
        for ta, tb in zip(a, b):
            assert torch.equal(bits(ta), bits(tb))
        assert int(torch.isnan(a[0]).sum()) == 4 and bool(torch.isfinite(a[0][~torch.isnan(a[0])]).all())
        # the per-pose minimum against numpy on the row errors (np.amin / np.argmin semantics incl. NaN), rows h * N + n - off
        e = np.full((H * N,), np.inf)
        e[off:off + B] = a[0].cpu().numpy()
        e = e.reshape(H, N)
        held = np.zeros((H * N,), bool); held[off:off + B] = True; held = held.reshape(H, N)
        for n in range(N):
            col = np.where(held[:, n], e[:, n], np.inf)
            want_h = int(np.argmin(col)) if not np.isnan(col).any() else int(np.flatnonzero(np.isnan(col))[0])
            assert int(a[2][n]) == want_h and (np.isnan(col[want_h]) and np.isnan(float(a[1][n])) or float(a[1][n]) == col[want_h])


def test_selection_pose_major_kernel_is_bitwise_the_row_major_pair(zh):
    """From 8 192 poses up (J = 17, 16-byte aligned rows) the row errors are computed POSE-MAJOR (row_error17_pose_major_kernel: 64 poses
    per wave, their ground truth staged once per chunk of hypotheses, the hypotheses' rows streamed one ahead) and the arg-min by the
    lane-per-pose kernel.  Against the row-major generic kernel (one lane per row straight from memory, taken for an unaligned row
    pointer): every row's error, the per-pose minimum and the arg-min BIT FOR BIT - P1 and P2, a shard that starts and ends inside a
    hypothesis, an odd pose count (every hypothesis' tile starts at another 4-byte alignment), NaN rows, a ragged last pose tile, more
    than one hypothesis chunk."""
    rng = np.random.default_rng(68)
    N, H = 8219, 5
    gt = dev(0.3 * rng.standard_normal((N, 17, 3)), torch.float64)
    bits = lambda t: t.view(torch.int64) if t.dtype == torch.float64 else t
    for off, B in ((0, N * H), (N + 777, 3 * N + 5)):
        x = (0.3 * rng.standard_normal((B, 17, 3))).astype(np.float32)
        x[11, 2, 0] = np.nan
        x[B - 3] = np.nan
        xa = dev(x)
        buf = torch.empty(B * 51 + 1, dtype=torch.float32, device="cuda")
        xb = buf[1:].view(B, 17, 3)
        xb.copy_(xa)
        assert xa.data_ptr() % 16 == 0 and xb.data_ptr() % 16 == 4
        for p2 in (False, True):
            a = zh.min_mpjpe(xa, gt, N, procrustes=p2, row_offset=off)
            b = zh.min_mpjpe(xb, gt, N, procrustes=p2, row_offset=off)
            for name, ta, tb in zip(("err", "best", "best_h"), a, b):
                assert torch.equal(bits(ta), bits(tb)), (off, B, p2, name, int((bits(ta) != bits(tb)).sum()))
            assert int(torch.isnan(a[0]).sum()) == 2 and int((a[2] < 0).sum()) == 0          # both shards hold at least one hypothesis of every pose


def test_pose_min_lane_per_pose_kernel_follows_numpy(zh):
    """From 8 192 poses up the arg-min over hypotheses runs one LANE per pose (coalesced reads of one hypothesis at a time) instead of one
    wavefront per pose; same total order (np.amin / np.argmin incl. NaN, ties to the lower hypothesis), on a shard that starts and ends
    inside a hypothesis, and with poses that hold no local hypothesis at all."""
    rng = np.random.default_rng(67)
    N, H = 8300, 5
    e = rng.random((H, N))
    e[2, 17] = np.nan; e[4, 17] = np.nan; e[1, 300] = e[3, 300] = e.min() / 2          # NaN wins (first NaN index); ties to the lower index
    flat = e.reshape(-1)
    for off, B in ((0, H * N), (N // 2 + 3, 3 * N + 11), (2 * N + 100, 200)):
        best, idx = zh.pose_min(dev(flat[off:off + B], torch.float64), N, row_offset=off)
        held = np.zeros((H * N,), bool); held[off:off + B] = True; held = held.reshape(H, N)
        col = np.where(held, e, np.inf)
        nan_any = np.isnan(col).any(0)
        want_h = np.where(nan_any, np.argmax(np.isnan(col), 0), np.argmin(np.where(np.isnan(col), np.inf, col), 0))
        none = ~held.any(0)
        want_h = np.where(none, -1, want_h)
        got_h, got = idx.cpu().numpy(), best.cpu().numpy()
        assert np.array_equal(got_h, want_h)
        ok = ~none & ~nan_any
        assert np.array_equal(got[ok], col[want_h[ok], np.flatnonzero(ok)]) and np.isnan(got[nan_any & ~none]).all() and np.isinf(got[none]).all()


@pytest.mark.parametrize("B", [1300, 2304, 4096, 5000, 7000, 8500, 10000, 16384, 18000, 22000])
def test_score_network_every_launch_shape(zh, W, weights0, B):
    """Row counts that take each tile-selection branch of the dense layers - exact fp32: 32 / 64 / 128-row tiles chosen by the
    cost model below one round, the pair launch with a short and a long remainder, exact rounds; f16x3: 64x64 tiles only (up to
    2 048 rows), 128x128 tiles (below 8 192), a partly filled round of 128x256 tiles with and without a last 64-row strip, one
    whole round, a whole round + a short remainder on 64x64 tiles (18 000) and + a long one on big tiles (22 000); the oracle
    is evaluated on a sample of rows (rows are independent) including the last ones."""
    import zedo_oracle as O
    rng = np.random.default_rng(B)
    x = (0.3 * rng.standard_normal((B, 17, 3))).astype(np.float32)
    pick = np.unique(np.concatenate([rng.integers(0, B, 48), np.arange(B - 40, B), np.arange(0, 8)]))
    s = zh.Schedule(W, np.array([0.07], np.float32))
    eps = zh.score_eps(W, s, 0, dev(x)).cpu().numpy()
    ref = O.score_model_forward(weights0, x[pick], np.float32(0.07) * np.float32(999))
    np.testing.assert_allclose(eps[pick], ref, atol=3e-6, rtol=0)
    assert np.isfinite(eps).all()


# ---------------------------------------------------------------- full-size properties

def test_rows_are_independent_at_full_size(zh, W):
    """At BASELINE size (B = 50 * 1015 rows) a row's trajectory must not depend on its neighbours or on
    where the row sits in a tile: run 3 steps on the full batch and on a 1000-row slice, bit for bit."""
    from lib.dataset import synthetic as syn
    import zedo_oracle as O
    H, N, S = 50, 1015, 5
    d = syn.make_poses(N, seed=1)
    rng = np.random.default_rng(11)
    x0 = (0.25 * rng.standard_normal((H * N, 17, 3))).astype(np.float32)
    T0 = np.tile(d["db_3d"][:, 0, :], (H, 1)).astype(np.float32)
    sched = zh.Schedule(W, O.oil_timestamps(S))
    geom = zh.reproj_prepare(dev(d["db_2d"][:, :, :2]), dev(d["camera_param"]), dev(d["db_2d"][:, :, 2]))
    x, T = dev(x0), dev(T0)
    zh.oil_run(W, sched, x, geom, T, 0, S, 2)
    lo, hi = 20011, 21011
    xs, Ts = dev(x0[lo:hi]), dev(T0[lo:hi])
    zh.oil_run(W, sched, xs, geom, Ts, 0, S, 2, row_offset=lo)
    assert torch.equal(x[lo:hi], xs) and torch.equal(T[lo:hi], Ts)
    assert torch.isfinite(x).all()
    lo2, hi2 = 49900, H * N                                   # rows served by the remainder tiles of the pair launches
    xs2, Ts2 = dev(x0[lo2:hi2]), dev(T0[lo2:hi2])
    zh.oil_run(W, sched, xs2, geom, Ts2, 0, S, 2, row_offset=lo2)
    assert torch.equal(x[lo2:hi2], xs2) and torch.equal(T[lo2:hi2], Ts2)
    # after the reprojection correction every joint lies on its ray: the correction is idempotent
    g1 = zh.reproj_grad(x, geom, T, False)
    x2 = x + g1
    g2 = zh.reproj_grad(x2, geom, T, False)
    assert g2.abs().max().item() < 1e-5 * max(1.0, x2.abs().max().item() + T.abs().max().item())


def test_selection_properties_at_full_size(zh):
    """BASELINE size (N = 1015, H = 50): the best-of-H selection must not depend on how the rows are sharded
    (min over shards == global min, lowest index on ties) nor on the order of the hypotheses."""
    from lib.dataset import synthetic as syn
    H, N = 50, 1015
    d = syn.make_poses(N, seed=5, dtype3d=np.float64)
    gt = d["db_3d"] - d["db_3d"][:, 0:1]
    rng = np.random.default_rng(3)
    rows = (np.tile(gt, (H, 1, 1)) + 0.08 * rng.standard_normal((H * N, 17, 3))).astype(np.float32)
    rows[7 * N:8 * N] = rows[3 * N:4 * N]                      # exact ties between hypotheses 3 and 7
    x, g = dev(rows), dev(gt, torch.float64)
    for p2 in (False, True):
        err, best, idx = zh.min_mpjpe(x, g, N, procrustes=p2)
        e = err.reshape(H, N)
        assert torch.equal(best, e.min(0).values) and torch.equal(idx.long(), e.argmin(0))
        assert not (idx == 7).any()                            # first minimum wins, like np.argmin
        # three uneven shards, combined on the host the way reduce_min_over_ranks does
        cuts = [0, 17 * N + 333, 31 * N + 1, H * N]
        sb = torch.full((3, N), float("inf"), dtype=torch.float64, device="cuda")
        si = torch.full((3, N), 2 ** 31 - 1, dtype=torch.int64, device="cuda")
        for k in range(3):
            lo, hi = cuts[k], cuts[k + 1]
            _, b, i = zh.min_mpjpe(x[lo:hi].contiguous(), g, N, procrustes=p2, row_offset=lo)
            sb[k] = b
            si[k] = torch.where(i < 0, torch.full_like(i, 2 ** 31 - 1), i).long()
        gb = sb.min(0).values
        gi = torch.where(sb == gb, si, torch.full_like(si, 2 ** 31 - 1)).min(0).values
        assert torch.equal(gb, best) and torch.equal(gi, idx.long())
        # hypothesis order: reversing the hypotheses permutes the errors and nothing else
        xr = x.reshape(H, N, 17, 3).flip(0).reshape(H * N, 17, 3).contiguous()
        err_r, best_r, _ = zh.min_mpjpe(xr, g, N, procrustes=p2)
        assert torch.equal(err_r.reshape(H, N).flip(0), e) and torch.equal(best_r, best)
        if p2:      # the alignment minimises the SQUARED error; the mean joint distance drops for all but a few rows
            assert err.mean() < err_p1.mean() and (err <= err_p1 + 1e-12).double().mean() > 0.99
        err_p1 = err


# ---------------------------------------------------------------- argument validation, chunking

def test_bad_arguments_are_rejected(zh, W):
    import ctypes
    import zedo_oracle as O
    lib = zh._lib
    s = zh.Schedule(W, O.oil_timestamps(10))
    x = torch.zeros(4, 17, 3, device="cuda")
    ws = zh.workspace(4)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    # step out of range, B = 0, NULL pointers, workspace too small -> negative ZEDO_E_* codes, nothing launched
    assert lib.zedo_sde_step(W._h, s._h, 10, P(x), 4, P(ws), ws.numel(), None) == -1
    assert lib.zedo_sde_step(W._h, s._h, 0, P(x), 0, P(ws), ws.numel(), None) == -1
    assert lib.zedo_sde_step(W._h, s._h, 0, None, 4, P(ws), ws.numel(), None) == -1
    assert lib.zedo_sde_step(W._h, s._h, 0, P(x), 4, P(ws), 16, None) == -3
    assert lib.zedo_oil_run(W._h, s._h, P(x), None, None, 0, 5, 1, 4, 4, 0, P(ws), ws.numel(), None) == -1
    assert lib.zedo_workspace_bytes(0) == 0
    # rows beyond H*N would index the cluster poses out of bounds (H = 2 hypotheses x N = 3 poses, 7 rows asked)
    x0 = torch.zeros(2, 17, 3, device="cuda"); uv = torch.zeros(3, 17, 2, device="cuda")
    K = torch.eye(3, device="cuda").repeat(3, 1, 1).contiguous()
    with pytest.raises(zh.ZedoError):
        zh.ipo_fit(x0, uv, K, [0, 1, 4], "z", 3.0, 0.5, 2.0, 1, 18, 7)
    with pytest.raises(zh.ZedoError):
        zh.ipo_fit(x0, uv, K, [0, 1, 4], "z", 3.0, 0.5, 2.0, 1, 18, 4, row_offset=3)
    with pytest.raises(zh.ZedoError):
        zh.rotate_init(x0, torch.zeros(7, 3, 3, device="cuda"), 3)
    with pytest.raises(zh.ZedoError):
        zh.Schedule(W, np.zeros(0, np.float32))
    with pytest.raises(zh.ZedoError):       # wrong parameter count
        zh.Weights({k: np.zeros(3, np.float32) for k in zh.param_names()})
    # the entry points of ABI version 3, raw
    geom = torch.zeros(3, 17, 8, device="cuda")
    cnt = ctypes.c_int(0)
    assert lib.zedo_reproj_degenerate(None, 3, 17, ctypes.cast(ctypes.byref(cnt), ctypes.c_void_p), None) == -1
    assert lib.zedo_reproj_degenerate(P(geom), 0, 17, ctypes.cast(ctypes.byref(cnt), ctypes.c_void_p), None) == -1
    assert lib.zedo_reproj_degenerate(P(geom), 3, 17, None, None) == -1
    err = torch.zeros(6, dtype=torch.float64, device="cuda"); best = torch.zeros(3, dtype=torch.float64, device="cuda")
    bh = torch.zeros(3, dtype=torch.int32, device="cuda")
    assert lib.zedo_pose_min(None, 6, 3, 0, P(best), P(bh), None) == -1
    assert lib.zedo_pose_min(P(err), 6, 3, -1, P(best), P(bh), None) == -1
    assert lib.zedo_pose_min(P(err), 0, 3, 0, P(best), P(bh), None) == -1
    assert lib.zedo_weights_set_math(None, 0, None) == -1 and lib.zedo_weights_set_math(W._h, 7, None) == -1
    assert lib.zedo_weights_get_math(None) == -1 and lib.zedo_weights_get_math(W._h) == zh.MATH_MODES[W.math]
    with pytest.raises(zh.ZedoError):       # host tensor where a device tensor is required
        zh.reproj_prepare(torch.zeros(2, 17, 2), torch.zeros(2, 3, 3))
    assert b"argument" in lib.zedo_error_string(-1)


def test_row_chunking_and_reprojection_fusion_are_bitwise_neutral(tmp_path):
    """ZEDO_CHUNK_ROWS bounds the workspace for very large batches (BASELINE config 5: 5 M rows); the chunk
    loop must give exactly the rows of the unchunked run.  Separate processes: the cap is read once."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(%r, "zedo-release_amd")); sys.path.insert(0, os.path.join(%r, "oracle"))
import zedo_hip as zh, zedo_oracle as O
from lib.dataset import synthetic as syn
H, N, S = 5, 301, 7
d = syn.make_poses(N, seed=4, conf_mode="uniform")
rng = np.random.default_rng(1)
x0 = (0.25 * rng.standard_normal((H * N, 17, 3))).astype(np.float32)
T0 = np.tile(d["db_3d"][:, 0, :], (H, 1)).astype(np.float32)
dev = lambda a: torch.tensor(a, device="cuda")
W = zh.Weights(syn.make_weights(0)); s = zh.Schedule(W, O.oil_timestamps(S))
geom = zh.reproj_prepare(dev(d["db_2d"][:, :, :2].copy()), dev(d["camera_param"]), dev(d["db_2d"][:, :, 2].copy()))
x, T = dev(x0), dev(T0)
zh.oil_run(W, s, x, geom, T, 0, S, 2)
np.savez(sys.argv[1], x=x.cpu().numpy(), T=T.cpu().numpy(), ws=zh.workspace_bytes(H * N))
''' % (root, root)
    outs = []
    # "unfused": the reprojection correction as its own launch every iteration (ZEDO_UNFUSED_REPROJ) instead of
    # riding in the epilogue of the previous iteration's post_dense launch - the same arithmetic, bit for bit
    for tag, env in (("full", {}), ("chunk", {"ZEDO_CHUNK_ROWS": "512"}), ("unfused", {"ZEDO_UNFUSED_REPROJ": "1"}),
                     ("unfused_chunk", {"ZEDO_UNFUSED_REPROJ": "1", "ZEDO_CHUNK_ROWS": "320"})):
        out = str(tmp_path / f"{tag}.npz")
        subprocess.run([sys.executable, "-c", code, out], check=True, env={**os.environ, **env})
        outs.append(np.load(out))
    for o in outs[1:]:
        assert np.array_equal(outs[0]["x"], o["x"]) and np.array_equal(outs[0]["T"], o["T"])
    assert int(outs[1]["ws"]) == 512 * (64 + 2048) * 4 < int(outs[0]["ws"])


def test_both_math_modes_are_fp32_accurate_against_the_fp64_oracle(zh, weights0):
    """What "fp32-level accuracy" of ZEDO_MATH_F16X3 means, measured: the network output of BOTH arithmetic modes against
    the oracle evaluated in float64 (not against each other) on 3 000 rows at three noise levels.  The split-fp16 mode
    must be no farther from exact arithmetic than the exact-fp32-MFMA mode is (x 1.25 + 1e-8: its measured error is in
    fact the smaller one - 64 block sums instead of 1024 chained roundings), and both far inside the single-call
    tolerance of the parity tests (2e-6).  Also: the mode is a property of the handle, switchable back and forth, and
    switching back reproduces the first result bit for bit."""
    import zedo_oracle as O
    g = np.random.Generator(np.random.Philox(key=[31, 7]))
    x = (0.3 * g.standard_normal((3000, 17, 3))).astype(np.float32)
    ts = np.array([0.1, 0.05, 0.011], np.float32)
    w64 = O.cast_weights(weights0, np.float64)
    W = zh.Weights(weights0, math="f32")
    s = zh.Schedule(W, ts)
    assert W.math == "f32"
    err = {}
    first = None
    for mode in ("f32", "f16x3", "f32"):
        W.set_math(mode)
        worst, sq, n = 0.0, 0.0, 0
        for i, t in enumerate(ts):
            eps = zh.score_eps(W, s, i, dev(x)).cpu().numpy()
            if mode == "f32" and i == 0:
                if first is None:
                    first = eps.copy()
                else:
                    assert np.array_equal(first, eps)
            ref = O.score_model_forward(w64, x.astype(np.float64), np.float64(t) * 999.0, dtype=np.float64)
            d = np.abs(eps.astype(np.float64) - ref)
            worst, sq, n = max(worst, float(d.max())), sq + float((d * d).sum()), n + d.size
        err[mode] = (worst, float(np.sqrt(sq / n)))
    _report("math_modes_vs_fp64", [dict(mode=m, max_abs=e[0], rms=e[1]) for m, e in err.items()])
    assert err["f32"][0] <= 2e-6 and err["f16x3"][0] <= 2e-6, err
    assert err["f16x3"][1] <= 1.25 * err["f32"][1] + 1e-8, err
    with pytest.raises(zh.ZedoError):
        W.set_math("bf16")


def _adversarial_weights(weights0, case):
    """Networks built to sit on the weak spots of the split-fp16 arithmetic (include/zedo_hip.h ZEDO_MATH_F16X3)."""
    w = {k: v.copy() for k, v in weights0.items()}
    if case == "denormal_high_pieces":        # SiLU(-14) = -1.2e-5, SiLU(-9) = -1.1e-3: whole activations / their low pieces below fp16's 6.1e-5
        w["pre_gnorm.bias"][:] = -14.0
        w["b1_gnorm1.bias"][:] = -9.0
    elif case == "denormal_low_pieces":       # activations ~ 5e-3: the high piece is normal, the low piece (x 2^-11) is not
        w["pre_gnorm.weight"][:] = 0.01
        w["pre_gnorm.bias"][:] = 0.0
        w["b2_gnorm1.weight"][:] = 0.02
    elif case == "scale_boundary":            # max |w| exactly on / one ulp below / one ulp above a power of two: the per-matrix shift changes by one
        for name, target in (("b1_dense1.weight", np.float32(2.0 ** -5)), ("b1_dense2.weight", np.nextafter(np.float32(2.0 ** -5), np.float32(0))),
                             ("b2_dense1.weight", np.nextafter(np.float32(2.0 ** -4), np.float32(1))), ("post_dense.weight", np.float32(2.0 ** -7))):
            m = np.abs(w[name]).max()
            w[name] = (w[name] * (target / m)).astype(np.float32)
            i = np.unravel_index(np.abs(w[name]).argmax(), w[name].shape)
            w[name][i] = np.sign(w[name][i]) * target
            assert np.abs(w[name]).max() == target
    elif case == "gamma_near_refusal":        # activation bound 32 237 + 11 + 11 < 32 768: accepted, and activations of ~2e4 really occur
        w["pre_gnorm.weight"][:] = 5790.0
    elif case == "range_2^30_in_a_group":     # pre-activations of one GroupNorm group spread over 2^30
        b = w["pre_dense.bias"]
        b[96:128] = np.where(np.arange(32) % 2 == 0, 2.0 ** 15, 2.0 ** -15).astype(np.float32)
        b[512:544] = -(2.0 ** np.linspace(-15, 15, 32)).astype(np.float32)
    elif case == "row_ratio_2^-7.9":          # half the rows of a matrix 2^-7.9 below its maximum: the smallest ratio the mode accepts
        w["b2_dense1.weight"][::2] *= np.float32(2.0 ** -7.9)
    else:
        raise KeyError(case)
    return w


ADVERSARIAL = ["denormal_high_pieces", "denormal_low_pieces", "scale_boundary", "gamma_near_refusal", "range_2^30_in_a_group", "row_ratio_2^-7.9"]


@pytest.mark.parametrize("case", ADVERSARIAL)
def test_f16x3_on_adversarial_networks_against_the_fp64_oracle(zh, weights0, case):
    """the split-fp16 mode on networks built to hurt it - fp16-denormal activations and low pieces, weight maxima
    on the power-of-two boundary of the per-matrix scale, GroupNorm gains just under the refusal threshold, 2^30 of dynamic range
    inside one GroupNorm group, rows at the smallest accepted ratio to their matrix maximum - against the oracle in float64, beside
    the exact-fp32 mode on the same network.  Bound: per dense layer |dy| <= sum_k |w_k| (2^-22 |a_k| + 2^-24)
    + 64 fp32 block additions, against 1024 chained fp32 roundings for the fma chain - so the mode must stay within 1.5 x the
    exact-fp32 mode's distance from float64 (+ 3e-8 of the output scale for the absolute 2^-24 term), on every case."""
    import zedo_oracle as O
    w = _adversarial_weights(weights0, case)
    g = np.random.Generator(np.random.Philox(key=[47, 3]))
    x = (0.3 * g.standard_normal((1500, 17, 3))).astype(np.float32)
    ts = np.array([0.1, 0.02], np.float32)
    w64 = O.cast_weights(w, np.float64)
    W = zh.Weights(w, math="f32")
    s = zh.Schedule(W, ts)
    err, scale = {}, 0.0
    for mode in ("f32", "f16x3"):
        W.set_math(mode)
        worst, sq, n = 0.0, 0.0, 0
        for i, t in enumerate(ts):
            eps = zh.score_eps(W, s, i, dev(x)).cpu().numpy()
            assert np.isfinite(eps).all(), (case, mode)
            ref = O.score_model_forward(w64, x.astype(np.float64), np.float64(t) * 999.0, dtype=np.float64)
            scale = max(scale, float(np.abs(ref).max()))
            d = np.abs(eps.astype(np.float64) - ref)
            worst, sq, n = max(worst, float(d.max())), sq + float((d * d).sum()), n + d.size
        err[mode] = (worst, float(np.sqrt(sq / n)))
    _report("f16x3_adversarial_" + case, [dict(mode=m, max_abs=e[0], rms=e[1], output_scale=scale) for m, e in err.items()])
    assert err["f16x3"][0] <= 1.5 * err["f32"][0] + 3e-8 * max(1.0, scale), (case, err, scale)
    assert err["f16x3"][1] <= 1.5 * err["f32"][1] + 1e-8 * max(1.0, scale), (case, err, scale)


def test_f16x3_is_refused_for_rows_far_below_their_matrix_maximum(zh, weights0):
    """Each matrix carries ONE power-of-two scale in the split-fp16 mode: a non-zero row whose largest weight is below 2^-8 of the
    matrix maximum would lose bits in its small elements (low pieces fp16-denormal) - refused, like the overflow case; all-zero rows
    (the padding of pre / post_dense, a pruned channel) are fine; the exact-fp32 mode takes everything."""
    w = {k: v.copy() for k, v in weights0.items()}
    w["b2_dense1.weight"][5] *= np.float32(2.0 ** -8.6)
    W = zh.Weights(w, math="f32")
    with pytest.raises(zh.ZedoError, match="bad argument"):
        W.set_math("f16x3")
    assert W.math == "f32"
    w = {k: v.copy() for k, v in weights0.items()}
    w["b2_dense1.weight"][5] = 0.0
    w["post_dense.weight"][50] = 0.0
    assert zh.Weights(w, math="f32").set_math("f16x3").math == "f16x3"


def test_f16x3_is_refused_for_a_network_that_could_overflow_fp16(zh, weights0):
    """The split-fp16 mode stores activations as unscaled fp16 pieces (max 65504).  A network whose GroupNorm parameters
    allow activations near that range (bound: max|gamma| sqrt(31) + max|beta| along the residual path >= 32768) must be
    refused when the mode is requested - loudly, not with an inf three layers later; so must non-finite weights.  The
    exact-fp32 mode takes both."""
    big = {k: v.copy() for k, v in weights0.items()}
    big["b1_gnorm2.weight"][7] = 9000.0                    # 9000 * sqrt(31) = 5.0e4 on the residual path
    W = zh.Weights(big, math="f32")
    with pytest.raises(zh.ZedoError, match="bad argument"):
        W.set_math("f16x3")
    assert W.math == "f32"
    # non-finite parameters: +-inf AND NaN (std::fmax drops a NaN operand - ADVICE r3 - so finiteness is tracked on its own),
    # in a weight matrix, in the thin layers and in a GroupNorm parameter
    for name, idx, bad in (("b2_dense1.weight", (3, 5), np.inf), ("b2_dense1.weight", (3, 5), np.nan), ("pre_dense.weight", (0, 0), np.nan),
                           ("post_dense.weight", (50, 1023), np.nan), ("b1_gnorm1.weight", (17,), np.nan), ("pre_gnorm.bias", (1,), -np.inf)):
        badw = {k: v.copy() for k, v in weights0.items()}
        badw[name][idx] = bad
        W2 = zh.Weights(badw, math="f32")
        with pytest.raises(zh.ZedoError, match="bad argument"):
            W2.set_math("f16x3")
        assert W2.math == "f32"
    ok = zh.Weights(weights0, math="f32").set_math("f16x3")
    assert ok.math == "f16x3"
