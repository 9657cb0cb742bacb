"""SURVEY.md section 7, parity stage (A), at BASELINE size: the OIL loop given the REFERENCE's own IPO output.

The end-to-end comparison at configs[2] size (tests/test_surface_gpu.py::test_fused_driver_at_baseline_size_matches_
reference) runs IPO + OIL together, and IPO - 500 Adam iterations at lr 0.1 on an L1 loss - does not converge: its last
iterate differs between any two fp32 implementations by a median 0.01 rad, which moves the unaligned MPJPE by more than
the 0.05 mm bar and drowns the comparison of the 1000-step loop.  Here the loop is fed (R, T) exactly as the
reference's run produced them (tests/golden/<capture>_ipo.npz: float32, bit for bit, tools/gen_golden.py::
_driver_ipo_pin) through zedo_rotate_init + zedo_oil_run, all 50 750 rows x 1000 steps, and the dataset means are held to
the north-star bar DIRECTLY: |MPJPE - reference| <= 0.05 mm and |PA-MPJPE - reference| <= 0.05 mm, no standard-error
escape (reference loop: run/opt_main.py:197-222).

Round 5: the dataset means are not the only assertion any more.  A mean over 50 750 rows forgives what moves rows in both
directions - the mutation table showed the conf^4 -> conf^2 mutant of the least-squares weight (simple_zeroshot_opt.py:73-93)
passing the means on the draw with uniform confidences.  tests/golden/<capture>_oil64.npz holds the reference's loop re-run in
FLOAT64 from the same (R, T); every row's error is held against it: the HIP loop's distance from exact arithmetic must be no
larger than the reference's own fp32 loop's - median <= 1.0 x, p90 and p99 <= 1.25 x - on every capture and both protocols."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAR_MM = 0.05

CAPTURES = [n for n in ("driver_h36m_full", "driver_pw3d_full", "driver_pw3d_full_b", "driver_pw3d_full_c")
            if os.path.exists(os.path.join(ROOT, "tests", "golden", n + "_ipo.npz"))]


def _sha(*arrs):
    import hashlib
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run_oil_from_pins(weights0, g, pin, math_mode=None):
    """-> (x rows [H*N,17,3] on the GPU, problem dict).  (R, T) of every row come from the fixture, nothing from
    zedo_ipo_fit."""
    import zedo_hip as zh
    from zedo_hip.pipeline import Pipeline, ZeDOConfig
    from lib.dataset import synthetic as syn
    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    h36m = str(g["dataset"]) == "h36m"
    d = syn.make_poses(N, seed=int(g["seed_pose"]), conf_mode=str(g["conf_mode"]), dtype3d=np.float64 if h36m else np.float32)
    cl = syn.make_clusters(H, seed=int(g["seed_cl"]))
    assert _sha(d["db_2d"], d["camera_param"], cl) == str(g["inputs_sha"]) == str(pin["inputs_sha"])
    cfg = ZeDOConfig(IPO_keylist=[int(k) for k in g["keylist"]], IPO_T=float(g["ipo_T"]), IPO_minScaleT=float(g["minT"]),
                     OIL_iterations=S)
    pipe = Pipeline(weights0, cfg, "cuda").load(cl, d["db_2d"], d["camera_param"])
    cs = pin["cs"].reshape(H * N, 2)
    R = np.zeros((H * N, 3, 3), np.float32)
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = cs[:, 0], -cs[:, 1], cs[:, 1], cs[:, 0], 1.0
    Rd = torch.tensor(R, device="cuda")
    T = torch.tensor(np.ascontiguousarray(pin["T"].reshape(H * N, 3)), device="cuda")
    x = zh.rotate_init(pipe.x0, Rd, N)                                     # opt_main.py:201
    zh.oil_run(pipe.weights, pipe.sched, x, pipe.geom, T, 0, S, S // 5)    # opt_main.py:202-220
    return x, dict(N=N, H=H, S=S, h36m=h36m, d=d)


@pytest.mark.parametrize("name", CAPTURES)
def test_oil_loop_from_the_reference_ipo_output_meets_the_bar(weights0, golden, name):
    import zedo_hip as zh
    from lib.dataset.h36m import H36MDataset3D
    from lib.dataset.pw3d import PW3D
    g, pin = golden(name), golden(name + "_ipo")
    x, P = run_oil_from_pins(weights0, g, pin)
    N, H, d = P["N"], P["H"], P["d"]
    assert x.shape == (H * N, 17, 3) and bool(torch.isfinite(x).all())
    if P["h36m"]:
        ds = H36MDataset3D.from_arrays(d["db_2d"], d["db_3d"] * 1000.0, d["camera_param"], 2 + (np.arange(N) % 15))
        gtc = (d["db_3d"] * 1000.0 - (d["db_3d"] * 1000.0)[:, 0:1]) / 1000.0
    else:
        ds = PW3D.from_arrays(d["db_2d"], d["db_3d"], d["camera_param"])
        gtc = (d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float64)
    p1 = ds.eval_multi(("rows", x), protocol2=False)
    p2 = ds.eval_multi(("rows", x), protocol2=True)
    rep = {"test": "stage_a:" + name, "N": N, "H": H, "S": P["S"], "mpjpe_hip": p1, "mpjpe_ref": float(g["mpjpe"]),
           "pa_hip": p2, "pa_ref": float(g["pa_mpjpe"]), "d_mpjpe_mm": (p1 - float(g["mpjpe"])) * 1e3,
           "d_pa_mpjpe_mm": (p2 - float(g["pa_mpjpe"])) * 1e3}
    gt = torch.as_tensor(gtc, device="cuda")
    assert os.path.exists(os.path.join(ROOT, "tests", "golden", name + "_oil64.npz")), "the float64-loop arbiter of this capture is missing"
    arb = golden(name + "_oil64")
    for key, proto in (("p1", False), ("p2", True)):
        err, best, idx = zh.min_mpjpe(x, gt, N, procrustes=proto)
        e = err.reshape(H, N).T.cpu().numpy()                               # [N, H]
        de = np.abs(e - g[f"err_{key}"].astype(np.float64)) * 1e3           # every (pose, hypothesis), mm
        db = (best.cpu().numpy() - g[f"best_{key}"]) * 1e3
        rep[key] = dict(argmin_agreement=float((idx.cpu().numpy() == g[f"argmin_{key}"]).mean()),
                        all_rows_abs_delta_mm=dict(median=float(np.median(de)), p90=float(np.percentile(de, 90)),
                                                   p99=float(np.percentile(de, 99)), max=float(de.max())),
                        best_delta_mm=dict(mean=float(db.mean()), abs_median=float(np.median(np.abs(db))),
                                           abs_p99=float(np.percentile(np.abs(db), 99)), abs_max=float(np.abs(db).max())))
        if arb is not None:      # the reference's loop in float64 from the same (R, T): whose fp32 loop is closer to it?
            assert str(arb["inputs_sha"]) == str(g["inputs_sha"])
            a = arb[f"err_{key}"].astype(np.float64)
            dh, dr = np.abs(e - a) * 1e3, np.abs(g[f"err_{key}"].astype(np.float64) - a) * 1e3
            rep[key]["vs_fp64_loop_mm"] = {
                "hip": dict(median=float(np.median(dh)), p90=float(np.percentile(dh, 90)), p99=float(np.percentile(dh, 99))),
                "reference_fp32": dict(median=float(np.median(dr)), p90=float(np.percentile(dr, 90)), p99=float(np.percentile(dr, 99))),
                "dataset_mean_hip": float((best.cpu().numpy().mean() - float(arb["mpjpe" if key == "p1" else "pa_mpjpe"])) * 1e3),
                "dataset_mean_reference_fp32": float((float(g["mpjpe" if key == "p1" else "pa_mpjpe"]) -
                                                      float(arb["mpjpe" if key == "p1" else "pa_mpjpe"])) * 1e3)}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.jsonl", "a") as f:
        f.write(json.dumps(rep) + "\n")
    print(json.dumps(rep))
    assert abs(rep["d_mpjpe_mm"]) <= BAR_MM, rep
    assert abs(rep["d_pa_mpjpe_mm"]) <= BAR_MM, rep
    # per row, against the float64 loop: not farther from exact arithmetic than the reference's own fp32 loop
    # (measured round 4: median 0.76-0.85 x, p90 0.90-1.12 x, p99 0.84-1.22 x for MPJPE; 0.43-0.47 x throughout for PA-MPJPE)
    for key in ("p1", "p2"):
        v = rep[key]["vs_fp64_loop_mm"]
        h, r = v["hip"], v["reference_fp32"]
        assert h["median"] <= 1.0 * r["median"], (name, key, h, r)
        assert h["p90"] <= 1.25 * r["p90"] and h["p99"] <= 1.25 * r["p99"], (name, key, h, r)
