"""A third fp32 implementation against the reference's IPO end-state ensembles (CPU only; fixtures from tools/gen_oracle_ipoens.py).

tests/test_ensemble_gpu.py::test_ipo_end_state_distribution_matches_the_references holds the HIP kernels' IPO end state (500 Adam
iterations on an L1 loss, reference run/opt_main.py:180-195, simple_zeroshot_opt.py:8-31) against the reference's as distributions:
201-point quantile functions of rotation angle, depth scale and end-state loss over the 50 750 fits of BASELINE configs[2], ensemble
mean against ensemble mean, and measures 3.6-6.2 pooled single-member standard deviations at the worst interior quantile (asserted:
<= 8).  Is that distance a property of the kernels, or of ANY other fp32 implementation of this chaotic fit?  The numpy oracle shares
no code with either side (its own pairwise sums, its own operation order), so its ensembles answer it: 12 oracle members per draw,
summarised by the same tests/_ipo_summary.py, against the same 16 / 8 / 8 reference members, by the same statistic and the same bars.

Measured (round 5): worst quantile oracle-vs-reference 4.1 / 3.7 / 4.5 (angle), 1.3 / 1.0 / 1.4 (scale), 6.9 / 5.3 / 5.5 (loss) member
standard deviations on draws A / b / c - against 5.5 / 4.2 / 3.6, 0.9 / 0.9 / 1.6, 6.2 / 5.5 / 5.5 for the HIP kernels: the same band.
The median quantile sits at ~1 sd for both.  The distance is what two fp32 summation orders are worth inside 500 x 17 operations of
a fit; the kernels are not farther from the reference than an independent reimplementation is."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
DRAWS = ["driver_pw3d_full", "driver_pw3d_full_b", "driver_pw3d_full_c"]


def distance(ref, other):
    """The statistic of the GPU test: per interior quantile |mean_ref - mean_other| in pooled single-member standard deviations."""
    out = {}
    sl = slice(2, -2)
    for k, floor in (("q_angle", 2e-4), ("q_scale", 1e-5), ("q_loss", 2e-4)):
        a, b = ref[k], other[k]
        d = np.abs(a.mean(0) - b.mean(0))[sl]
        sd = np.maximum(np.sqrt((a.var(0, ddof=1) + b.var(0, ddof=1)) / 2)[sl], floor)
        out[k] = dict(max_abs_diff=float(d.max()), max_in_member_sd=float((d / sd).max()), median_in_member_sd=float(np.median(d / sd)))
    return out


@pytest.mark.parametrize("name", DRAWS)
def test_the_numpy_oracle_sits_in_the_same_band_as_the_kernels(name):
    ref = np.load(os.path.join(GOLDEN, name + "_ipoens.npz"))
    orc = np.load(os.path.join(GOLDEN, name + "_ipoens_oracle.npz"))
    assert str(ref["inputs_sha"]) == str(orc["inputs_sha"]) and orc["q_loss"].shape[0] >= 12
    assert not set(int(m) for m in orc["members"]) & set(range(1, 200))           # perturbation streams disjoint from the reference's and the kernels'
    dist = distance(ref, orc)
    rec = {"test": "ipo_end_state_distribution_oracle", "capture": name, "members_oracle": int(orc["q_loss"].shape[0]),
           "members_ref": int(ref["q_loss"].shape[0]), **dist,
           "mean_loss_px": dict(ref=float(ref["mean_loss"].mean()), oracle=float(orc["mean_loss"].mean()),
                                diff=float(ref["mean_loss"].mean() - orc["mean_loss"].mean()))}
    # the record goes to the parity report only when asked for (ZEDO_PARITY_REPORT=<path>): a plain CPU pytest run leaves the tree alone;
    # how the oracle's distance compares with the kernels' is a report (tools/parity_summary.py reads both records), not an assertion
    # about committed fixtures
    if os.environ.get("ZEDO_PARITY_REPORT"):
        with open(os.environ["ZEDO_PARITY_REPORT"], "a") as f:
            f.write(json.dumps(rec) + "\n")
    # the kernels' own bars (tests/test_ensemble_gpu.py), applied to the oracle: an independent fp32 implementation passes them too
    for k, abs_tol in (("q_angle", 1.2e-2), ("q_scale", 3e-3), ("q_loss", 3e-2)):
        assert dist[k]["max_in_member_sd"] <= 8.0 and dist[k]["max_abs_diff"] <= abs_tol, (k, dist[k])
        assert dist[k]["median_in_member_sd"] <= 1.5, (k, dist[k])
    assert abs(rec["mean_loss_px"]["diff"]) <= 3e-3
