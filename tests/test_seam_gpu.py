"""The seam launch (-m gpu): post_dense + SDE update + next reprojection of OIL iteration i and pre_dense + GroupNorm + SiLU of
iteration i + 1 in ONE kernel (csrc/zedo_gemm.hip::seam_kernel, zedo_gemm16.hip::seam16_kernel; reference
lib/algorithms/advanced/model.py:264-269,290-291 and run/opt_main.py:203-220) against the two separate launches
(ZEDO_NO_SEAM=1, read once per process): the same tile code, the same products in the same order - the pose state x and the
translation T after a run that crosses the switch to the least-squares T must agree BIT FOR BIT, in both arithmetic modes, for a
whole batch, for a batch with a ragged last tile, and through the row-chunk loop; and the seam must really have been taken
(sampled launch counts of the library's own profiler)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import hashlib, json, os, sys
import numpy as np, torch
root = %r
sys.path.insert(0, os.path.join(root, "zedo-release_amd")); sys.path.insert(0, os.path.join(root, "oracle"))
import zedo_hip as zh, zedo_oracle as O
from lib.dataset import synthetic as syn
out = {}
W = zh.Weights(syn.make_weights(0))
for tag, H, N, S, sw in (("whole_tiles", 16, 1024, 6, 2), ("ragged", 17, 1003, 5, 3), ("configs2", 50, 1015, 4, 1)):
    d = syn.make_poses(N, seed=4, conf_mode="uniform")
    rng = np.random.default_rng(1)
    x0 = (0.25 * rng.standard_normal((H * N, 17, 3))).astype(np.float32)
    T0 = np.tile(d["db_3d"][:, 0, :], (H, 1)).astype(np.float32)
    dev = lambda a: torch.tensor(a, device="cuda")
    s = zh.Schedule(W, O.oil_timestamps(1000)[100:100 + S].copy())
    geom = zh.reproj_prepare(dev(d["db_2d"][:, :, :2].copy()), dev(d["camera_param"]), dev(d["db_2d"][:, :, 2].copy()))
    x, T = dev(x0), dev(T0)
    zh.profile_start(sample_every=1, max_samples=64)
    zh.oil_run(W, s, x, geom, T, 0, S, sw)
    torch.cuda.synchronize()
    pr = zh.profile_stop()
    h = hashlib.sha256(); h.update(x.cpu().numpy().tobytes()); h.update(T.cpu().numpy().tobytes())
    out[tag] = dict(sha=h.hexdigest(), finite=bool(torch.isfinite(x).all()), launches={k: v["launches"] for k, v in pr.items()}, steps=S,
                    chunks=-(-H * N // int(os.environ.get("ZEDO_CHUNK_ROWS", 1 << 20))))
print("RESULT " + json.dumps(out))
''' % ROOT


def _run(env):
    e = dict(os.environ)
    e.pop("ZEDO_NO_SEAM", None)
    e.pop("ZEDO_CHUNK_ROWS", None)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CODE], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_the_seam_launch_is_bitwise_the_two_launches(math_mode):
    seam = _run({})
    two = _run({"ZEDO_NO_SEAM": "1"})
    chunked = _run({"ZEDO_CHUNK_ROWS": "16640"})          # 65 x 256 rows per chunk: whole seam-eligible chunks + a short last one
    for tag in seam:
        a, b, c = seam[tag], two[tag], chunked[tag]
        assert a["finite"] and a["sha"] == b["sha"] == c["sha"], (tag, a["sha"], b["sha"], c["sha"])
        S = a["steps"]
        # the seam really ran: per chunk ONE stand-alone pre_dense (first iteration), ONE stand-alone post_dense (last), S - 1 seams
        assert a["launches"]["seam_post_pre"] == S - 1 and a["launches"]["pre_dense"] == 1 and a["launches"]["post_dense_sde"] == 1, a["launches"]
        assert a["launches"]["hidden_dense"] == 4 * S and a["launches"]["reproj"] == 1
        # ... and ZEDO_NO_SEAM=1 really is the two-launch sequence
        assert b["launches"]["seam_post_pre"] == 0 and b["launches"]["pre_dense"] == S and b["launches"]["post_dense_sde"] == S, b["launches"]
        # chunks below the seam's minimum size (a short last chunk) fall back to two launches inside the same call
        assert c["launches"]["seam_post_pre"] + c["launches"]["post_dense_sde"] == c["chunks"] * S, c["launches"]
        assert c["launches"]["seam_post_pre"] >= (c["chunks"] - 1) * (S - 1)
