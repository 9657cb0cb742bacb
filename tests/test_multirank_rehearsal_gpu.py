"""N REAL ranks on ONE MI355X (-m gpu): `bench.py --gpus N` and the drivers as N fresh processes, N members of one process
group, every rank on its own contiguous row shard - the rehearsal of the 8-GPU form the driver's 8-GPU node runs.

RCCL refuses two ranks on one device, so the exchange step travels over the gloo REHEARSAL transport
(ZEDO_DIST_BACKEND=gloo: the same all-reduce / all-gather calls on host copies, zedo_hip/pipeline.py) and every rank
binds device 0 (ZEDO_SHARE_DEVICE=1).  Everything else is the product path: the launcher (parent builds / verifies the
library once, ranks get ZEDO_NO_BUILD=1), rank environments, shard_rows, the per-rank kernels with row_offset, the global
IPO normaliser, reduce_min_over_ranks / gather_row_shards, max-over-ranks timing, rank 0's JSON line.  What stays
RCCL-only: the transport itself (nccl process group, device-side collectives over xGMI) - covered with one rank in
tests/test_rccl_gpu.py.  The N-rank results must be bit-identical to the 1-rank run of the same global problem."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _one_arithmetic_mode(math_mode):
    """Ranks, shards, transport and launcher do not depend on the arithmetic of the dense layers: rehearsed in the default
    (exact fp32) mode only; the sharded arithmetic of the f16x3 mode is covered by the bitwise slice / virtual-rank tests."""
    if math_mode != "f32":
        pytest.skip("multi-rank rehearsal runs in the default arithmetic mode only")


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(share):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "ZEDO_FORCE_DIST", "ZEDO_BENCH_FORCE_DIST",
              "ZEDO_NO_BUILD", "ZEDO_SHARE_DEVICE", "ZEDO_DIST_BACKEND", "ZEDO_BENCH_FAIL_RANK", "ZEDO_BENCH_CORRUPT_RANK", "ZEDO_BENCH_CORRUPT_KIND"):
        e.pop(k, None)
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if share:
        e.update(ZEDO_SHARE_DEVICE="1", ZEDO_DIST_BACKEND="gloo")
    return e


def a_hypo(args):
    return int(args[args.index("--hypo") + 1])


def _bench(args, share, extra_env=None, ok=True):
    e = _env(share)
    e.update(extra_env or {})
    r = subprocess.run([sys.executable, BENCH] + args + ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-alt-mode",
                                                         "--strong-poses", "61", "--strong-steps", "2"],
                       env=e, cwd=ROOT, capture_output=True, text=True, timeout=900)
    if not ok:
        return r
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line, from rank 0"
    return json.loads(lines[0])


CASES = [
    # (id, N-rank arguments, 1-rank arguments of the same global problem, key that must agree)
    ("w2_weak", ["--poses", "8", "--hypo", "5", "--oil", "20"], ["--scaling", "strong", "--poses", "64", "--hypo", "5", "--oil", "20"], "selection_sha16"),
    ("w2_strong", ["--scaling", "strong", "--poses", "61", "--hypo", "5", "--oil", "20"], ["--scaling", "strong", "--poses", "61", "--hypo", "5", "--oil", "20"],
     "selection_sha16"),
    ("w3_h36m_actionwise", ["--workload", "3", "--poses", "60", "--hypo", "4", "--oil", "15"], ["--workload", "3", "--poses", "60", "--hypo", "4", "--oil", "15"],
     "selection_sha16"),
    ("w4_inference_gather", ["--workload", "4", "--poses", "45", "--hypo", "3", "--oil", "10"], ["--workload", "4", "--poses", "45", "--hypo", "3", "--oil", "10"],
     "results_sha16"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_eight_ranks_on_one_gpu_equal_the_one_rank_run(case):
    _, multi, single, key = case
    a = _bench(["--gpus", "8"] + multi, share=True)
    b = _bench(["--gpus", "1"] + single, share=False)
    assert a["n_gpus"] == a["rccl_ranks"] == 8 and a["dist_backend"] == "gloo" and a["devices_shared"] is True
    assert b["n_gpus"] == 1 and b["dist_backend"] is None
    assert a["config"]["poses_total"] == b["config"]["poses_total"]
    assert a["config"]["rows_per_gpu"] == -(-b["config"]["rows_per_gpu"] // 8)          # shard_rows: ceil(rows / 8) on rank 0
    assert a[key] == b[key], (key, a[key], b[key])
    if key == "selection_sha16":
        assert a["mpjpe_best_of_H_m"] == b["mpjpe_best_of_H_m"] and a["pa_mpjpe_best_of_H_m"] == b["pa_mpjpe_best_of_H_m"]
    assert a["value"] > 0 and a["ms_per_step"] > 0 and a["scaling"] in ("weak", "strong")
    # the run checked itself before it timed anything: 8 ranks == 1 rank on the live transport, every rank's own pass time reported
    # ... for BOTH exchange kinds, whatever the workload (round 6): the MIN selection and the all-gather
    sc = a["multi_rank_selfcheck"]
    assert sc["ok"] is True and sc["ranks"] == 8 and sc["backend"] == "gloo"
    for kind in ("selection", "gather"):
        assert sc[kind]["ok"] is True and sc[kind]["sha"] == sc[kind]["sha_unsharded"] and len(sc[kind]["sha"]) == 16
    assert sc["selection"]["sha"] != sc["gather"]["sha"]
    assert b["multi_rank_selfcheck"] is None and b["strong"] is None
    # ... and, beside a workload-2 headline, the `strong` object: one fixed problem on rank 0 alone against the same problem split over
    # the 8 ranks + the exchange, timed back to back, the two selections bit-identical
    st = a["strong"]
    if a["config"]["baseline_config"] == 2:
        assert st["scaling"] == "strong" and st["n_gpus"] == 8 and st["steps"] == 2 and st["rows_per_gpu"] == -(-61 * a_hypo(multi) // 8)
        assert st["matches_one_rank"] is True and st["selection_sha16"] == st["one_rank_selection_sha16"] and len(st["selection_sha16"]) == 16
        assert st["ms_per_step"] > 0 and st["one_rank"]["ms_per_step"] > 0 and st["speedup_vs_one_rank"] > 0 and st["efficiency"] > 0
        assert len(st["rank_pass_s"]["all"]) == 8 and st["efficiency_vs_projection"] is None        # non-stated size: nothing to project from
    else:
        assert st is None
    rp = a["rank_pass_s"]
    assert len(rp["all"]) == 8 and rp["min"] <= rp["max"] and rp["all"][rp["argmax"]] == rp["max"] and abs(rp["max"] * 1e3 - a["ms_per_step"]) < 0.02


def test_a_failing_rank_other_than_zero_fails_the_launcher():
    """Rank 5 of 8 exits with code 9 after its first pass (test hook ZEDO_BENCH_FAIL_RANK); the other seven are waiting in
    the exchange step - the launcher must end them and exit 9, with no JSON line."""
    r = _bench(["--gpus", "8", "--poses", "8", "--hypo", "3", "--oil", "10"], share=True, extra_env={"ZEDO_BENCH_FAIL_RANK": "5"}, ok=False)
    assert r.returncode == 9 and "rank 5 exited with code 9" in r.stderr, (r.returncode, r.stderr[-2000:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("workload,kind", [("2", None), ("4", None), ("2", "gather"), ("4", "selection")])
def test_a_corrupted_shard_fails_the_selfcheck_before_anything_is_timed(workload, kind):
    """ZEDO_BENCH_CORRUPT_RANK=3: rank 3 of 8 moves its shard by a millimetre before the exchange step of the self-check; the
    sharded digest then differs from rank 0's unsharded one, every rank exits with code 4 and no JSON line is printed - a
    wrong N-rank result cannot yield a number.  Both exchange kinds are checked in every run (round 6): corrupting ONLY the
    all-gather's input fails a selection workload, corrupting ONLY the selection's input fails the gather workload."""
    env = {"ZEDO_BENCH_CORRUPT_RANK": "3"}
    if kind:
        env["ZEDO_BENCH_CORRUPT_KIND"] = kind
    r = _bench(["--gpus", "8", "--workload", workload, "--poses", "8", "--hypo", "3", "--oil", "10"], share=True, extra_env=env, ok=False)
    assert r.returncode == 4 and "multi_rank_selfcheck FAILED" in r.stderr, (r.returncode, r.stderr[-2000:])
    for k in ("selection", "gather"):
        assert (f"FAILED ({k})" in r.stderr) == (kind in (None, k)), (k, kind, r.stderr[-2000:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_ranks_without_the_gloo_switch_are_refused_on_one_device():
    e = {"ZEDO_SHARE_DEVICE": "1"}
    r = _bench(["--gpus", "2", "--poses", "8", "--hypo", "2", "--oil", "5"], share=False, extra_env=e, ok=False)
    assert r.returncode != 0 and "ZEDO_DIST_BACKEND=gloo" in r.stderr


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _driver(nranks, module, args, tmp):
    """run.opt_main / run.inference under torchrun (the README's recipe): no launcher of ours in front, the ranks
    serialise on the build lock themselves."""
    e = _env(nranks > 1)
    e["PYTHONPATH"] = os.path.join(ROOT, "zedo-release_amd") + os.pathsep + e.get("PYTHONPATH", "")
    if nranks > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(_port()), "-m", module] + args
    else:
        cmd = [sys.executable, "-m", module] + args
    r = subprocess.run(cmd, env=e, cwd=str(tmp), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_drivers_under_torchrun_with_four_ranks_equal_the_one_rank_run(tmp_path):
    import hashlib
    import numpy as np
    cfg = lambda n: os.path.join(ROOT, "zedo-release_amd", "configs", "optim", f"concat_pose_optimization_{n}.py")
    import re
    # every rank prints the two means, and four ranks' lines may interleave on one pipe: collect (name, value) pairs
    pick = lambda out: sorted(set(re.findall(r"mean (PA-MPJPE|MPJPE) : ([0-9.eE+-]+?)(?=mean|\s|$)", out)))
    outs, shas = {}, {}
    for n in (1, 4):
        d = tmp_path / f"r{n}"
        d.mkdir()
        outs[n] = pick(_driver(n, "run.opt_main", ["--config", cfg("pw3d"), "--hypo", "5", "--synthetic", "37", "--oil_iterations", "20"], d))
        _driver(n, "run.inference", ["--config", cfg("wild"), "--hypo", "3", "--synthetic", "26", "--oil_iterations", "10", "--out", str(d / "results.npy")], d)
        res = np.load(d / "results.npy")
        assert res.shape == (26, 3, 17, 3) and np.isfinite(res).all()
        shas[n] = hashlib.sha256(res.tobytes()).hexdigest()
        # a sampler configuration OUTSIDE the fused pipeline (reverse-diffusion predictor; deterministic under the forced
        # probability flow): the step-wise loop, the ranks sharing its hypothesis loop (5 hypotheses over 4 ranks: 2 + 2 + 1 + 0),
        # the all-gather of uneven whole-hypothesis shards (gather_row_shards(lo=...))
        cfg2 = d / "cfg_rd.py"
        cfg2.write_text("import importlib.util\n"
                        f"_s = importlib.util.spec_from_file_location('base_cfg', r'{cfg('h36m')}')\n"
                        "_m = importlib.util.module_from_spec(_s); _s.loader.exec_module(_m)\n"
                        "def get_config():\n"
                        "    c = _m.get_config()\n"
                        "    c.sampling.predictor = 'reverse_diffusion'\n"
                        "    return c\n")
        _driver(n, "run.inference", ["--config", str(cfg2), "--hypo", "5", "--synthetic", "9", "--oil_iterations", "4", "--out", str(d / "results_rd.npy")], d)
        res2 = np.load(d / "results_rd.npy")
        assert res2.shape == (9, 5, 17, 3) and np.isfinite(res2).all()
        shas[(n, "rd")] = hashlib.sha256(res2.tobytes()).hexdigest()
    assert len(outs[1]) == 2 and outs[1] == outs[4], (outs[1], outs[4])          # the printed dataset means, digit for digit
    assert shas[1] == shas[4] and shas[(1, "rd")] == shas[(4, "rd")]
