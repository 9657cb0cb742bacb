"""RCCL on the GPU with ONE rank (-m gpu): every collective of the sharded path - the process group over
backend "nccl" (= RCCL on ROCm), the fp64 MIN and int64 MIN all-reduces of reduce_min_over_ranks, the all-gather
of run.inference, the barrier / MAX of bench.py - executed on an MI355X in fresh child processes with
RANK=0 WORLD_SIZE=1 ZEDO_FORCE_DIST=1, and compared bit for bit with the same run without a process group.
The multi-rank arithmetic itself is covered on CPU over gloo (tests/test_distributed_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(dist):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ZEDO_FORCE_DIST", "ZEDO_BENCH_FORCE_DIST"):
        e.pop(k, None)
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if dist:
        e.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", ZEDO_FORCE_DIST="1", MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(_free_port()))
    return e


def _run(cmd, dist, cwd=ROOT):
    r = subprocess.run(cmd, env=_env(dist), cwd=cwd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_bench_exchange_step_on_rccl_is_bit_identical():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--oil", "20", "--poses", "64",
           "--hypo", "5", "--no-cpu-baseline", "--strong-poses", "96", "--strong-steps", "2"]
    lines = {}
    for dist in (False, True):
        out = _run(cmd, dist)
        lines[dist] = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    a, b = lines[False], lines[True]
    # the bench contract: one JSON line with the driver's keys, the roofline object and the cpu_baseline slot
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                   ("config", dict), ("roofline", dict)):
        assert isinstance(a[k], typ), (k, a[k])
    assert a["vs_baseline"] is None and "cpu_baseline" in a and a["scaling"] == "weak" and a["dtype"] == "f32"
    assert set(a["config"]) >= {"workload"} and not ({"model", "global_batch", "seq_len"} & set(a["config"]))
    r = a["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert set(r) >= {"traffic", "measured_live", "replayed", "box_mfma_peak_measured", "frac_of_box_peak"}
    assert a["n_gpus"] == b["n_gpus"] == 1
    # the line checks itself: per-class kernel times, less the calibrated cost of an empty event bracket,
    # over all launches of an OIL iteration cannot exceed the wall time of an iteration; the dominant kernel cannot beat what
    # this box's matrix pipe sustains
    for l in (a, b):
        assert 0 < l["event_bracket_ms"] < 0.02, l["event_bracket_ms"]
        assert 0 < l["sum_kernel_ms_per_oil_step"] <= l["wall_ms_per_oil_step"], (l["sum_kernel_ms_per_oil_step"], l["wall_ms_per_oil_step"])
        rf = l["roofline"]
        assert rf["flop_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e12 <= rf["box_mfma_peak_measured"]
        assert rf["avg_launch_ms"] <= rf["avg_launch_ms_bracketed"]
    assert a["selection_sha16"] == b["selection_sha16"], (a["selection_sha16"], b["selection_sha16"])
    assert a["mpjpe_best_of_H_m"] == b["mpjpe_best_of_H_m"] and a["pa_mpjpe_best_of_H_m"] == b["pa_mpjpe_best_of_H_m"]
    # a run with a process group (here: RCCL, one member) carries the self-check of BOTH exchange kinds and the `strong` object
    assert a["multi_rank_selfcheck"] is None and a["strong"] is None
    sc = b["multi_rank_selfcheck"]
    assert sc["ok"] and sc["backend"] == "nccl" and sc["ranks"] == 1
    for kind in ("selection", "gather"):
        assert sc[kind]["ok"] and sc[kind]["sha"] == sc[kind]["sha_unsharded"] and len(sc[kind]["sha"]) == 16
    st = b["strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 1 and st["matches_one_rank"] and st["selection_sha16"] == st["one_rank_selection_sha16"]
    assert st["ms_per_step"] > 0 and st["one_rank"]["ms_per_step"] > 0 and 0.2 < st["speedup_vs_one_rank"] < 5.0      # one rank against itself on a ~10 ms problem: a sanity check, not a measurement
    assert st["alt_mode"]["math"] == "f16x3" and st["alt_mode"]["matches_one_rank"]


DRIVER = r'''
import hashlib, json, os, sys
import numpy as np
root = %r
sys.path.insert(0, os.path.join(root, "zedo-release_amd"))
import torch
import run.opt_main as om, run.inference as inf
cfg = lambda n: os.path.join(root, "zedo-release_amd", "configs", "optim", "concat_pose_optimization_%%s.py" %% n)
p1, p2 = om.main(om.parse_args(["prog", "--config", cfg("pw3d"), "--hypo", "3", "--synthetic", "40", "--oil_iterations", "20"]))
if "MASTER_PORT" in os.environ:            # second process group of this process: fresh rendezvous port
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s_.getsockname()[1]); s_.close()
out = os.path.join(sys.argv[1], "results.npy")
res, errs = inf.main(inf.parse_args(["prog", "--config", cfg("h36m"), "--hypo", "2", "--synthetic", "30", "--oil_iterations", "10",
                                     "--eval", "--out", out]))
# a sampler configuration OUTSIDE the fused pipeline (reverse-diffusion predictor): the step-wise loop, ranks sharing its
# hypothesis loop, the all-gather of whole-hypothesis shards (gather_row_shards(lo=...))
cfg2 = os.path.join(sys.argv[1], "cfg_rd.py")
open(cfg2, "w").write(
    "import importlib.util\n"
    "_s = importlib.util.spec_from_file_location('base_cfg', r'%%s')\n"
    "_m = importlib.util.module_from_spec(_s); _s.loader.exec_module(_m)\n"
    "def get_config():\n"
    "    c = _m.get_config()\n"
    "    c.sampling.predictor = 'reverse_diffusion'\n"
    "    return c\n" %% cfg("h36m"))
if "MASTER_PORT" in os.environ:
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s_.getsockname()[1]); s_.close()
out2 = os.path.join(sys.argv[1], "results_rd.npy")
torch.manual_seed(0)
res2, _ = inf.main(inf.parse_args(["prog", "--config", cfg2, "--hypo", "3", "--synthetic", "12", "--oil_iterations", "5", "--out", out2]))
import torch.distributed as dist
print("RESULT " + json.dumps(dict(opt=[repr(p1), repr(p2)], inf=[repr(errs[0]), repr(errs[1])],
                                  sha=hashlib.sha256(np.load(out).tobytes()).hexdigest(), shape=list(res.shape),
                                  sha_stepwise=hashlib.sha256(np.load(out2).tobytes()).hexdigest(), shape_stepwise=list(res2.shape),
                                  group_was_used=os.environ.get("ZEDO_FORCE_DIST") == "1")))
'''


def test_drivers_on_rccl_are_bit_identical(tmp_path):
    """run.opt_main --synthetic and run.inference --synthetic --eval through init_process_group("nccl"): the
    sharded eval_multi (two MIN all-reduces per protocol) and the all-gather of the inference result."""
    res = {}
    for dist in (False, True):
        d = tmp_path / ("dist" if dist else "plain")
        d.mkdir()
        out = _run([sys.executable, "-c", DRIVER % ROOT, str(d)], dist)
        res[dist] = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res[True]["group_was_used"] and not res[False]["group_was_used"]
    for k in ("opt", "inf", "sha", "shape", "sha_stepwise", "shape_stepwise"):
        assert res[False][k] == res[True][k], (k, res[False][k], res[True][k])
    assert res[True]["shape"] == [30, 2, 17, 3] and res[True]["shape_stepwise"] == [12, 3, 17, 3]


INFER_SHARD = r'''
import hashlib, json, os, sys
import numpy as np
root = %r
sys.path.insert(0, os.path.join(root, "zedo-release_amd"))
import torch
import run.inference as inf
cfg = os.path.join(root, "zedo-release_amd", "configs", "optim", "concat_pose_optimization_wild.py")
out = os.path.join(sys.argv[1], "results.npy")
res, errs = inf.main(inf.parse_args(["prog", "--config", cfg, "--hypo", "50", "--synthetic", "12500", "--oil_iterations", "3", "--out", out]))
r = np.load(out, mmap_mode="r")
print("RESULT " + json.dumps(dict(shape=list(r.shape), finite=bool(np.isfinite(res).all()), errs=errs is None,
                                  sha=hashlib.sha256(np.ascontiguousarray(r[::97]).tobytes()).hexdigest(),
                                  group_was_used=os.environ.get("ZEDO_FORCE_DIST") == "1")))
'''


def test_inference_driver_at_the_config4_shard_size_through_the_all_gather(tmp_path):
    """BASELINE configs[4] as ONE GPU of eight sees it: run.inference on 12 500 synthetic detections x H = 50 =
    625 000 rows (run/inference.py:175-236; three OIL steps - the loop rate does not depend on S) through the
    driver itself, with and without the process group: under ZEDO_FORCE_DIST=1 the 625 000-row result travels through
    gather_row_shards' RCCL all-gather before results.npy is written; both runs must write the same file."""
    res = {}
    for dist in (False, True):
        d = tmp_path / ("dist" if dist else "plain")
        d.mkdir()
        out = _run([sys.executable, "-c", INFER_SHARD % ROOT, str(d)], dist)
        res[dist] = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res[True]["group_was_used"] and not res[False]["group_was_used"]
    assert res[True]["shape"] == [12500, 50, 17, 3] and res[True]["finite"] and res[True]["errs"]
    assert res[False]["sha"] == res[True]["sha"]


def test_bench_launcher_starts_its_rank_on_the_gpu():
    """`python bench.py --gpus N` without torchrun: the parent (no GPU call) starts the ranks as fresh processes.  With one GPU
    in the box the launcher is forced on for N = 1 (ZEDO_BENCH_FORCE_LAUNCH=1) and the rank takes the RCCL path
    (ZEDO_FORCE_DIST=1): the child initialises the process group from the environment the launcher built, runs, and rank 0's
    JSON line comes back through the parent, which exits 0."""
    e = _env(False)
    e.update(ZEDO_BENCH_FORCE_LAUNCH="1", ZEDO_FORCE_DIST="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--oil", "10", "--poses", "64",
           "--hypo", "3", "--no-cpu-baseline", "--no-alt-mode"]
    r = subprocess.run(cmd, env=e, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["config"]["rows_per_gpu"] == 64 * 3
    # a rank that fails makes the launcher fail: an impossible workload (0 OIL steps is rejected by the schedule)
    bad = subprocess.run(cmd[:-2] + ["--oil", "0"], env=e, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert bad.returncode != 0 and "rank 0 exited" in bad.stderr
