"""CPU: `python bench.py --gpus N` is a real launcher (one fresh process per GPU, RCCL rendezvous on 127.0.0.1) and the
row arithmetic of the eval surface's device-side subsampling."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env_without_ranks():
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_dry_launch_describes_one_rank_per_gpu():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "3", "--dry-launch"], capture_output=True, text=True,
                       env=_env_without_ranks(), timeout=300)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["dry_launch"] and d["gpus"] == 4 and len(d["ranks"]) == 4
    assert "--dry-launch" not in d["cmd"] and d["cmd"][-4:] == ["--gpus", "4", "--steps", "3"]
    ports = {x["MASTER_PORT"] for x in d["ranks"]}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for i, x in enumerate(d["ranks"]):
        assert x["RANK"] == x["LOCAL_RANK"] == str(i) and x["WORLD_SIZE"] == "4" and x["MASTER_ADDR"] == "127.0.0.1"
        assert x["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launcher_refuses_more_ranks_than_gpus_instead_of_benchmarking_one():
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, env=_env_without_ranks(), timeout=300)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr and not r.stdout.strip()


def test_launcher_propagates_a_failing_rank():
    sys.path.insert(0, ROOT)
    import bench
    # two "ranks" that are plain python processes: rank 1 fails, the launcher must end rank 0 and return non-zero
    envs = bench.rank_environments(2, bench.free_port(), base={"PATH": os.environ.get("PATH", "")})
    assert [e["RANK"] for e in envs] == ["0", "1"] and all(e["WORLD_SIZE"] == "2" for e in envs)
    code = "import os,sys,time; r=int(os.environ['RANK']); time.sleep(0.3 if r else 30); sys.exit(7 if r else 0)"
    procs = [subprocess.Popen([sys.executable, "-c", code], env=e) for e in envs]
    rc = None
    import time
    t0 = time.time()
    while rc is None and time.time() - t0 < 20:
        for p in procs:
            c = p.poll()
            if c not in (None, 0):
                rc = c
        time.sleep(0.05)
    for p in procs:
        if p.poll() is None:
            p.terminate()
        p.wait(timeout=10)
    assert rc == 7


def test_workload_table_matches_the_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    b = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "1000 steps" in b["configs"][2] and "H=50" in b["configs"][2]
    assert bench.WORKLOADS[2]["poses"] == 1015 and bench.WORKLOADS[2]["default_scaling"] == "weak"
    assert bench.WORKLOADS[3]["poses"] == 567040 and bench.WORKLOADS[3]["weak"] * 8 == 567040
    assert "100k" in b["configs"][4] and bench.WORKLOADS[4]["poses"] == 100000 and bench.WORKLOADS[4]["weak"] * 8 == 100000


@pytest.mark.parametrize("N,H,k", [(10, 3, 3), (7, 2, 2), (9, 4, 1), (5, 3, 7)])
def test_device_side_subsampling_of_row_shards(N, H, k):
    """eval_multi(sample_interval=k) on ("rows", ...) shards == on the [N,H,J,3] layout (reference h36m.py:386-387): the kept
    rows of every contiguous shard are the contiguous shard [new_offset, ...) of the subsampled problem."""
    sys.path.insert(0, os.path.join(ROOT, "zedo-release_amd"))
    from lib.dataset._eval import subsample
    rows = torch.arange(H * N * 2 * 3, dtype=torch.float32).reshape(H * N, 2, 3)        # row = h*N + n
    gt = np.arange(N * 2 * 3, dtype=np.float64).reshape(N, 2, 3)
    dense = rows.reshape(H, N, 2, 3).permute(1, 0, 2, 3)                                   # [N,H,J,3]
    ref_p, ref_gt, _ = subsample(dense, gt, k)
    ref_rows = ref_p.permute(1, 0, 2, 3).reshape(-1, 2, 3)                                 # rows of the subsampled problem
    for lo in range(0, H * N):
        for cnt in (1, 4, H * N - lo):
            cnt = min(cnt, H * N - lo)
            (tag, got), g2, off = subsample(("rows", rows[lo:lo + cnt]), gt, k, lo)
            assert tag == "rows" and np.array_equal(g2, ref_gt)
            assert torch.equal(got, ref_rows[off:off + got.shape[0]]), (lo, cnt, off)
