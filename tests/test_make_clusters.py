"""tools/make_clusters.py (optional cluster builder; CPU): k-means over a pose set -> clusters/<name>_cluster{H}.npy in
the layout the drivers load (reference run/opt_main.py:58-65,167-168: np.load -> [H,17,3], root joint subtracted by the consumer,
float32 required by the bmm with the fp32 rotation)."""
import os
import pickle
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_clusters as mc  # noqa: E402


def _pose_set(H=6, per=400, seed=3, noise=0.01):
    g = np.random.Generator(np.random.Philox(key=[seed, 1]))
    protos = 0.25 * g.standard_normal((H, 17, 3))
    protos[:, 0] = 0.0
    sizes = [per + 37 * k for k in range(H)]                  # distinct sizes: the output order (largest first) is defined
    poses = np.concatenate([protos[k][None] + noise * g.standard_normal((sizes[k], 17, 3)) for k in range(H)])
    root = g.standard_normal((len(poses), 1, 3))              # arbitrary root positions: the tool clusters root-centred poses
    perm = g.permutation(len(poses))
    return protos, sizes, (poses + root)[perm]


def test_kmeans_recovers_separated_modes_in_the_consumers_layout(tmp_path):
    protos, sizes, poses = _pose_set()
    src = tmp_path / "poses.npy"
    np.save(src, poses.astype(np.float32))
    path = mc.main([str(src), "--hypo", "6", "--name", "h36m", "--out-dir", str(tmp_path / "clusters")])
    sys.path.insert(0, os.path.join(ROOT, "zedo-release_amd"))
    from run._driver import cluster_file
    assert os.path.relpath(path, tmp_path) == cluster_file("h36m", 6) == cluster_file("3dpw", 6) == cluster_file("wild", 6)
    c = np.load(path)
    assert c.shape == (6, 17, 3) and c.dtype == np.float32 and np.isfinite(c).all()
    assert np.abs(c[:, 0]).max() == 0.0                                    # root-centred (the consumer subtracts joint 0 anyway)
    # largest cluster first, every prototype recovered to the noise of a mean over >= 400 members
    order = np.argsort(-np.array(sizes), kind="stable")
    want = protos[order] - protos[order][:, 0:1]
    assert np.abs(c - want).max() < 0.01 * 4 / np.sqrt(400) + 2e-3, np.abs(c - want).max()
    # deterministic; another seed finds the same partition here
    again, _, _ = mc.build(mc.load_poses(str(src)), 6, seed=0)
    assert np.array_equal(again, c)
    other, _, _ = mc.build(mc.load_poses(str(src)), 6, seed=5)
    assert np.abs(other - c).max() < 1e-5


def test_medoids_are_real_poses_and_inputs_in_the_assets_formats(tmp_path):
    protos, sizes, poses = _pose_set(H=4, per=150, seed=9)
    centred = poses - poses[:, 0:1]
    med, counts, _ = mc.build(poses, 4, seed=1, medoid=True)
    assert med.dtype == np.float32 and sorted(counts.tolist(), reverse=True) == counts.tolist() and counts.sum() == len(poses)
    for k in range(4):                                                     # every medoid is one of the input poses (root-centred)
        assert np.abs(centred - med[k][None].astype(np.float64)).reshape(len(poses), -1).max(1).min() < 1e-6
    # the H36M test-set layout: a pickled list of records with joint_3d_camera in millimetres (reference h36m.py:206-263)
    pkl = tmp_path / "h36m_like.pkl"
    with open(pkl, "wb") as f:
        pickle.dump([{"joint_3d_camera": (p * 1000.0).astype(np.float32), "action": 2} for p in poses], f)
    from_pkl, _, _ = mc.build(mc.load_poses(str(pkl)), 4, seed=1, medoid=True)
    assert np.abs(from_pkl - med).max() < 1e-6
    # an .npz with several arrays: the [M,17,3] one is found, or named
    npz = tmp_path / "set.npz"
    np.savez(npz, imgname=np.arange(3), joints=poses)
    assert np.array_equal(mc.build(mc.load_poses(str(npz)), 4, seed=1, medoid=True)[0], med)
    assert np.array_equal(mc.build(mc.load_poses(str(npz), key="joints"), 4, seed=1, medoid=True)[0], med)
    with pytest.raises(SystemExit):
        mc.build(poses[:3], 4)
    with pytest.raises(SystemExit):
        mc.load_poses(str(tmp_path / "nothing.txt"))


def test_more_clusters_than_modes_and_duplicate_poses_do_not_produce_empty_or_nan_clusters():
    g = np.random.Generator(np.random.Philox(key=[4, 4]))
    poses = np.repeat(0.2 * g.standard_normal((5, 17, 3)), 40, axis=0)    # five distinct poses, 40 copies each
    out, counts, inertia = mc.build(poses, 5, seed=2)
    assert np.isfinite(out).all() and (counts > 0).all() and inertia < 1e-10      # (the expanded-square distance cancels to ~1e-15)
    out8, counts8, _ = mc.build(poses + 1e-4 * g.standard_normal(poses.shape), 8, seed=2)
    assert np.isfinite(out8).all() and (counts8 > 0).all() and len(np.unique(out8.reshape(8, -1), axis=0)) == 8
