#!/usr/bin/env python3
"""Build a hypothesis cluster file for the ZeDO path from a 3D pose set ("optional k-means cluster builder";
the reference ships NO builder: its `clusters/<name>_cluster{H}.npy` files come from a Google-Drive folder, Readme.md:134-157).

The consumer (reference run/opt_main.py:58-65,167-168; here run/_driver.py::cluster_file) does

    sample_poses = np.load(f"clusters/{name}_cluster{H}.npy")                     # [H, 17, 3]
    x0[h] = torch.tensor(sample_poses - sample_poses[:, 0:1, :])[h]               # root-centred, must be float32 (bmm with fp32 R)

so the file is H poses x 17 joints x 3, float32, metres, in the dataset's joint order; the root joint need not be at the origin (the
consumer subtracts it) - this tool writes root-centred poses anyway.

Method: k-means (k-means++ seeding, Lloyd iterations, numpy, deterministic for a given --seed) over the ROOT-CENTRED poses flattened
to 51 values; empty clusters are re-seeded with the pose farthest from its centre.  --medoid writes, for every cluster, the real pose
nearest to the centre instead of the mean (a mean pose has slightly shortened limbs).  Clusters are ordered by size, largest first,
so that --hypo 1 style prefixes are the most populated modes.

Inputs: .npy [M,17,3]; .npz (key --key, default: the first [M,17,3] array); .pkl in the H36M test-set layout (list of dicts with
`joint_3d_camera` [17,3] in millimetres, reference lib/dataset/h36m.py:206-263) - converted to metres.

    python tools/make_clusters.py data/h36m/h36m_train.pkl --hypo 50 --name h36m          ->  clusters/h36m_cluster50.npy
"""
import argparse
import os
import pickle
import sys

import numpy as np


def load_poses(path, key=None, unit=None):
    """-> float64 [M,17,3] in metres."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        p = np.load(path)
    elif ext == ".npz":
        z = np.load(path, allow_pickle=True)
        if key is None:
            key = next((k for k in z.files if getattr(z[k], "ndim", 0) == 3 and z[k].shape[1:] == (17, 3)), None)
            if key is None:
                raise SystemExit(f"{path}: no [M,17,3] array (keys: {z.files}); pass --key")
        p = z[key]
    elif ext in (".pkl", ".pickle"):
        with open(path, "rb") as f:
            db = pickle.load(f)
        if isinstance(db, dict):
            db = db.get(key or "joint_3d_camera", db)
        if isinstance(db, (list, tuple)) and db and isinstance(db[0], dict):
            p = np.stack([np.asarray(it[key or "joint_3d_camera"], dtype=np.float64) for it in db])
            unit = unit or "mm"
        else:
            p = np.asarray(db)
    else:
        raise SystemExit(f"{path}: expected .npy, .npz or .pkl")
    p = np.asarray(p, dtype=np.float64)
    if p.ndim != 3 or p.shape[1:] != (17, 3):
        raise SystemExit(f"{path}: poses must be [M,17,3], got {p.shape}")
    if (unit or "m") == "mm":
        p = p / 1000.0
    if not np.isfinite(p).all():
        raise SystemExit(f"{path}: non-finite coordinates")
    return p


def _sqdist(X, C, chunk=65536):
    """[M,K] squared distances, chunked over M (a 1.5 M-pose training set x 50 centres stays below 1 GB)."""
    out = np.empty((X.shape[0], C.shape[0]), dtype=np.float64)
    cc = (C * C).sum(1)
    for lo in range(0, X.shape[0], chunk):
        x = X[lo:lo + chunk]
        out[lo:lo + chunk] = np.maximum((x * x).sum(1)[:, None] - 2.0 * (x @ C.T) + cc[None, :], 0.0)
    return out


def kmeans(X, K, seed=0, iters=100, tol=1e-7):
    """k-means++ seeding + Lloyd.  X [M,D] float64 -> (centres [K,D], labels [M], inertia)."""
    M = X.shape[0]
    if K > M:
        raise SystemExit(f"{K} clusters asked for, {M} poses given")
    rng = np.random.Generator(np.random.Philox(key=[int(seed), 2026]))
    C = np.empty((K, X.shape[1]), dtype=np.float64)
    C[0] = X[rng.integers(M)]
    d2 = _sqdist(X, C[:1])[:, 0]
    for k in range(1, K):
        tot = d2.sum()
        C[k] = X[rng.integers(M)] if not tot > 0 else X[min(int(np.searchsorted(np.cumsum(d2), rng.random() * tot)), M - 1)]
        d2 = np.minimum(d2, _sqdist(X, C[k:k + 1])[:, 0])
    labels = np.zeros(M, dtype=np.int64)
    inertia = np.inf
    for _ in range(iters):
        D = _sqdist(X, C)
        labels = D.argmin(1)
        best = D[np.arange(M), labels]
        counts = np.bincount(labels, minlength=K)
        newC = np.zeros_like(C)
        np.add.at(newC, labels, X)
        for k in range(K):
            if counts[k]:
                newC[k] /= counts[k]
            else:                       # empty cluster: re-seed with the pose farthest from its centre
                far = int(best.argmax())
                newC[k] = X[far]
                best[far] = 0.0
        new_inertia = float(best.sum())
        shift = float(np.abs(newC - C).max())
        C = newC
        if shift <= tol or abs(inertia - new_inertia) <= tol * max(new_inertia, 1e-30):
            inertia = new_inertia
            break
        inertia = new_inertia
    D = _sqdist(X, C)
    labels = D.argmin(1)
    return C, labels, float(D[np.arange(M), labels].sum())


def build(poses, H, seed=0, iters=100, medoid=False):
    """poses [M,17,3] metres -> float32 [H,17,3] root-centred cluster poses, largest cluster first."""
    P = poses - poses[:, 0:1, :]
    X = P.reshape(len(P), -1)
    C, labels, inertia = kmeans(X, H, seed=seed, iters=iters)
    counts = np.bincount(labels, minlength=H)
    if medoid:
        D = _sqdist(X, C)
        C = np.stack([X[int(D[:, k].argmin())] for k in range(H)])
    order = np.argsort(-counts, kind="stable")
    out = C[order].reshape(H, 17, 3)
    out = out - out[:, 0:1, :]
    return out.astype(np.float32), counts[order], inertia


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("poses", help=".npy / .npz / .pkl pose set [M,17,3]")
    ap.add_argument("--hypo", type=int, required=True, help="number of clusters H (the drivers' --hypo)")
    ap.add_argument("--name", default="h36m", help="file stem: h36m | 3dhp | h36m_sitting (run/opt_main.py:58-65)")
    ap.add_argument("--out-dir", default="clusters")
    ap.add_argument("--key", default=None, help="array name inside an .npz / field name inside a .pkl record")
    ap.add_argument("--unit", choices=("m", "mm"), default=None, help="unit of the input coordinates (default: m; .pkl records: mm)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--medoid", action="store_true", help="write the real pose nearest to every centre instead of the mean")
    a = ap.parse_args(argv)
    poses = load_poses(a.poses, a.key, a.unit)
    out, counts, inertia = build(poses, a.hypo, a.seed, a.iters, a.medoid)
    os.makedirs(a.out_dir, exist_ok=True)
    path = os.path.join(a.out_dir, f"{a.name}_cluster{a.hypo}.npy")
    np.save(path, out)
    rms = float(np.sqrt(inertia / (len(poses) * 17)))
    print(f"{path}: {a.hypo} clusters of {len(poses)} poses, float32 {out.shape}, sizes {int(counts[0])} .. {int(counts[-1])}, "
          f"rms joint distance to the assigned centre {rms * 1e3:.1f} mm")
    return path


if __name__ == "__main__":
    main()
