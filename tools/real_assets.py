"""Acceptance recipe for the day the REAL assets exist (reference Readme.md:134-162): stage 1 of 2 - capture.

The reference's evaluation needs four Google-Drive downloads that were never available offline: `checkpoint_1500.pth` (the trained
score network), `data/h36m/h36m_test.pkl` (+ `h36m_sh_dt_ft.pkl` detections) or `data/3dpw/pw3d_test.npz`, and
`clusters/h36m_cluster{H}.npy`.  Every parity number of this repository is therefore on random-init weights.  With the files in
`<assets>` (the reference's own directory layout, i.e. what `python -m run.opt_main ... --ckpt_dir <assets>/checkpoint/concatebb
--ckpt_name checkpoint_1500.pth` is run from):

  stage 1 (build container, has /root/reference, CPU):
      python tools/real_assets.py --assets <assets> --dataset h36m --hypo 5 --poses 160 --steps 1000 [--gt] --tag real_h36m
    runs the REFERENCE - its reader (h36m.py:206-263 / pw3d.py:177-227), its model with the checkpoint loaded the way
    run/opt_main.py:120-137 does, its IPO + OIL loop (opt_main.py:166-224 re-driven by tools/gen_golden.py) and its eval_multi - over a
    stated subsample of the test set (every ZeDO.sample-th frame as the config says, then `--poses` of those, evenly spaced) and writes
    tests/golden/<tag>.npz: the selection, the sha256 of every asset file, the dataset means, the per-(pose, hypothesis) errors, and
    what the split-fp16 mode's range guards will say about the trained GroupNorm parameters and weight rows.
  stage 2 (GPU box, the same <assets> beside the repo or in ZEDO_REAL_ASSETS): tests/test_real_assets_gpu.py consumes every
    tests/golden/real_*.npz: this repository's readers + checkpoint loader + fused pipeline + metric on the same files and selection,
    against the fixture (PA-MPJPE within 0.05 mm outright; MPJPE within 0.05 mm for a trained, contracting prior - round 4 measured
    0.0011 mm with a contractive stand-in - and within the bars of the synthetic-weight tests otherwise).

Both stages are exercised NOW on the synthetic files of the assets' formats under tests/golden/assets (fixtures
tests/golden/real_synth_h36m.npz, real_synth_3dpw.npz: `--assets tests/golden/assets --synthetic-checkpoint`)."""
import argparse
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETTINGS = {   # configs/optim/concat_pose_optimization_{h36m,pw3d}.py:72-81 of the reference
    "h36m": dict(keylist=[0, 1, 4], ipo_T=3.0, minT=0.5, sample=640, data=("data", "h36m"), files=["h36m_test.pkl", "h36m_sh_dt_ft.pkl"], cluster="h36m"),
    "3dpw": dict(keylist=list(range(17)), ipo_T=8.0, minT=0.2, sample=35, data=("data", "3dpw"), files=["pw3d_test.npz"], cluster="h36m"),
}


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def select(n_total, poses):
    """`poses` indices spread evenly over the (already frame-subsampled) test set: every action of H36M stays represented."""
    if not poses or poses >= n_total:
        return np.arange(n_total)
    return np.unique(np.linspace(0, n_total - 1, poses).round().astype(np.int64))


def f16x3_guards(w):
    """What zedo_weights_set_math(ZEDO_MATH_F16X3) will say (csrc/zedo_capi.hip): the activation bound from the GroupNorm parameters
    (< 32768) and the smallest ratio of a non-zero row's maximum to its matrix maximum (>= 2^-8)."""
    s31 = np.float32(5.5677643)
    gb = lambda g, b: float(np.abs(w[g]).max() * s31 + np.abs(w[b]).max())
    gn = [gb("pre_gnorm.weight", "pre_gnorm.bias")] + [gb(f"b{b}_gnorm{k}.weight", f"b{b}_gnorm{k}.bias") for b in (1, 2) for k in (1, 2)]
    act = max(gn[1], gn[3], gn[0] + gn[2] + gn[4])
    ratio = 1.0
    for name in ["pre_dense.weight", "post_dense.weight"] + [f"b{b}_dense{k}.weight" for b in (1, 2) for k in (1, 2)]:
        m = np.abs(w[name])
        rm = m.max(axis=1)
        if m.max() > 0 and (rm > 0).any():
            ratio = min(ratio, float((rm[rm > 0] / m.max()).min()))
    finite = all(np.isfinite(v).all() for v in w.values())
    return dict(activation_bound=act, min_row_ratio=ratio, finite=finite, accepted=bool(finite and act < 32768.0 and ratio >= 1.0 / 256.0))


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--assets", required=True)
    ap.add_argument("--dataset", choices=sorted(SETTINGS), default="h36m")
    ap.add_argument("--hypo", type=int, default=5)
    ap.add_argument("--poses", type=int, default=160)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--gt", action="store_true", help="ground-truth 2D instead of detections (run/opt_main.py --gt)")
    ap.add_argument("--ckpt", default=os.path.join("checkpoint", "concatebb", "checkpoint_1500.pth"), help="relative to --assets")
    ap.add_argument("--synthetic-checkpoint", action="store_true",
                    help="no trained checkpoint / cluster file in --assets: use the seeded random-init weights and clusters, written to "
                         "--scratch in the reference's file formats (exercise of the recipe; the GPU test regenerates the same files)")
    ap.add_argument("--scratch", default="/tmp/zedo_real_assets_scratch")
    ap.add_argument("--sample", type=int, default=None, help="override ZeDO.sample (frame interval of the reader); 0 = every frame")
    ap.add_argument("--tag", default=None)
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_golden as G          # imports the reference (build container only) and the shared re-driven loop
    import torch
    st = SETTINGS[a.dataset]
    assets = os.path.abspath(a.assets)
    droot = os.path.join(assets, *st["data"]) if os.path.isdir(os.path.join(assets, *st["data"])) else os.path.join(assets, st["data"][1])
    sample = st["sample"] if a.sample is None else (a.sample or None)

    # ---- the checkpoint, loaded the way run/opt_main.py:120-137 does
    ckpt_path = os.path.join(a.scratch if a.synthetic_checkpoint else assets, a.ckpt)
    if a.synthetic_checkpoint:
        syn = G.syn          # this repository's seeded generators (tools/gen_golden.py imports them beside the reference)
        os.makedirs(os.path.dirname(ckpt_path), exist_ok=True)
        sd = {"module." + k: torch.tensor(v) for k, v in syn.make_weights(seed=0).items()}
        sd["module.sigmas"] = torch.tensor(syn.sigmas_buffer())
        torch.save({"model_state_dict": sd, "ema": {"decay": 0.9999, "num_updates": 0, "shadow_params": []}, "step": 1500}, ckpt_path)
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in ckpt["model_state_dict"].items()}
    w = {k: v.detach().cpu().numpy() for k, v in sd.items() if k != "sigmas"}
    model = G.ref_model({k: v.astype(np.float32) for k, v in w.items()})

    # ---- the dataset through the REFERENCE's reader, then the stated selection
    if a.dataset == "h36m":
        ds = G.H36MDataset3D(droot, "test", gt2d=a.gt, abs_coord=True, sample_interval=sample, flip=False)
    else:
        ds = G.PW3D(droot, "test", gt2d=a.gt, abs_coord=True, sample_interval=sample, flip=False)
    sel = select(len(ds.db_2d), a.poses)
    for name in ("db_2d", "db_3d", "camera_param", "gt_dataset", "image_name", "w", "h"):
        if hasattr(ds, name):
            v = getattr(ds, name)
            setattr(ds, name, [v[i] for i in sel] if isinstance(v, list) else v[sel])
    gt_2d, K = np.asarray(ds.db_2d, dtype=np.float32), ds.camera_param
    N, H, S = len(sel), a.hypo, a.steps

    cl_path = os.path.join(a.scratch if a.synthetic_checkpoint else assets, "clusters", f"{st['cluster']}_cluster{H}.npy")
    if a.synthetic_checkpoint:
        syn = G.syn          # this repository's seeded generators (tools/gen_golden.py imports them beside the reference)
        os.makedirs(os.path.dirname(cl_path), exist_ok=True)
        np.save(cl_path, syn.make_clusters(H, seed=8))
    cl = np.load(cl_path).astype(np.float32)

    # ---- run/opt_main.py:166-224, re-driven (tools/gen_golden.py)
    res_all, cs, Ts = [], [], []
    for sid in range(H):
        noisy = (torch.ones((N, 17, 3)) * torch.tensor(cl - cl[:, 0:1, :])[sid:sid + 1]).float()
        r = G.run_ref_ipo(noisy.numpy(), gt_2d[:, :, :2], K, "z", st["keylist"], st["ipo_T"], st["minT"], 2.0, 500, trace_upto=1)
        x = torch.tensor(r["R"]).bmm(noisy.permute(0, 2, 1)).permute(0, 2, 1).contiguous().numpy()
        res, _, _ = G.run_ref_oil(model, x, gt_2d[:, :, :2], gt_2d[:, :, 2].copy(), K, r["T"], S, [])
        res_all.append(res)
        cs.append(np.stack([r["R"][:, 0, 0], r["R"][:, 1, 0]], -1))
        Ts.append(r["T"][:, 0, :])
        print(f"  hypothesis {sid + 1}/{H} done", flush=True)
    batch = np.swapaxes(np.array(res_all), 0, 1)                      # [N, H, 17, 3]
    p1 = float(ds.eval_multi(batch, protocol2=False))
    p2 = float(ds.eval_multi(batch, protocol2=True))
    files = {f: sha256_file(os.path.join(droot, f)) for f in st["files"] if os.path.exists(os.path.join(droot, f))}
    if not a.synthetic_checkpoint:          # (the synthetic stand-ins are regenerated from their seeds by the test: torch.save is not byte-stable)
        files[os.path.relpath(ckpt_path, assets)] = sha256_file(ckpt_path)
        files[os.path.relpath(cl_path, assets)] = sha256_file(cl_path)
    tag = a.tag or f"real_{a.dataset}"
    out = os.path.join(ROOT, "tests", "golden", tag + ".npz")
    guards = f16x3_guards(w)
    np.savez_compressed(out, dataset=np.array(a.dataset), gt2d=np.bool_(a.gt), N=np.int64(N), H=np.int64(H), S=np.int64(S), sel=sel, sample=np.int64(sample or 0),
                        keylist=np.array(st["keylist"]), ipo_T=np.float64(st["ipo_T"]), minT=np.float64(st["minT"]),
                        ckpt=np.array(a.ckpt), cluster_file=np.array(os.path.join("clusters", os.path.basename(cl_path))), data_dir=np.array(os.path.relpath(droot, assets)),
                        file_names=np.array(sorted(files)), file_sha256=np.array([files[k] for k in sorted(files)]),
                        trained=np.bool_(not a.synthetic_checkpoint), mpjpe=np.float64(p1), pa_mpjpe=np.float64(p2),
                        batch_results=batch.astype(np.float32), ipo_cs=np.stack(cs).astype(np.float32), ipo_T_out=np.stack(Ts).astype(np.float32),
                        f16x3_activation_bound=np.float64(guards["activation_bound"]), f16x3_min_row_ratio=np.float64(guards["min_row_ratio"]),
                        f16x3_accepted=np.bool_(guards["accepted"]))
    print(f"wrote {out}: N = {N} of the test set (sample {sample}), H = {H}, S = {S}: MPJPE {p1 * 1e3:.4f} mm, PA-MPJPE {p2 * 1e3:.4f} mm; "
          f"f16x3 guards: activation bound {guards['activation_bound']:.1f} (< 32768), min row ratio 2^{np.log2(guards['min_row_ratio']):.2f} (>= 2^-8): "
          f"{'accepted' if guards['accepted'] else 'REFUSED'}")


if __name__ == "__main__":
    main()
