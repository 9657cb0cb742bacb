#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

The reference (/root/reference, ipl-uw/ZeDO-Release) has no tests and ships no
fixtures, data or weights, so the parity pin of this repo is: seeded synthetic
inputs (zedo-release_amd/lib/dataset/synthetic.py) pushed through the
reference's own functions on CPU, outputs committed as small .npz fixtures.
This script is the only place that touches /root/reference; nothing under
tests/, bench.py or smoke() reads it at run time.

The reference cannot be imported as-is offline: torchvision (dead import at
lib/algorithms/advanced/model.py:20) and prettytable (lib/dataset/h36m.py:4)
are missing -> two tiny stubs under tools/ref_stubs/.  run/opt_main.py needs
absl + ml_collections, so its loop (opt_main.py:166-224) is re-driven here,
statement by statement, with a SimpleNamespace config holding the values of
configs/optim/concat_pose_optimization_*.py.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--only name]
"""
import argparse
import os
import sys
from types import SimpleNamespace as NS

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "ref_stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, os.path.join(ROOT, "zedo-release_amd", "lib", "dataset"))
sys.path.insert(2, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import synthetic as syn  # zedo-release_amd/lib/dataset/synthetic.py (numpy only)

from lib.algorithms.advanced.model import ScoreModelFC_Adv, get_timestep_embedding
from lib.algorithms.advanced import sde_lib, sampling
from lib.algorithms.advanced import utils as mutils
from lib.algorithms.advanced.simple_zeroshot_opt import gradient_field_gen, RotOpt
from lib.dataset.h36m import H36MDataset3D
from lib.dataset.pw3d import PW3D
from lib.utils.transforms import procrustes

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(int(os.environ.get("ZEDO_GOLDEN_THREADS", "8")))


def ref_config():
    return NS(
        device=torch.device("cpu"),
        model=NS(embedding_type="positional", sigma_max=50, sigma_min=0.01, num_scales=1000,
                 scale_by_sigma=False, beta_min=0.1, beta_max=20.0, t=0.1, ema_rate=0.9999),
        training=NS(cond_pose_mask_prob=0.0, cond_part_mask_prob=0.0, cond_joint_mask_prob=0.0,
                    sde="subvpsde", continuous=True),
        sampling=NS(method="pc", predictor="euler_maruyama", corrector="none", snr=0.16,
                    n_steps_each=1, probability_flow=True, noise_removal=True),
    )


def ref_model(weights, dtype=torch.float32):
    m = ScoreModelFC_Adv(ref_config(), n_joints=17, joint_dim=3, hidden_dim=1024, embed_dim=512, cond_dim=3)
    sd = {k: torch.tensor(v) for k, v in weights.items()}
    sd["sigmas"] = torch.tensor(syn.sigmas_buffer())
    m.load_state_dict(sd, strict=True)
    m.eval()
    m = m.to(dtype)
    if dtype != torch.float32:
        # fp64 arbiter only: the reference's positional embedding is hard-wired to fp32
        # (model.py:87-90), so cast its output for the double-precision copy of the model.
        m.posit_proj = lambda t: get_timestep_embedding(t, 512).to(dtype)
    return m


def ref_sde():
    return sde_lib.subVPSDE(beta_min=0.1, beta_max=20.0, N=1000, T=0.1)


def ref_sampling_fn(n):
    cfg = ref_config()
    return sampling.get_sampling_fn(cfg, ref_sde(), (n, 17, 3), lambda x: x, 0.01, device=torch.device("cpu"))


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------ captures

def gen_model():
    w = syn.make_weights(seed=0)
    m = ref_model(w)
    g = np.random.Generator(np.random.Philox(key=[7, 1]))
    x = (0.3 * g.standard_normal((8, 17, 3))).astype(np.float32)
    ts = np.array([0.1, 0.0555, 0.01], dtype=np.float32)
    eps = []
    tbias = []
    temb_all = []
    with torch.no_grad():
        for t in ts:
            labels = torch.ones(8) * torch.tensor(t) * 999
            eps.append(m(torch.tensor(x), labels, None, None).numpy())
            temb = m.shared_time_embed(get_timestep_embedding(labels[:1], 512))
            temb_all.append(temb.numpy()[0])
            rows = []
            for n in ["pre_dense", "b1_dense1", "b1_dense2", "b2_dense1", "b2_dense2"]:
                rows.append((getattr(m, n + "_t")(temb) + getattr(m, n).bias).numpy()[0])
            tbias.append(np.stack(rows))
    # a wider sweep of labels for the positional embedding alone
    lab = torch.linspace(0.1, 0.01, 1000) * 999
    pe = get_timestep_embedding(lab, 512).numpy()
    save("model_forward", weights_seed=np.int64(0), weights_sha=np.array(syn.weights_checksum(w)),
         x=x, ts=ts, eps=np.stack(eps), temb=np.stack(temb_all), tbias=np.stack(tbias),
         pe_labels=lab.numpy(), pe=pe[::37])


ALT_WEIGHTS = {"w1": dict(seed=1), "tied": dict(seed=0, prior="tied")}


def gen_weights_alt():
    """The single-call goldens again on OTHER weights (every capture used make_weights(seed=0)):
    a second random draw (seed 1) and the contractive "tied" prior - network output at three noise levels, the time-bias
    rows, pc_sampler's x_mean at four steps of the 1000-step schedule, and the 100-step loop from a pinned (R, T) on 12
    poses in fp32 and fp64 (snapshots 20 / 21 / 100: across the switch to the least-squares T)."""
    g = np.random.Generator(np.random.Philox(key=[7, 11]))
    x = (0.3 * g.standard_normal((8, 17, 3))).astype(np.float32)
    ts3 = np.array([0.1, 0.0555, 0.01], dtype=np.float32)
    N = 12
    d = syn.make_poses(N, seed=23, conf_mode="uniform")
    cl = syn.make_clusters(1, seed=8)
    x0 = np.broadcast_to((cl - cl[:, 0:1])[0][None], (N, 17, 3)).astype(np.float32).copy()
    ang = g.uniform(-np.pi, np.pi, N)
    R = np.zeros((N, 3, 3), np.float32)
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = np.cos(ang), -np.sin(ang), np.sin(ang), np.cos(ang), 1
    T = (d["db_3d"][:, 0:1, :] * (1 + 0.05 * g.standard_normal((N, 1, 1)))).astype(np.float32)
    xi = np.einsum("bij,bkj->bki", R, x0).astype(np.float32)
    out = dict(x=x, ts=ts3, db2d=d["db_2d"], K=d["camera_param"], x_init=xi, T_init=T, snap_steps=np.array([20, 21, 100]))
    fn = ref_sampling_fn(8)
    for tag, kw in ALT_WEIGHTS.items():
        w = syn.make_weights(**kw)
        m = ref_model(w)
        out[f"sha_{tag}"] = np.array(syn.weights_checksum(w))
        eps, tbias = [], []
        with torch.no_grad():
            for t in ts3:
                labels = torch.ones(8) * torch.tensor(t) * 999
                eps.append(m(torch.tensor(x), labels, None, None).numpy())
                temb = m.shared_time_embed(get_timestep_embedding(labels[:1], 512))
                tbias.append(np.stack([(getattr(m, n + "_t")(temb) + getattr(m, n).bias).numpy()[0]
                                       for n in ["pre_dense", "b1_dense1", "b1_dense2", "b2_dense1", "b2_dense2"]]))
        out[f"eps_{tag}"], out[f"tbias_{tag}"] = np.stack(eps), np.stack(tbias)
        tl = torch.linspace(0.1, 0.01, 1000)
        idx = [0, 199, 200, 999]
        out["idx_1000"] = np.array(idx)
        out[f"xmean_{tag}"] = np.stack([fn(m, condition=torch.zeros(8, 17, 2), denoise_x=torch.tensor(x), t=tl[i], t_step=i)[1] for i in idx])
        for dt_tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            mm = m if dt == torch.float32 else ref_model(w, torch.float64)
            res, Tf, got = run_ref_oil(mm, xi, d["db_2d"][:, :, :2], d["db_2d"][:, :, 2].copy(), d["camera_param"], T, 100, [20, 21, 100], dt)
            out[f"snaps_{tag}_{dt_tag}"] = np.stack([got[k] for k in (20, 21, 100)])
            out[f"T_final_{tag}_{dt_tag}"] = Tf
    save("weights_alt", **out)


def gen_pc_step():
    w = syn.make_weights(seed=0)
    m = ref_model(w)
    fn = ref_sampling_fn(8)
    g = np.random.Generator(np.random.Philox(key=[7, 2]))
    x = (0.3 * g.standard_normal((8, 17, 3))).astype(np.float32)
    out = {}
    for S in (1000, 100):
        ts = torch.linspace(0.1, 0.01, S)
        out[f"ts_{S}"] = ts.numpy()
        idx = [0, S // 5 - 1, S // 5, S - 1]
        res = []
        for i in idx:
            torch.manual_seed(i)  # output must not depend on this
            trajs, xm = fn(m, condition=torch.zeros(8, 17, 2), denoise_x=torch.tensor(x), t=ts[i], t_step=i)
            assert trajs.shape == (1, 8, 17, 3) and np.array_equal(trajs[0], xm)
            res.append(xm)
        out[f"idx_{S}"] = np.array(idx)
        out[f"xmean_{S}"] = np.stack(res)
    # score_fn surface
    sfn = mutils.get_score_fn(ref_sde(), m, train=False, continuous=True)
    with torch.no_grad():
        out["score_t0p05"] = sfn(torch.tensor(x), torch.ones(8) * 0.05, None, None).numpy()
    save("pc_step", x=x, **out)


def gen_reproj():
    d = syn.make_poses(16, seed=3, conf_mode="wild")
    g = np.random.Generator(np.random.Philox(key=[7, 3]))
    uv = d["db_2d"][:, :, :2]
    K = d["camera_param"]
    x = (0.25 * g.standard_normal((16, 17, 3))).astype(np.float32)
    x[:, 0] = 0
    Tgiven = (np.array([0.1, -0.2, 5.0]) + 0.1 * g.standard_normal((16, 1, 3))).astype(np.float32)
    out = {}
    for tag, conf in (("wild", d["db_2d"][:, :, 2].copy()), ("ones", np.ones((16, 17), np.float32)), ("none", None)):
        c = None if conf is None else torch.tensor(conf.copy())
        gT = gradient_field_gen(torch.tensor(uv), torch.tensor(x), torch.tensor(K), t=torch.tensor(Tgiven), conf=c)
        c2 = None if conf is None else torch.tensor(conf.copy())
        gS, Ts = gradient_field_gen(torch.tensor(uv), torch.tensor(x), torch.tensor(K), conf=c2, returnT=True)
        out[f"g_given_{tag}"] = gT.numpy()
        out[f"g_solve_{tag}"] = gS.numpy()
        out[f"T_solve_{tag}"] = Ts.numpy()
        if conf is not None:
            out[f"conf_after_{tag}"] = c2.numpy()  # in-place clamp is observable
    # sign-flip case (simple_zeroshot_opt.py:93): x matches the detections, but the detections are
    # mirrored through the principal point, so the least-squares depth comes out negative and is negated.
    xrel = (d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float32)
    uvneg = (2 * K[:, None, :2, 2] - uv).astype(np.float32)
    gN, TN = gradient_field_gen(torch.tensor(uvneg), torch.tensor(xrel), torch.tensor(K),
                                conf=torch.ones(16, 17), returnT=True)
    # far-away pose (depth ~12 m folded into x itself): small |T|, mixed signs before the fix
    xflip = x.copy()
    xflip[:, :, 2] *= -1
    gF, TF = gradient_field_gen(torch.tensor(uv), torch.tensor(xflip + np.array([0, 0, 12.0], np.float32)),
                                torch.tensor(K), conf=torch.ones(16, 17), returnT=True)
    save("reproj", uv=uv, K=K, x=x, T_given=Tgiven, conf_wild=d["db_2d"][:, :, 2],
         uv_neg=uvneg, x_rel=xrel, g_neg=gN.numpy(), T_neg=TN.numpy(),
         x_far=xflip + np.array([0, 0, 12.0], np.float32), g_far=gF.numpy(), T_far=TF.numpy(), **out)


def run_ref_ipo(x0, cond, K, axes, keylist, ipo_T, minT, maxT, iters, trace_upto=20, dtype=torch.float32):
    """opt_main.py:170-195 on CPU tensors.  Returns dict of captures.  dtype=float64: the same statements with
    every tensor and RotOpt().double() - the arbiter for the per-iteration IPO parity criterion."""
    denoise_x = torch.tensor(x0).to(dtype)
    condition = torch.tensor(cond).to(dtype)
    Kt = torch.tensor(K).to(dtype)
    pelvis = torch.cat((condition[:, 0, :], torch.ones((condition.shape[0], 1))), axis=-1)
    T = torch.inverse(Kt).bmm(pelvis[:, :, None]).permute(0, 2, 1)
    T = T / torch.norm(T, dim=-1, keepdim=True) * ipo_T
    T0 = T.clone()
    rot_opt = RotOpt(denoise_x.shape[0], axis=axes, minT=minT, maxT=maxT).to(dtype)
    opt = torch.optim.Adam(rot_opt.parameters(), lr=0.1)
    crit = torch.nn.L1Loss(reduction="none")
    tr_q, tr_s, tr_l = [], [], []
    for i in range(iters):
        opt.zero_grad()
        rot2d = rot_opt(denoise_x[:, keylist, :], T, Kt)
        loss = torch.mean(crit(rot2d[:, :, :2], condition[:, keylist, :2]))
        loss.backward()
        opt.step()
        if i < trace_upto:
            z = torch.zeros(denoise_x.shape[0], 1, dtype=dtype)
            q = torch.cat([rot_opt.rot_vect] + [getattr(rot_opt, "rot_vect_%s" % a, z) for a in "xyz"], -1)
            tr_q.append(q.detach().numpy().copy())
            tr_s.append(rot_opt.scale.detach().numpy().reshape(-1).copy())
            tr_l.append(float(loss))
    Tfin = (T * torch.clamp(rot_opt.scale, min=minT, max=maxT)).detach()
    R = rot_opt.generate_matrix().detach()
    return dict(T0=T0.numpy(), T=Tfin.numpy(), R=R.numpy(), loss=np.float32(float(loss)),
                trace_q=np.stack(tr_q), trace_scale=np.stack(tr_s),
                trace_loss=np.array(tr_l, np.float32 if dtype == torch.float32 else np.float64),
                scale=rot_opt.scale.detach().numpy().reshape(-1))


IPO_TRACE = 50


def gen_ipo():
    out = {}
    clusters = syn.make_clusters(2, seed=5)
    centred = clusters - clusters[:, 0:1]
    for N in (8, 64):
        d = syn.make_poses(N, seed=11 + N)
        out[f"db2d_{N}"] = d["db_2d"]
        out[f"K_{N}"] = d["camera_param"]
        x0 = np.broadcast_to(centred[0][None], (N, 17, 3)).astype(np.float32).copy()
        for axes in ("z", "xyz"):
            for kname, kl, ipoT, minT in (("h36m", [0, 1, 4], 3.0, 0.5), ("pw3d", list(range(17)), 8.0, 0.2)):
                r = run_ref_ipo(x0, d["db_2d"][:, :, :2], d["camera_param"], axes, kl, ipoT, minT, 2.0, 500,
                                trace_upto=IPO_TRACE)
                for k, v in r.items():
                    out[f"{k}_{N}_{axes}_{kname}"] = v
                # fp64 arbiter (RotOpt().double(), every tensor double) for the per-iteration criterion
                r64 = run_ref_ipo(x0, d["db_2d"][:, :, :2], d["camera_param"], axes, kl, ipoT, minT, 2.0, IPO_TRACE,
                                  trace_upto=IPO_TRACE, dtype=torch.float64)
                out[f"trace_q64_{N}_{axes}_{kname}"] = r64["trace_q"]
                out[f"trace_scale64_{N}_{axes}_{kname}"] = r64["trace_scale"]
                out[f"trace_loss64_{N}_{axes}_{kname}"] = r64["trace_loss"]
    save("ipo", cluster0=centred[0], **out)


# custom ZeDO.IPO_keylist values (config-reachable; round 6: every key-list length has its lane-per-row kernel): key lists of 1, 5, 8 and 12
# joints, with and without the root joint
IPO_CUSTOM_KEYLISTS = dict(k1=([4], 3.0, 0.5), k5=([0, 2, 5, 11, 14], 3.0, 0.5), k8=([0, 1, 4, 7, 8, 11, 14, 16], 8.0, 0.2),
                           k12=(list(range(1, 13)), 8.0, 0.2))


def gen_ipo_custom():
    """The reference's IPO loop (opt_main.py:170-195) in float64 - the arbiter of the per-iteration parity test - for key lists the shipped
    configurations do not use, on the problems of gen_ipo (same clusters, poses, intrinsics), plus the fp32 end state after 500 iterations.
    A file of its own: tests/golden/ipo.npz stays byte for byte what round 1-5 generated."""
    out = {}
    clusters = syn.make_clusters(2, seed=5)
    centred = clusters - clusters[:, 0:1]
    for N in (8, 64):
        d = syn.make_poses(N, seed=11 + N)
        out[f"db2d_{N}"] = d["db_2d"]
        out[f"K_{N}"] = d["camera_param"]
        x0 = np.broadcast_to(centred[0][None], (N, 17, 3)).astype(np.float32).copy()
        for axes in ("z", "xyz"):
            for kname, (kl, ipoT, minT) in IPO_CUSTOM_KEYLISTS.items():
                r64 = run_ref_ipo(x0, d["db_2d"][:, :, :2], d["camera_param"], axes, kl, ipoT, minT, 2.0, IPO_TRACE,
                                  trace_upto=IPO_TRACE, dtype=torch.float64)
                out[f"trace_q64_{N}_{axes}_{kname}"] = r64["trace_q"]
                out[f"trace_scale64_{N}_{axes}_{kname}"] = r64["trace_scale"]
                out[f"trace_loss64_{N}_{axes}_{kname}"] = r64["trace_loss"]
                r32 = run_ref_ipo(x0, d["db_2d"][:, :, :2], d["camera_param"], axes, kl, ipoT, minT, 2.0, 500, trace_upto=1)
                out[f"loss_{N}_{axes}_{kname}"] = r32["loss"]
                out[f"trace_q1_{N}_{axes}_{kname}"] = r32["trace_q"][0]
                out[f"trace_scale1_{N}_{axes}_{kname}"] = r32["trace_scale"][0]
                out[f"keylist_{kname}"] = np.array(kl, np.int32)
    save("ipo_custom", cluster0=centred[0], **out)


def run_ref_oil(m, x, cond, conf, K, T, S, snaps, dtype=torch.float32):
    """opt_main.py:197-222 (the torch.no_grad block), CPU tensors."""
    fn = ref_sampling_fn(x.shape[0])
    sde = ref_sde()
    condition = torch.tensor(cond).to(dtype)
    conf = torch.tensor(conf).to(dtype)
    Kt = torch.tensor(K).to(dtype)
    T = torch.tensor(T).to(dtype)
    denoise_x = torch.tensor(x).to(dtype)
    timestamp = torch.linspace(sde.T, 0.01, S)
    got = {}
    with torch.no_grad():
        for i in range(S):
            if i < S // 5:
                jg = gradient_field_gen(condition, denoise_x, Kt, t=T, conf=conf, returnT=False)
            else:
                jg, T = gradient_field_gen(condition, denoise_x, Kt, conf=conf, returnT=True)
            denoise_x += jg
            trajs, results = fn(m, condition=condition * 0, gradient=jg, denoise_x=denoise_x,
                                t=timestamp[i].to(dtype), t_step=i, args=None)
            denoise_x = torch.tensor(results)
            if (i + 1) in snaps:
                got[i + 1] = results.copy()
    return results, T.numpy(), got


def gen_oil():
    w = syn.make_weights(seed=0)
    m32 = ref_model(w)
    m64 = ref_model(w, torch.float64)
    N = 12
    d = syn.make_poses(N, seed=21, conf_mode="uniform")
    cl = syn.make_clusters(1, seed=6)
    x0 = np.broadcast_to((cl - cl[:, 0:1])[0][None], (N, 17, 3)).astype(np.float32).copy()
    # pinned (R,T): a z-rotation per pose and a plausible translation (no IPO here: SURVEY 7 protocol A)
    g = np.random.Generator(np.random.Philox(key=[7, 4]))
    ang = g.uniform(-np.pi, np.pi, N)
    R = np.zeros((N, 3, 3), np.float32)
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = np.cos(ang), -np.sin(ang), np.sin(ang), np.cos(ang), 1
    T = (d["db_3d"][:, 0:1, :] * (1 + 0.05 * g.standard_normal((N, 1, 1)))).astype(np.float32)
    x = np.einsum("bij,bkj->bki", R, x0).astype(np.float32)
    out = dict(db2d=d["db_2d"], K=d["camera_param"], x_init=x, T_init=T)
    for S, snaps in ((1000, [1, 10, 100, 200, 201, 500, 1000]), (100, [1, 10, 20, 21, 50, 100])):
        for tag, m, dt in (("f32", m32, torch.float32), ("f64", m64, torch.float64)):
            res, Tf, got = run_ref_oil(m, x, d["db_2d"][:, :, :2], d["db_2d"][:, :, 2], d["camera_param"], T, S, snaps, dt)
            out[f"snap_steps_{S}"] = np.array(snaps)
            out[f"snaps_{S}_{tag}"] = np.stack([got[s] for s in snaps])
            out[f"T_final_{S}_{tag}"] = Tf
    save("oil", **out)


def _h36m_obj(gt_mm, actions):
    ds = object.__new__(H36MDataset3D)
    ds.subset = "test"
    ds.seq5678 = False
    ds.gt_dataset = [dict(joint_3d_camera=gt_mm[i], action=int(actions[i])) for i in range(len(gt_mm))]
    return ds


def _pw3d_obj(db3d):
    ds = object.__new__(PW3D)
    ds.db_3d = db3d
    return ds


def gen_eval():
    N, H = 32, 5
    d = syn.make_poses(N, seed=31, dtype3d=np.float64)
    g = np.random.Generator(np.random.Philox(key=[7, 5]))
    gt_m = d["db_3d"]
    rel = gt_m - gt_m[:, 0:1]
    preds = (rel[:, None] + 0.05 * g.standard_normal((N, H, 17, 3))).astype(np.float32)
    # hypothesis 3 of every 4th pose is a mirrored pose -> Procrustes 'best' picks a reflection
    preds[::4, 3] = (rel[::4] * np.array([-1.0, 1.0, 1.0]) * 1.3 + 0.2).astype(np.float32)
    actions = 2 + (np.arange(N) % 15)
    gt_mm = gt_m * 1000.0
    h36 = _h36m_obj(gt_mm, actions)
    pw = _pw3d_obj(gt_m.astype(np.float32))
    out = dict(preds=preds, gt_mm_h36m=gt_mm, actions=actions, db3d_pw3d=gt_m.astype(np.float32))
    out["h36m_p1"] = np.float64(h36.eval_multi(preds, protocol2=False))
    out["h36m_p2"] = np.float64(h36.eval_multi(preds, protocol2=True))
    out["pw3d_p1"] = np.float64(pw.eval_multi(preds, protocol2=False))
    out["pw3d_p2"] = np.float64(pw.eval_multi(preds, protocol2=True))
    # per (n,h) errors with the reference's own inner statements (h36m.py:402-408)
    e1 = np.zeros((N, H))
    e2 = np.zeros((N, H))
    Z = np.zeros((N, H, 17, 3))
    for n in range(N):
        gt = (gt_mm[n] - gt_mm[n][0:1]) / 1000.0
        for h in range(H):
            e1[n, h] = np.mean(np.sqrt(np.square(preds[n, h] - gt).sum(axis=1)))
            Z[n, h] = procrustes(gt.copy(), preds[n, h].copy())[1]
            e2[n, h] = np.mean(np.sqrt(np.square(Z[n, h] - gt).sum(axis=1)))
    out.update(err_p1=e1, err_p2=e2, aligned=Z)
    # rank-2 alignments: the prediction, or the ground truth, lies in a plane, so A0^T B0 has a zero
    # singular value and numpy's SVD returns an arbitrary sign for the null direction (transforms.py:88-96).
    # Cases: plane z = 0 exactly (exact rank 2) and a random plane rounded to float32 (rank 2 to 1e-8).
    gd = np.random.Generator(np.random.Philox(key=[7, 51]))

    def flat(P, nrm):
        nrm = nrm / np.linalg.norm(nrm)
        return P - (P @ nrm)[..., None] * nrm

    gen_gt = rel[:6].copy()                                                  # float64, general position
    gen_pred = (rel[6:12] * 1.1 + 0.03 * gd.standard_normal((6, 17, 3))).astype(np.float32)
    pl_pred = (rel[:6] + 0.03 * gd.standard_normal((6, 17, 3)))
    pl_pred[:3, :, 2] = 0.0
    pl_pred[3:] = flat(pl_pred[3:], gd.standard_normal(3))
    pl_pred = pl_pred.astype(np.float32)
    pl_gt = rel[6:12].copy()
    pl_gt[:3, :, 2] = 0.0
    pl_gt[3:] = flat(pl_gt[3:], gd.standard_normal(3))
    deg = {}
    for tag, G, P in (("planar_pred", gen_gt, pl_pred), ("planar_gt", pl_gt, gen_pred)):
        e = np.zeros(6)
        Zd = np.zeros((6, 17, 3))
        for n in range(6):
            Zd[n] = procrustes(G[n].copy(), P[n].copy())[1]
            e[n] = np.mean(np.sqrt(np.square(Zd[n] - G[n]).sum(axis=1)))
        deg[f"deg_{tag}_gt"], deg[f"deg_{tag}_pred"], deg[f"deg_{tag}_err_p2"], deg[f"deg_{tag}_aligned"] = G, P, e, Zd
    out.update(deg)
    save("eval_multi", **out)


def gen_driver():
    """BASELINE config 1: N=64, H=1, S=100 through the re-driven opt_main loop."""
    w = syn.make_weights(seed=0)
    m = ref_model(w)
    N, H, S = 64, 1, 100
    d = syn.make_poses(N, seed=41)
    cl = syn.make_clusters(H, seed=8)
    gt_2d, K = d["db_2d"], d["camera_param"]
    batch_results = []
    Ts, Rs = [], []
    for sid in range(H):
        noisy = torch.ones_like(torch.tensor(d["db_3d"])) * torch.tensor(cl - cl[:, 0:1, :])[sid:sid + 1]
        r = run_ref_ipo(noisy.numpy(), gt_2d[:, :, :2], K, "z", [0, 1, 4], 3.0, 0.5, 2.0, 500)
        x = torch.tensor(r["R"]).bmm(noisy.permute(0, 2, 1)).permute(0, 2, 1).contiguous().numpy()
        res, Tf, _ = run_ref_oil(m, x, gt_2d[:, :, :2], gt_2d[:, :, 2], K, r["T"], S, [])
        batch_results.append(res)
        Ts.append(r["T"]); Rs.append(r["R"])
    batch_results = np.swapaxes(np.array(batch_results), 0, 1)
    pw = _pw3d_obj(d["db_3d"])
    p1 = pw.eval_multi(batch_results, protocol2=False)
    p2 = pw.eval_multi(batch_results, protocol2=True)
    save("driver_cfg1", db_2d=gt_2d, db_3d=d["db_3d"], K=K, clusters=cl, ipo_R=np.stack(Rs), ipo_T=np.stack(Ts),
         batch_results=batch_results, mpjpe=np.float64(p1), pa_mpjpe=np.float64(p2))



# ------------------------------------------------------------------ dataset files

ASSETS = os.path.join(OUT, "assets")
PW3D_ORDER = [5, 2, 6, 3, 11, 14, 12, 15, 13, 16, 1, 4, 8, 10, 0, 7, 9]


def write_assets(N=20, seed=77):
    """Small files in the formats the reference's readers parse (h36m.py:206-263, pw3d.py:177-227), filled
    with seeded synthetic poses.  They are DATA made here, not reference content."""
    import pickle
    d = syn.make_poses(N, seed=seed, conf_mode="uniform", dtype3d=np.float64)
    K = d["camera_param"].astype(np.float64)
    mm = d["db_3d"] * 1000.0
    g = np.random.Generator(np.random.Philox(key=[seed, 99]))
    items = []
    for i in range(N):
        img = np.concatenate([d["db_2d"][i, :, :2].astype(np.float64), mm[i, :, 2:3] - mm[i, 0:1, 2:3]], axis=1)
        items.append(dict(joint_3d_camera=mm[i], joint_3d_image=img,
                          camera_param=dict(fx=np.array([K[i, 0, 0]]), fy=np.array([K[i, 1, 1]]),
                                            cx=np.float64(K[i, 0, 2]), cy=np.float64(K[i, 1, 2])),
                          action=int(2 + i % 15), subaction=1 + i % 2, subject=9 + 2 * (i % 2), cam_id=i % 4,
                          image_path=f"s_{9 + 2 * (i % 2):02d}_act_{2 + i % 15:02d}_subact_01_ca_{i % 4 + 1:02d}_{i:06d}.jpg"))
    os.makedirs(os.path.join(ASSETS, "h36m"), exist_ok=True)
    os.makedirs(os.path.join(ASSETS, "3dpw"), exist_ok=True)
    with open(os.path.join(ASSETS, "h36m", "h36m_test.pkl"), "wb") as f:
        pickle.dump(items, f, protocol=4)
    det = d["db_2d"][:, :, :2].astype(np.float64) + 3.0 * g.standard_normal((N, 17, 2))
    dt = dict(test=dict(joint3d_image=np.concatenate([det, np.zeros((N, 17, 1))], axis=2),
                        confidence=d["db_2d"][:, :, 2:3].astype(np.float64)))
    with open(os.path.join(ASSETS, "h36m", "h36m_sh_dt_ft.pkl"), "wb") as f:
        pickle.dump(dt, f, protocol=4)
    # 3DPW: joints stored in the file's own order (order_change scatters file joint i to H36M joint ORDER[i])
    pose = d["db_3d"]                                     # H36M order, metres, camera frame
    in_file = pose[:, PW3D_ORDER, :]
    root = in_file[:, 14, :] + 0.01 * g.standard_normal((N, 3))
    rel = np.concatenate([in_file - root[:, None, :], np.ones((N, 17, 1))], axis=2)
    cam = dict(f=np.stack([K[:, 0, 0], K[:, 1, 1]], 1), c=np.stack([K[:, 0, 2], K[:, 1, 2]], 1))
    np.savez_compressed(os.path.join(ASSETS, "3dpw", "pw3d_test.npz"), keypoints3d17_relative=rel, root_cam=root,
                        cam_param=np.array(cam, dtype=object), image_width=np.full(N, 1920), image_height=np.full(N, 1080),
                        image_path=np.array([f"imageFiles/seq_{i % 3}/image_{i:05d}.jpg" for i in range(N)]))


def gen_datasets():
    write_assets()
    out = {}
    for tag, kw in (("gt", dict(gt2d=True)), ("dt", dict(gt2d=False)), ("gt_s3", dict(gt2d=True, sample_interval=3)),
                    ("dt_rel", dict(gt2d=False, abs_coord=False))):
        kw.setdefault("abs_coord", True)
        ds = H36MDataset3D(os.path.join(ASSETS, "h36m"), "test", flip=False, **kw)
        out[f"h36m_{tag}_db_2d"], out[f"h36m_{tag}_db_3d"], out[f"h36m_{tag}_camera_param"] = ds.db_2d, ds.db_3d, ds.camera_param
        out[f"h36m_{tag}_actions"] = np.array([it["action"] for it in ds.gt_dataset])
        out[f"h36m_{tag}_len"] = np.int64(len(ds))
    for tag, kw in (("abs", dict(abs_coord=True)), ("abs_s4", dict(abs_coord=True, sample_interval=4)), ("rel", dict(abs_coord=False))):
        ds = PW3D(os.path.join(ASSETS, "3dpw"), "test", gt2d=True, flip=False, **kw)
        out[f"pw3d_{tag}_db_2d"], out[f"pw3d_{tag}_db_3d"], out[f"pw3d_{tag}_camera_param"] = ds.db_2d, ds.db_3d, ds.camera_param
        out[f"pw3d_{tag}_w"], out[f"pw3d_{tag}_h"] = ds.w, ds.h
        out[f"pw3d_{tag}_image_name"] = np.array([str(s) for s in ds.image_name])
    save("datasets", **out)


def gen_driver_files():
    """The file-driven evaluation: reference readers on tests/golden/assets -> the re-driven opt_main loop
    (opt_main.py:166-224) with H36M settings -> H36MDataset3D.eval_multi (action-wise).  N=20, H=2, S=60."""
    write_assets()
    w = syn.make_weights(seed=0)
    m = ref_model(w)
    H, S = 2, 60
    out = {}
    for tag, gt2d in (("gt", True), ("dt", False)):
        ds = H36MDataset3D(os.path.join(ASSETS, "h36m"), "test", gt2d=gt2d, abs_coord=True, sample_interval=None, flip=False)
        gt_3d, K, gt_2d = ds.db_3d, ds.camera_param, ds.db_2d
        cl = syn.make_clusters(H, seed=8)
        batch_results = []
        for sid in range(H):
            noisy = (torch.ones_like(torch.tensor(gt_3d)) * torch.tensor(cl - cl[:, 0:1, :])[sid:sid + 1]).float()
            c2 = np.asarray(gt_2d, dtype=np.float32)
            r = run_ref_ipo(noisy.numpy(), c2[:, :, :2], K, "z", [0, 1, 4], 3.0, 0.5, 2.0, 500)
            x = torch.tensor(r["R"]).bmm(noisy.permute(0, 2, 1)).permute(0, 2, 1).contiguous().numpy()
            res, _, _ = run_ref_oil(m, x, c2[:, :, :2], c2[:, :, 2], K, r["T"], S, [])
            batch_results.append(res)
        batch_results = np.swapaxes(np.array(batch_results), 0, 1)
        out[f"{tag}_batch_results"] = batch_results
        out[f"{tag}_mpjpe"] = np.float64(ds.eval_multi(batch_results, protocol2=False))
        out[f"{tag}_pa_mpjpe"] = np.float64(ds.eval_multi(batch_results, protocol2=True))
    out["clusters"] = syn.make_clusters(H, seed=8)
    save("driver_files", **out)



# ------------------------------------------------------------------ other SDEs / update rules

def sampler_cases():
    """(sde name, ctor kwargs) x (predictor, probability_flow) and x (corrector) combinations captured below."""
    sdes = [("vpsde", dict(beta_min=0.1, beta_max=20.0, N=1000, T=1.0)),
            ("subvpsde", dict(beta_min=0.1, beta_max=20.0, N=1000, T=1.0)),
            ("vesde", dict(sigma_min=0.01, sigma_max=50.0, N=1000, T=1.0))]
    preds = [("euler_maruyama", False), ("euler_maruyama", True), ("reverse_diffusion", False),
             ("reverse_diffusion", True), ("ancestral_sampling", False)]
    corrs = ["langevin", "ald"]
    return sdes, preds, corrs


def sampler_inputs():
    g = np.random.Generator(np.random.Philox(key=[2024, 3]))
    x = g.standard_normal((6, 17, 3)).astype(np.float32)
    cond = g.standard_normal((6, 17, 3)).astype(np.float32)
    t = np.array([0.9, 0.5, 0.1, 0.013, 0.0005, 0.7], np.float32)     # 0.0005 -> discrete step 0 (VE adjacent sigma = 0)
    return x, cond, t


class DetNoise:
    """Replacement for torch.randn_like during capture and test: numpy Philox, independent of the torch build."""

    def __init__(self):
        self.calls = 0

    def __call__(self, x):
        g = np.random.Generator(np.random.Philox(key=[555, self.calls]))
        self.calls += 1
        return torch.tensor(g.standard_normal(tuple(x.shape)), dtype=x.dtype)


def analytic_score(x, t, condition, mask):
    return -(x - 0.3 * condition) / (0.5 + t)[:, None, None]


def gen_samplers():
    sdes, preds, corrs = sampler_cases()
    xn, cn, tn = sampler_inputs()
    x, cond, t = torch.tensor(xn), torch.tensor(cn), torch.tensor(tn)
    mask = torch.zeros_like(x)
    out = dict(x=xn, cond=cn, t=tn)
    orig = torch.randn_like
    for name, kw in sdes:
        cls = dict(vpsde=sde_lib.VPSDE, subvpsde=sde_lib.subVPSDE, vesde=sde_lib.VESDE)[name]
        sde = cls(**kw)
        out[f"{name}_drift"], out[f"{name}_diffusion"] = [a.numpy() for a in sde.sde(x, t)]
        out[f"{name}_mean"], out[f"{name}_std"] = [a.numpy() for a in sde.marginal_prob(x, t)]
        f, G = sde.discretize(x, t)
        out[f"{name}_disc_f"], out[f"{name}_disc_G"] = f.numpy(), G.numpy()
        for pf in (False, True):
            r = sde.reverse(analytic_score, pf)
            d, g_ = r.sde(x, t, cond, mask)
            out[f"{name}_rsde_drift_pf{int(pf)}"], out[f"{name}_rsde_diffusion_pf{int(pf)}"] = d.numpy(), g_.numpy()
            f, G = r.discretize(x, t, cond, mask)
            out[f"{name}_rdisc_f_pf{int(pf)}"], out[f"{name}_rdisc_G_pf{int(pf)}"] = f.numpy(), G.numpy()
        for pname, pf in preds:
            key = f"{name}_pred_{pname}_pf{int(pf)}"
            torch.randn_like = DetNoise()
            try:
                xn_, xm = sampling.get_predictor(pname)(sde, analytic_score, pf).update_fn(x, t, cond, mask)
                out[key + "_x"], out[key + "_mean"] = xn_.numpy(), xm.numpy()
            except (NotImplementedError, AssertionError, AttributeError) as e:
                out[key + "_raises"] = np.array(type(e).__name__)
            finally:
                torch.randn_like = orig
        for cname in corrs:
            key = f"{name}_corr_{cname}"
            torch.randn_like = DetNoise()
            try:
                xn_, xm = sampling.get_corrector(cname)(sde, analytic_score, 0.16, 2).update_fn(x, t, cond, mask)
                out[key + "_x"], out[key + "_mean"] = xn_.numpy(), xm.numpy()
            except (NotImplementedError, AssertionError, AttributeError) as e:
                out[key + "_raises"] = np.array(type(e).__name__)
            finally:
                torch.randn_like = orig
    save("samplers", **out)



PC_GENERIC_CASES = [
    # tag, sde, continuous, predictor, corrector, probability_flow, noise_removal, t
    ("vp_rd_langevin", "vpsde", True, "reverse_diffusion", "langevin", False, True, 0.31),
    ("vp_anc_none_disc", "vpsde", False, "ancestral_sampling", "none", False, True, 0.52),
    ("vp_em_none_pf", "vpsde", True, "euler_maruyama", "none", True, True, 0.2),
    ("ve_rd_ald", "vesde", True, "reverse_diffusion", "ald", False, True, 0.4),
    ("ve_anc_langevin", "vesde", True, "ancestral_sampling", "langevin", False, False, 0.15),
    ("subvp_em_none_sde", "subvpsde", True, "euler_maruyama", "none", False, False, 0.07),
    ("subvp_rd_none", "subvpsde", True, "reverse_diffusion", "none", False, True, 0.05),
]


def make_sde(name):
    if name == "vpsde":
        return sde_lib.VPSDE(beta_min=0.1, beta_max=20.0, N=1000, T=1.0)
    if name == "vesde":
        return sde_lib.VESDE(sigma_min=0.01, sigma_max=50.0, N=1000, T=1.0)
    return sde_lib.subVPSDE(beta_min=0.1, beta_max=20.0, N=1000, T=1.0)


def gen_pc_generic():
    """get_sampling_fn with the real network for the config-reachable, non-shipped SDE / predictor / corrector
    combinations (one pc_sampler call each; noise from DetNoise)."""
    m = ref_model(syn.make_weights(seed=0))
    g = np.random.Generator(np.random.Philox(key=[7, 12]))
    x = (0.3 * g.standard_normal((8, 17, 3))).astype(np.float32)
    out = dict(x=x)
    orig = torch.randn_like
    for tag, sname, cont, pred, corr, pf, denoise, t in PC_GENERIC_CASES:
        cfg = ref_config()
        cfg.training.sde, cfg.training.continuous = sname, cont
        cfg.sampling.predictor, cfg.sampling.corrector, cfg.sampling.probability_flow = pred, corr, pf
        cfg.sampling.noise_removal = denoise
        fn = sampling.get_sampling_fn(cfg, make_sde(sname), (8, 17, 3), lambda v: v, 0.01, device=torch.device("cpu"))
        torch.randn_like = DetNoise()
        try:
            trajs, res = fn(m, condition=torch.zeros(8, 17, 2), denoise_x=torch.tensor(x), t=torch.tensor(t), t_step=3)
        finally:
            torch.randn_like = orig
        out[f"{tag}_trajs"] = trajs
        out[f"{tag}_res"] = res if isinstance(res, np.ndarray) else res.numpy()
        out[f"{tag}_res_is_tensor"] = np.bool_(not isinstance(res, np.ndarray))
    save("pc_generic", **out)



# ------------------------------------------------------------------ 3DHP / SkiPose

def write_3dhp_asset(N=45, seed=91):
    import pickle
    d = syn.make_poses(N, seed=seed, dtype3d=np.float64)
    K = d["camera_param"].astype(np.float64)
    mm = d["db_3d"] * 1000.0
    items = []
    for i in range(N):
        j2 = np.concatenate([d["db_2d"][i, :, :2].astype(np.float64), np.zeros((17, 1))], axis=1)
        items.append(dict(joint_3d_camera=mm[i], joint_2d=j2, w=2048, h=2048,
                          camera_param=dict(fx=K[i, 0, 0], fy=K[i, 1, 1], cx=K[i, 0, 2], cy=K[i, 1, 2]),
                          imageid=f"TS{1 + i % 6}/imageSequence/img_{i:06d}.jpg", valid_i=float(i % 5 != 3),
                          action=1 + i % 7))
    os.makedirs(os.path.join(ASSETS, "3dhp"), exist_ok=True)
    with open(os.path.join(ASSETS, "3dhp", "mpii3d_test.pkl"), "wb") as f:
        pickle.dump(items, f, protocol=4)


def write_ski_asset(N=24, seed=93):
    """ski_test.h5 with the dataset keys the reader parses (skiPose.py:119-157): per frame 3D [51] (metres,
    camera frame), 2D [34] in 0..1 of the 256-pixel crop, cam_intrinsic [3,3] in crop units, seq / cam / frame.
    A genuine HDF5 file, written through libhdf5 (tools/ref_stubs/h5py.py binds the C library; h5py is not installed)."""
    import h5py
    d = syn.make_poses(N, seed=seed, dtype3d=np.float64)
    K = d["camera_param"].astype(np.float64)
    g = np.random.Generator(np.random.Philox(key=[seed, 7]))
    cam = K / 256.0
    cam[:, 2, 2] = 1.0 / 256.0 + 0.001 * g.standard_normal(N)       # the reader overwrites [2,2] with 1
    os.makedirs(os.path.join(ASSETS, "ski"), exist_ok=True)
    with h5py.File(os.path.join(ASSETS, "ski", "ski_test.h5"), "w") as f:
        for name, arr in (("3D", d["db_3d"].reshape(N, 51).astype(np.float32)),
                          ("2D", (d["db_2d"][:, :, :2].astype(np.float64) / 256.0).reshape(N, 34)),
                          ("cam_intrinsic", cam), ("seq", (np.arange(N) % 3).astype(np.float64)),
                          ("cam", (np.arange(N) % 6).astype(np.int64)), ("frame", (7 * np.arange(N)).astype(np.int64))):
            f.create_dataset(name, data=arr)


def gen_3dhp_ski():
    import contextlib
    import io
    import re
    from lib.dataset.mpii3dHP import MPII3DHP
    from lib.dataset.skiPose import skiPose
    write_3dhp_asset()
    out = {}
    for tag, kw in (("all", dict()), ("s2", dict(sample_interval=2)), ("rel_s3", dict(sample_interval=3, abs_coord=False))):
        kw.setdefault("abs_coord", True)
        ds = MPII3DHP(os.path.join(ASSETS, "3dhp"), "test", gt2d=True, flip=False, **kw)
        out[f"hp_{tag}_db_2d"], out[f"hp_{tag}_db_3d"], out[f"hp_{tag}_camera_param"] = ds.db_2d, ds.db_3d, ds.camera_param
        out[f"hp_{tag}_valid_id"] = ds.valid_id
        out[f"hp_{tag}_actions"] = np.array([it["action"] for it in ds.gt_dataset])
        out[f"hp_{tag}_image_path"] = np.array([str(s) for s in ds.image_path])
    # eval_multi on the valid, every-2nd subset: best-of-H action-wise error + the printed PCK / AUC
    ds = MPII3DHP(os.path.join(ASSETS, "3dhp"), "test", gt2d=True, abs_coord=True, sample_interval=2, flip=False)
    N, H = len(ds.db_3d), 4
    g = np.random.Generator(np.random.Philox(key=[91, 5]))
    rel = ds.db_3d - ds.db_3d[:, 0:1]
    preds = (rel[:, None] + 0.06 * g.standard_normal((N, H, 17, 3))).astype(np.float32)
    out["hp_preds"] = preds
    for proto in (False, True):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            err = ds.eval_multi(preds.copy(), protocol2=proto)
        txt = buf.getvalue()
        k = "p2" if proto else "p1"
        assert np.isfinite(err), "an action of the table has no sample"
        out[f"hp_{k}"] = np.float64(err)
        out[f"hp_{k}_pck"] = np.float64(re.search(r"PCK : ([0-9.eE+-]+)", txt).group(1))
        out[f"hp_{k}_auc"] = np.float64(re.search(r"AUC : ([0-9.eE+-]+)", txt).group(1))
    # PCK / AUC helpers directly
    a = (0.08 * g.standard_normal((30, 17, 3))).astype(np.float32)
    b = (a + 0.07 * g.standard_normal((30, 17, 3))).astype(np.float32)
    out["pck_gts"], out["pck_preds"] = a, b
    out["pck_150"] = np.float64(mutils.compute_PCK(a, b))
    out["pck_50_joints"] = np.float64(mutils.compute_PCK(a, b, eval_joints=[1, 2, 3, 14, 15, 16], threshold=50))
    out["auc"] = np.float64(mutils.compute_AUC(a, b))
    out["auc_joints"] = np.float64(mutils.compute_AUC(a, b, eval_joints=[0, 7, 8, 9, 10]))
    # SkiPose reader (skiPose.py:119-157) through the h5py stand-in on a synthetic ski_test.h5
    write_ski_asset()
    for tag, kw in (("abs", dict(abs_coord=True)), ("rel_s5", dict(abs_coord=False, sample_interval=5))):
        sds = skiPose(os.path.join(ASSETS, "ski"), "test", gt2d=True, flip=False, **kw)
        out[f"skir_{tag}_db_2d"], out[f"skir_{tag}_db_3d"], out[f"skir_{tag}_camera_param"] = sds.db_2d, sds.db_3d, sds.camera_param
        out[f"skir_{tag}_image_name"] = np.array([str(n) for n in sds.image_name])
    # SkiPose eval_multi
    sk = object.__new__(skiPose)
    sk.db_3d = (ds.db_3d + 0.5).astype(np.float32)
    out["ski_db_3d"] = sk.db_3d
    out["ski_p1"] = np.float64(sk.eval_multi(preds, protocol2=False))
    out["ski_p2"] = np.float64(sk.eval_multi(preds, protocol2=True))
    save("hp3d_ski", **out)



def gen_driver_full():
    """The full-length loop (S = 1000, H = 3) with the 3DPW settings of BASELINE configs[2] on N = 160 poses:
    reference loop + PW3D.eval_multi.  ~2 minutes of CPU."""
    w = syn.make_weights(seed=0)
    m = ref_model(w)
    N, H, S = 160, 3, 1000
    d = syn.make_poses(N, seed=57, conf_mode="uniform")
    cl = syn.make_clusters(H, seed=13)
    gt_2d, K = d["db_2d"], d["camera_param"]
    batch_results = []
    for sid in range(H):
        noisy = torch.ones_like(torch.tensor(d["db_3d"])) * torch.tensor(cl - cl[:, 0:1, :])[sid:sid + 1]
        r = run_ref_ipo(noisy.numpy(), gt_2d[:, :, :2], K, "z", list(range(17)), 8.0, 0.2, 2.0, 500)
        x = torch.tensor(r["R"]).bmm(noisy.permute(0, 2, 1)).permute(0, 2, 1).contiguous().numpy()
        res, _, _ = run_ref_oil(m, x, gt_2d[:, :, :2], gt_2d[:, :, 2].copy(), K, r["T"], S, [])
        batch_results.append(res)
    batch_results = np.swapaxes(np.array(batch_results), 0, 1)
    pw = _pw3d_obj(d["db_3d"])
    p1 = pw.eval_multi(batch_results, protocol2=False)
    p2 = pw.eval_multi(batch_results, protocol2=True)
    save("driver_full", db_2d=gt_2d, db_3d=d["db_3d"], K=K, clusters=cl, mpjpe=np.float64(p1), pa_mpjpe=np.float64(p2),
         batch_results=batch_results.astype(np.float32))


def gen_driver_full_env():
    """The reference's own fp32 reproducibility on the gen_driver_full problem (N = 160, H = 3, S = 1000, 3DPW settings): the
    same run on detections moved by -1/0/+1 ulp (syn.perturb_ulp streams 1..6) - dataset means only."""
    w = syn.make_weights(seed=0)
    m = ref_model(w)
    N, H, S = 160, 3, 1000
    d = syn.make_poses(N, seed=57, conf_mode="uniform")
    cl = syn.make_clusters(H, seed=13)
    K = d["camera_param"]
    pw = _pw3d_obj(d["db_3d"])
    p1s, p2s = [], []
    for run in range(1, 7):
        gt_2d = d["db_2d"].copy()
        gt_2d[:, :, :2] = syn.perturb_ulp(gt_2d[:, :, :2], run)
        batch_results = []
        for sid in range(H):
            noisy = torch.ones_like(torch.tensor(d["db_3d"])) * torch.tensor(cl - cl[:, 0:1, :])[sid:sid + 1]
            r = run_ref_ipo(noisy.numpy(), gt_2d[:, :, :2], K, "z", list(range(17)), 8.0, 0.2, 2.0, 500, trace_upto=1)
            x = torch.tensor(r["R"]).bmm(noisy.permute(0, 2, 1)).permute(0, 2, 1).contiguous().numpy()
            res, _, _ = run_ref_oil(m, x, gt_2d[:, :, :2], gt_2d[:, :, 2].copy(), K, r["T"], S, [])
            batch_results.append(res)
        br = np.swapaxes(np.array(batch_results), 0, 1)
        p1s.append(pw.eval_multi(br, protocol2=False))
        p2s.append(pw.eval_multi(br, protocol2=True))
        print(f"  driver_full env member {run}: {p1s[-1]:.6f} {p2s[-1]:.6f}", flush=True)
    save("driver_full_env", members=np.arange(1, 7), mpjpe=np.array(p1s, np.float64), pa_mpjpe=np.array(p2s, np.float64))


def _driver_full_size(tag, N, H, S, seed_pose, seed_cl, keylist, ipo_T, minT, conf_mode, dataset, cache_dir,
                      dtype=torch.float32, perturb=0, weights=None):
    """opt_main.py:166-228 at a BASELINE configuration's stated size.  One hypothesis at a time like the reference;
    each finished hypothesis is parked under cache_dir so that an interrupted capture resumes.  Only small arrays
    are committed: the per-(pose, hypothesis) errors, per-pose best / argmin, dataset means, the IPO outcome as
    (rotation angle about z, depth scale) and the seeds of the inputs."""
    w = weights if weights is not None else syn.make_weights(seed=0)
    m = ref_model(w, dtype)
    f64 = dtype == torch.float64          # the arbiter run: every tensor, RotOpt and the network in double
    d = syn.make_poses(N, seed=seed_pose, conf_mode=conf_mode, dtype3d=np.float64 if dataset == "h36m" else np.float32)
    cl = syn.make_clusters(H, seed=seed_cl)
    gt_2d, K = d["db_2d"], d["camera_param"]
    if perturb:      # detections moved by -1/0/+1 ulp (syn.perturb_ulp): a member of the reference's own fp32 ensemble
        gt_2d = gt_2d.copy()
        gt_2d[:, :, :2] = syn.perturb_ulp(gt_2d[:, :, :2], perturb)
    os.makedirs(cache_dir, exist_ok=True)
    batch_results, ang, scl, loss, cs_all, T_all = [], [], [], [], [], []
    import time
    if os.environ.get("ZEDO_GOLDEN_FILL") == "reverse":
        # helper process: fill the per-hypothesis cache from the LAST hypothesis downwards and stop - a second core for a
        # capture another process is running forwards (it finds the finished hypotheses in the cache when it gets there)
        for sid in reversed(range(H)):
            f = os.path.join(cache_dir, f"{tag}_h{sid:02d}.npz")
            if os.path.exists(f) or os.path.exists(f + ".claim"):
                continue
            open(f + ".claim", "w").close()
            t0 = time.time()
            noisy = (torch.ones((N, 17, 3)) * torch.tensor(cl - cl[:, 0:1, :])[sid:sid + 1]).to(dtype)
            r = run_ref_ipo(noisy.numpy(), gt_2d[:, :, :2], K, "z", keylist, ipo_T, minT, 2.0, 500, trace_upto=1, dtype=dtype)
            x = torch.tensor(r["R"]).bmm(noisy.permute(0, 2, 1)).permute(0, 2, 1).contiguous().numpy()
            res, _, _ = run_ref_oil(m, x, gt_2d[:, :, :2], gt_2d[:, :, 2].copy(), K, r["T"], S, [], dtype)
            np.savez(f + ".tmp.npz", res=res, R=r["R"], T=r["T"], T0=r["T0"], loss=r["loss"])
            os.replace(f + ".tmp.npz", f)
            os.remove(f + ".claim")
            print(f"  {tag}: (helper) hypothesis {sid + 1}/{H} in {time.time() - t0:.0f} s", flush=True)
        return
    for sid in range(H):
        f = os.path.join(cache_dir, f"{tag}_h{sid:02d}.npz")
        if os.path.exists(f):
            z = np.load(f)
            res, R, Tf, T0, ls = z["res"], z["R"], z["T"], z["T0"], z["loss"]
        else:
            t0 = time.time()
            noisy = (torch.ones((N, 17, 3)) * torch.tensor(cl - cl[:, 0:1, :])[sid:sid + 1]).to(dtype)
            r = run_ref_ipo(noisy.numpy(), gt_2d[:, :, :2], K, "z", keylist, ipo_T, minT, 2.0, 500, trace_upto=1, dtype=dtype)
            x = torch.tensor(r["R"]).bmm(noisy.permute(0, 2, 1)).permute(0, 2, 1).contiguous().numpy()
            res, _, _ = run_ref_oil(m, x, gt_2d[:, :, :2], gt_2d[:, :, 2].copy(), K, r["T"], S, [], dtype)
            R, Tf, T0, ls = r["R"], r["T"], r["T0"], r["loss"]
            np.savez(f, res=res, R=R, T=Tf, T0=T0, loss=ls)
            print(f"  {tag}: hypothesis {sid + 1}/{H} in {time.time() - t0:.0f} s", flush=True)
        batch_results.append(res)
        cs_all.append(np.stack([R[:, 0, 0], R[:, 1, 0]], -1))
        T_all.append(Tf[:, 0, :])
        ang.append(np.arctan2(R[:, 1, 0], R[:, 0, 0]))
        scl.append(Tf[:, 0, 2] / T0[:, 0, 2])
        loss.append(ls)
    batch_results = np.swapaxes(np.array(batch_results), 0, 1)           # [N, H, 17, 3]
    if dataset == "h36m":
        gt_mm = d["db_3d"] * 1000.0
        actions = 2 + (np.arange(N) % 15)
        ds = _h36m_obj(gt_mm, actions)
        gtc = ((gt_mm - gt_mm[:, 0:1]) / 1000.0)
    else:
        ds = _pw3d_obj(d["db_3d"])
        gtc = d["db_3d"] - d["db_3d"][:, 0:1]
    p1 = ds.eval_multi(batch_results, protocol2=False)
    p2 = ds.eval_multi(batch_results, protocol2=True)
    # per (n, h) errors with the reference's own inner statements (h36m.py:402-408 / pw3d.py:318-326)
    e1 = np.zeros((N, H))
    e2 = np.zeros((N, H))
    for n in range(N):
        for h in range(H):
            e1[n, h] = np.mean(np.sqrt(np.square(batch_results[n, h] - gtc[n]).sum(axis=1)))
            Z = procrustes(gtc[n].copy(), batch_results[n, h].copy())[1]
            e2[n, h] = np.mean(np.sqrt(np.square(Z - gtc[n]).sum(axis=1)))
    if f64:       # arbiter: only the metric side is kept
        save(tag, N=np.int64(N), H=np.int64(H), S=np.int64(S), mpjpe=np.float64(p1), pa_mpjpe=np.float64(p2),
             best_p1=e1.min(1), best_p2=e2.min(1), argmin_p1=e1.argmin(1).astype(np.int32),
             argmin_p2=e2.argmin(1).astype(np.int32), inputs_sha=np.array(_sha(gt_2d, K, cl)))
        return
    if perturb or weights is not None:
        # an ensemble member / another prior: the metric side + the IPO end state as quantile functions (tests/_ipo_summary.py)
        import _ipo_summary as ips
        x0c = (cl - cl[:, 0:1, :])[:, None]
        sm = ips.summary(np.stack(cs_all), np.stack(T_all), x0c, gt_2d[:, :, :2], K, keylist, ipo_T)
        save(tag, N=np.int64(N), H=np.int64(H), S=np.int64(S), perturb=np.int64(perturb), mpjpe=np.float64(p1), pa_mpjpe=np.float64(p2),
             best_p1=e1.min(1).astype(np.float32), best_p2=e2.min(1).astype(np.float32),
             argmin_p1=e1.argmin(1).astype(np.int8), argmin_p2=e2.argmin(1).astype(np.int8),
             ipo_loss=np.array(loss, np.float32), inputs_sha=np.array(_sha(gt_2d, K, cl)),
             weights_sha=np.array(syn.weights_checksum(w)), **sm)
        return
    save(tag, N=np.int64(N), H=np.int64(H), S=np.int64(S), seed_pose=np.int64(seed_pose), seed_cl=np.int64(seed_cl),
         conf_mode=np.array(conf_mode), dataset=np.array(dataset), keylist=np.array(keylist), ipo_T=np.float64(ipo_T),
         minT=np.float64(minT), mpjpe=np.float64(p1), pa_mpjpe=np.float64(p2),
         err_p1=e1.astype(np.float32), err_p2=e2.astype(np.float32),
         best_p1=e1.min(1), best_p2=e2.min(1), argmin_p1=e1.argmin(1).astype(np.int32), argmin_p2=e2.argmin(1).astype(np.int32),
         ipo_angle=np.stack(ang).astype(np.float32), ipo_scale=np.stack(scl).astype(np.float32),
         ipo_loss=np.array(loss, np.float32), inputs_sha=np.array(_sha(gt_2d, K, cl)))


def _sha(*arrs):
    import hashlib
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


CACHE = os.environ.get("ZEDO_GOLDEN_CACHE", "/tmp/zedo_golden_cache")


def gen_driver_h36m_full():
    """BASELINE configs[1] at its stated size: N = 886 (the H36M test set at ZeDO.sample = 640,
    configs/optim/concat_pose_optimization_h36m.py:72-81), H = 1, S = 1000, key list [0,1,4], IPO_T 3,
    H36MDataset3D.eval_multi (action-wise, millimetre float64 ground truth).  ~1 CPU-minute."""
    _driver_full_size("driver_h36m_full", 886, 1, 1000, 101, 17, [0, 1, 4], 3.0, 0.5, "uniform", "h36m", CACHE)


def gen_driver_h36m_full_f64():
    """The same run with the reference in float64 (arbiter for the dataset means: how far is the reference's own fp32
    run from exact arithmetic on this chaotic loop?)."""
    _driver_full_size("driver_h36m_full_f64", 886, 1, 1000, 101, 17, [0, 1, 4], 3.0, 0.5, "uniform", "h36m", CACHE,
                      dtype=torch.float64)


def gen_driver_pw3d_full_f64():
    """float64 arbiter of configs[2] (about twice the CPU time of the fp32 capture); --only driver_pw3d_full_f64."""
    _driver_full_size("driver_pw3d_full_f64", 1015, 50, 1000, 103, 19, list(range(17)), 8.0, 0.2, "uniform", "3dpw", CACHE,
                      dtype=torch.float64)


def gen_driver_pw3d_full_b():
    """A second, independent draw of configs[2]'s shape (other poses, other clusters, confidence 1 everywhere): one
    dataset mean is one sample of a chaotic quantity; a second one tells a bias from a fluctuation."""
    _driver_full_size("driver_pw3d_full_b", 1015, 50, 1000, 203, 29, list(range(17)), 8.0, 0.2, "ones", "3dpw", CACHE)


def gen_driver_pw3d_full_c():
    """Third independent draw (uniform confidences, other seeds)."""
    _driver_full_size("driver_pw3d_full_c", 1015, 50, 1000, 307, 31, list(range(17)), 8.0, 0.2, "uniform", "3dpw", CACHE)


def gen_driver_pw3d_full():
    """BASELINE configs[2] at its stated size: N = 1015, H = 50, S = 1000, 17-joint key list, IPO_T 8
    (configs/optim/concat_pose_optimization_pw3d.py:72-81), PW3D.eval_multi.  ~45 CPU-minutes: run once
    (python tools/gen_golden.py --only driver_pw3d_full); excluded from the default sweep."""
    _driver_full_size("driver_pw3d_full", 1015, 50, 1000, 103, 19, list(range(17)), 8.0, 0.2, "uniform", "3dpw", CACHE)


def gen_driver_pw3d_full_env():
    """The reference's OWN fp32 reproducibility on configs[2]: the capture of gen_driver_pw3d_full repeated on detections
    moved by -1/0/+1 ulp (syn.perturb_ulp, stream ZEDO_ENV_RUN = 1, 2, ...).  Thread count is NOT such a perturbation: the
    reference's IPO and loop are bit-identical on 1, 4 and 8 threads here (probed).  One member = 40 CPU-minutes on 8
    threads, 2.3 h on one; run the members as separate single-thread processes:
        ZEDO_ENV_RUN=3 ZEDO_GOLDEN_THREADS=1 python tools/gen_golden.py --only driver_pw3d_full_env"""
    run = int(os.environ["ZEDO_ENV_RUN"])
    assert run > 0
    draw = os.environ.get("ZEDO_ENV_DRAW", "a")          # a | b | c: which configs[2] capture (default: driver_pw3d_full)
    tag, seed_pose, seed_cl, conf_mode = {"a": ("driver_pw3d_full", 103, 19, "uniform"), "b": ("driver_pw3d_full_b", 203, 29, "ones"),
                                          "c": ("driver_pw3d_full_c", 307, 31, "uniform")}[draw]
    _driver_full_size(f"{tag}_env{run}", 1015, 50, 1000, seed_pose, seed_cl, list(range(17)), 8.0, 0.2, conf_mode, "3dpw", CACHE,
                      perturb=run)


def gen_driver_h36m_full_env():
    """The reference's own fp32 reproducibility on BASELINE configs[1] (gen_driver_h36m_full: 886 poses, one hypothesis, H36M
    settings, action-wise means): the same run on detections moved by -1/0/+1 ulp, stream ZEDO_ENV_RUN = 1, 2, ...  3 CPU-minutes
    per member on one thread."""
    run = int(os.environ["ZEDO_ENV_RUN"])
    assert run > 0
    _driver_full_size(f"driver_h36m_full_env{run}", 886, 1, 1000, 101, 17, [0, 1, 4], 3.0, 0.5, "uniform", "h36m", CACHE, perturb=run)


def gen_driver_pw3d_full_tied():
    """configs[2] with a CONTRACTIVE prior (syn.make_weights(prior="tied")): does the spread of the end-to-end MPJPE between
    two fp32 implementations - the IPO's chaotic last iterate carried through an expansive loop - collapse when the denoiser
    pulls towards an attractor, as a trained one does?  Same inputs as gen_driver_pw3d_full.  2.4 h on one thread."""
    _driver_full_size("driver_pw3d_full_tied", 1015, 50, 1000, 103, 19, list(range(17)), 8.0, 0.2, "uniform", "3dpw", CACHE,
                      weights=syn.make_weights(seed=0, prior="tied"))


def gen_driver_pw3d_ipoens():
    """The reference's IPO END STATE as a distribution: the 500 Adam iterations of run/opt_main.py:180-195 on configs[2]
    (50 hypotheses x 1015 poses) for ZEDO_IPOENS_MEMBERS ulp-perturbed copies of the detections (syn.perturb_ulp streams
    1..M; the IPO costs 2 s per hypothesis on a CPU, the 1000-step loop is not run), each summarised by the quantile
    functions of tests/_ipo_summary.py.  ZEDO_IPOENS_DRAW = a | b | c selects the capture (default a)."""
    import _ipo_summary as ips
    draw = os.environ.get("ZEDO_IPOENS_DRAW", "a")
    M = int(os.environ.get("ZEDO_IPOENS_MEMBERS", "16"))
    tag, seed_pose, seed_cl, conf_mode = {"a": ("driver_pw3d_full", 103, 19, "uniform"), "b": ("driver_pw3d_full_b", 203, 29, "ones"),
                                          "c": ("driver_pw3d_full_c", 307, 31, "uniform")}[draw]
    N, H, keylist, ipo_T, minT = 1015, 50, list(range(17)), 8.0, 0.2
    d = syn.make_poses(N, seed=seed_pose, conf_mode=conf_mode)
    cl = syn.make_clusters(H, seed=seed_cl)
    K = d["camera_param"]
    assert str(np.load(os.path.join(OUT, tag + ".npz"))["inputs_sha"]) == _sha(d["db_2d"], K, cl)
    x0c = (cl - cl[:, 0:1, :])[:, None]
    rows = []
    import time
    for run in range(1, M + 1):
        t0 = time.time()
        uv = syn.perturb_ulp(d["db_2d"][:, :, :2], run)
        cs, Ts = [], []
        for sid in range(H):
            noisy = (torch.ones((N, 17, 3)) * torch.tensor(cl - cl[:, 0:1, :])[sid:sid + 1])
            r = run_ref_ipo(noisy.numpy(), uv, K, "z", keylist, ipo_T, minT, 2.0, 500, trace_upto=1)
            cs.append(np.stack([r["R"][:, 0, 0], r["R"][:, 1, 0]], -1))
            Ts.append(r["T"][:, 0, :])
        rows.append(ips.summary(np.stack(cs), np.stack(Ts), x0c, uv, K, keylist, ipo_T))
        print(f"  {tag} ipo ensemble: member {run}/{M} in {time.time() - t0:.0f} s", flush=True)
    save(tag + "_ipoens", members=np.arange(1, M + 1), inputs_sha=np.array(_sha(d["db_2d"], K, cl)),
         **{k: np.stack([np.asarray(r[k]) for r in rows]) for k in rows[0]})


def _driver_ipo_pin(tag, N, H, seed_pose, seed_cl, keylist, ipo_T, minT, conf_mode, dataset, cache_dir):
    """SURVEY 7 parity stage (A) at BASELINE size: the reference's own IPO OUTPUT of the run captured by
    _driver_full_size(tag, ...) - per (hypothesis, pose) the rotation about z as (cos, sin) = (R[0,0], R[1,0]) exactly as
    RotOpt.generate_matrix() returned it (the other entries of R are exact 0 / 1 / copies: asserted) and
    T = T0 * clamp(scale) (opt_main.py:194-195), both float32 - so that the OIL loop can be run from the REFERENCE's
    (R, T) and compared with the reference's final poses without the IPO's chaotic last iterate in between.
    Read from the per-hypothesis cache of the full capture; a missing cache entry re-runs the (deterministic) IPO."""
    d = syn.make_poses(N, seed=seed_pose, conf_mode=conf_mode, dtype3d=np.float64 if dataset == "h36m" else np.float32)
    cl = syn.make_clusters(H, seed=seed_cl)
    gt_2d, K = d["db_2d"], d["camera_param"]
    full = np.load(os.path.join(OUT, tag + ".npz"))
    assert str(full["inputs_sha"]) == _sha(gt_2d, K, cl)
    cs, Ts = [], []
    for sid in range(H):
        f = os.path.join(cache_dir, f"{tag}_h{sid:02d}.npz")
        if os.path.exists(f):
            z = np.load(f)
            R, Tf = z["R"], z["T"]
        else:
            noisy = (torch.ones((N, 17, 3)) * torch.tensor(cl - cl[:, 0:1, :])[sid:sid + 1])
            r = run_ref_ipo(noisy.numpy(), gt_2d[:, :, :2], K, "z", keylist, ipo_T, minT, 2.0, 500, trace_upto=1)
            R, Tf = r["R"], r["T"]
        assert R.dtype == np.float32 and Tf.dtype == np.float32
        assert (np.array_equal(R[:, 0, 0], R[:, 1, 1]) and np.array_equal(R[:, 0, 1], -R[:, 1, 0]) and np.all(R[:, 2, 2] == 1)
                and not R[:, :2, 2].any() and not R[:, 2, :2].any())
        # the capture's own summary of this IPO run must be what the cache holds
        assert np.array_equal(np.arctan2(R[:, 1, 0], R[:, 0, 0]).astype(np.float32), full["ipo_angle"][sid])
        cs.append(np.stack([R[:, 0, 0], R[:, 1, 0]], -1))
        Ts.append(Tf[:, 0, :])
    save(tag + "_ipo", cs=np.stack(cs).astype(np.float32), T=np.stack(Ts).astype(np.float32), inputs_sha=full["inputs_sha"])


def gen_driver_ipo_pins():
    _driver_ipo_pin("driver_h36m_full", 886, 1, 101, 17, [0, 1, 4], 3.0, 0.5, "uniform", "h36m", CACHE)
    _driver_ipo_pin("driver_pw3d_full", 1015, 50, 103, 19, list(range(17)), 8.0, 0.2, "uniform", "3dpw", CACHE)
    _driver_ipo_pin("driver_pw3d_full_b", 1015, 50, 203, 29, list(range(17)), 8.0, 0.2, "ones", "3dpw", CACHE)
    _driver_ipo_pin("driver_pw3d_full_c", 1015, 50, 307, 31, list(range(17)), 8.0, 0.2, "uniform", "3dpw", CACHE)



def _driver_oil_f64_from_pins(tag, N, H, S, seed_pose, seed_cl, conf_mode, cache_dir, h36m=False):
    """Arbiter of parity stage (A): the reference's OIL loop (opt_main.py:197-222) re-run in float64 FROM THE SAME
    (R, T) the reference's fp32 run produced (tests/golden/<tag>_ipo.npz) - i.e. exact arithmetic on the inputs the fp32
    reference run and the HIP loop both start from.  How far the reference's own fp32 loop drifts from it is the
    yardstick for the HIP loop's drift.  Per-(pose, hypothesis) errors against the root-centred ground truth (3DPW: float32
    metres; h36m=True: float64 millimetres centred the way h36m.py:400-401 does).  ~80 CPU-minutes for configs[2]; resumable."""
    w = syn.make_weights(seed=0)
    m = ref_model(w, torch.float64)
    d = syn.make_poses(N, seed=seed_pose, conf_mode=conf_mode, dtype3d=np.float64 if h36m else np.float32)
    cl = syn.make_clusters(H, seed=seed_cl)
    gt_2d, K = d["db_2d"], d["camera_param"]
    pin = np.load(os.path.join(OUT, tag + "_ipo.npz"))
    assert str(pin["inputs_sha"]) == _sha(gt_2d, K, cl)
    os.makedirs(cache_dir, exist_ok=True)
    import time
    res_all = []
    for sid in range(H):
        f = os.path.join(cache_dir, f"{tag}_oil64_h{sid:02d}.npz")
        if os.path.exists(f):
            res = np.load(f)["res"]
        else:
            t0 = time.time()
            c, s_ = pin["cs"][sid, :, 0], pin["cs"][sid, :, 1]
            R = np.zeros((N, 3, 3), np.float32)
            R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = c, -s_, s_, c, 1
            noisy = (torch.ones((N, 17, 3)) * torch.tensor(cl - cl[:, 0:1, :])[sid:sid + 1])
            x = torch.tensor(R).bmm(noisy.permute(0, 2, 1)).permute(0, 2, 1).contiguous().numpy()     # fp32, opt_main.py:201
            res, _, _ = run_ref_oil(m, x, gt_2d[:, :, :2], gt_2d[:, :, 2].copy(), K, pin["T"][sid][:, None, :], S, [],
                                    torch.float64)
            np.savez(f, res=res)
            print(f"  {tag} oil64: hypothesis {sid + 1}/{H} in {time.time() - t0:.0f} s", flush=True)
        res_all.append(res)
    br = np.swapaxes(np.array(res_all), 0, 1)
    if h36m:
        mm = d["db_3d"] * 1000.0
        gtc = (mm - mm[:, 0:1]) / 1000.0
    else:
        gtc = (d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float64)
    e1 = np.linalg.norm(br - gtc[:, None], axis=-1).mean(-1)
    e2 = np.zeros((N, H))
    for n in range(N):
        for h in range(H):
            Z = procrustes(gtc[n].copy(), br[n, h].copy())[1]
            e2[n, h] = np.mean(np.sqrt(np.square(Z - gtc[n]).sum(axis=1)))
    save(tag + "_oil64", N=np.int64(N), H=np.int64(H), S=np.int64(S), err_p1=e1.astype(np.float32), err_p2=e2.astype(np.float32),
         best_p1=e1.min(1), best_p2=e2.min(1), argmin_p1=e1.argmin(1).astype(np.int32), argmin_p2=e2.argmin(1).astype(np.int32),
         mpjpe=np.float64(e1.min(1).mean()), pa_mpjpe=np.float64(e2.min(1).mean()), inputs_sha=pin["inputs_sha"])


def gen_driver_h36m_full_oil64():
    _driver_oil_f64_from_pins("driver_h36m_full", 886, 1, 1000, 101, 17, "uniform", CACHE, h36m=True)


def gen_driver_pw3d_full_oil64():
    _driver_oil_f64_from_pins("driver_pw3d_full", 1015, 50, 1000, 103, 19, "uniform", CACHE)


def gen_driver_pw3d_full_b_oil64():
    _driver_oil_f64_from_pins("driver_pw3d_full_b", 1015, 50, 1000, 203, 29, "ones", CACHE)


def gen_driver_pw3d_full_c_oil64():
    _driver_oil_f64_from_pins("driver_pw3d_full_c", 1015, 50, 1000, 307, 31, "uniform", CACHE)



GENS = dict(model=gen_model, weights_alt=gen_weights_alt, pc_step=gen_pc_step, reproj=gen_reproj, ipo=gen_ipo, ipo_custom=gen_ipo_custom, oil=gen_oil,
            eval=gen_eval, driver=gen_driver, datasets=gen_datasets,
            driver_files=gen_driver_files, samplers=gen_samplers, pc_generic=gen_pc_generic, hp3d_ski=gen_3dhp_ski, driver_full=gen_driver_full,
            driver_h36m_full=gen_driver_h36m_full, driver_pw3d_full=gen_driver_pw3d_full,
            driver_h36m_full_f64=gen_driver_h36m_full_f64, driver_pw3d_full_f64=gen_driver_pw3d_full_f64,
            driver_pw3d_full_b=gen_driver_pw3d_full_b, driver_pw3d_full_c=gen_driver_pw3d_full_c,
            driver_ipo_pins=gen_driver_ipo_pins, driver_h36m_full_oil64=gen_driver_h36m_full_oil64, driver_pw3d_full_oil64=gen_driver_pw3d_full_oil64, driver_pw3d_full_b_oil64=gen_driver_pw3d_full_b_oil64,
            driver_pw3d_full_c_oil64=gen_driver_pw3d_full_c_oil64, driver_pw3d_full_env=gen_driver_pw3d_full_env,
            driver_pw3d_ipoens=gen_driver_pw3d_ipoens, driver_pw3d_full_tied=gen_driver_pw3d_full_tied, driver_full_env=gen_driver_full_env, driver_h36m_full_env=gen_driver_h36m_full_env)
SLOW = {"driver_h36m_full_oil64", "driver_pw3d_full", "driver_pw3d_full_f64", "driver_pw3d_full_b", "driver_pw3d_full_c", "driver_ipo_pins", "driver_pw3d_full_oil64", "driver_pw3d_full_b_oil64",
        "driver_pw3d_full_c_oil64", "driver_pw3d_full_env", "driver_pw3d_ipoens", "driver_pw3d_full_tied", "driver_full_env", "driver_h36m_full_env"}     # only with --only

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for k, f in GENS.items():
        if a.only == k or (a.only is None and k not in SLOW):
            print("==", k)
            f()
    for p in [os.path.join(dp, f) for dp, _, fs in os.walk("/root/reference") for f in fs if f.endswith(".pyc")]:
        print("WARNING: bytecode written into the reference tree:", p)
