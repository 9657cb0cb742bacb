"""ctypes binding of libzedo_hip.so (C ABI: include/zedo_hip.h) for torch tensors on an MI355X.

PyTorch is plumbing here: it owns device memory and the HIP stream; every arithmetic step of the
ZeDO hot path runs in the hand-written gfx950 kernels behind the C ABI.  There is NO fallback:
importing this module without the built library, or calling it without a GPU, raises.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load_zedo_build():
    """zedo_build.py sits beside this package; it is loaded BY LOCATION so that importing zedo_hip never edits sys.path
    (the package root also holds the generic top-level names `lib` and `run`, which must not shadow a host application's)."""
    import importlib.util
    import sys
    if "zedo_build" in sys.modules:
        return sys.modules["zedo_build"]
    spec = importlib.util.spec_from_file_location("zedo_build", os.path.join(os.path.dirname(_HERE), "zedo_build.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["zedo_build"] = mod
    spec.loader.exec_module(mod)
    return mod


_zb = _load_zedo_build()

# Not a fallback: the only way to get the hot path is to compile the HIP sources (hipcc cross-compiles gfx950 without a
# GPU).  A missing library is built once, under a lock (N ranks importing at the same time build it once); with
# ZEDO_NO_BUILD=1 (ranks of a launcher that has built already) or when the build fails this raises ImportError.
LIB_PATH = _zb.ensure_library()

_lib = ctypes.CDLL(LIB_PATH)

_vp, _i, _f, _d, _ll, _sz = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double,
                             ctypes.c_longlong, ctypes.c_size_t)

# name -> (restype, argtypes): every symbol declared in include/zedo_hip.h
SIGNATURES = {
    "zedo_abi_version": (_i, []),
    "zedo_error_string": (ctypes.c_char_p, [_i]),
    "zedo_weights_create": (_i, [_vp, _sz, _i, _i, _i, _i, _i, _vp, ctypes.POINTER(_vp)]),
    "zedo_weights_destroy": (None, [_vp]),
    "zedo_weights_set_math": (_i, [_vp, _i, _vp]),
    "zedo_weights_get_math": (_i, [_vp]),
    "zedo_schedule_create": (_i, [_vp, _vp, _i, _f, _f, _f, _i, _vp, ctypes.POINTER(_vp)]),
    "zedo_schedule_destroy": (None, [_vp]),
    "zedo_schedule_read": (_i, [_vp, _vp, _vp, _vp]),
    "zedo_workspace_bytes": (_sz, [_i]),
    "zedo_reproj_prepare": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "zedo_reproj_degenerate": (_i, [_vp, _i, _i, _vp, _vp]),
    "zedo_reproj_grad": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _ll, _vp]),
    "zedo_score_eps": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "zedo_sde_step": (_i, [_vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "zedo_oil_run": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ll, _vp, _sz, _vp]),
    "zedo_ipo_fit": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _i, _d, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _vp]),
    "zedo_ipo_fit_resume": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _i, _d, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                 _i, _ll, _vp]),
    "zedo_rotate_init": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _ll, _vp]),
    "zedo_min_mpjpe": (_i, [_vp, _vp, _i, _i, _i, _ll, _i, _vp, _vp, _vp, _vp]),
    "zedo_pose_min": (_i, [_vp, _i, _i, _ll, _vp, _vp, _vp]),
    "zedo_probe_mfma_peak": (_i, [_i, _vp, _vp, _vp]),
    "zedo_probe_mfma_peak_f16": (_i, [_i, _vp, _vp, _vp]),
    "zedo_profile_start": (_i, [_i, _i]),
    "zedo_profile_stop": (_i, [_vp, _vp, _vp]),
    "zedo_profile_shader_ghz": (_d, []),
    "zedo_profile_bracket_ms": (_d, []),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(_lib, _name)  # AttributeError here = library/header mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

N_JOINTS, JOINT_DIM, HIDDEN_DIM, EMBED_DIM = 17, 3, 1024, 512
GEOM_F = 8


class ZedoError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise ZedoError(f"libzedo_hip: {_lib.zedo_error_string(rc).decode()} (code {rc})")


_gpu_seen = False


def _need_gpu():
    global _gpu_seen
    if not _gpu_seen:           # torch.cuda.is_available() costs ~20 us per call; a GPU does not go away
        if not torch.cuda.is_available():
            raise ZedoError("libzedo_hip needs an MI355X (gfx950): no GPU is visible and there is no CPU path")
        _gpu_seen = True


def _p(t, dtype=torch.float32):
    """device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == dtype):
        raise ZedoError(f"expected a contiguous {dtype} CUDA tensor, got {type(t)} "
                        f"{getattr(t, 'dtype', None)} {getattr(t, 'device', None)}")
    return ctypes.c_void_p(t.data_ptr())


# ---- streams, devices, threads ------------------------------------------------------------------------------------------
# The C ABI is re-entrant across distinct streams (include/zedo_hip.h); this mirror honours that: every call is
# enqueued on the CURRENT stream of the device its tensors live on (not of whatever device happens to be current), runs
# with that device current (the library's per-device launch state follows hipGetDevice), takes its workspace from torch's
# stream-aware caching allocator per call - so two streams or two host threads never share scratch memory, and a block is
# only handed out again in stream order - and refuses arguments that live on different devices.
def _device_of(*objs):
    """The one CUDA device of the given tensors / handles (None entries skipped); mixed devices are an error."""
    dev = None
    for o in objs:
        if o is None:
            continue
        d = o.device if isinstance(o, (torch.Tensor, Weights, Schedule)) else None
        if d is None:
            continue
        d = torch.device(d)
        if d.type != "cuda":
            raise ZedoError(f"expected CUDA tensors, got one on {d}")
        if d.index is None:
            d = torch.device("cuda", torch.cuda.current_device())
        if dev is None:
            dev = d
        elif d != dev:
            raise ZedoError(f"arguments live on different devices ({dev} and {d}): one call runs on one GPU")
    if dev is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _norm_device(device=None):
    """torch.device("cuda", index) for `device` (None / "cuda" = the current device)."""
    d = torch.device("cuda") if device is None else torch.device(device)
    if d.type != "cuda":
        raise ZedoError(f"libzedo_hip runs on CUDA (HIP) devices only, not on {d}")
    return torch.device("cuda", torch.cuda.current_device()) if d.index is None else d


def _stream(device=None):
    """hipStream_t of torch's current stream ON `device` (default: the current device)."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def abi_version():
    return _lib.zedo_abi_version()


# state-dict order of ScoreModelFC_Adv (reference lib/algorithms/advanced/model.py:113-152)
def param_names(n_blocks=2):
    names = ["pre_dense.weight", "pre_dense.bias", "pre_dense_t.weight", "pre_dense_t.bias",
             "pre_gnorm.weight", "pre_gnorm.bias", "shared_time_embed.0.weight", "shared_time_embed.0.bias"]
    for b in range(1, n_blocks + 1):
        for k in (1, 2):
            names += [f"b{b}_dense{k}.weight", f"b{b}_dense{k}.bias", f"b{b}_dense{k}_t.weight",
                      f"b{b}_dense{k}_t.bias", f"b{b}_gnorm{k}.weight", f"b{b}_gnorm{k}.bias"]
    return names + ["post_dense.weight", "post_dense.bias"]


MATH_MODES = {"f32": 0, "f16x3": 1}      # ZEDO_MATH_F32 / ZEDO_MATH_F16X3 of include/zedo_hip.h


def default_math():
    """The arithmetic of the hidden layers for handles created from now on: environment ZEDO_MATH = f32 (default: exact
    fp32 MFMA) | f16x3 (split-fp16 operands on the fp16 matrix pipe, fp32-level accuracy; opt-in)."""
    m = os.environ.get("ZEDO_MATH", "f32").strip().lower() or "f32"
    if m not in MATH_MODES:
        raise ZedoError(f"ZEDO_MATH={m!r}: expected one of {sorted(MATH_MODES)}")
    return m


class Weights:
    """Device copy of a ScoreModelFC_Adv state dict, repacked for the kernels (zedo_weights_create).
    math: "f32" | "f16x3" | None (= default_math(), i.e. the ZEDO_MATH environment variable)."""

    def __init__(self, state_dict, n_joints=N_JOINTS, joint_dim=JOINT_DIM, hidden=HIDDEN_DIM, embed=EMBED_DIM,
                 n_blocks=2, math=None, device=None):
        _need_gpu()
        self.device = _norm_device(device)
        flat = []
        for name in param_names(n_blocks):
            v = state_dict[name]
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            flat.append(np.ascontiguousarray(v, dtype=np.float32).reshape(-1))
        flat = np.concatenate(flat)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(_lib.zedo_weights_create(flat.ctypes.data_as(_vp), flat.size, n_joints, joint_dim, hidden, embed,
                                            n_blocks, _stream(self.device), ctypes.byref(self._h)))
        self.n_joints, self.joint_dim, self.hidden, self.embed, self.n_blocks = n_joints, joint_dim, hidden, embed, n_blocks
        self.set_math(default_math() if math is None else math)

    def set_math(self, math):
        """Not re-entrant with calls that use this handle: switch modes between runs, not during them."""
        if math not in MATH_MODES:
            raise ZedoError(f"math mode {math!r}: expected one of {sorted(MATH_MODES)}")
        with torch.cuda.device(self.device):
            _check(_lib.zedo_weights_set_math(self._h, MATH_MODES[math], _stream(self.device)))
        self.math = math
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:   # _lib is None during interpreter shutdown
            _lib.zedo_weights_destroy(h)
            self._h = None


class Schedule:
    """Per-step tables (time-bias rows, a_i, c_i) for one timestamp vector (zedo_schedule_create)."""

    def __init__(self, weights, ts, beta_min=0.1, beta_max=20.0, n_sde=1000, label_scale=999.0):
        _need_gpu()
        ts = np.ascontiguousarray(np.asarray(ts, dtype=np.float32).reshape(-1))
        self.S = int(ts.size)
        self.ts = ts
        self.weights = weights  # keep alive
        self.device = weights.device
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(_lib.zedo_schedule_create(weights._h, ts.ctypes.data_as(_vp), self.S, label_scale, beta_min, beta_max,
                                             n_sde, _stream(self.device), ctypes.byref(self._h)))

    def read(self):
        nl = 1 + 2 * self.weights.n_blocks
        tb = np.empty((self.S, nl, self.weights.hidden), np.float32)
        a = np.empty(self.S, np.float32)
        c = np.empty(self.S, np.float32)
        _check(_lib.zedo_schedule_read(self._h, tb.ctypes.data_as(_vp), a.ctypes.data_as(_vp), c.ctypes.data_as(_vp)))
        return tb, a, c

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:
            _lib.zedo_schedule_destroy(h)
            self._h = None


def workspace_bytes(B):
    return int(_lib.zedo_workspace_bytes(int(B)))


def workspace(B, device=None):
    """A uint8 CUDA tensor large enough for B rows, owned by the CALL that asked for it: a fresh block of torch's caching
    allocator, tagged with the current stream of `device` (so the allocator re-issues it only in stream order once the call's
    reference is gone - the kernels of the call may still be running then) and shared with no other stream or thread.  After
    the first call of a size the block comes out of the allocator's cache: microseconds, no hipMalloc."""
    _need_gpu()
    device = _norm_device(device)
    with torch.cuda.device(device):
        return torch.empty(workspace_bytes(B), dtype=torch.uint8, device=device)


def reproj_prepare(uv, K, conf=None, conf_clamped_out=None):
    """uv [N,J,2], K [N,3,3], conf [N,J] or None -> geom [N,J,8]."""
    _need_gpu()
    dev = _device_of(uv, K, conf, conf_clamped_out)
    N, J = uv.shape[0], uv.shape[1]
    with torch.cuda.device(dev):
        geom = torch.empty((N, J, GEOM_F), dtype=torch.float32, device=dev)
        _check(_lib.zedo_reproj_prepare(_p(uv), _p(K), _p(conf), N, J, _p(geom), _p(conf_clamped_out), _stream(dev)))
    return geom


def reproj_degenerate(geom):
    """Number of poses whose least-squares system for T is singular (zedo_reproj_degenerate; synchronises)."""
    _need_gpu()
    dev = _device_of(geom)
    n = ctypes.c_int(0)
    with torch.cuda.device(dev):
        _check(_lib.zedo_reproj_degenerate(_p(geom), geom.shape[0], geom.shape[1], ctypes.cast(ctypes.byref(n), _vp), _stream(dev)))
    return int(n.value)


SINGULAR_MSG = ("gradient_field_gen: the least-squares system for T is singular for {n} pose(s) - every camera ray of the "
                "pose coincides; the reference's torch.inverse(AtA) raises here (simple_zeroshot_opt.py:89-92)")


def reproj_grad(x, geom, T, solve_T, row_offset=0):
    """gradient_field_gen body: returns g [B,J,3]; T [B,3] is overwritten when solve_T."""
    _need_gpu()
    dev = _device_of(x, geom, T)
    B, J = x.shape[0], x.shape[1]
    with torch.cuda.device(dev):
        g = torch.empty_like(x)
        _check(_lib.zedo_reproj_grad(_p(x), _p(geom), _p(T), int(bool(solve_T)), _p(g), B, geom.shape[0], J,
                                     int(row_offset), _stream(dev)))
    return g


def score_eps(weights, sched, step, x):
    _need_gpu()
    dev = _device_of(weights, sched, x)
    B = x.shape[0]
    with torch.cuda.device(dev):
        ws = workspace(B, dev)
        eps = torch.empty_like(x)
        _check(_lib.zedo_score_eps(weights._h, sched._h, int(step), _p(x), _p(eps), B, _p(ws, torch.uint8), ws.numel(),
                                   _stream(dev)))
    return eps


def sde_step(weights, sched, step, x):
    """x <- a x + c eps(x), in place."""
    _need_gpu()
    dev = _device_of(weights, sched, x)
    B = x.shape[0]
    with torch.cuda.device(dev):
        ws = workspace(B, dev)
        _check(_lib.zedo_sde_step(weights._h, sched._h, int(step), _p(x), B, _p(ws, torch.uint8), ws.numel(), _stream(dev)))
    return x


def oil_run(weights, sched, x, geom, T, step_begin, step_end, switch_step, row_offset=0):
    """Fused OIL loop over steps [step_begin, step_end); x [B,J,3] and T [B,3] updated in place."""
    _need_gpu()
    dev = _device_of(weights, sched, x, geom, T)
    B = x.shape[0]
    with torch.cuda.device(dev):
        ws = workspace(B, dev)
        _check(_lib.zedo_oil_run(weights._h, sched._h, _p(x), _p(geom), _p(T), int(step_begin), int(step_end),
                                 int(switch_step), B, geom.shape[0], int(row_offset), _p(ws, torch.uint8), ws.numel(),
                                 _stream(dev)))
    return x, T


def axes_mask(axes):
    return sum({"x": 1, "y": 2, "z": 4}[a] for a in set(axes))


def ipo_fit(x0, uv, K, keylist, axes, ipo_T, min_scale, max_scale, iters, normaliser, B, row_offset=0,
            return_params=False, state=None, it_begin=0):
    """x0 [H,J,3] centred cluster poses, uv [N,J,2], K [N,3,3] -> R [B,3,3], T [B,3] (, q [B,4], scale [B]).
    state [B,15] (optional, in/out) + it_begin: resume from a captured Adam state (zedo_ipo_fit_resume)."""
    _need_gpu()
    dev = _device_of(x0, uv, K, state)
    N, J, H = uv.shape[0], uv.shape[1], x0.shape[0]
    kl = (ctypes.c_int * len(keylist))(*[int(k) for k in keylist])
    with torch.cuda.device(dev):
        R = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
        T = torch.empty((B, 3), dtype=torch.float32, device=dev)
        q = torch.empty((B, 4), dtype=torch.float32, device=dev) if return_params else None
        sc = torch.empty((B,), dtype=torch.float32, device=dev) if return_params else None
        if state is not None:
            _check(_lib.zedo_ipo_fit_resume(_p(x0), _p(uv), _p(K), ctypes.cast(kl, _vp), len(keylist), axes_mask(axes),
                                            float(ipo_T), float(min_scale), float(max_scale), int(iters), float(normaliser),
                                            _p(R), _p(T), _p(q), _p(sc), _p(state), int(it_begin), B, H, N, J,
                                            int(row_offset), _stream(dev)))
        else:
            _check(_lib.zedo_ipo_fit(_p(x0), _p(uv), _p(K), ctypes.cast(kl, _vp), len(keylist), axes_mask(axes),
                                     float(ipo_T), float(min_scale), float(max_scale), int(iters), float(normaliser),
                                     _p(R), _p(T), _p(q), _p(sc), B, H, N, J, int(row_offset), _stream(dev)))
    return (R, T, q, sc) if return_params else (R, T)


def rotate_init(x0, R, N, row_offset=0):
    _need_gpu()
    dev = _device_of(x0, R)
    B, J = R.shape[0], x0.shape[1]
    with torch.cuda.device(dev):
        x = torch.empty((B, J, 3), dtype=torch.float32, device=dev)
        _check(_lib.zedo_rotate_init(_p(x0), _p(R), _p(x), B, x0.shape[0], N, J, int(row_offset), _stream(dev)))
    return x


def min_mpjpe(pred, gt_centred, N, procrustes=False, row_offset=0):
    """pred [B,J,3] fp32 rows (h,n); gt_centred [N,J,3] float64 -> (err [B], best [N], best_h [N])."""
    _need_gpu()
    dev = _device_of(pred, gt_centred)
    B, J = pred.shape[0], pred.shape[1]
    with torch.cuda.device(dev):
        err = torch.empty((B,), dtype=torch.float64, device=dev)
        best = torch.empty((N,), dtype=torch.float64, device=dev)
        best_h = torch.empty((N,), dtype=torch.int32, device=dev)
        _check(_lib.zedo_min_mpjpe(_p(pred), _p(gt_centred, torch.float64), B, N, J, int(row_offset), int(bool(procrustes)),
                                   _p(err, torch.float64), _p(best, torch.float64), _p(best_h, torch.int32), _stream(dev)))
    return err, best, best_h


def pose_min(err, N, row_offset=0):
    """Per-pose minimum / first arg-min over the hypotheses of the (possibly edited) per-row errors (zedo_pose_min)."""
    _need_gpu()
    dev = _device_of(err)
    with torch.cuda.device(dev):
        best = torch.empty((N,), dtype=torch.float64, device=dev)
        best_h = torch.empty((N,), dtype=torch.int32, device=dev)
        _check(_lib.zedo_pose_min(_p(err, torch.float64), err.shape[0], N, int(row_offset), _p(best, torch.float64),
                                  _p(best_h, torch.int32), _stream(dev)))
    return best, best_h


def probe_mfma_peak(iters=100000):
    """-> (sustained fp32-MFMA TFLOP/s, shader clock GHz) of this box right now (zedo_probe_mfma_peak)."""
    _need_gpu()
    tf, ghz = ctypes.c_double(), ctypes.c_double()
    _check(_lib.zedo_probe_mfma_peak(int(iters), ctypes.cast(ctypes.byref(tf), _vp), ctypes.cast(ctypes.byref(ghz), _vp), _stream()))
    return tf.value, ghz.value


def probe_mfma_peak_f16(iters=400000):
    """-> (sustained fp16-MFMA TFLOP/s, shader clock GHz): a bare v_mfma_f32_32x32x16_f16 stream on this box right now
    (zedo_probe_mfma_peak_f16) - the attainable ceiling of the split-fp16 mode's matrix pipe under power management."""
    _need_gpu()
    tf, ghz = ctypes.c_double(), ctypes.c_double()
    _check(_lib.zedo_probe_mfma_peak_f16(int(iters), ctypes.cast(ctypes.byref(tf), _vp), ctypes.cast(ctypes.byref(ghz), _vp), _stream()))
    return tf.value, ghz.value


PROF_CLASSES = ("hidden_dense", "pre_dense", "post_dense_sde", "reproj")


def profile_start(sample_every=16, max_samples=8192):
    _check(_lib.zedo_profile_start(int(sample_every), int(max_samples)))


def profile_stop():
    """-> {class: dict(total_ms, samples, launches, avg_ms, avg_ms_raw)} for the sampled kernel launches.  avg_ms is the
    bracketed duration minus what an empty bracket measures on the same stream (zedo_profile_bracket_ms, also returned
    under the key "bracket_ms" of the hidden_dense entry); avg_ms_raw is the bracketed duration itself."""
    n = len(PROF_CLASSES)
    tot = (ctypes.c_double * n)()
    cnt = (ctypes.c_longlong * n)()
    seen = (ctypes.c_longlong * n)()
    _check(_lib.zedo_profile_stop(ctypes.cast(tot, _vp), ctypes.cast(cnt, _vp), ctypes.cast(seen, _vp)))
    br = float(_lib.zedo_profile_bracket_ms())
    out = {k: dict(total_ms=tot[i], samples=int(cnt[i]), launches=int(seen[i]),
                   avg_ms_raw=(tot[i] / cnt[i] if cnt[i] else None),
                   avg_ms=(max(tot[i] / cnt[i] - br, 0.0) if cnt[i] else None)) for i, k in enumerate(PROF_CLASSES)}
    out["hidden_dense"]["shader_clock_ghz"] = float(_lib.zedo_profile_shader_ghz()) or None
    out["hidden_dense"]["bracket_ms"] = br
    return out
