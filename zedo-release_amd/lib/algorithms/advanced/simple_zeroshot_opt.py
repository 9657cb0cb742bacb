"""Reprojection optimiser of the drop-in surface (reference lib/algorithms/advanced/simple_zeroshot_opt.py).

`gradient_field_gen` keeps the reference's signature and return values and runs in libzedo_hip.so
(zedo_reproj_prepare + zedo_reproj_grad).  `RotOpt` keeps the parameter names of the reference module; the
500-iteration Adam fit that the reference drives through it (run/opt_main.py:180-195) is one kernel here:
`RotOpt.fit(...)` -> zedo_ipo_fit.  `RotOpt.forward` remains available as ordinary differentiable torch code
for callers that use the module directly.
"""
import threading

import torch
import torch.nn as nn

from .utils import quaternion_to_matrix


class RotOpt(nn.Module):
    def __init__(self, batch_size=100, axis="y", minT=0.5, maxT=2):
        super().__init__()
        self.rot_vect = nn.Parameter(torch.ones((batch_size, 1)))
        for a in axis:
            setattr(self, "rot_vect_%s" % a, nn.Parameter(torch.zeros((batch_size, 1))))
        self.identity = nn.Parameter(torch.eye(3), requires_grad=False)
        self.scale = nn.Parameter(torch.ones((batch_size, 1, 1)))
        self.batch_size, self.axis, self.minT, self.maxT = batch_size, axis, minT, maxT

    def quaternion(self):
        z = torch.zeros((self.batch_size, 1), device=self.rot_vect.device)
        return torch.cat([self.rot_vect] + [getattr(self, "rot_vect_%s" % a, z) for a in "xyz"], dim=-1)

    def generate_matrix(self):
        return quaternion_to_matrix(self.quaternion())

    def forward(self, x, T, K):
        """uv of R x + T clamp(scale) under intrinsics K (reference :20-25); x [B,k,3], T [B,1,3], K [B,3,3]."""
        p = self.generate_matrix().bmm(x.permute(0, 2, 1)) + (T * torch.clamp(self.scale, self.minT, self.maxT)).permute(0, 2, 1)
        p = K.bmm(p).permute(0, 2, 1)
        return p[:, :, :2] / p[:, :, 2:]

    @torch.no_grad()
    def fit(self, x0, cond, K, keylist, ipo_T, iters=500, normaliser=None):
        """The IPO loop of run/opt_main.py:177-195 in one HIP kernel.  x0 [B,J,3] (or [1,J,3] shared),
        cond [B,J,2], K [B,3,3].  Updates this module's parameters and returns (R [B,3,3], T [B,1,3])."""
        import zedo_hip
        B = cond.shape[0]
        shared = x0.shape[0] == 1 or bool((x0 == x0[:1]).all())
        if not shared:
            raise NotImplementedError("per-pose initial poses: zedo_ipo_fit takes one cluster pose per hypothesis")
        norm = B * len(keylist) * 2 if normaliser is None else normaliser
        R, T, q, sc = zedo_hip.ipo_fit(x0[:1].float().contiguous(), cond.float().contiguous(), K.float().contiguous(),
                                       list(keylist), self.axis, ipo_T, self.minT, self.maxT, iters, norm, B,
                                       return_params=True)
        self.rot_vect.copy_(q[:, 0:1])
        for i, a in enumerate("xyz"):
            if hasattr(self, "rot_vect_%s" % a):
                getattr(self, "rot_vect_%s" % a).copy_(q[:, i + 1:i + 2])
        self.scale.copy_(sc.reshape(-1, 1, 1))
        return R, T.reshape(B, 1, 3)


def perpendicular_distance(point, vector):
    """(p . r) r - p (reference :33-36); host helper, the sampling path uses the fused kernel."""
    return torch.sum(point * vector, dim=-1, keepdim=True) * vector - point


_geom_tls = threading.local()     # per calling thread: no sharing, no locking
_GEOM_SLOTS = 4                   # a loop alternating between a few (key2d, K, conf) triples keeps all of them


def invalidate_ray_cache():
    """Drop the cached rays of the calling thread.  The cache recognises a change through torch's version counters;
    call this after writing into key2d / K / conf in a way they do not see (`t.data` writes, raw-pointer kernels)."""
    _geom_tls.entries = []


def _rays(key2d, K, conf):
    """zedo_reproj_prepare for the CALLER's (key2d, K, conf), reused while the SAME tensor objects come back unmodified
    (object identity + storage address + version counter + shape + dtype of the tensors the caller passed - not of
    converted copies, which would be new objects on every call and never hit): the reference's loop hands
    gradient_field_gen identical condition / K / conf tensors 1000 times per hypothesis (run/opt_main.py:203-206).
    Conversion to contiguous fp32, the build, the singular-system count (one small device allocation + a stream sync)
    and the in-place clamp of conf (:64-66; idempotent, so skipping it on a hit changes nothing) happen on a miss only.
    Up to _GEOM_SLOTS triples per thread, least recently used out first; the caller's tensors are held (so that ids
    stay unique) until they are displaced or invalidate_ray_cache() is called."""
    import zedo_hip

    def fingerprint():
        return tuple((id(a), a.data_ptr(), a._version, tuple(a.shape), a.dtype) if a is not None else None
                     for a in (key2d, K, conf))

    entries = getattr(_geom_tls, "entries", None)
    if entries is None:
        entries = _geom_tls.entries = []
    key = fingerprint()
    for i, e in enumerate(entries):
        if e[0] == key:
            entries.append(entries.pop(i))
            return e[2], e[3]
    uv = key2d.float().contiguous()
    Kc = K.float().contiguous()
    cc = None
    if conf is not None:
        cc = conf if (conf.is_contiguous() and conf.dtype == torch.float32) else conf.float().contiguous()
    geom = zedo_hip.reproj_prepare(uv, Kc, cc, cc)
    if cc is not None and cc is not conf:
        conf.copy_(cc)                                       # the clamp is observable on the caller's tensor (:64-66)
    singular = zedo_hip.reproj_degenerate(geom)              # once per (key2d, K, conf): step-invariant
    entries.append((fingerprint(), (key2d, K, conf), geom, singular))      # fingerprint AFTER the in-place clamp of conf
    del entries[:-_GEOM_SLOTS]
    return geom, singular


def gradient_field_gen(key2d, key3d, K, noise_type=None, t=None, conf=None, returnT=False, norm_true=None,
                       previous_T=None):
    """Gradient of the 3D keypoints towards their camera rays (reference :46-125).

    key2d [B,J,2], key3d [B,J,3], K [B,3,3], conf [B,J] or None (clamped to [1e-4,1] IN PLACE, like the
    reference :64-66), t [B,1,3] or None (None -> weighted least-squares translation, sign-fixed).
    """
    import zedo_hip
    std = 0.0001
    B, J = key3d.shape[0], key3d.shape[1]
    x = key3d.float().contiguous()
    geom, singular = _rays(key2d, K, conf)
    if t is None:
        # torch.inverse(AtA) of the reference raises on a singular system (:89-92).  Here "singular" is the exact test
        # den == 0 of the closed form, i.e. ONLY a pose whose camera rays all coincide exactly raises; a nearly singular
        # system returns a huge T, as torch.inverse does above its own pivot threshold
        if singular:
            raise torch.linalg.LinAlgError(zedo_hip.SINGULAR_MSG.format(n=singular))
        T = torch.empty((B, 3), dtype=torch.float32, device=x.device)      # written by the kernel (solve_T)
        g = zedo_hip.reproj_grad(x, geom, T, True)
    else:
        T = t.reshape(B, 3).float().contiguous().clone()
        g = zedo_hip.reproj_grad(x, geom, T, False)
    if noise_type == "gaussian":
        g = g + std * torch.randn_like(g) * t
    elif noise_type == "uniform":
        g = g + std * (torch.randn_like(g) - 0.5)
    if returnT:
        return g, T.reshape(B, 1, 3)
    return g
