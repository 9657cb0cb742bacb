"""ScoreModelFC_Adv with the reference's constructor, parameter names and forward signature
(reference lib/algorithms/advanced/model.py:97-298), evaluated by libzedo_hip.so.

The module holds ordinary torch Parameters so that `load_state_dict` accepts the reference's checkpoints
unchanged (34 tensors + the float64 `sigmas` buffer, reference run/opt_main.py:120-137); they are repacked
into the kernels' layout on first use and again whenever they change.  `forward` has no PyTorch compute
path: it needs an MI355X and the built library.
"""
import math

import numpy as np
import torch
import torch.nn as nn


def get_sigmas(config):
    """reference model.py:68-78"""
    return np.exp(np.linspace(np.log(config.model.sigma_max), np.log(config.model.sigma_min), config.model.num_scales))


def get_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
    """Sinusoidal embedding (reference model.py:81-95).  Host-side helper kept for API compatibility; the
    sampling path computes it on the device (zedo_geom.hip: posemb_kernel)."""
    assert timesteps.dim() == 1
    half = embedding_dim // 2
    scale = math.log(max_positions) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.float32, device=timesteps.device) * -scale)
    arg = timesteps.float()[:, None] * freq[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb


class ScoreModelFC_Adv(nn.Module):
    def __init__(self, config, n_joints=17, joint_dim=3, hidden_dim=64, embed_dim=32, cond_dim=2, n_blocks=2):
        super().__init__()
        self.config = config
        self.n_joints, self.joint_dim, self.n_blocks = n_joints, joint_dim, n_blocks
        self.hidden_dim, self.embed_dim = hidden_dim, embed_dim
        self.time_embedding_type = config.model.embedding_type.lower()
        if self.time_embedding_type != "positional":
            raise NotImplementedError("the HIP path implements the 'positional' time embedding of the shipped configs "
                                      "(configs/optim/*.py: model.embedding_type)")
        d = n_joints * joint_dim
        self.act = nn.SiLU()
        self.pre_dense = nn.Linear(d, hidden_dim)
        self.pre_dense_t = nn.Linear(embed_dim, hidden_dim)
        self.pre_gnorm = nn.GroupNorm(32, num_channels=hidden_dim)
        self.dropout = nn.Dropout(p=0.25)      # identity in eval mode; kept for attribute parity
        self.shared_time_embed = nn.Sequential(nn.Linear(embed_dim, embed_dim), self.act)
        self.register_buffer("sigmas", torch.tensor(get_sigmas(config)))
        for b in range(1, n_blocks + 1):
            for k in (1, 2):
                setattr(self, f"b{b}_dense{k}", nn.Linear(hidden_dim, hidden_dim))
                setattr(self, f"b{b}_dense{k}_t", nn.Linear(embed_dim, hidden_dim))
                setattr(self, f"b{b}_gnorm{k}", nn.GroupNorm(32, num_channels=hidden_dim))
        self.post_dense = nn.Linear(hidden_dim, d)
        self._packed = None          # (version key, zedo_hip.Weights)
        self._sched_cache = {}
        self._plist = None           # the parameter tensors, walked once (Module.parameters() costs ~0.3 ms per walk)

    # ---- device-side copy of the parameters --------------------------------------------------------
    def _apply(self, fn, *a, **k):   # .to() / .cuda() / .float(): parameter storage may move
        self._plist = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):      # assign=True replaces the parameter OBJECTS
        self._plist = None
        return super().load_state_dict(*a, **k)

    def __setattr__(self, name, value):      # a reassigned parameter / sub-module is a new tensor object
        if name != "_plist" and isinstance(value, (nn.Parameter, nn.Module)):
            object.__setattr__(self, "_plist", None)
        super().__setattr__(name, value)

    def invalidate_hip_weights(self):
        """Force a repack on the next use - for writers that bypass the version counters (`p.data.copy_`, raw pointers)."""
        self._plist, self._packed, self._sched_cache = None, None, {}

    def _version(self):
        """Cheap fingerprint of the parameter values: storage address + in-place version counter of each tensor
        (load_state_dict, optimiser steps and in-place ops under no_grad bump the counter; writes through `p.data` do
        NOT - `.data` carries a counter of its own - call invalidate_hip_weights() after those)."""
        if self._plist is None:
            self._plist = list(self.parameters())
        return tuple((p.data_ptr(), p._version) for p in self._plist)

    def hip_weights(self):
        import zedo_hip
        key = (self._version(), zedo_hip.default_math())      # ZEDO_MATH may change between calls (tests run both modes)
        if self._packed is None or self._packed[0] != key:
            sd = {k: v for k, v in self.state_dict().items() if k != "sigmas"}
            self._packed = (key, zedo_hip.Weights(sd, self.n_joints, self.joint_dim, self.hidden_dim, self.embed_dim,
                                                  self.n_blocks))
            self._sched_cache = {}
        return self._packed[1]

    def hip_schedule(self, values, label_scale, beta_min=0.1, beta_max=20.0, n_sde=1000):
        """Cached zedo_hip.Schedule for a tuple of times (label_scale 999) or labels (label_scale 1)."""
        import zedo_hip
        w = self.hip_weights()
        key = (tuple(np.asarray(values, np.float32).tobytes()), float(label_scale), beta_min, beta_max, n_sde)
        s = self._sched_cache.get(key)
        if s is None:
            if len(self._sched_cache) > 64:
                self._sched_cache.clear()
            s = zedo_hip.Schedule(w, np.asarray(values, np.float32), beta_min, beta_max, n_sde, label_scale)
            self._sched_cache[key] = s
        return s

    def forward(self, batch, t, condition=None, mask=None):
        """batch [B,j,3], t [B] = labels (999 * SDE time), condition / mask ignored as in the reference
        (model.py:239-291).  Returns eps [B,j,3]."""
        import zedo_hip
        if self.training:
            raise RuntimeError("the HIP path evaluates the network in eval() mode only (dropout = identity)")
        x = batch.reshape(batch.shape[0], self.n_joints, self.joint_dim).float().contiguous()
        labels = t.reshape(-1).float()
        uniq = torch.unique(labels)
        out = torch.empty_like(x)
        for u in uniq.tolist():       # the sampler always passes one repeated value (advanced/sampling.py:497)
            sel = labels == u
            sched = self.hip_schedule([u], 1.0)
            if bool(sel.all()):
                out = zedo_hip.score_eps(self.hip_weights(), sched, 0, x)
            else:
                out[sel] = zedo_hip.score_eps(self.hip_weights(), sched, 0, x[sel].contiguous())
        if self.config.model.scale_by_sigma:
            out = out / self.sigmas.to(out.device)[t.reshape(-1).long()].reshape(-1, 1, 1).to(out.dtype)   # model.py:254,294
        return out.reshape(batch.shape)
