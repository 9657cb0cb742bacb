"""Multi-threaded CPU port of the per-step part of the ZeDO sampling path (torch CPU tensors, all host cores).

TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Used only by bench.py's ``cpu_baseline`` leg (and by the test that
checks it against oracle/zedo_oracle.py); the product package never imports it.

The numpy oracle (zedo_oracle.py) is the parity reference: it is written for clarity and single-threaded
element-wise arithmetic makes it a pessimistic CPU baseline (GroupNorm / SiLU on one core).  This file restates
the SAME functions - one OIL iteration = gradient_field_gen + pc_step - with torch CPU operators, which are
what the reference itself runs on a CPU (addmm, native_group_norm, silu on every core), so that the CPU number
quoted next to the GPU is a fair one.  Parity with the numpy oracle: tests/test_oracle_golden.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

import zedo_oracle as O


class StepPort:
    """One OIL iteration on CPU tensors (fp32): x <- pc_step(x + g(x)), T <- T or its least-squares update."""

    def __init__(self, weights, key2d, K, conf, n_blocks=2, threads=None):
        if threads:
            torch.set_num_threads(int(threads))
        self.w = {k: torch.tensor(np.asarray(v, np.float32)) for k, v in weights.items()}
        self.n_blocks = n_blocks
        key2d, K = torch.tensor(np.asarray(key2d, np.float32)), torch.tensor(np.asarray(K, np.float32))
        hom = torch.cat([key2d, torch.ones_like(key2d[:, :, :1])], dim=-1)
        ray = torch.einsum("bij,bkj->bki", torch.linalg.inv(K), hom)
        self.ray = ray / ray[:, :, 2:]                                   # simple_zeroshot_opt.py:61-71
        self.rn = self.ray / torch.linalg.norm(self.ray, dim=-1, keepdim=True)
        c = torch.clamp(torch.tensor(np.asarray(conf, np.float32)), 1e-4, 1.0)
        self.c2 = (c * c)[:, :, None]                                    # weights of the normal equations (:75-84)

    def _lin(self, name, x):
        return F.linear(x, self.w[name + ".weight"], self.w[name + ".bias"])

    def eps(self, x, label):
        """zedo_oracle.score_model_forward (model.py:215-298, eval mode)."""
        w, B = self.w, x.shape[0]
        pe = torch.tensor(O.timestep_embedding(np.float32(label), w["shared_time_embed.0.weight"].shape[0]))
        temb = F.silu(self._lin("shared_time_embed.0", pe))
        h = self._lin("pre_dense", x.reshape(B, -1)) + self._lin("pre_dense_t", temb)
        h = F.silu(F.group_norm(h, 32, w["pre_gnorm.weight"], w["pre_gnorm.bias"], 1e-5))
        for b in range(1, self.n_blocks + 1):
            h1 = self._lin(f"b{b}_dense1", h) + self._lin(f"b{b}_dense1_t", temb)
            h1 = F.silu(F.group_norm(h1, 32, w[f"b{b}_gnorm1.weight"], w[f"b{b}_gnorm1.bias"], 1e-5))
            h2 = self._lin(f"b{b}_dense2", h1) + self._lin(f"b{b}_dense2_t", temb)
            h2 = F.silu(F.group_norm(h2, 32, w[f"b{b}_gnorm2.weight"], w[f"b{b}_gnorm2.bias"], 1e-5))
            h = h + h2
        return self._lin("post_dense", h).reshape(x.shape)

    def gradient(self, x, T, solve):
        """zedo_oracle.gradient_field_gen (simple_zeroshot_opt.py:46-125)."""
        ray = self.ray
        if solve:
            B, J, _ = x.shape
            A = torch.zeros(B, 2 * J, 3)
            b = torch.zeros(B, 2 * J, 1)
            b[:, 0::2, 0] = x[:, :, 0] - x[:, :, 2] * ray[:, :, 0]
            b[:, 1::2, 0] = x[:, :, 1] - x[:, :, 2] * ray[:, :, 1]
            A[:, 0::2, 0], A[:, 0::2, 2] = -1, ray[:, :, 0]
            A[:, 1::2, 1], A[:, 1::2, 2] = -1, ray[:, :, 1]
            A[:, 0::2, :] *= self.c2
            A[:, 1::2, :] *= self.c2
            b[:, 0::2, :] *= self.c2
            b[:, 1::2, :] *= self.c2
            At = A.transpose(1, 2)
            T = (torch.linalg.inv(At @ A) @ (At @ b)).transpose(1, 2)
            T = torch.where(T[:, :, 2:3] < 0, -T, T)
        p = x + T
        return (p * self.rn).sum(-1, keepdim=True) * self.rn - p, T

    def step(self, x, T, t, solve, n_sde=1000, beta_0=0.1, beta_1=20.0):
        """x + g, then zedo_oracle.pc_step (sampling.py:450-527 in the shipped configuration)."""
        with torch.no_grad():
            g, T = self.gradient(x, T, solve)
            x = x + g
            t = np.float32(t)
            beta = np.float32(beta_0) + t * np.float32(beta_1 - beta_0)
            disc = np.float32(1.0) - np.exp(np.float32(-2 * beta_0) * t - np.float32(beta_1 - beta_0) * t ** 2)
            std = O.subvp_marginal_std(t, beta_0, beta_1)
            score = -self.eps(x, t * np.float32(999)) / float(std)
            drift = float(np.float32(-0.5) * beta) * x - float(beta * disc) * score
            return x + drift * float(np.float32(-1.0 / n_sde)), T
