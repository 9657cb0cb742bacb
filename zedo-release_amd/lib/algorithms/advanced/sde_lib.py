"""SDE objects of the drop-in surface (mirror of reference lib/algorithms/advanced/sde_lib.py).

Only scalar schedule arithmetic lives here (host-side plumbing on tiny [B] tensors); the score network
evaluation that dominates `RSDE.sde` runs in libzedo_hip.so through the `score_fn` handed to `reverse`.
The shipped configurations use subVPSDE only (configs/optim/*.py: training.sde = 'subvpsde'); VPSDE and
VESDE keep their schedule methods so that config switches fail late and clearly, not at import.
"""
import abc

import numpy as np
import torch


class SDE(abc.ABC):
    """Forward SDE dx = f(x,t) dt + g(t) dw on mini-batches (reference sde_lib.py:7-69)."""

    def __init__(self, N):
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        ...

    @abc.abstractmethod
    def sde(self, x, t):
        ...

    @abc.abstractmethod
    def marginal_prob(self, x, t):
        ...

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi) - torch.sum(z.reshape(z.shape[0], -1) ** 2, dim=1) / 2.0

    def discretize(self, x, t):
        """Euler-Maruyama discretisation x_{i+1} = x_i + f_i + G_i z (reference :52-69)."""
        dt = 1.0 / self.N
        drift, diffusion = self.sde(x, t)
        return drift * dt, diffusion * float(np.sqrt(dt))

    def reverse(self, score_fn, probability_flow=False):
        """Reverse-time SDE / probability-flow ODE (reference :71-109).  `score_fn(x, t, condition, mask)`."""
        fwd = self

        class RSDE(fwd.__class__):
            def __init__(self):
                self.N = fwd.N
                self.probability_flow = probability_flow

            @property
            def T(self):
                return fwd.T

            def sde(self, x, t, condition, mask):
                drift, diffusion = fwd.sde(x, t)
                score = score_fn(x, t, condition, mask)
                drift = drift - diffusion[:, None, None] ** 2 * score
                if self.probability_flow:      # the reference zeroes the diffusion for the ODE (:99)
                    diffusion = torch.zeros(1, device=drift.device)
                return drift, diffusion

            def discretize(self, x, t, condition, mask):
                f, G = fwd.discretize(x, t)
                rev_f = f - G[:, None, None] ** 2 * score_fn(x, t, condition, mask)
                return rev_f, (torch.zeros_like(G) if self.probability_flow else G)

        return RSDE()


class subVPSDE(SDE):
    """sub-VP SDE (reference :168-206): beta(t) linear, diffusion^2 = beta (1 - exp(-2 int beta))."""

    def __init__(self, beta_min=0.1, beta_max=20, N=1000, T=1):
        super().__init__(N)
        self.beta_0, self.beta_1, self._T = beta_min, beta_max, T

    @property
    def T(self):
        return self._T

    def beta(self, t):
        return self.beta_0 + t * (self.beta_1 - self.beta_0)

    def sde(self, x, t):
        b = self.beta(t)
        discount = 1.0 - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t ** 2)
        return -0.5 * b[:, None, None] * x, torch.sqrt(b * discount)

    def log_mean_coeff(self, t):
        return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0

    def marginal_prob(self, x, t):
        lmc = self.log_mean_coeff(t)
        return torch.exp(lmc)[:, None, None] * x, 1 - torch.exp(2.0 * lmc)   # note: not a square root (:197)


class VPSDE(SDE):
    """Variance-preserving SDE (reference :112-165); selectable by config, unused by the shipped ones."""

    def __init__(self, beta_min=0.1, beta_max=20, N=1000, T=1):
        super().__init__(N)
        self.beta_0, self.beta_1, self._T = beta_min, beta_max, T
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1.0 - self.discrete_betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)

    @property
    def T(self):
        return self._T

    def sde(self, x, t):
        b = self.beta_0 + t * (self.beta_1 - self.beta_0)
        return -0.5 * b[:, None, None] * x, torch.sqrt(b)

    def marginal_prob(self, x, t):
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.exp(lmc[:, None, None]) * x, torch.sqrt(1.0 - torch.exp(2.0 * lmc))

    def discretize(self, x, t):
        step = (t * (self.N - 1) / self.T).long()
        beta = self.discrete_betas.to(x.device)[step]
        alpha = self.alphas.to(x.device)[step]
        return torch.sqrt(alpha)[:, None, None] * x - x, torch.sqrt(beta)


class VESDE(SDE):
    """Variance-exploding SDE (reference :209-261); selectable by config, unused by the shipped ones."""

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000, T=1):
        super().__init__(N)
        self.sigma_min, self.sigma_max, self._T = sigma_min, sigma_max, T
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))

    @property
    def T(self):
        return self._T

    def sde(self, x, t):
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        # the constant is rounded to fp32 before the square root, as in the reference (:241-242)
        g = sigma * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min)), device=t.device))
        return torch.zeros_like(x), g

    def marginal_prob(self, x, t):
        return x, self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def prior_sampling(self, shape):
        return torch.randn(*shape) * self.sigma_max

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        s2 = self.sigma_max ** 2
        return -n / 2.0 * np.log(2 * np.pi * s2) - torch.sum(z.reshape(z.shape[0], -1) ** 2, dim=1) / (2 * s2)

    def discretize(self, x, t):
        """SMLD discretisation (reference :253-261): f = 0, G^2 = sigma_i^2 - sigma_{i-1}^2 (sigma_{-1} = 0)."""
        step = (t * (self.N - 1) / self.T).long()
        sig = self.discrete_sigmas.to(t.device)
        prev = torch.where(step == 0, torch.zeros_like(t), sig[step - 1])
        return torch.zeros_like(x), torch.sqrt(sig[step] ** 2 - prev ** 2)
