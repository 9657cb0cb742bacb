"""VPSDE / VESDE / subVPSDE schedule methods, their reverse-time forms and every registered
predictor / corrector update rule against captures from the reference (tools/gen_golden.py::gen_samplers,
reference sde_lib.py:112-261, sampling.py:180-331).  The score function is analytic and torch.randn_like is
replaced by a numpy-Philox stream in both the capture and this test, so only the update arithmetic is compared.
Host-side torch code on [B]-sized tensors: runs on CPU, fp32 tolerance 2e-6 relative (most entries are bit-equal)."""
import numpy as np
import pytest
import torch

from lib.algorithms.advanced import sampling, sde_lib

SDES = dict(vpsde=(sde_lib.VPSDE, dict(beta_min=0.1, beta_max=20.0, N=1000, T=1.0)),
            subvpsde=(sde_lib.subVPSDE, dict(beta_min=0.1, beta_max=20.0, N=1000, T=1.0)),
            vesde=(sde_lib.VESDE, dict(sigma_min=0.01, sigma_max=50.0, N=1000, T=1.0)))
PREDS = [("euler_maruyama", False), ("euler_maruyama", True), ("reverse_diffusion", False),
         ("reverse_diffusion", True), ("ancestral_sampling", False)]
ERRORS = dict(NotImplementedError=NotImplementedError, AssertionError=AssertionError, AttributeError=AttributeError)


class DetNoise:
    def __init__(self):
        self.calls = 0

    def __call__(self, x):
        g = np.random.Generator(np.random.Philox(key=[555, self.calls]))
        self.calls += 1
        return torch.tensor(g.standard_normal(tuple(x.shape)), dtype=x.dtype)


def analytic_score(x, t, condition, mask):
    return -(x - 0.3 * condition) / (0.5 + t)[:, None, None]


def close(a, b):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(np.broadcast_to(a, b.shape) if a.shape != b.shape else a, b, rtol=2e-6, atol=1e-7)


@pytest.fixture()
def inputs(golden):
    g = golden("samplers")
    return g, torch.tensor(g["x"]), torch.tensor(g["cond"]), torch.tensor(g["t"])


@pytest.mark.parametrize("name", list(SDES))
def test_sde_schedule_methods(inputs, name):
    g, x, cond, t = inputs
    sde = SDES[name][0](**SDES[name][1])
    for got, key in zip(sde.sde(x, t), ("drift", "diffusion")):
        close(got, g[f"{name}_{key}"])
    for got, key in zip(sde.marginal_prob(x, t), ("mean", "std")):
        close(got, g[f"{name}_{key}"])
    for got, key in zip(sde.discretize(x, t), ("disc_f", "disc_G")):
        close(got, g[f"{name}_{key}"])
    mask = torch.zeros_like(x)
    for pf in (False, True):
        r = sde.reverse(analytic_score, pf)
        assert r.N == sde.N and r.T == sde.T
        for got, key in zip(r.sde(x, t, cond, mask), ("rsde_drift", "rsde_diffusion")):
            close(got, g[f"{name}_{key}_pf{int(pf)}"])
        for got, key in zip(r.discretize(x, t, cond, mask), ("rdisc_f", "rdisc_G")):
            close(got, g[f"{name}_{key}_pf{int(pf)}"])


@pytest.mark.parametrize("name", list(SDES))
@pytest.mark.parametrize("pname,pf", PREDS)
def test_predictor_update_rules(inputs, monkeypatch, name, pname, pf):
    g, x, cond, t = inputs
    sde = SDES[name][0](**SDES[name][1])
    key = f"{name}_pred_{pname}_pf{int(pf)}"
    monkeypatch.setattr(torch, "randn_like", DetNoise())
    if key + "_raises" in g.files:
        with pytest.raises(ERRORS[str(g[key + "_raises"])]):
            sampling.get_predictor(pname)(sde, analytic_score, pf).update_fn(x, t, cond, torch.zeros_like(x))
        return
    xn, xm = sampling.get_predictor(pname)(sde, analytic_score, pf).update_fn(x, t, cond, torch.zeros_like(x))
    close(xn, g[key + "_x"])
    close(xm, g[key + "_mean"])


@pytest.mark.parametrize("name", list(SDES))
@pytest.mark.parametrize("cname", ["langevin", "ald"])
def test_corrector_update_rules(inputs, monkeypatch, name, cname):
    g, x, cond, t = inputs
    sde = SDES[name][0](**SDES[name][1])
    key = f"{name}_corr_{cname}"
    monkeypatch.setattr(torch, "randn_like", DetNoise())
    if key + "_raises" in g.files:
        with pytest.raises(ERRORS[str(g[key + "_raises"])]):
            sampling.get_corrector(cname)(sde, analytic_score, 0.16, 2).update_fn(x, t, cond, torch.zeros_like(x))
        return
    xn, xm = sampling.get_corrector(cname)(sde, analytic_score, 0.16, 2).update_fn(x, t, cond, torch.zeros_like(x))
    close(xn, g[key + "_x"])
    close(xm, g[key + "_mean"])


def test_ancestral_sampling_rejects_probability_flow():
    with pytest.raises(AssertionError):
        sampling.get_predictor("ancestral_sampling")(sde_lib.VPSDE(), analytic_score, True)


def test_registries_hold_the_reference_names():
    assert sorted(sampling._PREDICTORS) == ["ancestral_sampling", "euler_maruyama", "none", "reverse_diffusion"]
    assert sorted(sampling._CORRECTORS) == ["ald", "langevin", "none"]
