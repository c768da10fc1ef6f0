"""MC Fisher (``stochastic=True``; laplace/curvature/curvature.py:341-364, KFAC: curvlinops FisherType.MC).

A stochastic estimator cannot be compared draw-for-draw with the reference's RNG stream, so parity is pinned in
three ways: (1) GIVEN the sampled functional gradients, diag / full / KFAC equal the oracle's restatement of the
reference formulas on the same draws; (2) the draws have the reference's distribution (structure of
``softmax(f) - onehot``, moments); (3) with many draws the estimate converges to the exact GGN.
`not gpu`: kernel emulation; `gpu`: HIP kernels.
"""
import os

import pytest
import torch

from oracle import curvature_oracle as O
from oracle.fixtures import FIXTURES
from tests.conftest import golden_model, load_golden

LIKS = ("classification", "regression")


def rel(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return (got - want).abs().max().item() / (want.abs().max().item() + 1e-30)


@pytest.fixture
def emulated():
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    yield
    _lib.set_kernels_for_testing(prev)


def _draws(f64, lik, S, seed):
    g = torch.Generator().manual_seed(seed)
    B, C = f64.shape
    if lik == "regression":
        return -torch.randn(S, B, C, generator=g, dtype=torch.float64)
    p = torch.softmax(f64, -1)
    idx = torch.multinomial(p, S, replacement=True, generator=g)
    return p.unsqueeze(0) - torch.nn.functional.one_hot(idx.t(), C).to(torch.float64)


def check_given_draws(name, lik, dev):
    from laplace_amd import HipGGN

    g = load_golden(name, lik)
    model64, X64, y64 = golden_model(name, g, dtype=torch.float64)
    model, X, y = golden_model(name, g, dtype=torch.float32, device=dev)
    S, N = 3, 40
    with torch.no_grad():
        f64 = model64(X64)
    draws = _draws(f64, lik, S, seed=5)
    Js, _ = O.jacobians(model64, X64)
    F_mid = O.mc_functional_fisher(draws)
    b = HipGGN(model, lik, stochastic=True, num_samples=S)
    b._mc_functional_grads = lambda f: draws.to(f.dtype).to(f.device)
    loss, h = b.diag(X, y)
    assert rel(h, O.ggn_diag(Js, F_mid)) < 1e-4
    loss_f, H = b.full(X, y)
    assert rel(H, O.ggn_full(Js, F_mid)) < 1e-4
    want_loss = O.loss_sum(f64, y64, lik)  # includes the likelihood factor
    assert rel(loss, want_loss) < 1e-5 and rel(loss_f, want_loss) < 1e-5
    loss_k, Hk = b.kron(X, y, N=N)
    ol, ok = O.kfac_ggn(model64, X64, y64, N, lik, mc_grads=draws)
    assert rel(loss_k, ol) < 1e-5
    assert len(Hk.kfacs) == len(ok)
    for F_, G_ in zip(Hk.kfacs, ok):
        for a, w in zip(F_, G_):
            assert rel(a, w) < 1e-4
    # last-layer mode goes through the generic row path (the structured dense kernel is exact-GGN only)
    import copy

    from laplace_amd.mirror import FeatureExtractor

    bl = HipGGN(FeatureExtractor(copy.deepcopy(model)).to(dev), lik, last_layer=True, stochastic=True, num_samples=S)
    bl._mc_functional_grads = lambda f: draws.to(f.dtype).to(f.device)
    _, Hl = bl.full(X, y)
    with torch.no_grad():
        phi = model64[:-1](X64)
    Jl = O.last_layer_jacobians(phi.reshape(phi.shape[0], -1), f64.shape[1], model64[-1].bias is not None)
    assert rel(Hl, O.ggn_full(Jl, F_mid)) < 1e-4


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("lik", LIKS)
def test_mc_fisher_given_draws_on_emulation(emulated, name, lik):
    check_given_draws(name, lik, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("lik", LIKS)
def test_mc_fisher_given_draws_gpu(name, lik):
    check_given_draws(name, lik, "cuda")


def check_sampler_and_convergence(dev):
    from laplace_amd import HipGGN

    g = load_golden("mlp", "classification")
    model, X, y = golden_model("mlp", g, dtype=torch.float32, device=dev)
    S = 6000
    b = HipGGN(model, "classification", stochastic=True, num_samples=S)
    b.generator = torch.Generator(device=dev).manual_seed(11)
    with torch.no_grad():
        f = model(X)
    gs = b._mc_functional_grads(f)
    p = torch.softmax(f, -1)
    assert gs.shape == (S, X.shape[0], f.shape[1])
    assert gs.sum(-1).abs().max() < 1e-5                       # p - onehot sums to zero
    onehot = p.unsqueeze(0) - gs
    assert ((onehot - onehot.round()).abs().max() < 1e-5) and (onehot.round().sum(-1) == 1).all()
    assert (onehot.mean(0) - p).abs().max() < 5.0 / S ** 0.5    # E[onehot] = p
    # reproducible with the generator, and convergent to the exact GGN
    b.generator.manual_seed(3)
    _, h1 = b.diag(X, y)
    b.generator.manual_seed(3)
    _, h2 = b.diag(X, y)
    assert torch.equal(h1, h2)
    exact = HipGGN(model, "classification")
    _, he = exact.diag(X, y)
    assert rel(h1, he) < 0.1
    _, Hk = b.kron(X, y, N=20)
    _, He = exact.kron(X, y, N=20)
    for F_, G_ in zip(Hk.kfacs, He.kfacs):
        assert rel(F_[0], G_[0]) < 0.1
    # regression draws: standard normal
    gr = load_golden("mlp", "regression")
    mr, Xr, yr = golden_model("mlp", gr, dtype=torch.float32, device=dev)
    br = HipGGN(mr, "regression", stochastic=True, num_samples=S)
    with torch.no_grad():
        e = br._mc_functional_grads(mr(Xr))
    assert abs(e.mean().item()) < 0.02 and abs(e.var().item() - 1.0) < 0.05


def test_mc_sampler_and_convergence_on_emulation(emulated):
    check_sampler_and_convergence("cpu")


@pytest.mark.gpu
def test_mc_sampler_and_convergence_gpu():
    check_sampler_and_convergence("cuda")
