"""fit -> posterior -> GLM predictive -> marginal likelihood through laplace_amd's lean drivers,
against the golden outputs of the reference's own Laplace classes (tests/golden/*.npz:
``la.<subset>.<structure>.*``, produced by oracle/make_golden.py).

`not gpu`: host logic on the CPU kernel emulation.  `gpu`: the same assertions on the HIP kernels.
"""
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

from oracle.fixtures import FIXTURES
from oracle.make_golden import PRIOR_PREC, SIGMA_NOISE
from tests.conftest import golden_kfacs, golden_model, load_golden

LIKS = ("classification", "regression")
CASES = [(n, l, s, h) for n in FIXTURES for l in LIKS for s in ("all", "last_layer") for h in ("diag", "full", "kron")]


def rel(got, want):
    got = torch.as_tensor(got).detach().double().cpu()
    want = torch.as_tensor(want).detach().double().cpu()
    return (got - want).abs().max().item() / (want.abs().max().item() + 1e-30)


def run_case(name, lik, sow, hs, dev, tol=1e-4, fused=True):
    from laplace_amd.laplace import HipLaplace

    g = load_golden(name, lik)
    model, X, y = golden_model(name, g, dtype=torch.float32, device=dev)
    sig = SIGMA_NOISE if lik == "regression" else 1.0
    la = HipLaplace(model, lik, sow, hs, prior_precision=PRIOR_PREC, sigma_noise=sig)
    if hs == "kron":
        la.fit(DataLoader(TensorDataset(X, y), batch_size=5), fused=fused)
    else:
        la.fit(DataLoader(TensorDataset(X, y), batch_size=5))
    tag = f"la.{sow}.{hs}"
    assert rel(la.loss, g[f"{tag}.loss"]) < tol, "loss"
    if hs == "kron":
        for F_, G_ in zip(la.H_facs.kfacs, golden_kfacs(g, f"{tag}.H")):
            for a, w in zip(F_, G_):
                assert rel(a, w) < tol, "accumulated kfacs"
    else:
        assert rel(la.H, g[f"{tag}.H"]) < tol, "accumulated H"
    f_mu, f_var = la._glm_predictive_distribution(X)
    assert rel(f_mu, g[f"{tag}.f_mu"]) < tol, "f_mu"
    assert rel(f_var, g[f"{tag}.f_var"]) < tol, f"f_var {rel(f_var, g[f'{tag}.f_var']):.2e}"
    assert rel(la.log_det_posterior_precision, g[f"{tag}.logdet_post"]) < tol, "logdet"
    assert rel(la.log_marginal_likelihood(), g[f"{tag}.marglik"]) < tol, "marglik"
    # link function sanity (baselaplace.py:650-664)
    out = la(X)
    if lik == "classification":
        assert torch.allclose(out.sum(-1), torch.ones(len(X), device=dev), atol=1e-5)


@pytest.fixture
def emulated():
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    yield
    _lib.set_kernels_for_testing(prev)


@pytest.mark.parametrize("name,lik,sow,hs", CASES)
def test_e2e_host_logic_on_emulation(emulated, name, lik, sow, hs):
    run_case(name, lik, sow, hs, "cpu")


@pytest.mark.parametrize("name", FIXTURES)
def test_e2e_kron_unfused_loop_on_emulation(emulated, name):
    """the reference's literal `self.H += backend.kron(...)` loop gives the same factors"""
    run_case(name, "classification", "all", "kron", "cpu", fused=False)


@pytest.mark.gpu
@pytest.mark.parametrize("name,lik,sow,hs", CASES)
def test_e2e_gpu(name, lik, sow, hs):
    run_case(name, lik, sow, hs, "cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_e2e_kron_unfused_gpu(name):
    run_case(name, "regression", "all", "kron", "cuda", fused=False)


def test_marglik_prior_optimisation_moves_uphill(emulated):
    from laplace_amd.laplace import HipLaplace

    g = load_golden("mlp", "regression")
    model, X, y = golden_model("mlp", g, dtype=torch.float32)
    la = HipLaplace(model, "regression", "all", "kron", prior_precision=1.0, sigma_noise=0.8)
    la.fit(DataLoader(TensorDataset(X, y), batch_size=5))
    before = la.log_marginal_likelihood().item()
    la.optimize_prior_precision(method="marglik", n_steps=30, lr=0.1, prior_structure="scalar")
    assert la.log_marginal_likelihood().item() > before


def _sampling_smoke(dev):
    """posterior samples have the posterior's moments; sampling predictives are valid and leave the model intact"""
    from laplace_amd.laplace import HipLaplace

    for hs in ("kron", "diag", "full"):
        g = load_golden("mlp", "classification")
        model, X, y = golden_model("mlp", g, dtype=torch.float32, device=dev)
        la = HipLaplace(model, "classification", "all", hs, prior_precision=PRIOR_PREC)
        la.fit(DataLoader(TensorDataset(X, y), batch_size=5))
        gen = torch.Generator(device=dev).manual_seed(0)
        s = la.sample(20000, generator=gen)
        assert s.shape == (20000, la.n_params) and torch.isfinite(s).all()
        assert (s.mean(0) - la.mean).abs().max() < 0.05 * (1 + la.mean.abs().max())
        if hs == "diag":
            assert rel(s.var(0), la.posterior_variance) < 0.1
        elif hs == "full":
            assert rel(torch.cov(s.T), la.posterior_covariance) < 0.1
        else:
            want = la.posterior_precision.to_matrix(exponent=-1)
            assert rel(torch.cov(s.T), want) < 0.1
        p = la.predictive_samples(X, pred_type="nn", n_samples=7, generator=gen)
        assert p.shape == (7, len(X), 2) and torch.allclose(p.sum(-1), torch.ones(7, len(X), device=dev), atol=1e-5)
        p = la(X, pred_type="glm", link_approx="mc", n_samples=50, generator=gen)
        assert torch.allclose(p.sum(-1), torch.ones(len(X), device=dev), atol=1e-5)
        assert torch.allclose(torch.nn.utils.parameters_to_vector(la.params), la.mean)


def test_sampling_smoke_on_emulation(emulated):
    _sampling_smoke("cpu")


@pytest.mark.gpu
def test_sampling_smoke_gpu():
    _sampling_smoke("cuda")


def test_parameters_outside_linear_and_conv_take_the_generic_route():
    """`jacobians` / `diag` / `full` of a model with parameters the kernels have no rule for (LayerNorm, an unfrozen
    BatchNorm affine): the reference's functorch route (laplace/curvature/curvature.py:88-129, 375-433) serves them — no
    NotImplementedError — and only `kron` refuses, as the reference does (docs/index.md:364-366)."""
    import pytest
    from torch.func import functional_call, jacrev, vmap

    from laplace_amd import HipGGN, _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        torch.manual_seed(0)
        for m in (torch.nn.Sequential(torch.nn.Linear(4, 6), torch.nn.LayerNorm(6), torch.nn.Tanh(), torch.nn.Linear(6, 3)),
                  torch.nn.Sequential(torch.nn.Conv2d(2, 4, 3, padding=1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                                      torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(4, 3)).eval()):
            conv = isinstance(m[0], torch.nn.Conv2d)
            X = torch.randn(5, 2, 6, 6) if conv else torch.randn(5, 4)
            y = torch.randint(3, (5,))
            b = HipGGN(m, "classification")
            assert b._tape().uncovered
            Js, f = b.jacobians(X)
            params = {k: v for k, v in m.named_parameters()}
            buffers = {k: v for k, v in m.named_buffers()}
            J = vmap(lambda x: jacrev(lambda p: functional_call(m, (p, buffers), (x[None],))[0])(params))(X)
            want = torch.cat([J[k].reshape(5, 3, -1) for k in params], -1)
            assert torch.allclose(Js, want, atol=1e-6)
            _, d = b.diag(X, y)
            _, H = b.full(X, y)
            assert torch.allclose(torch.diagonal(H), d, rtol=1e-5, atol=1e-7)
            p = torch.softmax(f, -1)
            Hw = torch.einsum("ncp,nck,nkq->pq", want, torch.diag_embed(p) - p[:, :, None] * p[:, None, :], want)
            assert torch.allclose(H, Hw, rtol=1e-4, atol=1e-6)
            with pytest.raises(NotImplementedError, match="KFAC supports"):
                b.kron(X, y, 5)
    finally:
        _lib.set_kernels_for_testing(prev)


@pytest.mark.parametrize("name,lik", [("mlp", "regression"), ("conv", "classification")])
@pytest.mark.parametrize("fused", [True, False])
def test_fp64_model_through_the_default_kron_fit_and_predictive(emulated, name, lik, fused):
    """ADVICE (round 4): the fp32 twin wrapped `kron` / `diag` / `full` / `jacobians` only — `HipLaplace(model.double())
    .fit(loader)` (the default fused accumulator) and the fused predictive ended in TypeError, which nothing falls back
    from.  The accumulator now runs on the twin (factors come back in the model's dtype) and the fused predictive
    hands a non-fp32 model to the reference's route.  Reference: its tests run fp64 models through every backend
    (tests/test_baselaplace.py:895-934)."""
    from laplace_amd.laplace import HipLaplace

    g = load_golden(name, lik)
    model, X, y = golden_model(name, g, dtype=torch.float64, device="cpu")
    sig = SIGMA_NOISE if lik == "regression" else 1.0
    la = HipLaplace(model, lik, "all", "kron", prior_precision=PRIOR_PREC, sigma_noise=sig)
    la.fit(DataLoader(TensorDataset(X, y), batch_size=5), fused=fused)
    tag = "la.all.kron"
    for F_, G_ in zip(la.H_facs.kfacs, golden_kfacs(g, f"{tag}.H")):
        for a, w in zip(F_, G_):
            assert a.dtype == torch.float64 and rel(a, w) < 1e-4
    f_mu, f_var = la._glm_predictive_distribution(X)
    assert f_mu.dtype == torch.float64 and f_var.dtype == torch.float64
    assert rel(f_mu, g[f"{tag}.f_mu"]) < 1e-4 and rel(f_var, g[f"{tag}.f_var"]) < 1e-4


def test_dict_style_inputs_are_cast_for_the_fp32_twin():
    """ADVICE (round 4): `_to32` returned a non-dict MutableMapping (UserDict, BatchEncoding) unchanged and did not look
    into tuples / lists"""
    from collections import UserDict

    from laplace_amd.backend import _HipCurvatureMixin as M

    d = UserDict({"input_ids": torch.ones(2, 3, dtype=torch.long), "feat": torch.ones(2, 3, dtype=torch.float64)})
    out = M._to32(d)
    assert isinstance(out, UserDict) and out["feat"].dtype == torch.float32 and out["input_ids"].dtype == torch.long
    assert d["feat"].dtype == torch.float64  # (the caller's container is not modified)
    t = M._to32((torch.ones(2, dtype=torch.float16), [torch.ones(2, dtype=torch.float64)]))
    assert t[0].dtype == torch.float32 and t[1][0].dtype == torch.float32 and isinstance(t, tuple) and isinstance(t[1], list)
