"""Posterior sampling / sampling predictives / grid search of the lean drivers against the UNMODIFIED reference
classes run on the same fit with the same torch generator (draw-for-draw where the reference is deterministic given
the generator, moments otherwise).  `not gpu`: kernel emulation, fp64 so that the same randn stream is drawn."""
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

from tests.conftest import golden_model, load_golden


@pytest.fixture
def emulated(reference_dropin):
    yield


def _pair(name, lik, hs, sow="all"):
    """(ours, reference) fitted on the same data; the reference runs its own in-tree backend where it can."""
    from laplace_amd import HipGGN
    from laplace_amd.laplace import HipLaplace

    import laplace as ref

    g = load_golden(name, lik)
    model, X, y = golden_model(name, g, dtype=torch.float32)
    loader = DataLoader(TensorDataset(X, y), batch_size=5)
    ours = HipLaplace(model, lik, sow, hs, prior_precision=0.7)
    ours.fit(loader)
    theirs = ref.Laplace(model, lik, subset_of_weights=sow, hessian_structure=hs, prior_precision=0.7, backend=HipGGN)
    theirs.fit(loader)
    return ours, theirs, X, y


@pytest.mark.parametrize("hs", ["kron", "diag", "full"])
@pytest.mark.parametrize("lik", ["classification", "regression"])
def test_sample_matches_reference_draw_for_draw(emulated, hs, lik):
    ours, theirs, X, y = _pair("mlp", lik, hs)
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    s1, s2 = ours.sample(7, generator=g1), theirs.sample(7, generator=g2)
    assert s1.shape == s2.shape == (7, ours.n_params)
    torch.testing.assert_close(s1, s2, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("hs", ["kron", "diag"])
def test_sampling_predictives_match_reference(emulated, hs):
    ours, theirs, X, y = _pair("mlp", "classification", hs)
    for pred_type in ("glm", "nn"):
        g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
        p1 = ours.predictive_samples(X, pred_type=pred_type, n_samples=6, generator=g1)
        p2 = theirs.predictive_samples(X, pred_type=pred_type, n_samples=6, generator=g2)
        torch.testing.assert_close(p1, p2, rtol=5e-4, atol=5e-5)
        f1 = ours.functional_samples(X, pred_type=pred_type, n_samples=6, generator=torch.Generator().manual_seed(2))
        f2 = theirs.functional_samples(X, pred_type=pred_type, n_samples=6, generator=torch.Generator().manual_seed(2))
        torch.testing.assert_close(f1, f2, rtol=5e-4, atol=5e-5)
    # the model's parameters are restored after NN sampling
    torch.testing.assert_close(torch.nn.utils.parameters_to_vector(ours.params), ours.mean)
    # __call__ with the MC link / NN predictive: valid distributions close to the probit answer
    probit = ours(X)
    torch.manual_seed(0)
    mc = ours(X, link_approx="mc", n_samples=4000)
    nn_ = ours(X, pred_type="nn", link_approx="mc", n_samples=2000)
    for p in (mc, nn_):
        assert torch.allclose(p.sum(-1), torch.ones(len(X)), atol=1e-5)
    assert (mc - probit).abs().max() < 0.05  # (the NN predictive is a different, non-linearised, quantity)


def test_regression_nn_predictive_and_gridsearch(emulated):
    ours, theirs, X, y = _pair("mlp", "regression", "kron")
    s1 = ours._nn_predictive_samples(X, 50, torch.Generator().manual_seed(3))
    s2 = theirs._nn_predictive_samples(X, 50, torch.Generator().manual_seed(3))
    torch.testing.assert_close(s1, s2, rtol=5e-4, atol=5e-5)
    torch.manual_seed(1)
    mu, var = ours(X, pred_type="nn", link_approx="mc", n_samples=50)
    assert mu.shape == s1.shape[1:] and var.shape == s1.shape[1:] and (var >= 0).all()
    # grid search: the chosen grid point minimises the validation loss, and the search leaves it installed
    val = DataLoader(TensorDataset(X, y), batch_size=5)
    best = ours.gridsearch_prior_precision(val, -2, 2, 9)
    losses = []
    for pp in torch.logspace(-2, 2, 9):
        ours.prior_precision = pp
        losses.append(float(((ours(X)[0] - y) ** 2).mean()))
    assert abs(float(best) - float(torch.logspace(-2, 2, 9)[int(torch.tensor(losses).argmin())])) < 1e-6


@pytest.mark.parametrize("hs", ["kron", "diag", "full"])
@pytest.mark.parametrize("sow", ["all", "last_layer"])
def test_joint_regression_predictive_matches_reference(emulated, hs, sow):
    """`joint=True` (baselaplace.py:1329-1331): the [B*C, B*C] GLM covariance over a batch."""
    ours, theirs, X, y = _pair("mlp", "regression", hs, sow)
    mu1, cov1 = ours(X, joint=True)
    mu2, cov2 = theirs(X, joint=True)
    n = X.shape[0] * mu2.numel() // X.shape[0]
    assert cov1.shape == cov2.shape == (n, n) and mu1.shape == mu2.shape
    torch.testing.assert_close(mu1, mu2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cov1, cov2, rtol=5e-4, atol=1e-6)
    # its diagonal blocks are the per-sample covariances of the ordinary call
    f_mu, f_var = ours(X)
    C = f_mu.shape[1]
    for i in range(X.shape[0]):
        torch.testing.assert_close(cov1[i * C:(i + 1) * C, i * C:(i + 1) * C], f_var[i], rtol=5e-4, atol=1e-6)


@pytest.mark.parametrize("hs", ["kron", "diag", "full"])
@pytest.mark.parametrize("sow", ["all", "last_layer"])
def test_state_dict_interchange_with_reference(emulated, hs, sow, tmp_path):
    """Checkpoints are interchangeable in both directions (baselaplace.py:1509-1557; Kron stores the factors,
    :1867-1879) and survive torch.save / torch.load."""
    import laplace as ref
    from laplace_amd import HipGGN
    from laplace_amd.laplace import HipLaplace

    ours, theirs, X, y = _pair("mlp", "classification", hs, sow)
    want = ours(X)
    # ours -> file -> reference class
    path = tmp_path / "la.pt"
    torch.save(ours.state_dict(), path)
    fresh_ref = ref.Laplace(theirs.model if sow == "all" else theirs.model.model, "classification", subset_of_weights=sow,
                            hessian_structure=hs, backend=HipGGN)
    fresh_ref.load_state_dict(torch.load(path, weights_only=False))
    torch.testing.assert_close(fresh_ref(X), want, rtol=5e-4, atol=1e-5)
    # reference -> ours
    fresh = HipLaplace(ours.model if sow == "all" else ours.model.model, "classification", sow, hs)
    fresh.load_state_dict(theirs.state_dict())
    torch.testing.assert_close(fresh(X), want, rtol=5e-4, atol=1e-5)
    torch.testing.assert_close(fresh.log_marginal_likelihood(), ours.log_marginal_likelihood(), rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        other = HipLaplace(ours.model if sow == "all" else ours.model.model, "classification", sow,
                           "diag" if hs != "diag" else "kron")
        other.load_state_dict(ours.state_dict())
