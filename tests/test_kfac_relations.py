"""KFAC relations R1-R10 that the reference's tests pin independently of any
third-party backend, asserted on the oracle's KFAC restatement.  The same relations are
re-run against the HIP backend in tests/test_gpu_backend.py."""
import pytest
import torch

from oracle import curvature_oracle as co
from oracle.fixtures import make_fixture

LIKS = ("classification", "regression")


def _setup(name, lik):
    model, X, y_cls, y_reg = make_fixture(name)
    return model, X, (y_cls if lik == "classification" else y_reg)


def _diag_ggn(model, X, lik):
    Js, f = co.jacobians(model, X)
    return co.ggn_diag(Js, co.functional_hessian(f, lik))


@pytest.mark.parametrize("lik", LIKS)
def test_R1_single_datum_exact(lik):
    """tests/test_curv_backends_backpack.py:120-128,156-164."""
    model, X, y = _setup("mlp", lik)
    _, kf = co.kfac_ggn(model, X[:1], y[:1], 1, lik)
    torch.testing.assert_close(co.kron_diag(kf), _diag_ggn(model, X[:1], lik), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("lik", LIKS)
def test_R2_R3_repeated_datum(lik):
    """tests/test_curv_backends_asdl.py:296-315; tests/test_curv_backends_curvlinops.py:308-333."""
    model, X, y = _setup("mlp", lik)
    X7, y7 = X[:1].repeat(7, 1), y[:1].repeat(7, *([1] * (y.ndim - 1)))
    l7, k7 = co.kfac_ggn(model, X7, y7, 7, lik)
    l1, k1 = co.kfac_ggn(model, X[:1], y[:1], 1, lik)
    torch.testing.assert_close(co.kron_diag(k7), _diag_ggn(model, X7, lik), rtol=1e-9, atol=1e-12)
    torch.testing.assert_close(7 * co.kron_diag(k1), co.kron_diag(k7), rtol=1e-9, atol=1e-12)
    torch.testing.assert_close(7 * l1, l7)


@pytest.mark.parametrize("name", ["mlp", "conv", "resnetish", "seqlin"])
@pytest.mark.parametrize("lik", LIKS)
def test_R4_minibatch_additivity(name, lik):
    """tests/test_curv_backends_curvlinops.py:207-238,277-293 (factors add with a fixed N)."""
    model, X, y = _setup(name, lik)
    N = X.shape[0]
    lf, kf = co.kfac_ggn(model, X, y, N, lik)
    la, ka = co.kfac_ggn(model, X[:3], y[:3], N, lik)
    lb, kb = co.kfac_ggn(model, X[3:], y[3:], N, lik)
    torch.testing.assert_close(la + lb, lf)
    # G adds exactly; A adds exactly because both carry 1/N  => the diag adds only when
    # expressed through the summed factors
    ks = co.kron_add(ka, kb)
    for F_, G_ in zip(ks, kf):
        for a, b in zip(F_, G_):
            torch.testing.assert_close(a, b, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("lik", LIKS)
def test_R5_norm_ratio(lik):
    """:241-247 (MLP within 10 %), :263-274 (conv single datum 10 %), :296-305 (conv batch 1 %)."""
    model, X, y = _setup("mlp", lik)
    _, kf = co.kfac_ggn(model, X, y, X.shape[0], lik)
    r = co.kron_diag(kf).norm() / _diag_ggn(model, X, lik).norm()
    assert abs(r - 1) < 0.1
    model, X, y = _setup("conv", lik)
    _, kf = co.kfac_ggn(model, X[:1], y[:1], 1, lik)
    r = co.kron_diag(kf).norm() / _diag_ggn(model, X[:1], lik).norm()
    assert abs(r - 1) < 0.1
    _, kf = co.kfac_ggn(model, X, y, X.shape[0], lik)
    r = co.kron_diag(kf).norm() / _diag_ggn(model, X, lik).norm()
    assert abs(r - 1) < 0.011


@pytest.mark.parametrize("name", ["mlp", "conv", "resnetish", "seqlin"])
def test_R6_block_shapes(name):
    """tests/test_matrix.py:32-48 / utils/matrix.py:33-77."""
    model, X, y = _setup(name, "classification")
    _, kf = co.kfac_ggn(model, X, y, X.shape[0], "classification")
    params = co.trainable_params(model)
    assert len(kf) == len(params)
    for F_, p in zip(kf, params):
        if p.ndim == 1:
            assert len(F_) == 1 and F_[0].shape == (p.shape[0],) * 2
        else:
            d_in = p[0].numel()
            assert F_[0].shape == (p.shape[0],) * 2 and F_[1].shape == (d_in, d_in)
    assert co.kron_diag(kf).numel() == sum(p.numel() for p in params)


def test_R7_expand_vs_reduce():
    """tests/test_curv_backends_curvlinops.py:179-192."""
    model, X, y = _setup("conv", "classification")
    le, ke = co.kfac_ggn(model, X, y, X.shape[0], "classification", kfac_approx="expand")
    lr, kr = co.kfac_ggn(model, X, y, X.shape[0], "classification", kfac_approx="reduce")
    torch.testing.assert_close(le, lr)
    assert not torch.allclose(co.kron_diag(ke), co.kron_diag(kr))


@pytest.mark.parametrize("name", ["mlp", "conv", "resnetish", "seqlin"])
@pytest.mark.parametrize("lik", LIKS)
def test_R9_bias_block_is_exact(name, lik):
    """Linear bias block [G] == bias-bias block of the dense GGN (curvature.py:375-411).
    (A conv bias under 'expand' carries sum_{n,l} g g^T, which is NOT the exact block - the
    exact one sums over positions first - so conv biases are excluded; likewise the bias of a
    Linear applied along a sequence, the `seqlin` fixture's first layer.)"""
    model, X, y = _setup(name, lik)
    _, kf = co.kfac_ggn(model, X, y, X.shape[0], lik)
    Js, f = co.jacobians(model, X)
    H = co.ggn_full(Js, co.functional_hessian(f, lik))
    off = 0
    conv_biases = {id(m.bias) for m in model.modules() if isinstance(m, torch.nn.Conv2d) and m.bias is not None}
    if name == "seqlin":
        conv_biases.add(id(model[0].bias))
    for F_, p in zip(kf, co.trainable_params(model)):
        n = p.numel()
        if p.ndim == 1 and id(p) not in conv_biases:
            torch.testing.assert_close(F_[0], H[off : off + n, off : off + n], rtol=1e-9, atol=1e-12)
        off += n


def test_R10_last_layer_regression_blocks():
    """Linear last layer, MSE: G (x) A == weight-weight block, [G] == bias-bias block."""
    model, X, _, y = make_fixture("mlp")
    last = model[2]
    for p in model[0].parameters():
        p.requires_grad_(False)
    _, kf = co.kfac_ggn(model, X, y, X.shape[0], "regression")
    Js, f = co.jacobians(model, X)
    H = co.ggn_full(Js, co.functional_hessian(f, "regression"))
    nw = last.weight.numel()
    torch.testing.assert_close(co.kron_product(kf[0][0], kf[0][1]), H[:nw, :nw], rtol=1e-9, atol=1e-12)
    torch.testing.assert_close(kf[1][0], H[nw:, nw:], rtol=1e-9, atol=1e-12)
