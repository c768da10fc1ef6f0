"""nn.Linear applied along a sequence (weight sharing over T positions, the transformer case; curvlinops' KFAC-expand
treats it like a convolution's positions): Jacobians, diagonal / dense GGN, KFAC factors and the
Jacobian-free Kron / diag GLM predictive against the fp64 oracle.  Host logic on the kernel emulation + GPU variant."""
import pytest
import torch
from torch import nn

from oracle import curvature_oracle as co


class SeqNet(nn.Module):
    """[B, T, 5] -> Linear -> tanh -> Linear (both shared over T) -> mean over T -> [B, C]"""

    def __init__(self, C=3, bias=True):
        super().__init__()
        self.l1 = nn.Linear(5, 6, bias=bias)
        self.act = nn.Tanh()
        self.l2 = nn.Linear(6, C, bias=bias)

    def forward(self, x):
        return self.l2(self.act(self.l1(x))).mean(1)


class _Loader(list):
    pass


def rel(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return (got - want).abs().max().item() / (want.abs().max().item() + 1e-30)


def _run(dev, lik, bias):
    from laplace_amd import HipGGN
    from laplace_amd.laplace import HipLaplace

    torch.manual_seed(3)
    C = 3 if lik == "classification" else 2
    model = SeqNet(C, bias).to(dev)
    X = torch.randn(9, 4, 5, device=dev)
    y = torch.randint(C, (9,), device=dev) if lik == "classification" else torch.randn(9, C, device=dev)
    m64 = SeqNet(C, bias).double()
    m64.load_state_dict({k: v.double().cpu() for k, v in model.state_dict().items()})
    X64, y64 = X.double().cpu(), (y.cpu() if lik == "classification" else y.double().cpu())
    Js64, f64 = co.jacobians(m64, X64)
    Hl = co.functional_hessian(f64, lik)

    backend = HipGGN(model, lik)
    Js, f = backend.jacobians(X)
    assert rel(f, f64) < 1e-5 and rel(Js, Js64) < 1e-4
    _, h = backend.diag(X, y)
    assert rel(h, co.ggn_diag(Js64, Hl)) < 1e-4  # (MSE-sum Hessian 2I x factor 1/2 = I: no extra factor)
    _, H = backend.full(X, y)
    assert rel(H, co.ggn_full(Js64, Hl)) < 1e-4

    loader = _Loader([(X[:5], y[:5]), (X[5:], y[5:])])
    loader.dataset = range(9)
    # KFAC factors, accumulated over two minibatches
    la = HipLaplace(model, lik, "all", "kron", prior_precision=0.7)
    la.fit(loader)
    want = None
    for xb, yb in ((X64[:5], y64[:5]), (X64[5:], y64[5:])):
        _, kf = co.kfac_ggn(m64, xb, yb, 9, lik)
        want = kf if want is None else co.kron_add(want, kf)
    for F_, G_ in zip(la.H_facs.kfacs, want):
        for a_, w_ in zip(F_, G_):
            assert rel(a_, w_) < 1e-4
    # Jacobian-free Kron predictive == J P^-1 J^T with the autograd Jacobian
    f_mu, f_var = la._glm_predictive_distribution(X)
    Qs, ls = co.kron_decompose(want)
    sig = float(la.sigma_noise)
    ref = co.functional_variance_kron(Js64, Qs, ls, 0.7, h_factor=1.0 / sig**2)
    assert rel(f_var, ref) < 1e-4
    # diagonal posterior
    ld = HipLaplace(model, lik, "all", "diag", prior_precision=0.7)
    ld.fit(loader)
    _, f_var_d = ld._glm_predictive_distribution(X)
    post_var = 1.0 / (co.ggn_diag(Js64, Hl) / sig**2 + 0.7)
    assert rel(f_var_d, co.functional_variance_diag(Js64, post_var)) < 1e-4


@pytest.mark.parametrize("lik", ["classification", "regression"])
@pytest.mark.parametrize("bias", [True, False])
def test_sequence_linear_on_emulation(lik, bias):
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        _run("cpu", lik, bias)
    finally:
        _lib.set_kernels_for_testing(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("lik", ["classification", "regression"])
def test_sequence_linear_gpu(lik):
    _run("cuda", lik, True)


class WideConvNet(nn.Module):
    """a conv wide enough (32 x 72) for the MFMA-tile route of the exact diagonal"""

    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(8, 32, 3, padding=1)
        self.act = nn.Tanh()
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.flat = nn.Flatten()
        self.fc = nn.Linear(32, 3)

    def forward(self, x):
        return self.fc(self.flat(self.pool(self.act(self.c1(x)))))


def _run_wide_conv(dev):
    from laplace_amd import HipEF, HipGGN

    torch.manual_seed(5)
    model = WideConvNet().to(dev)
    X = torch.randn(6, 8, 5, 5, device=dev)
    y = torch.randint(3, (6,), device=dev)
    m64 = WideConvNet().double()
    m64.load_state_dict({k: v.double().cpu() for k, v in model.state_dict().items()})
    Js64, f64 = co.jacobians(m64, X.double().cpu())
    _, h = HipGGN(model, "classification").diag(X, y)
    assert rel(h, co.ggn_diag(Js64, co.functional_hessian(f64, "classification"))) < 1e-4
    Gs, _ = co.per_sample_gradients(m64, X.double().cpu(), y.cpu(), "classification")
    _, h_ef = HipEF(model, "classification").diag(X, y)
    assert rel(h_ef, co.ef_diag(Gs, "classification")) < 1e-4


def test_wide_conv_exact_diagonal_on_emulation():
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        _run_wide_conv("cpu")
    finally:
        _lib.set_kernels_for_testing(prev)


@pytest.mark.gpu
def test_wide_conv_exact_diagonal_gpu():
    _run_wide_conv("cuda")


class ManyOutputsNet(nn.Module):
    """13 outputs: more than the predictive kernel holds in accumulators at once (10) -> 5-output block pairs"""

    def __init__(self, seq: bool):
        super().__init__()
        self.seq = seq
        self.body = nn.Linear(5, 7) if seq else nn.Conv2d(2, 6, 3, padding=1)
        self.act = nn.Tanh()
        self.head = nn.Linear(7 if seq else 6, 13)

    def forward(self, x):
        h = self.act(self.body(x))
        return self.head(h.mean(1) if self.seq else h.mean((2, 3)))


def _run_many_outputs(dev, seq):
    from laplace_amd.laplace import HipLaplace

    torch.manual_seed(9)
    model = ManyOutputsNet(seq).to(dev)
    X = (torch.randn(6, 4, 5) if seq else torch.randn(6, 2, 4, 4)).to(dev)
    y = torch.randint(13, (6,)).to(dev)
    m64 = ManyOutputsNet(seq).double()
    m64.load_state_dict({k: v.double().cpu() for k, v in model.state_dict().items()})
    Js64, f64 = co.jacobians(m64, X.double().cpu())
    loader = _Loader([(X, y)])
    loader.dataset = range(6)
    la = HipLaplace(model, "classification", "all", "kron", prior_precision=0.5)
    la.fit(loader)
    _, f_var = la._glm_predictive_distribution(X)
    _, kf = co.kfac_ggn(m64, X.double().cpu(), y.cpu(), 6, "classification")
    Qs, ls = co.kron_decompose(kf)
    assert rel(f_var, co.functional_variance_kron(Js64, Qs, ls, 0.5)) < 1e-4
    ld = HipLaplace(model, "classification", "all", "diag", prior_precision=0.5)
    ld.fit(loader)
    _, f_var_d = ld._glm_predictive_distribution(X)
    post_var = 1.0 / (co.ggn_diag(Js64, co.functional_hessian(f64, "classification")) + 0.5)
    assert rel(f_var_d, co.functional_variance_diag(Js64, post_var)) < 1e-4


@pytest.mark.parametrize("seq", [True, False])
def test_more_outputs_than_accumulators_on_emulation(seq):
    """Kron and diagonal GLM predictive with 13 outputs == J P^-1 J^T on the autograd Jacobian (fp64)"""
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        _run_many_outputs("cpu", seq)
    finally:
        _lib.set_kernels_for_testing(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("seq", [True, False])
def test_more_outputs_than_accumulators_gpu(seq):
    _run_many_outputs("cuda", seq)
