"""Eigenbasis algebra of KronDecomposed on the MI355X (-m gpu): the batched fp32-MFMA GEMM with the element-wise epilogue
(lk_gemm_f32), `_bmm` at exponents -1 / -1/2 / 1 built on it (matrix.py:406-461), the posterior samples
(baselaplace.py:1845-1858) and the joint GLM predictive (:1837-1843) of the lean drivers — against the fp64 oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    from tests.parity_log import record_error

    return record_error((a - b).abs().max().item() / (b.abs().max().item() + 1e-300))


@pytest.mark.parametrize("M,N,K,batch,ta,tb,use_e", [
    (64, 64, 16, 1, False, False, False), (70, 33, 19, 3, False, False, True), (33, 70, 45, 2, True, False, True),
    (129, 5, 7, 4, False, True, False), (5, 200, 130, 2, True, True, True), (300, 300, 300, 1, False, True, True),
])
def test_batched_gemm_with_epilogue_weight(M, N, K, batch, ta, tb, use_e):
    from laplace_amd._lib import get_kernels

    Kn = get_kernels()
    torch.manual_seed(M + N + K)
    A = torch.randn(batch, K, M, device=DEV) if ta else torch.randn(batch, M, K, device=DEV)
    B = torch.randn(1, N, K, device=DEV) if tb else torch.randn(1, K, N, device=DEV)  # shared operand (stride 0)
    E = torch.randn(M, N, device=DEV) if use_e else None
    C = torch.full((batch, M, N), 7.0, device=DEV)
    Kn.gemm(A, B, C, batch, M, N, K, A.shape[-1], B.shape[-1], N, sa=M * K, sb=0, sc=M * N, ta=ta, tb=tb, E=E,
            lde=N if use_e else 0, alpha=0.5)
    Ad = (A.transpose(1, 2) if ta else A).double()
    Bd = (B.transpose(1, 2) if tb else B).double()
    want = 0.5 * Ad @ Bd
    if use_e:
        want = want * E.double()
    assert rel(C, want) < 1e-6
    C2 = torch.ones(batch, M, N, device=DEV)
    Kn.gemm(A, B, C2, batch, M, N, K, A.shape[-1], B.shape[-1], N, sa=M * K, sb=0, sc=M * N, ta=ta, tb=tb, accumulate=True)
    assert rel(C2, 1.0 + Ad @ Bd) < 1e-6


def _random_decomposed(dims, damping=False):
    from laplace_amd.kron import HipKron

    torch.manual_seed(3)
    kfacs = []
    for d in dims:
        F_ = []
        for n in d:
            X = torch.randn(n + 5, n)
            F_.append((X.T @ X / (n + 5)).to(DEV))
        kfacs.append(F_)
    H = HipKron(kfacs)
    dec = H.decompose(damping=damping)
    dec.check_converged()
    return H, dec


@pytest.mark.parametrize("damping", [False, True])
def test_krondecomposed_bmm_on_the_gemm_kernel(damping):
    from oracle import curvature_oracle as co

    dims = [(10, 27), (7,), (33, 50), (1, 1)]
    H, dec = _random_decomposed(dims, damping)
    post = dec * 3.0 + torch.tensor([0.7, 0.2, 1.3, 0.5], device=DEV)
    P = sum(d[0] * (d[1] if len(d) == 2 else 1) for d in dims)
    W = torch.randn(6, 3, P, device=DEV)
    kf64 = [[M.double().cpu() for M in F_] for F_ in H.kfacs]
    Qs, ls = co.kron_decompose(kf64)
    ls = co.krondecomposed_scale(ls, 3.0)
    deltas = [0.7, 0.2, 1.3, 0.5]
    for e in (-1.0, -0.5, 1.0):
        got = post._bmm(W, exponent=e)
        want = co.krondecomposed_bmm(Qs, ls, deltas, W.double().cpu(), exponent=e, damping=damping)
        assert rel(got, want) < 1e-4, e
    assert rel(post.inv_square_form(W), co.krondecomposed_inv_square_form(Qs, ls, deltas, W.double().cpu(), damping)) < 1e-4
    assert rel(post.bmm(W[0, 0], exponent=-0.5), co.krondecomposed_bmm(Qs, ls, deltas, W[:1, :1].double().cpu(), -0.5, damping)[0, 0]) < 1e-4


@pytest.mark.parametrize("lik", ["classification", "regression"])
def test_posterior_samples_and_joint_predictive_on_the_device(lik):
    """`sample` = mean + P^{-1/2} z block-wise through the eigendecomposition; the joint predictive = inv_square_form on
    the stacked Jacobian rows — lean drivers on the device against the oracle fed the SAME normal draws."""
    from laplace_amd.laplace import HipLaplace
    from oracle import curvature_oracle as co
    from tests.conftest import golden_model, load_golden

    g = load_golden("mlp", lik)
    model, X, y = golden_model("mlp", g, dtype=torch.float32, device=DEV)
    m64, X64, y64 = golden_model("mlp", g, dtype=torch.float64)

    class L(list):
        dataset = list(range(len(X)))

    la = HipLaplace(model, lik, "all", "kron", prior_precision=0.7, sigma_noise=0.8 if lik == "regression" else 1.0)
    la.fit(L([(X[:5], y[:5]), (X[5:], y[5:])]))
    gen = torch.Generator(device=DEV).manual_seed(5)
    state = gen.get_state()
    s = la.sample(7, generator=gen)
    gen.set_state(state)
    z = la._randn(7, la.n_params, generator=gen)
    kf_ref = None
    for i in (0, 5):
        _, kb = co.kfac_ggn(m64, X64[i:i + 5], y64[i:i + 5], len(X), lik)
        kf_ref = kb if kf_ref is None else co.kron_add(kf_ref, kb)
    Qs, ls = co.kron_decompose(kf_ref)
    hf = float(la._H_factor)
    ls = co.krondecomposed_scale(ls, hf)
    want = la.mean.double().cpu().reshape(1, -1) + co.krondecomposed_bmm(Qs, ls, 0.7, z.double().cpu().unsqueeze(1), -0.5).squeeze(1)
    assert rel(s, want) < 1e-4
    # joint GLM predictive
    f_mu, f_cov = la._glm_joint_distribution(X[:4]) if hasattr(la, "_glm_joint_distribution") else (None, None)
    Js, f = co.jacobians(m64, X64[:4])
    n, c, p = Js.shape
    want_cov = co.krondecomposed_inv_square_form(Qs, ls, 0.7, Js.reshape(1, n * c, p)).squeeze(0)
    assert rel(f_cov, want_cov) < 1e-4
    assert rel(f_mu, f.flatten()) < 1e-4


@pytest.mark.parametrize("n", [1, 2, 10, 257, 576])
def test_packed_upper_triangle_round_trip(n):
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    A = torch.randn(n, n, device=DEV)
    packed = torch.empty(n * (n + 1) // 2, device=DEV)
    K.pack_upper(A, packed)
    i, j = torch.triu_indices(n, n)
    assert torch.equal(packed.cpu(), A.cpu()[i, j])
    B = torch.full((n, n), -1.0, device=DEV)
    K.unpack_upper(packed, B)
    assert torch.equal(torch.triu(B), torch.triu(A))
    assert (torch.tril(B, -1) == torch.tril(torch.full((n, n), -1.0, device=DEV), -1)).all()
