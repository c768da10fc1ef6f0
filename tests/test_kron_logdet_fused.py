"""Whole-posterior logdet (one `lk_kron_logdet_blocks_f32` call for all blocks) and the pending scalar of
`H * H_factor` on HipKronDecomposed (laplace/utils/matrix.py:342-404, baselaplace.py:1820): host logic on the kernel
emulation.  The kernel itself is checked against fp64 in tests/test_gpu_kernels.py::test_kron_logdet_blocks."""
import pytest
import torch

from laplace_amd import _lib
from tests.emulated_kernels import EmulatedKernels


@pytest.fixture(autouse=True)
def emulation():
    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    yield
    _lib.set_kernels_for_testing(prev)


def _cls():
    # looked up at call time: the drop-in tests rebuild laplace_amd.kron on top of the reference's classes
    from laplace_amd.kron import HipKronDecomposed

    return HipKronDecomposed


def _post(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(4, 7), (4,), (3, 5), (6, 2), (6,)]
    vals, vecs = [], []
    for sh in shapes:
        vals.append([torch.rand(n, generator=g) + 0.01 for n in sh])
        vecs.append([torch.linalg.qr(torch.randn(n, n, generator=g))[0] for n in sh])
    return _cls()(vecs, vals)


def _dense_logdet(vals, scale, deltas):
    tot = torch.zeros((), dtype=torch.float64)
    for ls, d in zip(vals, deltas.double()):
        lam = ls[0].double() if len(ls) == 1 else torch.outer(ls[0].double(), ls[1].double())
        tot = tot + torch.log(scale * lam + d).sum()
    return tot


def test_fused_logdet_matches_blockwise_formula_and_reference_algebra():
    H = _post()
    post = H * 0.37 + torch.tensor(2.5)
    assert isinstance(post, _cls())
    want = _dense_logdet(H.eigenvalues, 0.37, torch.full((5,), 2.5))
    assert abs(post.logdet().item() - want.item()) < 1e-4 * abs(want.item())
    # the materialised view is the reference's scalar split: scalar^(1/len) on every factor
    for ls, ms in zip(H.eigenvalues, post.eigenvalues):
        for l, m in zip(ls, ms):
            assert torch.allclose(m, 0.37 ** (1 / len(ls)) * l, rtol=1e-6)
    # chained products and sums, tensor scalars, per-layer deltas
    deltas = torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5])
    post2 = (H * torch.tensor(0.5)) * 3.0 + deltas + torch.tensor([1.0])
    want2 = _dense_logdet(H.eigenvalues, 1.5, deltas + 1.0)
    assert abs(post2.logdet().item() - want2.item()) < 1e-4 * abs(want2.item())
    # H itself is untouched by the products
    assert H._scale is None and post._scale == pytest.approx(0.37)


@pytest.mark.parametrize("per_layer", [False, True])
def test_fused_logdet_gradient_in_the_prior(per_layer):
    H = _post(1)
    log_pp = torch.zeros(5 if per_layer else 1, requires_grad=True)
    val = (H * 0.8 + log_pp.exp()).logdet()
    val.backward()
    lp64 = torch.zeros(5 if per_layer else 1, dtype=torch.float64, requires_grad=True)
    want = _dense_logdet(H.eigenvalues, 0.8, lp64.exp().expand(5))
    want.backward()
    assert abs(val.item() - want.item()) < 1e-4 * abs(want.item())
    assert torch.allclose(log_pp.grad.double(), lp64.grad, rtol=1e-4)


def test_blockwise_path_still_serves_damping_and_differentiable_eigenvalues():
    H = _post(2)
    Hd = _cls()(H.eigenvectors, H.eigenvalues, damping=True)
    post = Hd * 0.5 + torch.tensor(0.3)
    sd = 0.3 ** 0.5
    want = sum(
        (torch.log(ls[0] * 0.5 + 0.3).sum() if len(ls) == 1
         else torch.log(torch.outer(0.5 ** 0.5 * ls[0] + sd, 0.5 ** 0.5 * ls[1] + sd)).sum())
        for ls in H.eigenvalues)
    assert abs(post.logdet().item() - float(want)) < 1e-4 * abs(float(want))
    # eigenvalues that require grad (not the case after decompose, but allowed) go block by block with d/dl
    vals = [[l.clone().requires_grad_(True) for l in ls] for ls in H.eigenvalues]
    Hg = _cls()(H.eigenvectors, vals)
    out = (Hg + torch.tensor(1.0)).logdet()
    out.backward()
    assert all(l.grad is not None and torch.isfinite(l.grad).all() for ls in vals for l in ls)
