"""Non-convergence semantics of the eigendecomposition seam (utils/utils.py:208-222 of the reference: jitter, retry, fail
loudly): the solver's status words are read at the first use of a `HipKronDecomposed`; a failed factor is re-solved as
`M + I`; a second failure raises RuntimeError (what the reference's prior gridsearch catches, baselaplace.py:545-551)."""
import pytest
import torch

from laplace_amd import _lib
from laplace_amd.kron import HipKron
from tests.emulated_kernels import EmulatedKernels


class FlakySolver(EmulatedKernels):
    """reports 'ran out of sweeps' (and returns garbage) for the first `fail` solves of matrices of size `n_bad`"""

    def __init__(self, n_bad, fail):
        self.n_bad, self.fail, self.calls = n_bad, fail, 0

    def syevj_batched(self, mats, clamp=True, max_sweeps=0, streams=None):
        out = super().syevj_batched(mats, clamp=clamp, max_sweeps=max_sweeps, streams=streams)
        res = []
        for M, (l, Q, info) in zip(mats, out):
            if M.shape[0] == self.n_bad and self.calls < self.fail:
                self.calls += 1
                info = torch.ones_like(info)
                l, Q = torch.full_like(l, float("nan")), torch.zeros_like(Q)
            res.append((l, Q, info))
        return res


def _kron():
    torch.manual_seed(0)
    A = torch.randn(6, 6)
    B = torch.randn(4, 4)
    return HipKron([[A @ A.T, B @ B.T], [A @ A.T]])


@pytest.mark.parametrize("fail,ok", [(0, True), (1, True), (2, False)])
def test_first_use_checks_retries_with_jitter_then_raises(fail, ok):
    prev = _lib.set_kernels_for_testing(FlakySolver(4, fail))
    try:
        H = _kron()
        dec = H.decompose()  # never raises: the status words are still on the device
        post = dec * 2.0 + torch.tensor(0.5)
        if not ok:
            with pytest.raises(RuntimeError, match="did not converge"):
                post.logdet()
            return
        want = sum(torch.log(2.0 * torch.outer(*[torch.linalg.eigvalsh(M) for M in F]) + 0.5).sum() if len(F) == 2
                   else torch.log(2.0 * torch.linalg.eigvalsh(F[0]) + 0.5).sum() for F in H.kfacs)
        assert abs(float(post.logdet()) - float(want)) < 1e-3 * abs(float(want))
        l2 = dec.eigenvalues[0][1]
        assert torch.allclose(torch.sort(l2)[0], torch.linalg.eigvalsh(H.kfacs[0][1]).clamp(min=0), atol=1e-4)
    finally:
        _lib.set_kernels_for_testing(prev)
