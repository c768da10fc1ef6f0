"""The UNMODIFIED reference's classes and the real HIP kernels in one process: ``laplace.Laplace(model, ...,
backend=HipGGN)`` on the MI355X, against the golden outputs of the reference's own backends.  Needs a GPU AND the
reference's package: on the GPU box (no /root/reference) it is unpacked from the archive that `build()` stages under
the git-ignored oracle/_ref/ (oracle/ref_import.py: stage_reference — made from the sources where they lie, never
committed).  `tests/test_dropin_reference.py` covers the same seam on the kernel emulation, `tests/test_gpu_backend.py`
the same kernels behind the mirrored classes."""
import importlib

import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

from oracle.ref_import import reference_available
from tests.conftest import golden_kfacs, golden_model, load_golden

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not reference_available(), reason="neither /root/reference nor oracle/_ref/laplace_reference.tgz")]
DEV = "cuda"


@pytest.fixture(scope="module")
def ref():
    from oracle.ref_import import import_reference

    import_reference()
    import laplace_amd
    import laplace_amd.backend
    import laplace_amd.kron
    import laplace_amd.refapi as refapi

    if not refapi.HAVE_REFERENCE:  # laplace_amd was imported before the reference: re-derive the boundary classes
        importlib.reload(refapi)
        importlib.reload(laplace_amd.kron)
        importlib.reload(laplace_amd.backend)
        importlib.reload(laplace_amd)
    yield


def rel(got, want):
    got = torch.as_tensor(got).detach().double().cpu()
    want = torch.as_tensor(want).detach().double().cpu()
    from tests.parity_log import record_error

    return record_error((got - want).abs().max().item() / (want.abs().max().item() + 1e-30))


@pytest.mark.parametrize("name", ["mlp", "conv", "bnres"])
@pytest.mark.parametrize("lik", ["classification", "regression"])
@pytest.mark.parametrize("sow,hs", [("all", "kron"), ("all", "diag"), ("last_layer", "kron"), ("last_layer", "full")])
def test_reference_laplace_on_the_hip_kernels(ref, name, lik, sow, hs):
    from laplace import Laplace
    from laplace.utils.matrix import Kron, KronDecomposed

    import laplace_amd
    from laplace_amd import HipGGN, HipKron, HipKronDecomposed
    from laplace_amd import _lib
    from oracle.make_golden import PRIOR_PREC, SIGMA_NOISE

    assert isinstance(_lib.get_kernels(), _lib.HipKernels), "the real library, not the emulation"
    assert issubclass(HipKron, Kron) and issubclass(HipKronDecomposed, KronDecomposed)
    g = load_golden(name, lik)
    model, X, y = golden_model(name, g, dtype=torch.float32, device=DEV)
    sig = SIGMA_NOISE if lik == "regression" else 1.0
    la = Laplace(model, lik, subset_of_weights=sow, hessian_structure=hs, prior_precision=PRIOR_PREC, sigma_noise=sig,
                 backend=HipGGN)
    la.fit(DataLoader(TensorDataset(X, y), batch_size=5))
    tag = f"la.{sow}.{hs}"
    if hs == "kron":
        assert isinstance(la.H_facs, HipKron) and isinstance(la.posterior_precision, HipKronDecomposed)
        for F_, G_ in zip(la.H_facs.kfacs, golden_kfacs(g, f"{tag}.H")):
            for a, w in zip(F_, G_):
                assert rel(a, w) < 1e-4
    else:
        assert rel(la.H, g[f"{tag}.H"]) < 1e-4
    assert rel(la.loss, g[f"{tag}.loss"]) < 1e-4
    f_mu, f_var = laplace_amd.glm_predictive(la, X)
    assert rel(f_mu, g[f"{tag}.f_mu"]) < 1e-4
    assert rel(f_var, g[f"{tag}.f_var"]) < 1e-4
    assert rel(la.log_marginal_likelihood(), g[f"{tag}.marglik"]) < 1e-4
