"""Import the UNMODIFIED reference package from /root/reference (this container), or from the archive of it that
`stage_reference()` (run by the test session in this container, tests/conftest.py) packed into the git-ignored oracle/_ref/ (GPU box).

TEST INFRASTRUCTURE - not product code.  Only `oracle/make_golden.py` and the
`not gpu` oracle-pinning tests use this, and only where `/root/reference` exists.

The reference hard-imports five third-party distributions that are neither installed
nor installable here (no network): torchmetrics, opt_einsum, asdl (asdfghjkl 0.1a4),
backpack, curvlinops.  None of them is executed on the in-tree torch.func path
(`laplace/curvature/curvature.py` GGNInterface/EFInterface, `laplace/utils/matrix.py`
Kron/KronDecomposed, `laplace/baselaplace.py` Diag/Full/KronLaplace), so empty module
shells carrying the imported *names* are enough.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
#: ``oracle/_ref/`` is git-ignored (never in history) but travels with a `gpurun` snapshot: `stage_reference()` packs the
#: reference's `laplace/` package there so that the ONE test of the reference's own classes on the real kernels
#: (tests/test_gpu_dropin_reference.py) can execute on the GPU box, where /root/reference does not exist.  The archive
#: is made from the sources where they lie; nothing of it is ever committed.
STAGED_ARCHIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "laplace_reference.tgz")
_unpacked = None


def stage_reference() -> str | None:
    """(this container only) pack /root/reference/laplace into oracle/_ref/; returns the archive path or None"""
    import tarfile

    src = os.path.join(REFERENCE_ROOT, "laplace")
    if not os.path.isdir(src):
        return None
    os.makedirs(os.path.dirname(STAGED_ARCHIVE), exist_ok=True)
    newest = max(os.path.getmtime(os.path.join(d, f)) for d, _, fs in os.walk(src) for f in fs if f.endswith(".py"))
    if os.path.exists(STAGED_ARCHIVE) and os.path.getmtime(STAGED_ARCHIVE) >= newest:
        return STAGED_ARCHIVE
    with tarfile.open(STAGED_ARCHIVE, "w:gz") as tar:
        tar.add(src, arcname="laplace", filter=lambda ti: None if "__pycache__" in ti.name else ti)
    return STAGED_ARCHIVE


def reference_root() -> str | None:
    """directory that holds the reference's `laplace` package: /root/reference, or the staged archive unpacked into a
    temporary directory (GPU box)"""
    global _unpacked
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "laplace")):
        return REFERENCE_ROOT
    if _unpacked is not None:
        return _unpacked
    if os.path.exists(STAGED_ARCHIVE):
        import tarfile
        import tempfile

        dst = tempfile.mkdtemp(prefix="lk_reference_")
        with tarfile.open(STAGED_ARCHIVE, "r:gz") as tar:
            tar.extractall(dst, filter="data")
        _unpacked = dst
        return dst
    return None


def reference_available() -> bool:
    return reference_root() is not None


def _shell(name: str, **attrs) -> types.ModuleType:
    mod = sys.modules.get(name)
    if mod is None:
        mod = types.ModuleType(name)
        mod.__path__ = []  # behave like a package so submodule imports resolve
        sys.modules[name] = mod
    for key, val in attrs.items():
        setattr(mod, key, val)
    return mod


def _install_stubs() -> None:
    import torch

    class Metric(torch.nn.Module):
        """Just enough of torchmetrics.Metric for RunningNLLMetric (utils/metrics.py:6-21)."""

        def __init__(self, *a, **kw):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default)

    class _Unavailable:
        def __init__(self, *a, **kw):
            raise RuntimeError("third-party backend is stubbed (not installable here)")

    if "torchmetrics" not in sys.modules:
        _shell("torchmetrics", Metric=Metric, MeanSquaredError=_Unavailable)
    if "opt_einsum" not in sys.modules:
        _shell("opt_einsum", contract=lambda expr, *ops: torch.einsum(expr, *ops))
    if "asdl" not in sys.modules:
        _shell("asdl")
        _shell("asdl.fisher", FisherConfig=_Unavailable, get_fisher_maker=_Unavailable)
        _shell("asdl.grad_maker", LOSS_CROSS_ENTROPY="cross_entropy", LOSS_MSE="mse")
        _shell("asdl.gradient", batch_gradient=_Unavailable)
        _shell("asdl.hessian", HessianConfig=_Unavailable, HessianMaker=_Unavailable)
        _shell(
            "asdl.matrices",
            FISHER_EMP="fisher_emp", FISHER_EXACT="fisher_exact", FISHER_MC="fisher_mc",
            SHAPE_DIAG="diag", SHAPE_FULL="full", SHAPE_KRON="kron",
        )
    if "backpack" not in sys.modules:
        _shell("backpack", backpack=_Unavailable, extend=_Unavailable, memory_cleanup=_Unavailable)
        _shell("backpack.context", CTX=_Unavailable)
        _shell(
            "backpack.extensions",
            KFAC=_Unavailable, KFLR=_Unavailable, BatchGrad=_Unavailable,
            DiagGGNExact=_Unavailable, DiagGGNMC=_Unavailable, SumGradSquared=_Unavailable,
        )
    if "curvlinops" not in sys.modules:
        _shell(
            "curvlinops",
            EFLinearOperator=_Unavailable, FisherMCLinearOperator=_Unavailable,
            FisherType=_Unavailable, GGNLinearOperator=_Unavailable,
            HessianLinearOperator=_Unavailable, KFACLinearOperator=_Unavailable,
        )
        _shell("curvlinops._base", _LinearOperator=_Unavailable)


def import_reference():
    """Return the reference `laplace` package (raises if /root/reference is absent)."""
    root = reference_root()
    if root is None:
        raise ImportError(f"{REFERENCE_ROOT} not present on this machine (and no staged archive under oracle/_ref/)")
    _install_stubs()
    if root not in sys.path:
        sys.path.insert(0, root)
    import laplace  # noqa: WPS433

    return laplace
