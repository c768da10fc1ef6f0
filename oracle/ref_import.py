"""Import the UNMODIFIED reference package from /root/reference (this container only).

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


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "laplace"))


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
    if not reference_available():
        raise ImportError(f"{REFERENCE_ROOT} not present on this machine")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import laplace  # noqa: WPS433

    return laplace
