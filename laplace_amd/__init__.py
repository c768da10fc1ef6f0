"""laplace_amd — MI355X-native curvature backend for laplace-torch (aleximmer/Laplace).

Drop-in: ``Laplace(model, likelihood, ..., backend=laplace_amd.HipGGN)``.
"""
from laplace_amd.backend import HipEF, HipGGN
from laplace_amd.kron import HipKron, HipKronDecomposed
from laplace_amd.refapi import HAVE_REFERENCE


def fit_kron(la, train_loader, process_group=None, distributed=None):
    """Fused ``fit`` for the reference's ``KronLaplace`` objects — see :func:`laplace_amd.laplace.fit_kron`."""
    from laplace_amd.laplace import fit_kron as _fit

    return _fit(la, train_loader, process_group=process_group, distributed=distributed)


def glm_predictive(la, X, diagonal_output=False):
    """Fused GLM predictive for the reference's Laplace objects — see :func:`laplace_amd.laplace.glm_predictive`."""
    from laplace_amd.laplace import glm_predictive as _glm

    return _glm(la, X, diagonal_output=diagonal_output)


def Laplace(model, likelihood, subset_of_weights="last_layer", hessian_structure="kron", *args, **kwargs):
    """``laplace.Laplace``'s call shape, returning the fused subclasses of the reference's classes — see
    :mod:`laplace_amd.dropin`."""
    from laplace_amd.dropin import Laplace as _L

    return _L(model, likelihood, subset_of_weights, hessian_structure, *args, **kwargs)


__all__ = ["HipGGN", "HipEF", "HipKron", "HipKronDecomposed", "HAVE_REFERENCE", "fit_kron", "glm_predictive", "Laplace"]
__version__ = "0.1.0"
