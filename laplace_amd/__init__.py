"""laplace_amd — MI355X-native curvature backend for laplace-torch (aleximmer/Laplace).

Drop-in: ``Laplace(model, likelihood, ..., backend=laplace_amd.HipGGN)``.
"""
from laplace_amd.backend import HipEF, HipGGN
from laplace_amd.kron import HipKron, HipKronDecomposed
from laplace_amd.refapi import HAVE_REFERENCE

__all__ = ["HipGGN", "HipEF", "HipKron", "HipKronDecomposed", "HAVE_REFERENCE"]
__version__ = "0.1.0"
