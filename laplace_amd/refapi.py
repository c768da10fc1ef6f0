"""Resolve the base classes of the drop-in boundary.

If ``laplace-torch`` is importable, our backend subclasses the reference's own
``GGNInterface`` / ``EFInterface`` / ``Kron`` / ``KronDecomposed`` (so isinstance checks inside
``laplace/baselaplace.py`` hold and ``Laplace(..., backend=HipGGN)`` is a true drop-in).
Otherwise the local mirror (:mod:`laplace_amd.mirror`) provides the same contract.
"""
from __future__ import annotations

try:  # pragma: no cover - depends on the environment
    from laplace.curvature.curvature import CurvatureInterface, EFInterface, GGNInterface
    from laplace.utils.matrix import Kron, KronDecomposed

    HAVE_REFERENCE = True
except Exception:  # laplace-torch (or one of its hard dependencies) is not installed
    from laplace_amd.mirror import (  # noqa: F401
        CurvatureInterface,
        EFInterface,
        GGNInterface,
        Kron,
        KronDecomposed,
    )

    HAVE_REFERENCE = False

__all__ = ["CurvatureInterface", "GGNInterface", "EFInterface", "Kron", "KronDecomposed", "HAVE_REFERENCE"]
