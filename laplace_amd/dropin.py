"""Subclasses of the REFERENCE's own Laplace classes with the fused paths wired in (the "tertiary seam" of SURVEY.md §8b).

``Laplace(model, ..., backend=HipGGN)`` already runs the reference's classes on our kernels, but two of their methods
cannot be reached through the backend alone:

* ``KronLaplace.fit`` (laplace/baselaplace.py:1779-1809, via ParametricLaplace.fit :904-987) builds one ``Kron`` per
  minibatch and adds it — here ``fit`` accumulates in place through :class:`laplace_amd.backend.KronAccumulator`
  (upper triangles, pixel-pair A factors, one symmetrise / permute per fit);
* ``_glm_predictive_distribution`` (:1306-1342, lllaplace.py:212-237) materialises ``Js [B, C, P]`` — 447 MB per sample
  for ResNet-18 — before ``functional_variance``; here the per-layer factors of the Jacobian go straight into the
  quadratic-form kernels (:mod:`laplace_amd.predictive`).

Everything else (priors, marginal likelihood, link approximations, sampling, serialisation) is inherited untouched.
The classes carry their own ``_key`` so that the reference's factory ``laplace.Laplace`` (laplace/laplace.py:41-46, which
collects every subclass that has a ``_key``) keeps resolving to the reference's classes.  Needs laplace-torch; without
it (the GPU box) :func:`laplace_amd.laplace.HipLaplace` provides the lean stand-ins.
"""
from __future__ import annotations

from laplace_amd.backend import HipGGN
from laplace_amd.refapi import HAVE_REFERENCE

_CLASSES: dict = {}


def _ours(la) -> bool:
    return hasattr(la.backend, "kron_accumulator") and hasattr(la.backend, "_forward")


def reference_classes() -> dict:
    """``{(subset_of_weights, hessian_structure): class}`` — built on first use (imports laplace-torch)."""
    if _CLASSES:
        return _CLASSES
    if not HAVE_REFERENCE:
        raise ImportError("laplace-torch is not importable: use laplace_amd.laplace.HipLaplace (lean drivers) instead")
    from laplace.baselaplace import DiagLaplace, FullLaplace, KronLaplace
    from laplace.lllaplace import DiagLLLaplace, FullLLLaplace, KronLLLaplace

    from laplace_amd.laplace import fit_kron, glm_predictive

    class _FusedPredictive:
        def __init__(self, model, likelihood, *args, backend=HipGGN, **kwargs):
            super().__init__(model, likelihood, *args, backend=backend, **kwargs)

        def _glm_predictive_distribution(self, X, joint: bool = False, diagonal_output: bool = False):
            if joint or self.enable_backprop or not _ours(self):
                return super()._glm_predictive_distribution(X, joint=joint, diagonal_output=diagonal_output)
            return glm_predictive(self, X, diagonal_output=diagonal_output,
                                  fallback=super()._glm_predictive_distribution)

    class _FusedKronFit(_FusedPredictive):
        def fit(self, train_loader, override: bool = True, progress_bar: bool = False):
            # (online continuation re-weights the old factors: left to the reference's own loop)
            if not override or self.enable_backprop:
                return super().fit(train_loader, override=override, progress_bar=progress_bar)
            fit_kron(self, train_loader)

    def make(base, mixin, name):
        key = tuple(base._key) + ("laplace_amd",)
        return type(name, (mixin, base), {"_key": key, "__doc__": f"{base.__name__} with the fused laplace_amd paths",
                                          "__module__": __name__})

    _CLASSES[("all", "kron")] = make(KronLaplace, _FusedKronFit, "HipKronLaplace")
    _CLASSES[("all", "diag")] = make(DiagLaplace, _FusedPredictive, "HipDiagLaplace")
    _CLASSES[("all", "full")] = make(FullLaplace, _FusedPredictive, "HipFullLaplace")
    _CLASSES[("last_layer", "kron")] = make(KronLLLaplace, _FusedKronFit, "HipKronLLLaplace")
    _CLASSES[("last_layer", "diag")] = make(DiagLLLaplace, _FusedPredictive, "HipDiagLLLaplace")
    _CLASSES[("last_layer", "full")] = make(FullLLLaplace, _FusedPredictive, "HipFullLLLaplace")
    return _CLASSES


def Laplace(model, likelihood, subset_of_weights="last_layer", hessian_structure="kron", *args, **kwargs):
    """Call shape of ``laplace.Laplace`` (laplace/laplace.py:13-47).  With laplace-torch installed: the subclasses of its
    classes defined here (``backend`` defaults to :class:`HipGGN`); without: the lean drivers."""
    if HAVE_REFERENCE:
        classes = reference_classes()
        if (subset_of_weights, hessian_structure) not in classes:
            raise ValueError(f"no fused flavour for {(subset_of_weights, hessian_structure)}; use laplace.Laplace")
        return classes[(subset_of_weights, hessian_structure)](model, likelihood, *args, **kwargs)
    from laplace_amd.laplace import HipLaplace

    return HipLaplace(model, likelihood, subset_of_weights, hessian_structure, *args, **kwargs)
