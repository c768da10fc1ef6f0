"""Host-side mirror of the reference's plug-in interface for the curvature path.

Used ONLY when ``laplace-torch`` itself is not importable (e.g. on the GPU box, where the
reference checkout does not exist); when it is, :mod:`laplace_amd.refapi` hands out the
reference's own classes so that ``Laplace(model, ..., backend=HipGGN)`` is a true drop-in.

Same names, constructor arguments, attributes and error behaviour as
``laplace/curvature/curvature.py`` (CurvatureInterface :12-86, GGNInterface :294-339,
EFInterface :436-465) — written from the documented contract, not copied.  The generic
Jacobian / per-sample-gradient paths are host-side torch.func code exactly as in the
reference (they are the non-accelerated fallback for layers our kernels do not cover).
"""
from __future__ import annotations

from collections.abc import MutableMapping
from typing import Any

import torch
from torch import nn

HAVE_REFERENCE = False


class CurvatureInterface:
    """Plug-in contract of ``laplace.curvature.CurvatureInterface`` (curvature.py:12-292)."""

    def __init__(
        self,
        model: nn.Module,
        likelihood: str,
        last_layer: bool = False,
        subnetwork_indices: torch.LongTensor | None = None,
        dict_key_x: str = "input_ids",
        dict_key_y: str = "labels",
    ):
        if likelihood not in ("regression", "classification"):
            raise AssertionError(f"unsupported likelihood {likelihood!r}")
        self.likelihood = likelihood
        self.model = model
        self.last_layer = last_layer
        self.subnetwork_indices = subnetwork_indices
        self.dict_key_x = dict_key_x
        self.dict_key_y = dict_key_y
        if likelihood == "regression":
            self.lossfunc = nn.MSELoss(reduction="sum")
            self.factor = 0.5  # N(f, 1) from MSELoss(sum)
        else:
            self.lossfunc = nn.CrossEntropyLoss(reduction="sum")
            self.factor = 1.0
        self.params = [p for p in self._model.parameters() if p.requires_grad]
        self.params_dict = {k: v for k, v in self._model.named_parameters() if v.requires_grad}
        self.buffers_dict = dict(self.model.named_buffers())

    @property
    def _model(self) -> nn.Module:
        return self.model.last_layer if self.last_layer else self.model

    # -- generic host paths (torch.func), used for layers the HIP kernels do not cover -----------
    def jacobians(self, x, enable_backprop: bool = False):
        def fwd(params, buffers):
            out = torch.func.functional_call(self.model, (params, buffers), x)
            return out, out

        Jd, f = torch.func.jacrev(fwd, has_aux=True)(self.params_dict, self.buffers_dict)
        Js = torch.cat([j.flatten(start_dim=-p.dim()) for j, p in zip(Jd.values(), self.params_dict.values())], dim=-1)
        if self.subnetwork_indices is not None:
            Js = Js[:, :, self.subnetwork_indices]
        return (Js, f) if enable_backprop else (Js.detach(), f.detach())

    def last_layer_jacobians(self, x, enable_backprop: bool = False):
        f, phi = self.model.forward_with_features(x)
        B = phi.shape[0]
        C = f.numel() // B
        eye = torch.eye(C, device=phi.device, dtype=phi.dtype)
        Js = (eye[None, :, :, None] * phi[:, None, None, :]).reshape(B, C, -1)
        if self.model.last_layer.bias is not None:
            Js = torch.cat([Js, eye.expand(B, C, C)], dim=2)
        return (Js, f) if enable_backprop else (Js.detach(), f.detach())

    def gradients(self, x, y):
        def one(xi, yi, params, buffers):
            out = torch.func.functional_call(self.model, (params, buffers), xi.unsqueeze(0))
            loss = self.lossfunc(out, yi.unsqueeze(0))
            return loss, loss

        g, losses = torch.func.vmap(torch.func.grad(one, argnums=2, has_aux=True), in_dims=(0, 0, None, None))(
            x, y, self.params_dict, self.buffers_dict
        )
        Gs = torch.cat([v.flatten(start_dim=1) for v in g.values()], dim=1)
        if self.subnetwork_indices is not None:
            Gs = Gs[:, self.subnetwork_indices]
        return Gs, losses.sum(0)

    def full(self, x, y, **kwargs):
        raise NotImplementedError

    def kron(self, x, y, N, **kwargs):
        raise NotImplementedError

    def diag(self, x, y, **kwargs):
        raise NotImplementedError

    functorch_jacobians = jacobians


class GGNInterface(CurvatureInterface):
    """``laplace.curvature.GGNInterface`` constructor contract (curvature.py:294-339)."""

    def __init__(self, model, likelihood, last_layer=False, subnetwork_indices=None,
                 dict_key_x="input_ids", dict_key_y="labels", stochastic=False, num_samples=1):
        self.stochastic = stochastic
        self.num_samples = num_samples
        super().__init__(model, likelihood, last_layer, subnetwork_indices, dict_key_x, dict_key_y)

    # generic (materialised-Jacobian) GGN, the reference's fallback semantics (curvature.py:366-433)
    def _get_functional_hessian(self, f):
        if self.likelihood == "regression":
            return None
        p = torch.softmax(f, dim=-1)
        return torch.diag_embed(p) - p.unsqueeze(2) * p.unsqueeze(1)

    def _generic_js(self, x):
        return self.last_layer_jacobians(x) if self.last_layer else self.jacobians(x)

    def full(self, x, y, **kwargs):
        Js, f = self._generic_js(x)
        Lam = self._get_functional_hessian(f)
        LJ = Js if Lam is None else Lam @ Js
        H = torch.einsum("bcp,bcq->pq", Js, LJ)
        return (self.factor * self.lossfunc(f, y)).detach(), H.detach()

    def diag(self, x, y, **kwargs):
        Js, f = self._generic_js(x)
        Lam = self._get_functional_hessian(f)
        LJ = Js if Lam is None else Lam @ Js
        return (self.factor * self.lossfunc(f, y)).detach(), (Js * LJ).sum((0, 1)).detach()


class EFInterface(CurvatureInterface):
    """``laplace.curvature.EFInterface`` (curvature.py:436-505), generic per-sample-gradient form."""

    def full(self, x, y, **kwargs):
        Gs, loss = self.gradients(x, y)
        Gs = Gs.detach()
        return self.factor * loss.detach(), self.factor * (Gs.T @ Gs)

    def diag(self, x, y, **kwargs):
        Gs, loss = self.gradients(x, y)
        return self.factor * loss.detach(), self.factor * (Gs.detach() ** 2).sum(0)


class Kron:
    """Marker base; :class:`laplace_amd.kron.HipKron` implements the whole ``Kron`` contract
    (laplace/utils/matrix.py:16-279) itself."""


class KronDecomposed:
    """Marker base; see :class:`laplace_amd.kron.HipKronDecomposed` (matrix.py:282-560)."""


class FeatureExtractor(nn.Module):
    """Minimal stand-in for ``laplace.utils.feature_extractor.FeatureExtractor`` (:13-216):
    wraps a model, exposes ``last_layer`` and ``forward_with_features`` through a forward hook on
    the named (or last ``nn.Linear``) module."""

    def __init__(self, model: nn.Module, last_layer_name: str | None = None):
        super().__init__()
        self.model = model
        self._features = None
        if last_layer_name is None:
            cands = [(n, m) for n, m in model.named_modules() if isinstance(m, nn.Linear)]
            if not cands:
                raise ValueError("model has no nn.Linear to use as last layer")
            last_layer_name = cands[-1][0]
        self._last_layer_name = last_layer_name
        self.last_layer = dict(model.named_modules())[last_layer_name]
        self.last_layer.register_forward_hook(self._hook)

    def _hook(self, module, inp, out):
        self._features = inp[0].detach()

    def forward(self, x):
        return self.model(x)

    def forward_with_features(self, x: torch.Tensor | MutableMapping[str, Any]):
        out = self.model(x)
        return out, self._features
