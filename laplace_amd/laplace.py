"""Lean callers of the hot path: fit loop (+ data-parallel sharding), posterior, GLM predictive,
marginal likelihood — the "next" rows either side of the curvature path.

With laplace-torch installed none of this is needed: ``Laplace(model, ..., backend=HipGGN)`` drives
the same backend through the reference's own classes.  These drivers exist (a) so the path runs
end to end where the reference is not installed (the GPU box), and (b) to add what the reference
does not have: *fused* predictive variances that never materialise ``[B, C, P]`` Jacobians and a
multi-GPU ``fit`` (one process per GPU, minibatches sharded by rank, ONE RCCL all-reduce of the
accumulated curvature at the end of the epoch).

Names, arguments and semantics mirror ``laplace/baselaplace.py`` (ParametricLaplace.fit :904-987,
KronLaplace :1704-1879, DiagLaplace :2048-2136, FullLaplace :1572-1701, log marginal likelihood
:1074-1109, GLM predictive :598-695,1306-1342) and ``laplace/lllaplace.py`` (last-layer flavours).
"""
from __future__ import annotations

from math import pi, sqrt

import torch
import torch.distributed as dist
from torch import nn
from torch.nn.utils import parameters_to_vector, vector_to_parameters

from laplace_amd import predictive as _pred
from laplace_amd.backend import HipGGN
from laplace_amd.kron import HipKron


class _StridedBatches:
    """Batch sampler of one rank: batches ``rank, rank + world, ...`` of the wrapped batch sampler.  Only index lists
    pass through here — the other ranks' samples are never loaded or collated."""

    def __init__(self, batch_sampler, rank: int, world_size: int):
        self.batch_sampler, self.rank, self.world_size = batch_sampler, rank, world_size

    def __iter__(self):
        for i, idx in enumerate(self.batch_sampler):
            if i % self.world_size == self.rank:
                yield idx

    def __len__(self):
        return (len(self.batch_sampler) - self.rank + self.world_size - 1) // self.world_size


class ShardedLoader:
    """Minibatches ``rank, rank + world, ...`` of ``loader`` while still reporting the GLOBAL dataset
    (``len(loader.dataset)`` is the ``N`` every rank passes to ``kron``: the A factor is scaled by ``M / N``,
    laplace/curvature/curvlinops.py:46-53, so partial sums add up exactly).

    The shard is taken where it is cheap: a ``torch.utils.data.DataLoader`` is rebuilt around a strided batch sampler
    (a rank never loads, collates or transfers another rank's samples; with ``shuffle=True`` give the loader a
    ``generator`` seeded identically on every rank so that all ranks cut the same permutation), an indexable
    collection of ready batches is sliced, and only a plain iterable is walked with the foreign batches skipped."""

    def __init__(self, loader, rank: int, world_size: int):
        self.loader, self.rank, self.world_size = loader, rank, world_size
        self.dataset = loader.dataset

    def _mine(self):
        ld = self.loader
        from torch.utils.data import DataLoader

        if isinstance(ld, DataLoader) and ld.batch_sampler is not None:
            from torch.utils.data import RandomSampler

            smp = getattr(ld.batch_sampler, "sampler", None) or ld.sampler
            if isinstance(smp, RandomSampler) and smp.generator is None and self.world_size > 1:
                # every rank would draw ITS OWN permutation from its default RNG: the strided shards would overlap and
                # miss samples, and the all-reduced curvature would be silently wrong
                raise ValueError("ShardedLoader: the wrapped DataLoader shuffles without a `generator`; the ranks "
                                 "cannot cut the same permutation. Use shuffle=False (the order of the data does not "
                                 "matter to a curvature fit) or pass a torch.Generator seeded identically on all ranks.")
            return DataLoader(ld.dataset, batch_sampler=_StridedBatches(ld.batch_sampler, self.rank, self.world_size),
                              num_workers=ld.num_workers, collate_fn=ld.collate_fn, pin_memory=ld.pin_memory,
                              timeout=ld.timeout, worker_init_fn=ld.worker_init_fn,
                              multiprocessing_context=ld.multiprocessing_context, generator=ld.generator,
                              prefetch_factor=ld.prefetch_factor if ld.num_workers > 0 else None,
                              persistent_workers=ld.persistent_workers)
        if hasattr(ld, "__getitem__") and hasattr(ld, "__len__"):
            return (ld[i] for i in range(self.rank, len(ld), self.world_size))
        return (b for i, b in enumerate(ld) if i % self.world_size == self.rank)

    def __iter__(self):
        return iter(self._mine())

    def __len__(self):
        n = len(self.loader)
        return (n - self.rank + self.world_size - 1) // self.world_size


def loader_is_sharded(loader) -> bool:
    """Does every rank see only ITS part of the data through ``loader``?  (:class:`ShardedLoader`, or a DataLoader on
    a ``DistributedSampler``.)"""
    if isinstance(loader, ShardedLoader):
        return True
    from torch.utils.data.distributed import DistributedSampler

    for attr in ("sampler", "batch_sampler"):
        smp = getattr(loader, attr, None)
        if isinstance(smp, DistributedSampler) or isinstance(getattr(smp, "sampler", None), DistributedSampler):
            return True
    return False


def resolve_distributed(train_loader, distributed, group=None) -> bool:
    """The all-reduce of a fit is OPT-IN: explicit ``distributed=True``, or — with ``distributed=None`` — an
    initialised process group of more than one rank AND a loader that is visibly sharded.  A DDP-style script in which
    every rank runs ``la.fit(full_loader)`` (how the single-process reference is used) must not have its curvature
    silently multiplied by the world size: that case warns and stays local."""
    if distributed is not None:
        return bool(distributed)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return False
    if loader_is_sharded(train_loader):
        return True
    import warnings

    warnings.warn("torch.distributed is initialised but the training loader is not sharded (no ShardedLoader / "
                  "DistributedSampler): fitting locally WITHOUT an all-reduce. Wrap the loader in "
                  "laplace_amd.ShardedLoader(loader, rank, world_size) or pass distributed=True.", stacklevel=3)
    return False


def allreduce_curvature(tensors: list[torch.Tensor], group=None, mirror: bool = True) -> dict | None:
    """Sum the accumulated curvature over ranks with ONE collective (RCCL over xGMI when the backend is "nccl"; gloo in
    the CPU tests).  Square fp32 matrices are symmetric factors: only their packed upper triangles travel (half the
    bytes; ``lk_pack_upper_f32`` writes them straight into the exchange buffer, the all-reduce runs on that buffer in
    place, ``lk_unpack_upper_f32`` writes the sums back) — no concatenated copy of the squares.  ``mirror``: restore the
    lower triangles afterwards (callers that symmetrise later anyway pass ``False``).  Returns what was exchanged
    (``{"bytes": .., "tensors": .., "packed": ..}``; ``None`` without a process group) so that a launch script can assert
    the message size it expects (ResNet-18: 188 MB of packed upper triangles)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    tensors = list(tensors)
    sym = [t.dim() == 2 and t.shape[0] == t.shape[1] and t.shape[0] > 1 and t.dtype == torch.float32 and t.is_contiguous()
           for t in tensors]
    sizes = [t.shape[0] * (t.shape[0] + 1) // 2 if s_ else t.numel() for t, s_ in zip(tensors, sym)]
    flat = torch.empty(sum(sizes), dtype=tensors[0].dtype, device=tensors[0].device)
    off = 0
    for t, s_, n in zip(tensors, sym, sizes):
        if s_:
            K.pack_upper(t, flat[off:off + n])
        else:
            flat[off:off + n].copy_(t.reshape(-1))
        off += n
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t, s_, n in zip(tensors, sym, sizes):
        if s_:
            K.unpack_upper(flat[off:off + n], t)
            if mirror:
                K.symmetrize(t)
        else:
            t.copy_(flat[off:off + n].view_as(t))
        off += n
    return {"bytes": int(flat.numel() * flat.element_size()), "tensors": len(tensors), "packed": int(sum(sym))}


def expected_exchange_bytes(model_or_shapes) -> int:
    """bytes of ONE data-parallel curvature exchange of a KFAC fit: packed upper triangles of every factor (G, A per
    Linear / Conv2d weight) + the loss word, fp32 — what :func:`allreduce_curvature` must report (asserted by
    ``bench.py --gpus N`` and tools/launch_multi_gpu.sh).  ``model_or_shapes``: a module or a list of factor sizes."""
    if isinstance(model_or_shapes, nn.Module):
        sizes = []
        for m in model_or_shapes.modules():
            if isinstance(m, nn.Linear) and m.weight.requires_grad:
                sizes += [m.out_features, m.in_features]
            elif isinstance(m, nn.Conv2d) and m.weight.requires_grad:
                sizes += [m.out_channels, m.in_channels * m.kernel_size[0] * m.kernel_size[1]]
    else:
        sizes = list(model_or_shapes)
    return 4 * (sum(n * (n + 1) // 2 if n > 1 else 1 for n in sizes) + 1)  # + the loss word


def _share_n_outputs(la, group=None) -> None:
    """A rank whose shard was empty never ran the model: it learns the output width (``n_outputs`` /
    ``model.output_size``, which the reference's ``fit`` sets from its first minibatch, baselaplace.py:955-963) from
    the ranks that did.  One tiny MAX all-reduce, issued by every rank right after the curvature exchange."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    dev = la._device if dist.get_backend(group) == "nccl" else "cpu"
    n = torch.tensor([int(la.n_outputs or 0)], dtype=torch.int64, device=dev)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    if la.n_outputs is None and int(n.item()) > 0:
        la.n_outputs = int(n.item())
        setattr(la.model, "output_size", la.n_outputs)


class _HipLaplace:
    """Shared part of the Kron / diag / full flavours (subset_of_weights 'all' or 'last_layer')."""

    def __init__(self, model: nn.Module, likelihood: str, subset_of_weights: str = "all", sigma_noise=1.0,
                 prior_precision=1.0, temperature: float = 1.0, backend=HipGGN, backend_kwargs=None,
                 last_layer_name: str | None = None):
        if likelihood not in ("classification", "regression"):
            raise ValueError(f"Invalid likelihood type {likelihood}")
        if subset_of_weights not in ("all", "last_layer"):
            raise ValueError("subset_of_weights must be 'all' or 'last_layer'")
        self.likelihood = likelihood
        self.subset_of_weights = subset_of_weights
        self._last_layer_name = last_layer_name  # as given (lllaplace.py:335-339 stores the constructor argument)
        self.data = None  # first training datum of a last-layer fit (lllaplace.py: lazy head discovery on load)
        self.temperature = temperature
        self._backend_cls = backend
        self._backend_kwargs = dict(backend_kwargs or {})
        self._backend = None
        if subset_of_weights == "last_layer":
            from laplace_amd.mirror import FeatureExtractor  # (an already wrapped model is used as is)

            self.model = model if hasattr(model, "forward_with_features") else FeatureExtractor(model, last_layer_name)
            if getattr(self.model, "last_layer", None) is None:
                raise ValueError("give last_layer_name (lazy last-layer discovery is not supported here)")
            self.params = [p for p in self.model.last_layer.parameters() if p.requires_grad]
            self._backend_kwargs["last_layer"] = True
        else:
            self.model = model
            self.params = [p for p in model.parameters() if p.requires_grad]
        self.n_params = sum(p.numel() for p in self.params)
        self.n_layers = len(self.params)
        p0 = next(model.parameters())
        self._device, self._dtype = p0.device, p0.dtype
        self.sigma_noise = sigma_noise
        self.prior_precision = prior_precision
        self.prior_mean = 0.0
        self.loss = 0.0
        self.n_data = 0
        self.n_outputs = None
        self.H = None

    # ---- hyper-parameters ---------------------------------------------------------------------------
    @property
    def backend(self):
        if self._backend is None:
            self._backend = self._backend_cls(self.model, self.likelihood, **self._backend_kwargs)
        return self._backend

    @property
    def sigma_noise(self):
        return self._sigma_noise

    @sigma_noise.setter
    def sigma_noise(self, v):
        v = torch.as_tensor(v, device=self._device, dtype=self._dtype)
        if v.ndim == 1:
            if len(v) > 1:
                raise ValueError("Only homoscedastic output noise supported.")
            v = v[0]
        if self.likelihood != "regression" and float(v) != 1.0:
            raise ValueError("Sigma noise != 1 only available for regression.")
        self._sigma_noise = v
        self._posterior_cache = None

    @property
    def prior_precision(self):
        return self._prior_precision

    @prior_precision.setter
    def prior_precision(self, v):
        v = torch.as_tensor(v, device=self._device, dtype=self._dtype)
        if v.ndim == 0:
            v = v.reshape(1)
        if v.ndim != 1 or len(v) not in (1, self.n_layers, self.n_params):
            raise ValueError("Length of prior precision does not align with architecture.")
        self._prior_precision = v
        self._posterior_cache = None

    @property
    def prior_precision_diag(self) -> torch.Tensor:
        pp = self.prior_precision
        if len(pp) == 1:
            return pp * torch.ones(self.n_params, device=self._device, dtype=self._dtype)
        if len(pp) == self.n_params:
            return pp
        return torch.cat([d * torch.ones(p.numel(), device=self._device, dtype=self._dtype)
                          for d, p in zip(pp, self.params)])

    @property
    def _H_factor(self):
        return 1 / self.sigma_noise.square() / self.temperature

    # ---- fit (baselaplace.py:904-987) + sharding --------------------------------------------------------
    def _init_H(self):
        raise NotImplementedError

    def _curv_closure(self, X, y, N):
        raise NotImplementedError

    def _curvature_tensors(self) -> list[torch.Tensor]:
        raise NotImplementedError

    def fit(self, train_loader, override: bool = True, process_group=None, distributed: bool | None = None):
        """Accumulate the curvature over ``train_loader``.

        Data parallel: when ``torch.distributed`` is initialised (or ``distributed=True``) every rank
        consumes its shard of minibatches (wrap the loader in :class:`ShardedLoader`; ``N`` stays the
        global dataset size) and the packed curvature + loss are summed with one all-reduce.
        """
        # override=False (baselaplace.py:904-987): online continuation, the new curvature is ADDED to what is there.
        # It is accumulated separately first, so that a data-parallel continuation all-reduces the new part only.
        old = None
        if not override and getattr(self, "n_data", 0) and getattr(self, "H", None) is not None:
            old = (self.H, self.loss, self.n_data)
        self._init_H()
        self.loss = torch.zeros((), device=self._device, dtype=self._dtype)
        self.n_data = 0
        self.model.eval()
        self.mean = parameters_to_vector(self.params).detach()
        N = len(train_loader.dataset)
        for data in train_loader:
            if isinstance(data, dict) or hasattr(data, "keys"):
                X, y = data, data[self.backend.dict_key_y].to(self._device)
            else:
                X, y = data
                X, y = X.to(self._device), y.to(self._device)
            if self.n_outputs is None:
                with torch.no_grad():
                    self.n_outputs = self.model(X[:1] if torch.is_tensor(X) else X).shape[-1]
                setattr(self.model, "output_size", self.n_outputs)
                if self.subset_of_weights == "last_layer" and torch.is_tensor(X):
                    self.data = (X[:1].detach().cpu(), y[:1].detach().cpu())
            loss_b, H_b = self._curv_closure(X, y, N)
            self.loss = self.loss + loss_b
            self.H += H_b
        distributed = resolve_distributed(train_loader, distributed, process_group)
        if distributed:
            loss_t = self.loss.reshape(1).clone()
            allreduce_curvature(self._curvature_tensors() + [loss_t], group=process_group)
            self.loss = loss_t[0]
        self.n_data += N
        if old is not None:
            self.H += old[0]
            self.loss = self.loss + old[1]
            self.n_data += old[2]
        self._posterior_cache = None

    # ---- marginal likelihood (baselaplace.py:214-241,1003-1037,1074-1109) ------------------------------------
    @property
    def log_likelihood(self):
        factor = -self._H_factor
        if self.likelihood == "regression":
            c = self.n_data * self.n_outputs * torch.log(self.sigma_noise * sqrt(2 * pi))
            return factor * self.loss - c
        return factor * self.loss

    @property
    def scatter(self):
        delta = self.mean - self.prior_mean
        return (delta * self.prior_precision_diag) @ delta

    @property
    def log_det_prior_precision(self):
        return self.prior_precision_diag.log().sum()

    @property
    def log_det_ratio(self):
        return self.log_det_posterior_precision - self.log_det_prior_precision

    def log_marginal_likelihood(self, prior_precision=None, sigma_noise=None):
        if prior_precision is not None:
            self.prior_precision = prior_precision
        if sigma_noise is not None:
            if self.likelihood != "regression":
                raise ValueError("Can only change sigma_noise for regression.")
            self.sigma_noise = sigma_noise
        return self.log_likelihood - 0.5 * (self.log_det_ratio + self.scatter)

    def optimize_prior_precision(self, pred_type: str = "glm", method: str = "marglik", n_steps: int = 100,
                                 lr: float = 1e-1, init_prior_prec=1.0, prior_structure: str = "diag", val_loader=None,
                                 loss=None, log_prior_prec_min: float = -4, log_prior_prec_max: float = 4,
                                 grid_size: int = 100, link_approx: str = "probit", n_samples: int = 100,
                                 verbose: bool = False, progress_bar: bool = False):
        """Same call as the reference's (baselaplace.py:363-509).  ``method='marglik'``: Adam on the log prior
        precision (scalar / layer-wise / per-parameter: a scalar ``init_prior_prec`` is expanded to
        ``prior_structure``, a tensor keeps its own structure); every step is one ``lk_kron_logdet_blocks_f32`` call
        over all blocks of the posterior for the Kron flavours, with the analytic derivative in the prior precision.
        ``method='gridsearch'``: :meth:`gridsearch_prior_precision` on ``val_loader``."""
        if method == "marglik":
            pp = torch.as_tensor(init_prior_prec, device=self._device, dtype=self._dtype).reshape(-1)
            if len(pp) == 1 and prior_structure != "scalar":
                if prior_structure not in ("layerwise", "diag"):
                    raise ValueError(f"Invalid prior structure {prior_structure}.")
                pp = pp.repeat(self.n_layers if prior_structure == "layerwise" else self.n_params)
            self.prior_precision = pp
            log_pp = self.prior_precision.log().clone().requires_grad_(True)
            opt = torch.optim.Adam([log_pp], lr=lr)
            for _ in range(n_steps):
                opt.zero_grad()
                neg = -self.log_marginal_likelihood(prior_precision=log_pp.exp())
                neg.backward()
                opt.step()
            self.prior_precision = log_pp.detach().exp()
        elif method == "gridsearch":
            if val_loader is None:
                raise ValueError("gridsearch requires a validation set DataLoader")
            self.gridsearch_prior_precision(val_loader, log_prior_prec_min, log_prior_prec_max, grid_size, pred_type,
                                            link_approx, n_samples, loss)
        else:
            raise ValueError("For now only marglik and gridsearch is implemented.")
        if verbose:
            print(f"Optimized prior precision is {self.prior_precision}.")
        return self.prior_precision

    # ---- GLM predictive (baselaplace.py:598-695,1306-1342) ------------------------------------------------------
    def _glm_predictive_distribution(self, X, diagonal_output: bool = False):
        raise NotImplementedError

    def functional_covariance(self, Js: torch.Tensor) -> torch.Tensor:
        """``[B*C, B*C]`` joint GLM covariance ``Js P^-1 Js^T`` (baselaplace.py:1418-1437)."""
        raise NotImplementedError

    def _jacobians(self, X):
        return (self.backend.last_layer_jacobians(X) if self.subset_of_weights == "last_layer"
                else self.backend.jacobians(X))

    @torch.no_grad()
    def _glm_joint_distribution(self, X):
        """baselaplace.py:1329-1331: the joint predictive over a batch needs the materialised Jacobians (the
        cross-sample blocks couple all rows), assembled by the HIP Jacobian kernels."""
        Js, f_mu = self._jacobians(X)
        return f_mu.flatten().detach(), self.functional_covariance(Js).detach()

    @torch.no_grad()
    def __call__(self, x, pred_type: str = "glm", joint: bool = False, link_approx: str = "probit", n_samples: int = 100,
                 diagonal_output: bool = False, generator: torch.Generator | None = None):
        """Posterior predictive (baselaplace.py:598-695,1112-1208): ``pred_type`` 'glm' with the 'probit', 'mc', 'bridge'
        or 'bridge_norm' link approximation, or 'nn' (weight-space sampling, 'mc' only)."""
        if pred_type not in ("glm", "nn"):
            raise ValueError("Only glm and nn supported as prediction types.")
        if link_approx not in ("probit", "mc", "bridge", "bridge_norm"):
            raise ValueError(f"Unsupported link approximation {link_approx}.")
        if pred_type == "nn":
            if link_approx != "mc":
                raise ValueError("Only mc link approximation is supported for nn prediction type.")
            if self.likelihood == "regression":
                samples = self._nn_predictive_samples(x, n_samples, generator)
                return samples.mean(dim=0), samples.var(dim=0)
            return self._nn_predictive_samples(x, n_samples, generator).mean(dim=0)
        if joint and self.likelihood == "regression":  # joint=True only applies to regression (baselaplace.py:646-648)
            return self._glm_joint_distribution(x)
        f_mu, f_var = self._glm_predictive_distribution(x, diagonal_output=diagonal_output)
        if self.likelihood == "regression":
            return f_mu, f_var
        if link_approx == "mc":
            fv = f_var if not diagonal_output else torch.diag_embed(f_var)
            return self._glm_predictive_samples(f_mu, fv, n_samples, diagonal_output, generator).mean(dim=0)
        if link_approx in ("bridge", "bridge_norm"):
            if diagonal_output:
                f_mu, f_var = self._glm_predictive_distribution(x)  # the bridge needs the full output covariance
            return self._laplace_bridge(f_mu, f_var, normalise=link_approx == "bridge_norm")
        var_diag = f_var if diagonal_output else torch.diagonal(f_var, dim1=1, dim2=2)
        kappa = 1 / torch.sqrt(1.0 + pi / 8 * var_diag)
        return torch.softmax(kappa * f_mu, dim=-1)

    def _laplace_bridge(self, f_mu, f_var, normalise: bool):
        """Dirichlet mean of the Laplace bridge (baselaplace.py:665-692): project the logit Gaussian onto the
        zero-sum subspace, optionally rescale it to a mean output variance of sqrt(K/2), map it to Dirichlet
        concentrations and normalise."""
        K_out = f_mu.shape[-1]
        row, col, tot = f_var.sum(-1), f_var.sum(-2), f_var.sum((1, 2))
        mu = f_mu - row * (f_mu.sum(-1) / tot).unsqueeze(-1)
        var = torch.diagonal(f_var, dim1=1, dim2=2) - row * col / tot.unsqueeze(-1)
        if normalise:
            scale = var.mean(dim=1, keepdim=True) / (K_out / 2) ** 0.5
            mu = mu / scale.sqrt()
            var = var / scale
        alpha = (1 - 2 / K_out + mu.exp() / K_out**2 * torch.exp(-mu).sum(dim=1, keepdim=True)) / var
        return torch.nan_to_num(alpha / alpha.sum(dim=1, keepdim=True), nan=1.0)

    # ---- serialisation, interchangeable with the reference's classes (baselaplace.py:1509-1557,1867-1879) ----------
    _REF_NAMES = {"kron": "Kron", "diag": "Diag", "full": "Full"}

    @property
    def _ref_cls_name(self) -> str:
        """name of the reference class a checkpoint of this object loads into (and comes from)"""
        return self._REF_NAMES[self._structure] + ("LLLaplace" if self.subset_of_weights == "last_layer" else "Laplace")

    def _H_for_state(self):
        return self.H

    def state_dict(self) -> dict:
        if self.H is None:
            raise AttributeError("Laplace not fit. Run fit() first.")
        sd = {"mean": self.mean, "H": self._H_for_state(), "loss": self.loss, "prior_mean": self.prior_mean,
              "prior_precision": self.prior_precision, "sigma_noise": self.sigma_noise, "n_data": self.n_data,
              "n_outputs": self.n_outputs, "likelihood": self.likelihood, "temperature": self.temperature,
              "enable_backprop": False, "cls_name": self._ref_cls_name}
        if self.subset_of_weights == "last_layer":  # lllaplace.py:335-339
            sd["data"] = self.data
            sd["_last_layer_name"] = self._last_layer_name
        return sd

    def _set_H_from_state(self, H):
        self.H = H

    def load_state_dict(self, state_dict: dict) -> None:
        if state_dict["cls_name"] != self._ref_cls_name:
            raise ValueError("Loading a wrong Laplace type. Make sure `subset_of_weights` and `hessian_structure` "
                             "are correct!")
        if self.subset_of_weights == "last_layer":
            if state_dict.get("_last_layer_name", self._last_layer_name) != self._last_layer_name:
                raise ValueError("Different `last_layer_name` detected!")
            self.data = state_dict.get("data")
        if len(state_dict["mean"]) != self.n_params:
            raise ValueError("Attempting to load Laplace with different number of parameters than the model.")
        if str(getattr(state_dict["likelihood"], "value", state_dict["likelihood"])) != self.likelihood:
            raise ValueError("Different likelihoods detected!")
        to = lambda t: t.to(self._device, self._dtype) if torch.is_tensor(t) else t  # noqa: E731
        self.mean = to(state_dict["mean"])
        self.loss = to(state_dict["loss"])
        self.prior_mean = to(state_dict["prior_mean"])
        self.prior_precision = to(state_dict["prior_precision"])
        self.sigma_noise = to(state_dict["sigma_noise"])
        self.n_data = state_dict["n_data"]
        self.n_outputs = state_dict["n_outputs"]
        setattr(self.model, "output_size", self.n_outputs)
        self.temperature = state_dict["temperature"]
        self._set_H_from_state(state_dict["H"])
        self._posterior_cache = None

    # ---- sampling predictives (baselaplace.py:697-841,1210-1394) ------------------------------------------------
    def sample(self, n_samples: int = 100, generator: torch.Generator | None = None) -> torch.Tensor:
        """``[n_samples, P]`` draws from the Laplace posterior N(mean, posterior_precision^-1)."""
        raise NotImplementedError

    def _randn(self, *shape, generator=None):
        return torch.randn(*shape, device=self._device, dtype=self._dtype, generator=generator)

    @torch.no_grad()
    def _nn_functional_samples(self, X, n_samples: int = 100, generator=None) -> torch.Tensor:
        fs = []
        try:
            for sample in self.sample(n_samples, generator):
                vector_to_parameters(sample, self.params)
                fs.append(self.model(X.to(self._device) if torch.is_tensor(X) else X).detach())
        finally:
            vector_to_parameters(self.mean, self.params)
        return torch.stack(fs)

    def _nn_predictive_samples(self, X, n_samples: int = 100, generator=None) -> torch.Tensor:
        fs = self._nn_functional_samples(X, n_samples, generator)
        return torch.softmax(fs, dim=-1) if self.likelihood == "classification" else fs

    def _glm_functional_samples(self, f_mu, f_var, n_samples, diagonal_output=False, generator=None):
        """utils/utils.py:337-378 (normal_samples): the same ``[C, n_samples]`` standard-normal draw is shared by
        all points of the batch, exactly as in the reference."""
        if f_var.shape != (f_mu.shape[0], f_mu.shape[1], f_mu.shape[1]):
            raise ValueError("f_var must be [batch, outputs, outputs]")
        z = torch.randn(f_mu.shape[1], n_samples, device=f_mu.device, dtype=f_mu.dtype, generator=generator)
        if diagonal_output:
            scaled = torch.diagonal(f_var, dim1=1, dim2=2).sqrt().unsqueeze(-1) * z.unsqueeze(0)
        else:
            scaled = torch.matmul(torch.linalg.cholesky(f_var), z.unsqueeze(0))
        return (f_mu.unsqueeze(-1) + scaled).permute(2, 0, 1)

    def _glm_predictive_samples(self, f_mu, f_var, n_samples, diagonal_output=False, generator=None):
        fs = self._glm_functional_samples(f_mu, f_var, n_samples, diagonal_output, generator)
        return torch.softmax(fs, dim=-1) if self.likelihood == "classification" else fs

    @torch.no_grad()
    def functional_samples(self, x, pred_type: str = "glm", n_samples: int = 100, diagonal_output: bool = False,
                           generator=None) -> torch.Tensor:
        if pred_type == "glm":
            f_mu, f_var = self._glm_predictive_distribution(x)
            return self._glm_functional_samples(f_mu, f_var, n_samples, diagonal_output, generator)
        if pred_type == "nn":
            return self._nn_functional_samples(x, n_samples, generator)
        raise ValueError("Only glm and nn supported as prediction types.")

    @torch.no_grad()
    def predictive_samples(self, x, pred_type: str = "glm", n_samples: int = 100, diagonal_output: bool = False,
                           generator=None) -> torch.Tensor:
        if pred_type == "glm":
            f_mu, f_var = self._glm_predictive_distribution(x)
            return self._glm_predictive_samples(f_mu, f_var, n_samples, diagonal_output, generator)
        if pred_type == "nn":
            return self._nn_predictive_samples(x, n_samples, generator)
        raise ValueError("Only glm and nn supported as prediction types.")

    # ---- prior-precision grid search on a validation set (baselaplace.py:487-561) --------------------------------
    @torch.no_grad()
    def gridsearch_prior_precision(self, val_loader, log_prior_prec_min: float = -4, log_prior_prec_max: float = 4,
                                   grid_size: int = 100, pred_type: str = "glm", link_approx: str = "probit",
                                   n_samples: int = 100, loss=None):
        """Pick the scalar prior precision of ``logspace(min, max, grid_size)`` with the best validation loss
        (default: NLL for classification, MSE for regression, as the reference's ``RunningNLLMetric`` /
        ``MeanSquaredError``).  A grid point whose posterior is not positive definite scores ``inf``."""
        from collections.abc import MutableMapping

        def batches():
            for data in val_loader:
                if isinstance(data, MutableMapping):  # HuggingFace-style batch: labels under dict_key_y
                    yield data, data[self.backend.dict_key_y].to(self._device)
                else:
                    yield data[0].to(self._device), data[1].to(self._device)

        cached = None
        if self.subset_of_weights == "last_layer" and pred_type == "glm" and self.backend.last_layer:
            # the backbone does not depend on the prior: one feature pass per validation batch for the whole grid
            cached = [(self.backend.cache_features(X), y) for X, y in batches()]
        interval = torch.logspace(log_prior_prec_min, log_prior_prec_max, grid_size)
        results = []
        for pp in interval:
            self.prior_precision = pp
            try:
                tot = torch.zeros((), dtype=torch.float64, device=self._device)
                cnt = 0
                for X, y in (cached if cached is not None else batches()):
                    out = self(X, pred_type=pred_type, link_approx=link_approx, n_samples=n_samples)
                    if loss is not None:
                        tot += loss(out, y) * len(y)
                    elif self.likelihood == "regression":
                        tot += ((out[0] - y.reshape(out[0].shape)) ** 2).sum()
                    else:
                        tot += -torch.log(out[torch.arange(len(y), device=y.device), y.long()].clamp_min(1e-30)).sum()
                    cnt += len(y)
                res = float(tot) / max(cnt, 1)  # the one host read-back of this grid point
                results.append(res if res == res else float("inf"))
            except RuntimeError as err:  # torch.linalg.LinAlgError is a RuntimeError
                if "positive" in str(err) or "singular" in str(err) or isinstance(err, torch.linalg.LinAlgError):
                    results.append(float("inf"))
                else:
                    raise
        best = int(torch.tensor(results).argmin())
        self.prior_precision = interval[best]
        return self.prior_precision


class HipKronLaplace(_HipLaplace):
    _structure = "kron"

    """KFAC Laplace (KronLaplace, baselaplace.py:1704-1879) on the HIP kernels."""

    def __init__(self, *args, damping: bool = False, **kwargs):
        self.damping = damping
        self.H_facs = None
        super().__init__(*args, **kwargs)

    def _init_H(self):
        self.H = HipKron.init_from_model(self.params, self._device, self._dtype)

    def _curv_closure(self, X, y, N):
        return self.backend.kron(X, y, N=N)

    def _curvature_tensors(self):
        return [Hi for F in self.H.kfacs for Hi in F]

    @staticmethod
    def _rescale_factors(kron, factor: float):
        """``A *= factor`` on the two-factor blocks (baselaplace.py:1773-1777)"""
        for F in kron.kfacs:
            if len(F) == 2:
                F[1] *= factor
        return kron

    def fit(self, train_loader, override: bool = True, process_group=None, distributed=None, fused: bool = True):
        """``fused=True`` (default) accumulates in place through :class:`KronAccumulator` (upper
        triangles, native conv order; one symmetrise/permute per fit).  ``fused=False`` is the
        reference's literal loop ``self.H += backend.kron(X, y, N)`` (baselaplace.py:969-985)."""
        # online continuation (baselaplace.py:1785-1806): the A factors carry 1/N, so the old ones are discounted by
        # n_old / (n_old + n_new) and the new ones (computed with N = n_new) by n_new / (n_old + n_new)
        old = None
        distributed = resolve_distributed(train_loader, distributed, process_group)
        if not override and self.H_facs is not None:
            n_old, n_new = self.n_data, len(train_loader.dataset)
            old = (self._rescale_factors(self.H_facs, n_old / (n_old + n_new)), self.loss, n_old, n_new)
        if not (fused and hasattr(self.backend, "kron_accumulator")):
            super().fit(train_loader, override=True, process_group=process_group, distributed=distributed)
        else:
            self.model.eval()
            self.mean = parameters_to_vector(self.params).detach()
            N = len(train_loader.dataset)
            acc = self.backend.kron_accumulator(N)
            for data in train_loader:
                if isinstance(data, dict) or hasattr(data, "keys"):
                    X, y = data, data[self.backend.dict_key_y].to(self._device)
                else:
                    X, y = data
                    X, y = X.to(self._device), y.to(self._device)
                if self.n_outputs is None:
                    with torch.no_grad():
                        self.n_outputs = self.model(X[:1] if torch.is_tensor(X) else X).shape[-1]
                    setattr(self.model, "output_size", self.n_outputs)
                    if self.subset_of_weights == "last_layer" and torch.is_tensor(X):
                        self.data = (X[:1].detach().cpu(), y[:1].detach().cpu())
                acc.add_batch(X, y)
            if distributed:
                acc.ensure_allocated(self._device)  # a rank whose shard is empty contributes zeros
                allreduce_curvature(acc.tensors(), group=process_group, mirror=False)  # finalize() mirrors
                _share_n_outputs(self, process_group)
            self.loss, self.H = acc.finalize()
            self.n_data = N
            self._posterior_cache = None
        if old is None:
            self.H_facs = self.H
        else:
            facs, loss_old, n_old, n_new = old
            facs += self._rescale_factors(self.H, n_new / (n_new + n_old))
            self.H_facs = facs
            self.loss = self.loss + loss_old
            self.n_data = n_old + n_new
        # HIP eigensolver per factor; after a data-parallel fit the factors are sharded over the ranks
        self.H = self.H_facs.decompose(damping=self.damping, distributed=bool(distributed), process_group=process_group)

    @property
    def posterior_precision(self):
        pp = self.prior_precision
        if len(pp) not in (1, self.n_layers):
            raise ValueError("Prior precision for Kron either scalar or per-layer.")
        return self.H * self._H_factor + pp

    @property
    def log_det_posterior_precision(self):
        return self.posterior_precision.logdet()

    def functional_variance(self, Js: torch.Tensor) -> torch.Tensor:
        return self.posterior_precision.inv_square_form(Js)

    def _H_for_state(self):
        return self.H_facs.kfacs  # the factors, not the eigendecomposition (baselaplace.py:1867-1871)

    def _set_H_from_state(self, kfacs):
        to = lambda t: t.to(self._device, self._dtype)  # noqa: E731
        self.H_facs = HipKron([[to(Hi) for Hi in F] for F in kfacs])
        self.H = self.H_facs.decompose(damping=self.damping)

    def functional_covariance(self, Js: torch.Tensor) -> torch.Tensor:
        """baselaplace.py:1837-1843"""
        n_batch, n_outs, n_params = Js.shape
        return self.posterior_precision.inv_square_form(Js.reshape(1, n_batch * n_outs, n_params)).squeeze(0)

    def sample(self, n_samples: int = 100, generator=None) -> torch.Tensor:
        """baselaplace.py:1845-1858: ``mean + P^{-1/2} z`` block-wise through the eigendecomposition."""
        z = self._randn(n_samples, self.n_params, generator=generator)
        z = self.posterior_precision.bmm(z, exponent=-0.5)
        return self.mean.reshape(1, self.n_params) + z.reshape(n_samples, self.n_params)

    def _glm_predictive_distribution(self, X, diagonal_output: bool = False):
        post = self.posterior_precision
        try:
            f_mu, f_var = _pred.glm_variance_kron(self.backend, X, post)  # fused, no [B, C, P] Jacobian
        except NotImplementedError:
            Js, f_mu = (self.backend.last_layer_jacobians(X) if self.subset_of_weights == "last_layer"
                        else self.backend.jacobians(X))
            f_var = post.inv_square_form(Js)
        if diagonal_output:
            f_var = torch.diagonal(f_var, dim1=-2, dim2=-1)
        return f_mu.detach(), f_var.detach()


class HipDiagLaplace(_HipLaplace):
    _structure = "diag"

    """Diagonal Laplace (DiagLaplace, baselaplace.py:2048-2136)."""

    def _init_H(self):
        self.H = torch.zeros(self.n_params, device=self._device, dtype=self._dtype)

    def _curv_closure(self, X, y, N):
        return self.backend.diag(X, y, N=N)

    def _curvature_tensors(self):
        return [self.H]

    @property
    def posterior_precision(self):
        return self._H_factor * self.H + self.prior_precision_diag

    @property
    def posterior_variance(self):
        return 1 / self.posterior_precision

    @property
    def log_det_posterior_precision(self):
        return self.posterior_precision.log().sum()

    def functional_covariance(self, Js: torch.Tensor) -> torch.Tensor:
        """baselaplace.py:2117-2122"""
        n_batch, n_outs, n_params = Js.shape
        Js = Js.reshape(n_batch * n_outs, n_params)
        return torch.einsum("np,p,mp->nm", Js, self.posterior_variance, Js)

    def sample(self, n_samples: int = 100, generator=None) -> torch.Tensor:
        """baselaplace.py:2124-2131"""
        z = self._randn(n_samples, self.n_params, generator=generator)
        return self.mean.reshape(1, self.n_params) + z * self.posterior_variance.sqrt().reshape(1, self.n_params)

    def functional_variance(self, Js):
        return torch.einsum("ncp,p,nkp->nck", Js, self.posterior_variance, Js)

    def _glm_predictive_distribution(self, X, diagonal_output: bool = False):
        try:
            f_mu, f_var = _pred.glm_variance_diag(self.backend, X, self.posterior_variance)
        except NotImplementedError:
            Js, f_mu = (self.backend.last_layer_jacobians(X) if self.subset_of_weights == "last_layer"
                        else self.backend.jacobians(X))
            f_var = self.functional_variance(Js)
        if diagonal_output:
            f_var = torch.diagonal(f_var, dim1=-2, dim2=-1)
        return f_mu.detach(), f_var.detach()


class HipFullLaplace(_HipLaplace):
    _structure = "full"

    """Dense Laplace (FullLaplace, baselaplace.py:1572-1701).  The one-off ``P^3`` factorisation of
    the posterior precision stays a library call (torch.linalg.cholesky -> rocSOLVER), exactly where
    the reference calls it (utils/utils.py:118-129); accumulation and the predictive are HIP."""

    def _init_H(self):
        self.H = torch.zeros(self.n_params, self.n_params, device=self._device, dtype=self._dtype)

    def _curv_closure(self, X, y, N):
        return self.backend.full(X, y, N=N)

    def _curvature_tensors(self):
        return [self.H]

    @property
    def posterior_precision(self):
        return self._H_factor * self.H + torch.diag(self.prior_precision_diag)

    @property
    def posterior_covariance(self):
        if self._posterior_cache is None:
            L = torch.linalg.cholesky(self.posterior_precision)
            self._posterior_cache = torch.cholesky_inverse(L)
        return self._posterior_cache

    @property
    def log_det_posterior_precision(self):
        return self.posterior_precision.logdet()

    def functional_variance(self, Js):
        return torch.einsum("ncp,pq,nkq->nck", Js, self.posterior_covariance, Js)

    def functional_covariance(self, Js: torch.Tensor) -> torch.Tensor:
        """baselaplace.py:1686-1689"""
        n_batch, n_outs, n_params = Js.shape
        Js = Js.reshape(n_batch * n_outs, n_params)
        return torch.einsum("np,pq,mq->nm", Js, self.posterior_covariance, Js)

    def sample(self, n_samples: int = 100, generator=None) -> torch.Tensor:
        """baselaplace.py:1691-1703: ``mean + z L^T``; ``L`` = lower Cholesky factor of the posterior covariance
        (what ``invsqrt_precision`` / ``_precision_to_scale_tril`` return, utils/utils.py:118-129)."""
        z = self._randn(n_samples, self.n_params, generator=generator)
        scale = torch.linalg.cholesky(self.posterior_covariance)
        return self.mean.reshape(1, self.n_params) + z @ scale.T

    def _glm_predictive_distribution(self, X, diagonal_output: bool = False):
        if self.subset_of_weights == "last_layer":
            f_mu, f_var = _pred.glm_variance_full_last_layer(self.backend, X, self.posterior_covariance)
        else:
            Js, f_mu = self.backend.jacobians(X)
            f_var = self.functional_variance(Js)
        if diagonal_output:
            f_var = torch.diagonal(f_var, dim1=-2, dim2=-1)
        return f_mu.detach(), f_var.detach()


def fit_kron(la, train_loader, process_group=None, distributed: bool | None = None):
    """Fused ``fit`` for the REFERENCE's own ``KronLaplace`` / ``KronLLLaplace`` objects built with
    ``backend=HipGGN`` (or ``HipEF``)::

        la = Laplace(model, "classification", "all", "kron", backend=HipGGN)
        laplace_amd.fit_kron(la, train_loader)        # instead of la.fit(train_loader)

    Same result as ``la.fit(train_loader)`` with ``override=True`` (laplace/baselaplace.py:904-987,1785-1809), but the
    minibatches go through :class:`KronAccumulator` — in-place accumulation, pixel-pair A factors assembled once,
    one symmetrise / permute per fit — instead of one ``Kron`` per minibatch, and on several ranks
    (``torch.distributed`` initialised) the factors are summed with one all-reduce and decomposed sharded."""
    from collections.abc import MutableMapping

    la.model.eval()
    if hasattr(la.model, "find_last_layer") and getattr(la.model, "last_layer", None) is None:
        # last-layer flavours discover the head on the first batch (laplace/lllaplace.py:189-203)
        la.data = next(iter(train_loader))
        la._find_last_layer(la.data)
        la.params = [p for p in la.model.last_layer.parameters() if p.requires_grad]
        la.n_params = len(parameters_to_vector(la.model.last_layer.parameters()))
        la.n_layers = len(list(la.model.last_layer.parameters()))
        la.prior_precision = la._prior_precision
        la.prior_mean = la._prior_mean
    backend = la.backend  # (instantiated lazily by the reference; for last-layer flavours it needs the head)
    if not hasattr(backend, "kron_accumulator"):
        raise TypeError("fit_kron needs a laplace_amd backend (HipGGN / HipEF)")
    la.mean = parameters_to_vector(la.params).detach()
    N = len(train_loader.dataset)
    acc = backend.kron_accumulator(N, **getattr(la, "_asdl_fisher_kwargs", {}))
    first = True
    for data in train_loader:
        if isinstance(data, MutableMapping):
            X, y = data, data[la.dict_key_y].to(la._device)
        else:
            X, y = data
            X, y = X.to(la._device), y.to(la._device)
        if first:
            with torch.no_grad():
                out = la.model(X if isinstance(X, MutableMapping) else X[:1])
            la.n_outputs = out.shape[-1]
            setattr(la.model, "output_size", la.n_outputs)
            first = False
        acc.add_batch(X, y)
    distributed = resolve_distributed(train_loader, distributed, process_group)
    if distributed:
        acc.ensure_allocated(la._device)  # a rank whose shard is empty contributes zeros
        allreduce_curvature(acc.tensors(), group=process_group, mirror=False)  # finalize() mirrors
        _share_n_outputs(la, process_group)
    la.loss, la.H_facs = acc.finalize()
    la.n_data = N
    la.H = la.H_facs.decompose(damping=la.damping, distributed=bool(distributed), process_group=process_group)
    return la


def glm_predictive(la, X, diagonal_output: bool = False, fallback=None):
    """Fused GLM predictive ``(f_mu, f_var)`` for the REFERENCE's own Laplace objects built with ``backend=HipGGN``::

        f_mu, f_var = laplace_amd.glm_predictive(la, x_test)     # instead of la._glm_predictive_distribution(x_test)

    Same result as ``_glm_predictive_distribution`` (laplace/baselaplace.py:1306-1342, lllaplace.py:212-237), which
    materialises ``Js [B, C, P]`` (447 MB per sample for ResNet-18) before ``functional_variance``; here the layer
    factors of the Jacobian go straight into the quadratic-form kernels (:mod:`laplace_amd.predictive`).  Posteriors
    the fused kernels do not cover (dense full-network, damping, models with parameters outside Linear/Conv2d) fall
    through to the reference's own method."""
    backend = la.backend
    if not hasattr(backend, "_forward"):
        raise TypeError("glm_predictive needs a laplace_amd backend (HipGGN / HipEF)")
    la.model.eval()
    try:
        if hasattr(la, "H_facs"):
            f_mu, f_var = _pred.glm_variance_kron(backend, X, la.posterior_precision)
        elif torch.is_tensor(la.H) and la.H.ndim == 1:
            f_mu, f_var = _pred.glm_variance_diag(backend, X, la.posterior_variance)
        else:
            f_mu, f_var = _pred.glm_variance_full_last_layer(backend, X, la.posterior_covariance)
    except NotImplementedError:
        return (fallback or la._glm_predictive_distribution)(X, diagonal_output=diagonal_output)
    if diagonal_output:
        f_var = torch.diagonal(f_var, dim1=-2, dim2=-1)
    return f_mu.detach(), f_var.detach()


_FLAVOURS = {"kron": HipKronLaplace, "diag": HipDiagLaplace, "full": HipFullLaplace}


def HipLaplace(model, likelihood, subset_of_weights="all", hessian_structure="kron", **kwargs):
    """Factory with the call shape of ``laplace.Laplace`` (laplace/laplace.py:13-47)."""
    if hessian_structure not in _FLAVOURS:
        raise ValueError(f"hessian_structure must be one of {sorted(_FLAVOURS)}")
    return _FLAVOURS[hessian_structure](model, likelihood, subset_of_weights=subset_of_weights, **kwargs)
