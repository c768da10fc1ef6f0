"""CPU restatement of the reference's curvature hot path (torch-CPU, any float dtype).

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Every function cites the reference
file:line (relative to /root/reference) whose behaviour it restates.  The code is written
from the *mathematical definition* the reference implements, with plain autograd loops
instead of ``torch.func`` so that it is an independent check of the vectorised paths.

Conventions (reference: laplace/curvature/curvature.py:46-86):
  * ``params`` = ``[p for p in model.parameters() if p.requires_grad]`` in that order, each
    flattened row-major; ``P`` = total count.
  * ``likelihood`` in {"classification", "regression"}; loss = CrossEntropyLoss(sum) with
    ``factor`` 1.0, resp. MSELoss(sum) with ``factor`` 0.5 (curvature.py:63-72).
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
from torch import nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# basics
# --------------------------------------------------------------------------------------
def trainable_params(model: nn.Module) -> list[nn.Parameter]:
    """curvature.py:74-76."""
    return [p for p in model.parameters() if p.requires_grad]


def likelihood_factor(likelihood: str) -> float:
    """curvature.py:63-72."""
    return 0.5 if likelihood == "regression" else 1.0


def loss_sum(f: torch.Tensor, y: torch.Tensor, likelihood: str) -> torch.Tensor:
    """``factor * lossfunc(f, y)`` with sum reduction (curvature.py:63-72,408,417)."""
    if likelihood == "regression":
        return 0.5 * ((f - y) ** 2).sum()
    logp = torch.log_softmax(f, dim=-1)
    return -logp.gather(1, y.view(-1, 1).long()).sum()


def jacobians(model: nn.Module, X, params: Sequence[nn.Parameter] | None = None):
    """Per-sample output Jacobians ``Js[B, C, P]`` and outputs ``f[B, C]``.

    Restates CurvatureInterface.jacobians (curvature.py:88-129) the way the reference's own
    tests check it: one reverse pass per (sample, output) (tests/utils.py:85-106).
    """
    params = list(params) if params is not None else trainable_params(model)
    f = model(X)
    if f.ndim == 1:
        f = f.unsqueeze(-1)
    B, C = f.shape
    rows = []
    for n in range(B):
        per_out = []
        for c in range(C):
            grads = torch.autograd.grad(f[n, c], params, retain_graph=True, allow_unused=True)
            per_out.append(
                torch.cat(
                    [
                        (g if g is not None else torch.zeros_like(p)).reshape(-1)
                        for g, p in zip(grads, params)
                    ]
                )
            )
        rows.append(torch.stack(per_out))
    return torch.stack(rows).detach(), f.detach()


def last_layer_jacobians(phi: torch.Tensor, n_outputs: int, bias: bool):
    """Closed-form last-layer Jacobians (curvature.py:131-167).

    ``Js[n, i, j*D + p] = delta_ij * phi[n, p]`` (weight block, row-major ``[C, D]``), then the
    bias block ``delta_ij``.
    """
    B, D = phi.shape
    C = n_outputs
    Js = torch.zeros(B, C, C * D + (C if bias else 0), dtype=phi.dtype)
    for i in range(C):
        Js[:, i, i * D : (i + 1) * D] = phi
        if bias:
            Js[:, i, C * D + i] = 1.0
    return Js


def per_sample_gradients(model: nn.Module, X, y, likelihood: str, params=None):
    """``Gs[B, P]`` of the *unscaled* torch loss and its sum (curvature.py:169-210)."""
    params = list(params) if params is not None else trainable_params(model)
    f = model(X)
    Gs = []
    total = 0.0
    for n in range(f.shape[0]):
        if likelihood == "regression":
            ln = ((f[n] - y[n]) ** 2).sum()
        else:
            ln = -torch.log_softmax(f[n], dim=-1)[y[n].long()]
        grads = torch.autograd.grad(ln, params, retain_graph=True, allow_unused=True)
        Gs.append(
            torch.cat(
                [(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(grads, params)]
            )
        )
        total = total + ln.detach()
    return torch.stack(Gs).detach(), total


# --------------------------------------------------------------------------------------
# likelihood Hessians in function space
# --------------------------------------------------------------------------------------
def functional_hessian(f: torch.Tensor, likelihood: str):
    """``Lambda_n = diag(p) - p p^T`` for softmax-CE; identity for MSE (curvature.py:366-373)."""
    B, C = f.shape
    if likelihood == "regression":
        return torch.eye(C, dtype=f.dtype).expand(B, C, C).clone()
    p = torch.softmax(f, dim=-1)
    return torch.diag_embed(p) - p.unsqueeze(2) * p.unsqueeze(1)


def functional_hessian_sqrt(f: torch.Tensor, likelihood: str):
    """A square root ``S_n`` with ``S_n S_n^T = Lambda_n``; ``S[n, :, c]`` is column ``c``.

    softmax-CE: ``S = diag(sqrt p) - p sqrt(p)^T`` (valid because ``sum p = 1``); MSE: identity.
    The GGN/KFAC results are invariant to the choice of root.
    """
    B, C = f.shape
    if likelihood == "regression":
        return torch.eye(C, dtype=f.dtype).expand(B, C, C).clone()
    p = torch.softmax(f, dim=-1)
    sp = p.sqrt()
    return torch.diag_embed(sp) - p.unsqueeze(2) * sp.unsqueeze(1)


# --------------------------------------------------------------------------------------
# dense / diagonal GGN and EF
# --------------------------------------------------------------------------------------
def mc_functional_fisher(grad_samples: torch.Tensor):
    """Middle matrix of the MC Fisher GIVEN the sampled functional gradients ``[S, B, C]``
    (curvature.py:341-364: ``F += 1/num_samples * einsum('bc,bk->bck', grad_sample, grad_sample)``; the sampling
    itself — ``f - y~`` with ``y~ ~ N(f, 1)``, or ``softmax(f) - y~`` with ``y~ ~ Multinomial(logits=f)`` — is done
    by the caller so that a test can hand the same draws to both sides)."""
    S = grad_samples.shape[0]
    F_ = torch.zeros(grad_samples.shape[1], grad_samples.shape[2], grad_samples.shape[2], dtype=grad_samples.dtype)
    for s in range(S):
        g = grad_samples[s]
        F_ += g.unsqueeze(2) * g.unsqueeze(1) / S
    return F_


def ggn_full(Js: torch.Tensor, H_lik: torch.Tensor):
    """``H = sum_n J_n^T Lambda_n J_n`` (curvature.py:375-411, einsum ``bcp,bck,bkq->pq``)."""
    H = torch.zeros(Js.shape[-1], Js.shape[-1], dtype=Js.dtype)
    for n in range(Js.shape[0]):
        H += Js[n].T @ H_lik[n] @ Js[n]
    return H


def ggn_diag(Js: torch.Tensor, H_lik: torch.Tensor):
    """Diagonal of :func:`ggn_full` (curvature.py:413-433, einsum ``bcp,bck,bkp->p``)."""
    h = torch.zeros(Js.shape[-1], dtype=Js.dtype)
    for n in range(Js.shape[0]):
        h += ((H_lik[n] @ Js[n]) * Js[n]).sum(0)
    return h


def ef_full(Gs: torch.Tensor, likelihood: str):
    """``factor * Gs^T Gs`` (curvature.py:467-492)."""
    return likelihood_factor(likelihood) * Gs.T @ Gs


def ef_diag(Gs: torch.Tensor, likelihood: str):
    """``factor * sum_n Gs_n^2`` (curvature.py:494-505)."""
    return likelihood_factor(likelihood) * (Gs**2).sum(0)


# --------------------------------------------------------------------------------------
# KFAC (curvlinops 2.0.0 semantics through laplace/curvature/curvlinops.py:46-108)
# --------------------------------------------------------------------------------------
def _supported(module: nn.Module) -> bool:
    return isinstance(module, (nn.Linear, nn.Conv2d))


def kfac_modules(model: nn.Module, params: Sequence[nn.Parameter]):
    """Supported modules whose weight is Laplace-tracked, in ``named_modules`` order
    (curvlinops.py:55-75)."""
    ids = {id(p) for p in params}
    out = []
    for name, mod in model.named_modules():
        if _supported(mod) and id(mod.weight) in ids:
            out.append((name, mod))
    return out


def _patch_rows(mod: nn.Module, a: torch.Tensor, kfac_approx: str):
    """Rows of the layer-input matrix whose Gram matrix is the A factor, and L (positions)."""
    if isinstance(mod, nn.Conv2d):
        cols = F.unfold(a, mod.kernel_size, dilation=mod.dilation, padding=mod.padding, stride=mod.stride)
        B, D, L = cols.shape  # D = C_in*kh*kw in F.unfold order
        if kfac_approx == "expand":
            return cols.permute(0, 2, 1).reshape(B * L, D), L
        return cols.mean(dim=2), 1  # 'reduce': average over positions first
    if a.ndim > 2:  # weight sharing over extra dims of a Linear input
        B = a.shape[0]
        L = int(a[0].numel() // a.shape[-1])
        if kfac_approx == "expand":
            return a.reshape(-1, a.shape[-1]), L
        return a.reshape(B, L, -1).mean(dim=1), 1
    return a, 1


def _grad_rows(mod: nn.Module, g: torch.Tensor, kfac_approx: str):
    """Rows of the output-gradient matrix whose Gram matrix is the G factor."""
    if isinstance(mod, nn.Conv2d):
        B, D = g.shape[:2]
        g = g.reshape(B, D, -1)
        if kfac_approx == "expand":
            return g.permute(0, 2, 1).reshape(-1, D)
        return g.sum(dim=2)
    if g.ndim > 2:
        if kfac_approx == "expand":
            return g.reshape(-1, g.shape[-1])
        return g.reshape(g.shape[0], -1, g.shape[-1]).sum(dim=1)
    return g


def kfac_ggn(
    model: nn.Module,
    X,
    y: torch.Tensor,
    N: int,
    likelihood: str,
    params: Sequence[nn.Parameter] | None = None,
    kfac_approx: str = "expand",
    empirical: bool = False,
    mc_grads: torch.Tensor | None = None,
):
    """``(loss, kfacs)`` exactly as ``CurvlinopsInterface.kron`` returns them.

    Follows laplace/curvature/curvlinops.py:77-108 step for step:
      1. forward with input hooks -> per-layer ``A = (1/(M L)) sum a a^T``            (K1)
      2. C backward passes seeded with the columns of the loss-Hessian square root
         (TYPE2/exact; MSE-sum Hessian is 2I, CE-sum Hessian is Lambda) -> ``G = sum g g^T``
         (``empirical=True``: one pass seeded with the loss gradient, FisherType.EMPIRICAL;
         ``mc_grads`` [S, B, C]: FisherType.MC with ``mc_samples = S`` (curvlinops.py:150-164 passes
         ``fisher_type=MC if stochastic``) given the sampled functional gradients in the convention of
         curvature.py:341-364 — curvlinops samples the *loss* gradient, which for ``MSELoss(sum)`` is the
         functional gradient times sqrt(2) in distribution (covariance 2I), and averages over the draws)
      3. ordering ``[G, A]`` then ``[G]`` for the bias (curvlinops.py:55-75)             (K2)
      4. ``A *= M/N`` on two-factor blocks (curvlinops.py:46-53)                        (K3)
      5. ``kron *= factor`` = every factor of a 2-block times ``factor**0.5``, 1-blocks times
         ``factor`` (utils/matrix.py:100-118)                                           (K4)
      6. a second forward for the loss (curvlinops.py:106)
    """
    params = list(params) if params is not None else trainable_params(model)
    mods = kfac_modules(model, params)
    ids = {id(p) for p in params}
    taps: dict[str, dict] = {}
    handles = []
    for name, mod in mods:
        def hook(m, inp, out, name=name):
            # gradient edge of the output AS PRODUCED (curvlinops registers its tensor hook at this point too; a
            # model that later updates the tensor in place must not change which gradient is meant)
            taps[name] = {"a": inp[0].detach(), "out": torch.autograd.graph.get_gradient_edge(out)}
        handles.append(mod.register_forward_hook(hook))
    try:
        f = model(X)
    finally:
        for h in handles:
            h.remove()
    B, C = f.shape
    M = B
    outs = [taps[name]["out"] for name, _ in mods]

    if mc_grads is not None:
        scale = (math.sqrt(2.0) if likelihood == "regression" else 1.0) / math.sqrt(mc_grads.shape[0])
        seeds = [scale * mc_grads[s].to(f.dtype) for s in range(mc_grads.shape[0])]
    elif empirical:
        if likelihood == "regression":
            seeds = [2.0 * (f.detach() - y)]
        else:
            p = torch.softmax(f.detach(), dim=-1)
            seeds = [p - F.one_hot(y.long(), C).to(f.dtype)]
    else:
        S = functional_hessian_sqrt(f.detach(), likelihood)
        if likelihood == "regression":
            S = S * math.sqrt(2.0)  # Hessian of MSELoss(sum) w.r.t. f is 2I
        seeds = [S[:, :, c] for c in range(C)]

    G = {name: 0.0 for name, _ in mods}
    for seed in seeds:
        grads = torch.autograd.grad(f, outs, grad_outputs=seed, retain_graph=True)
        for (name, mod), g in zip(mods, grads):
            rows = _grad_rows(mod, g.detach(), kfac_approx)
            G[name] = G[name] + rows.T @ rows

    fac = likelihood_factor(likelihood)
    kfacs = []
    for name, mod in mods:
        rows, L = _patch_rows(mod, taps[name]["a"], kfac_approx)
        A = rows.T @ rows / (M * L)
        A = A * (M / N)
        has_bias = mod.bias is not None and id(mod.bias) in ids
        if G[name].numel() == 1 and A.numel() == 1 and not has_bias:
            kfacs.append([fac * G[name] * A])
            continue
        kfacs.append([math.sqrt(fac) * G[name], math.sqrt(fac) * A])
        if has_bias:
            kfacs.append([fac * G[name]])
    with torch.no_grad():
        loss = loss_sum(model(X), y, likelihood)
    return loss, kfacs


# --------------------------------------------------------------------------------------
# Kron / KronDecomposed algebra (laplace/utils/matrix.py) — dense, definition-level
# --------------------------------------------------------------------------------------
def kron_product(t1: torch.Tensor, t2: torch.Tensor):
    """Kronecker product (utils/utils.py:148-173)."""
    return torch.einsum("ij,kl->ikjl", t1, t2).reshape(t1.shape[0] * t2.shape[0], t1.shape[1] * t2.shape[1])


def kron_to_matrix(kfacs):
    """Block-diagonal dense matrix of a Kron (utils/matrix.py:255-275)."""
    blocks = [F_[0] if len(F_) == 1 else kron_product(F_[0], F_[1]) for F_ in kfacs]
    return torch.block_diag(*blocks)


def kron_diag(kfacs):
    """utils/matrix.py:238-253."""
    out = []
    for F_ in kfacs:
        if len(F_) == 1:
            out.append(F_[0].diagonal())
        else:
            out.append(torch.outer(F_[0].diagonal(), F_[1].diagonal()).reshape(-1))
    return torch.cat(out)


def kron_add(k1, k2):
    """utils/matrix.py:79-98."""
    return [[a + b for a, b in zip(F1, F2)] for F1, F2 in zip(k1, k2)]


def kron_scale(kfacs, scalar: float):
    """utils/matrix.py:100-118: each factor of a block times ``scalar**(1/len(block))``."""
    return [[(scalar ** (1.0 / len(F_))) * Hi for Hi in F_] for F_ in kfacs]


def symeig(M: torch.Tensor):
    """utils/utils.py:193-228: eigh on the UPPER triangle, ascending, eigenvalues clamped >= 0."""
    Mu = torch.triu(M) + torch.triu(M, 1).T
    l, Q = torch.linalg.eigh(Mu)
    l = torch.nan_to_num(l.clamp(min=0.0))
    return l, torch.nan_to_num(Q)


def kron_decompose(kfacs):
    """utils/matrix.py:123-150 -> (eigenvectors, eigenvalues) nested lists."""
    Qs, ls = [], []
    for F_ in kfacs:
        q, l = [], []
        for Hi in F_:
            li, Qi = symeig(Hi)
            q.append(Qi)
            l.append(li)
        Qs.append(q)
        ls.append(l)
    return Qs, ls


def _block_eigvals(ls, delta, damping: bool):
    if len(ls) == 1:
        return ls[0] + delta
    l1, l2 = ls
    if damping:
        sd = math.sqrt(float(delta)) if not torch.is_tensor(delta) else delta.sqrt()
        return torch.outer(l1 + sd, l2 + sd)
    return torch.outer(l1, l2) + delta


def _expand_deltas(deltas, n_blocks, dtype):
    deltas = torch.as_tensor(deltas, dtype=dtype).reshape(-1)
    if deltas.numel() == 1:
        deltas = deltas.expand(n_blocks)
    return deltas


def krondecomposed_scale(eigvals, scalar: float):
    """utils/matrix.py:342-363."""
    return [[(scalar ** (1.0 / len(ls))) * l for l in ls] for ls in eigvals]


def krondecomposed_logdet(eigvals, deltas, damping: bool = False):
    """utils/matrix.py:381-404."""
    deltas = _expand_deltas(deltas, len(eigvals), eigvals[0][0].dtype)
    total = torch.zeros((), dtype=eigvals[0][0].dtype)
    for ls, d in zip(eigvals, deltas):
        total = total + torch.log(_block_eigvals(ls, d, damping)).sum()
    return total


def krondecomposed_to_matrix(eigvecs, eigvals, deltas, exponent: float = 1.0, damping: bool = False):
    """utils/matrix.py:524-556."""
    deltas = _expand_deltas(deltas, len(eigvals), eigvals[0][0].dtype)
    blocks = []
    for Qs, ls, d in zip(eigvecs, eigvals, deltas):
        lam = _block_eigvals(ls, d, damping) ** exponent
        Q = Qs[0] if len(Qs) == 1 else kron_product(Qs[0], Qs[1])
        blocks.append(Q @ torch.diag(lam.reshape(-1)) @ Q.T)
    return torch.block_diag(*blocks)


def krondecomposed_bmm(eigvecs, eigvals, deltas, W: torch.Tensor, exponent: float = -1.0, damping: bool = False):
    """``(K + delta)^exponent @ W`` row-wise for ``W[B, K, P]`` (utils/matrix.py:406-456),
    evaluated through the dense matrix (definition-level check of the structured form)."""
    M = krondecomposed_to_matrix(eigvecs, eigvals, deltas, exponent, damping)
    return W @ M.T


def krondecomposed_inv_square_form(eigvecs, eigvals, deltas, W: torch.Tensor, damping: bool = False):
    """``W P^{-1} W^T`` per batch item (utils/matrix.py:458-461)."""
    SW = krondecomposed_bmm(eigvecs, eigvals, deltas, W, -1.0, damping)
    return torch.bmm(W, SW.transpose(1, 2))


def krondecomposed_diag(eigvecs, eigvals, deltas, exponent: float = 1.0, damping: bool = False):
    """utils/matrix.py:490-522."""
    return krondecomposed_to_matrix(eigvecs, eigvals, deltas, exponent, damping).diagonal()


# --------------------------------------------------------------------------------------
# GLM predictive variances (laplace/baselaplace.py)
# --------------------------------------------------------------------------------------
def functional_variance_diag(Js: torch.Tensor, post_var: torch.Tensor):
    """``einsum('ncp,p,nkp->nck')`` (baselaplace.py:2113-2115)."""
    return torch.einsum("ncp,p,nkp->nck", Js, post_var, Js)


def functional_variance_full(Js: torch.Tensor, post_cov: torch.Tensor):
    """``einsum('ncp,pq,nkq->nck')`` (baselaplace.py:1683-1684)."""
    return torch.einsum("ncp,pq,nkq->nck", Js, post_cov, Js)


def functional_variance_kron(Js, eigvecs, eigvals, prior_prec, h_factor: float = 1.0, damping: bool = False):
    """``posterior_precision.inv_square_form(Js)`` with ``P = H * h + delta``
    (baselaplace.py:1812-1835; utils/matrix.py:342-376,458-461)."""
    ev = krondecomposed_scale(eigvals, h_factor)
    return krondecomposed_inv_square_form(eigvecs, ev, prior_prec, Js, damping)


def posterior_covariance_full(H: torch.Tensor, prior_prec_diag: torch.Tensor, h_factor: float = 1.0):
    """``(h*H + diag(prior))^{-1}`` (baselaplace.py:1634-1673; utils/utils.py:118-129 computes it
    through a Cholesky-based scale; the inverse is the same matrix)."""
    Pm = h_factor * H + torch.diag(prior_prec_diag)
    return torch.linalg.inv(Pm)


def krondecomposed_inv_square_form_blocks(eigvecs, eigvals, deltas, W: torch.Tensor, damping: bool = False):
    """``W P^{-1} W^T`` per batch item WITHOUT the dense ``P x P`` matrix — what the reference itself evaluates
    (utils/matrix.py:406-461): per Kronecker block the slice ``W_p [B*K, p_in, p_out]`` is rotated into the
    eigenbasis (``Q1^T W_p Q2``), weighted by ``(l1 (x) l2 + delta)^-1`` and contracted with the rotated slice
    again (rotating back and contracting with ``W_p`` is the same number because ``Q1``, ``Q2`` are orthogonal).
    Needed where ``krondecomposed_inv_square_form``'s dense route cannot be held (ResNet-18: P = 11.2 M).
    Device-agnostic: runs where ``W`` lives."""
    B, K, P = W.shape
    deltas = _expand_deltas(deltas, len(eigvals), W.dtype).to(W.device)
    out = torch.zeros(B, K, K, dtype=W.dtype, device=W.device)
    off = 0
    for Qs, ls, d in zip(eigvecs, eigvals, deltas):
        lam = _block_eigvals(ls, d, damping)
        if len(ls) == 1:
            p = ls[0].numel()
            R = W[:, :, off:off + p] @ Qs[0]                       # [B, K, p]
            out += torch.einsum("bkp,p,bcp->bkc", R, 1.0 / lam, R)
        else:
            p_in, p_out = ls[0].numel(), ls[1].numel()
            p = p_in * p_out
            Wp = W[:, :, off:off + p].reshape(B * K, p_in, p_out)
            R = (Qs[0].T @ Wp @ Qs[1]).reshape(B, K, p)
            out += torch.einsum("bkp,p,bcp->bkc", R, 1.0 / lam.reshape(-1), R)
        off += p
    assert off == P
    return out
