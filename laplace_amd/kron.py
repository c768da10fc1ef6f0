"""``Kron`` / ``KronDecomposed`` with the hot operations on the HIP kernels.

``HipKron`` implements the full contract of ``laplace.utils.matrix.Kron``
(laplace/utils/matrix.py:16-279) and ``HipKronDecomposed`` that of ``KronDecomposed``
(:282-560).  Both subclass the reference's classes when laplace-torch is importable
(:mod:`laplace_amd.refapi`), so they survive the reference's own
``self.H += H_batch`` (laplace/baselaplace.py:985), ``H_facs.decompose()`` (:1809),
``H * _H_factor + prior_precision`` (:1820) and ``state_dict`` (:1867-1879) unchanged:
``kfacs`` stay plain, mutable torch tensors.

Hot paths on HIP: ``decompose`` (batched symmetric eigensolver ``lk_syevj_f32``), ``logdet``
(``lk_kron_logdet_f32`` with analytic derivatives wrapped in an autograd Function so the
marginal-likelihood optimisation of baselaplace.py:466-485 keeps working).  The generic
``_bmm`` on a materialised ``[B, K, P]`` operand is plain GEMM plumbing (torch.matmul →
rocBLAS); the structure-exploiting predictive that avoids materialising the Jacobian lives in
:mod:`laplace_amd.predictive`.
"""
from __future__ import annotations

from math import pow
from typing import Iterable

import numpy as np
import torch
from torch import nn

from laplace_amd._lib import get_kernels
from laplace_amd.refapi import Kron as _KronBase
from laplace_amd.refapi import KronDecomposed as _KronDecomposedBase


def _is_valid_scalar(s) -> bool:
    if np.isscalar(s) and np.isreal(s):
        return True
    if torch.is_tensor(s) and s.ndim <= 1:
        return not (s.ndim == 1 and len(s) != 1)
    return False


def _kron2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return torch.einsum("ij,kl->ikjl", a, b).reshape(a.shape[0] * b.shape[0], a.shape[1] * b.shape[1])


_STREAM_POOL: dict = {}


def _side_streams(dev, n: int):
    """A small persistent pool of HIP streams per device (created once, reused by every decompose)."""
    pool = _STREAM_POOL.setdefault(dev.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(dev))
    return pool[:n]


class HipKron(_KronBase):
    """Kronecker-factored curvature: ``kfacs[i]`` is ``[G, A]`` (weight) or ``[B]`` (bias)."""

    def __init__(self, kfacs, pending=None):
        #: ``pending``: a :class:`laplace_amd.backend.KronAccumulator` holding minibatches in the accumulator's raw form
        #: (upper triangles, native column order, deferred scales).  ``backend.kron`` hands its minibatch over like this,
        #: ``+=`` merges raw forms with one multi-tensor add, and the symmetrise / permute into the public layout
        #: happens ONCE, the first time ``kfacs`` is read (decompose, state_dict, any of the Kron algebra) — so the
        #: reference's literal loop ``self.H += backend.kron(X, y, N)`` (baselaplace.py:969-985) does per minibatch what
        #: the fused accumulator does, plus one add.
        self._kfacs = kfacs
        self._pending = pending

    @property
    def kfacs(self):
        if self._pending is not None:
            self._materialize()
        return self._kfacs

    @kfacs.setter
    def kfacs(self, value):
        self._kfacs = value
        self._pending = None

    def __getstate__(self):  # pickling / deepcopy see the public layout only (the raw form holds streams and the backend)
        return {"kfacs": self.kfacs}

    def __setstate__(self, state):
        self._kfacs, self._pending = state["kfacs"], None

    def _materialize(self):
        acc, self._pending = self._pending, None
        fresh = acc.finalize()[1]._kfacs
        if self._kfacs is None:
            self._kfacs = fresh
        else:
            for Fi, Fj in zip(self._kfacs, fresh):
                for Hi, Hj in zip(Fi, Fj):
                    Hi.add_(Hj)

    def _absorb(self, other: "HipKron") -> bool:
        """fold ``other``'s raw minibatches into this object's without touching ``other``; False if ``other`` has none"""
        acc = other._pending
        if acc is None:
            return False
        if self._pending is None:
            self._pending = acc.clone()
        elif not self._pending.merge_(acc):  # different structure (cannot happen for one model): go through the layout
            return False
        return True

    @classmethod
    def init_from_model(cls, model: nn.Module | Iterable[nn.Parameter], device, dtype) -> "HipKron":
        """Zero factors shaped after the parameters (matrix.py:33-77)."""
        params = model.parameters() if isinstance(model, nn.Module) else model
        kfacs = []
        for p in params:
            if p.ndim == 1:
                kfacs.append([torch.zeros(p.size(0), p.size(0), device=device, dtype=dtype)])
            elif 2 <= p.ndim <= 4:
                d_out, d_in = p.shape[0], int(np.prod(p.shape[1:]))
                kfacs.append([torch.zeros(d_out, d_out, device=device, dtype=dtype),
                              torch.zeros(d_in, d_in, device=device, dtype=dtype)])
            else:
                raise ValueError("Invalid parameter shape in network.")
        return cls(kfacs)

    # -- accumulation (matrix.py:79-118) ---------------------------------------------------------
    def __add__(self, other):
        if not isinstance(other, _KronBase) or not (isinstance(other, HipKron) or hasattr(other, "kfacs")):
            raise ValueError("Can only add Kron to Kron.")
        if self._pending is not None or getattr(other, "_pending", None) is not None:
            # keep raw minibatches raw: raw parts are merged (copies: neither operand changes), public-layout parts summed
            out, ok = HipKron(None), True
            for part in (self, other):
                pend, plain = (part._pending, part._kfacs) if isinstance(part, HipKron) else (None, part.kfacs)
                if pend is not None:
                    if out._pending is None:
                        out._pending = pend.clone()
                    elif not out._pending.merge_(pend):
                        ok = False
                        break
                if plain is not None:
                    if out._kfacs is None:
                        out._kfacs = [[Hi.clone() for Hi in F] for F in plain]
                    else:
                        for Fi, Fj in zip(out._kfacs, plain):
                            for Hi, Hj in zip(Fi, Fj):
                                Hi.add_(Hj)
            if ok:
                return out
        return HipKron([[Hi.add(Hj) for Hi, Hj in zip(Fi, Fj)] for Fi, Fj in zip(self.kfacs, other.kfacs)])

    __radd__ = __add__

    def __iadd__(self, other):
        if not isinstance(other, _KronBase) or not (isinstance(other, HipKron) or hasattr(other, "kfacs")):
            raise ValueError("Can only add Kron to Kron.")
        if isinstance(other, HipKron) and other._pending is not None and self._absorb(other):
            if other._kfacs is not None:  # (a lazily summed Kron that also carries a public-layout part)
                if self._kfacs is None:
                    self._kfacs = [[Hi.clone() for Hi in F] for F in other._kfacs]
                else:
                    for Fi, Fj in zip(self._kfacs, other._kfacs):
                        for Hi, Hj in zip(Fi, Fj):
                            Hi.add_(Hj)
            return self
        for Fi, Fj in zip(self.kfacs, other.kfacs):
            for Hi, Hj in zip(Fi, Fj):
                Hi.add_(Hj)
        return self

    def __mul__(self, scalar):
        if not _is_valid_scalar(scalar):
            raise ValueError("Input not valid python or torch scalar.")
        return HipKron([[pow(scalar, 1 / len(F)) * Hi for Hi in F] for F in self.kfacs])

    __rmul__ = __mul__

    def __len__(self) -> int:
        return len(self.kfacs)

    # -- eigendecomposition (matrix.py:123-150; utils/utils.py:193-228) -----------------------------
    @staticmethod
    def shard_factors(sizes: list[int], world: int) -> list[int]:
        """Owner rank of every dense factor: longest-processing-time-first on the eigensolver's ~n^3 cost.
        Deterministic, so every rank computes the same assignment without communicating."""
        load = [0.0] * world
        owner = [0] * len(sizes)
        for i in sorted(range(len(sizes)), key=lambda i: (-sizes[i], i)):
            r = min(range(world), key=lambda r: (load[r], r))
            owner[i] = r
            load[r] += float(sizes[i]) ** 3
        return owner

    def decompose(self, damping: bool = False, n_streams: int = 3, process_group=None,
                  distributed: bool = False) -> "HipKronDecomposed":
        """Eigendecompose every dense factor with the HIP block-Jacobi solver.

        The solver of one matrix is a long chain of small launches (latency-bound pivot solves), so all factors
        iterate together: every launch serves the current round of every matrix still running (largest first;
        ``lk_syevj_batched_f32``), in two lanes on two side streams (the pivot solves of one lane beside the tile
        updates of the other), with the final refinement of finished matrices on a third (``n_streams`` = 3; 2 = one
        lane, 1 = everything on one stream); the calling stream waits for all of them before returning.

        ``distributed=True`` (every rank of ``process_group`` holds the SAME factors, i.e. after the fit's
        all-reduce): the factors are sharded over the ranks (:meth:`shard_factors`), each rank solves its
        share, and one all-reduce of a packed buffer — owners contribute their eigenpairs, everybody else
        zeros — hands every rank the full decomposition.  The reference has no counterpart (single process).
        """
        K = get_kernels()
        dense = [(Hi.shape[0], bi, fi) for bi, F in enumerate(self.kfacs) for fi, Hi in enumerate(F) if Hi.ndim > 1]
        dense.sort(reverse=True)
        import torch.distributed as dist

        world = dist.get_world_size(process_group) if (distributed and dist.is_available() and dist.is_initialized()) else 1
        mine = dense
        if world > 1:
            rank = dist.get_rank(process_group)
            owner = self.shard_factors([n for n, _, _ in dense], world)
            mine = [d for d, o in zip(dense, owner) if o == rank]
        all_dense, dense = dense, mine
        results = {}
        infos = []
        if dense:
            # (the solver is fp32: factors in another floating dtype — an fp64 model, see backend._twin — are solved in fp32
            #  and their eigenpairs handed back in their own dtype)
            f_dtype = self.kfacs[dense[0][1]][dense[0][2]].dtype
            mats = [self.kfacs[bi][fi].to(torch.float32).contiguous() for _, bi, fi in dense]
            streams = None
            if mats[0].is_cuda and n_streams > 1 and len(mats) > 1:
                dev = mats[0].device
                main = torch.cuda.current_stream(dev)
                streams = _side_streams(dev, min(n_streams, len(mats)))
            # one call for all factors: the native scheduler interleaves the solves over the streams
            solved = K.syevj_batched(mats, clamp=True, streams=streams)
            for (_, bi, fi), (l, Q, info) in zip(dense, solved):
                results[(bi, fi)] = (l.to(f_dtype), Q.to(f_dtype))
                infos.append(info)
            if streams is not None:
                for st in streams:
                    main.wait_stream(st)
        if world > 1 and all_dense:
            # exchange: every owner packs [eigenvalues | eigenvectors | status] of ITS factors into one buffer and
            # broadcasts it; a rank receives exactly the bytes it does not have (an all-gather with unequal shares),
            # nothing is zero-padded or summed
            ref_t = self.kfacs[all_dense[0][1]][all_dense[0][2]]
            by_owner: dict[int, list] = {r: [] for r in range(world)}
            for d, o in zip(all_dense, owner):
                by_owner[o].append(d)
            bufs = {}
            for r in range(world):
                total = sum(n + n * n + 1 for n, _, _ in by_owner[r])
                bufs[r] = torch.empty(total, dtype=ref_t.dtype, device=ref_t.device)
            off = 0
            info_of = {(bi, fi): info for (_, bi, fi), info in zip(dense, infos)}
            for n, bi, fi in by_owner[rank]:
                l, Q = results[(bi, fi)]
                bufs[rank][off:off + n] = l
                bufs[rank][off + n:off + n + n * n] = Q.reshape(-1)
                bufs[rank][off + n + n * n] = info_of[(bi, fi)][0].to(ref_t.dtype)
                off += n + n * n + 1
            works = []
            for r in range(world):
                if bufs[r].numel():
                    src = dist.get_global_rank(process_group, r) if process_group is not None else r
                    works.append(dist.broadcast(bufs[r], src=src, group=process_group, async_op=True))
            for w in works:
                w.wait()
            results, infos = {}, []
            slot = {}
            for r in range(world):
                off = 0
                for n, bi, fi in by_owner[r]:
                    slot[(bi, fi)] = (r, off)
                    off += n + n * n + 1
            for n, bi, fi in all_dense:
                r, off = slot[(bi, fi)]
                flat = bufs[r]
                results[(bi, fi)] = (flat[off:off + n], flat[off + n:off + n + n * n].view(n, n))
                infos.append(flat[off + n + n * n:off + n + n * n + 1])
        eigvecs, eigvals = [], []
        for bi, F in enumerate(self.kfacs):
            Qs, ls = [], []
            for fi, Hi in enumerate(F):
                if Hi.ndim > 1:
                    l, Q = results[(bi, fi)]
                else:  # diagonal factor
                    l, Q = Hi, torch.eye(len(Hi), dtype=Hi.dtype, device=Hi.device)
                Qs.append(Q)
                ls.append(l)
            eigvecs.append(Qs)
            eigvals.append(ls)
        out = HipKronDecomposed(eigvecs, eigvals, damping=damping)
        out._eig_info = infos  # device flags: read (one synchronisation) at the first use of the decomposition
        out._conv = {"checked": False, "source": self, "slots": [(bi, fi) for _, bi, fi in all_dense]}
        return out

    # -- generic algebra: the differentiable torch route behind the HIP kernels (same results as utils/matrix.py:152-279) -----
    # Everything below is phrased over BLOCK VIEWS of the flattened parameter axis: block b of the Kronecker matrix acts on
    # the columns [off, off + size) of W, seen as a [rows, d_1, ..., d_k] tensor whose axis i is contracted with factor i
    # (a vector stands for a diagonal factor).
    def _block_views(self):
        """(column offset, column count, factors) of every block along the flattened parameter axis"""
        off = 0
        for Fs in self.kfacs:
            size = 1
            for F_ in Fs:
                size *= len(F_)
            yield off, size, Fs
            off += size

    @staticmethod
    def _contract_axis(X: torch.Tensor, F_: torch.Tensor, axis: int) -> torch.Tensor:
        """``Y[..., i, ...] = sum_j F[i, j] X[..., j, ...]`` along ``axis`` (a 1-D ``F`` is a diagonal: a scaling)"""
        if F_.ndim == 1:
            shape = [1] * X.ndim
            shape[axis] = -1
            return X * F_.reshape(shape)
        return torch.movedim(torch.tensordot(X, F_, dims=([axis], [1])), -1, axis)

    def _bmm(self, W: torch.Tensor) -> torch.Tensor:
        assert W.ndim == 3
        rows = W.reshape(-1, W.shape[-1])
        out = torch.empty_like(rows)
        for off, size, Fs in self._block_views():
            X = rows[:, off:off + size].reshape(rows.shape[0], *[len(F_) for F_ in Fs])
            for i, F_ in enumerate(Fs):
                X = self._contract_axis(X, F_, i + 1)
            out[:, off:off + size] = X.reshape(rows.shape[0], size)
        return out.reshape(W.shape)

    def bmm(self, W: torch.Tensor, exponent: float = 1) -> torch.Tensor:
        if exponent != 1:
            raise ValueError("Only supported after decomposition.")
        if W.ndim not in (1, 2, 3):
            raise ValueError("Invalid shape for W")
        lead = {1: (1, 1), 2: (W.shape[0], 1), 3: tuple(W.shape[:2])}[W.ndim]
        res = self._bmm(W.reshape(*lead, W.shape[-1]))
        return res if W.ndim == 3 else res.squeeze()

    @staticmethod
    def _factor_logdet(F_: torch.Tensor) -> torch.Tensor:
        return F_.logdet() if F_.ndim > 1 else F_.log().sum()

    def logdet(self) -> torch.Tensor:
        # log det (F_1 (x) ... (x) F_k) = sum_i (size / d_i) log det F_i
        total = 0
        for _, size, Fs in self._block_views():
            for F_ in Fs:
                total = total + (size // len(F_)) * self._factor_logdet(F_)
        return total

    def diag(self) -> torch.Tensor:
        parts = []
        for _, _, Fs in self._block_views():
            d = None
            for F_ in Fs:
                fd = F_.diagonal() if F_.ndim > 1 else F_
                d = fd if d is None else (d.unsqueeze(-1) * fd).reshape(-1)
            parts.append(d)
        return torch.cat(parts)

    def to_matrix(self) -> torch.Tensor:
        blocks = []
        for _, _, Fs in self._block_views():
            M = None
            for F_ in Fs:
                F2 = F_ if F_.ndim > 1 else F_.diag()
                M = F2 if M is None else _kron2(M, F2)
            blocks.append(M)
        return torch.block_diag(*blocks)


class _KronLogdet(torch.autograd.Function):
    """sum_ij log(l1_i l2_j + delta) on HIP, differentiable in l1, l2, delta."""

    @staticmethod
    def forward(ctx, l1, l2, delta):
        need = any(t is not None and t.requires_grad for t in (l1, l2, delta))
        d = delta.detach().reshape(1).to(torch.float32).contiguous()
        val, d1, d2, dd = get_kernels().kron_logdet(
            l1.detach().contiguous(), None if l2 is None else l2.detach().contiguous(), d, False, want_grads=need
        )
        ctx.has_l2 = l2 is not None
        ctx.delta_shape = delta.shape
        ctx.save_for_backward(*(t for t in (d1, d2, dd) if t is not None))
        ctx.need = need
        return val.reshape(())

    @staticmethod
    def backward(ctx, grad):
        if not ctx.need:
            return None, None, None
        saved = list(ctx.saved_tensors)
        d1 = saved.pop(0)
        d2 = saved.pop(0) if ctx.has_l2 else None
        dd = saved.pop(0)
        return grad * d1, (grad * d2 if d2 is not None else None), (grad * dd).reshape(ctx.delta_shape)


class _KronLogdetBlocks(torch.autograd.Function):
    """``sum_b sum_ij log(s * l1_i l2_j + delta_b)`` over all blocks of a :class:`HipKronDecomposed` in one
    ``lk_kron_logdet_blocks_f32`` call; differentiable in the per-block deltas (the prior precision)."""

    @staticmethod
    def forward(ctx, deltas, post):
        need = deltas.requires_grad
        d = deltas.detach().to(torch.float32).contiguous()
        s = post._scale
        if s is not None:
            s = (s if torch.is_tensor(s) else torch.tensor(s)).to(device=d.device, dtype=torch.float32).reshape(1)
        blocks = [tuple(l.contiguous() for l in ls) for ls in post._base_eigenvalues]
        val, dd, _ = get_kernels().kron_logdet_blocks(blocks, d, s, want_grads=need)
        ctx.need = need
        ctx.dtype = deltas.dtype
        if need:
            ctx.save_for_backward(dd)
        return val.reshape(()).to(deltas.dtype)

    @staticmethod
    def backward(ctx, grad):
        if not ctx.need:
            return None, None
        (dd,) = ctx.saved_tensors
        return (grad * dd).to(ctx.dtype), None


class HipKronDecomposed(_KronDecomposedBase):
    """Eigendecomposed Kronecker factors + per-block additive ``deltas`` (matrix.py:282-560)."""

    #: ``False`` evaluates ``logdet`` block by block (``lk_kron_logdet_f32``; A/B switch for tools/marglik_bench.py)
    fused_logdet = True

    def __init__(self, eigenvectors, eigenvalues, deltas: torch.Tensor | None = None, damping: bool = False):
        self._conv = {"checked": True}  # shared by everything derived from one decomposition (`_like`)
        self._eig_info = []
        self.eigenvectors = eigenvectors
        self.eigenvalues = eigenvalues
        device, dtype = eigenvectors[0][0].device, eigenvectors[0][0].dtype
        if deltas is None:
            self.deltas = torch.zeros(len(self), device=device, dtype=dtype)
        else:
            self._check_deltas(deltas)
            self.deltas = deltas
        self.damping = damping

    # ``H * scalar`` (matrix.py:366-376) is kept as a pending scalar on the eigenvalue PRODUCT of every block: the
    # fused logdet takes it as a kernel argument, everything else sees the materialised ``scalar^(1/len) * l``.
    @property
    def _base_eigenvalues(self):
        if not self._conv["checked"]:
            self._ensure_converged()
        return self._base_ev

    @property
    def eigenvalues(self):
        if self._scale is None:
            return self._base_eigenvalues
        if self._scaled is None:
            s = self._scale
            self._scaled = [[(s ** (1 / len(ls)) if torch.is_tensor(s) else pow(s, 1 / len(ls))) * l for l in ls]
                            for ls in self._base_eigenvalues]
        return self._scaled

    @eigenvalues.setter
    def eigenvalues(self, value):
        self._base_ev = value
        self._scale = None
        self._scaled = None

    def check_converged(self) -> None:
        """Raise (never ``exit()``, cf. utils/utils.py:208-222) if an eigensolve ran out of sweeps, after the
        reference's own remedy has been tried.  Synchronises with the device once."""
        self._conv["checked"] = False
        self._ensure_converged()

    def _ensure_converged(self) -> None:
        """The solver's status words are read at the FIRST use of the decomposition (one host read of a packed flag
        vector).  A factor that ran out of sweeps gets the reference's treatment (utils/utils.py:208-222): solve
        ``M + I`` instead, subtract 1 from the eigenvalues; if that fails too: RuntimeError — the reference's prior
        gridsearch catches RuntimeError (baselaplace.py:545-551); it never exits the process."""
        conv = self._conv
        conv["checked"] = True
        infos = self._eig_info
        if not infos:
            return
        flags = torch.stack([i.reshape(-1)[0].to(torch.float32) for i in infos]).cpu()
        bad = [k for k in range(len(infos)) if float(flags[k]) != 0.0]
        if not bad:
            return
        src, slots = conv.get("source"), conv.get("slots")
        if src is None or slots is None:
            raise RuntimeError("lk_syevj_f32: eigendecomposition did not converge")
        K = get_kernels()
        for k in bad:
            bi, fi = slots[k]
            M = src.kfacs[bi][fi]
            Mj = (M + torch.eye(M.shape[0], device=M.device, dtype=M.dtype)).contiguous()
            (l, Q, info), = K.syevj_batched([Mj], clamp=False)
            if int(info.reshape(-1)[0].item()) != 0:
                raise RuntimeError(f"lk_syevj_f32: eigendecomposition of a {M.shape[0]} x {M.shape[0]} factor did not "
                                   "converge, also not with jitter")
            l = torch.nan_to_num((l - 1.0).clamp(min=0.0))
            self._base_ev[bi][fi].copy_(l)
            self.eigenvectors[bi][fi].copy_(torch.nan_to_num(Q))
            infos[k].zero_()

    def detach(self):
        self.deltas = self.deltas.detach()
        return self

    def _check_deltas(self, deltas) -> None:
        if not isinstance(deltas, torch.Tensor):
            raise ValueError("Can only add torch.Tensor to KronDecomposed.")
        if deltas.ndim == 0 or (deltas.ndim == 1 and (len(deltas) == 1 or len(deltas) == len(self))):
            return
        raise ValueError("Invalid shape of delta added to KronDecomposed.")

    def _like(self, deltas, scale=None):
        out = HipKronDecomposed(self.eigenvectors, self._base_ev, deltas, self.damping)
        out._scale = scale
        out._eig_info = self._eig_info
        out._conv = self._conv
        return out

    def __add__(self, deltas: torch.Tensor):
        self._check_deltas(deltas)
        return self._like(self.deltas + deltas, self._scale)

    def __mul__(self, scalar):
        if not _is_valid_scalar(scalar):
            raise ValueError("Invalid argument, can only multiply Kron with scalar.")
        # the reference's math.pow (matrix.py:373) takes the VALUE of a tensor scalar: no gradient flows through it
        s = scalar.detach().reshape(()) if torch.is_tensor(scalar) else float(scalar)
        return self._like(self.deltas, s if self._scale is None else self._scale * s)

    __radd__ = __add__
    __rmul__ = __mul__

    def __len__(self) -> int:
        return len(self.eigenvalues)

    # -- logdet (matrix.py:381-404) on HIP ----------------------------------------------------------
    def logdet(self) -> torch.Tensor:
        base = self._base_eigenvalues
        fused = (self.fused_logdet and not self.damping and self.deltas.ndim == 1 and len(self.deltas) == len(base)
                 and all(len(ls) in (1, 2) for ls in base)
                 and all(l.dtype == torch.float32 and l.ndim == 1 and not l.requires_grad for ls in base for l in ls))
        if fused:  # every block in one pass (three launches), differentiable in the deltas
            return _KronLogdetBlocks.apply(self.deltas, self)
        total = 0
        for ls, delta in zip(self.eigenvalues, self.deltas):
            if any(l.dtype != torch.float32 for l in ls):  # (fp64 / fp16 posteriors: the formula itself, matrix.py:381-404)
                if len(ls) == 1:
                    total = total + torch.log(ls[0] + delta).sum()
                elif self.damping:
                    sd = torch.sqrt(delta)
                    total = total + torch.log(torch.outer(ls[0] + sd, ls[1] + sd)).sum()
                else:
                    total = total + torch.log(torch.outer(ls[0], ls[1]) + delta).sum()
            elif len(ls) == 1:
                total = total + _KronLogdet.apply(ls[0], None, delta)
            elif len(ls) == 2:
                l1, l2 = ls
                if self.damping:
                    sd = torch.sqrt(delta)
                    total = total + torch.log(torch.outer(l1 + sd, l2 + sd)).sum()
                else:
                    total = total + _KronLogdet.apply(l1, l2, delta)
            else:
                raise ValueError("Too many Kronecker factors. Something went wrong.")
        return total

    # -- generic (materialised-operand) algebra: GEMM plumbing ---------------------------------------
    def _block_pow(self, ls, delta, exponent):
        if len(ls) == 1:
            return torch.pow(ls[0] + delta, exponent)
        l1, l2 = ls
        if self.damping:
            sd = torch.sqrt(delta)
            return torch.pow(torch.outer(l1 + sd, l2 + sd), exponent)
        return torch.pow(torch.outer(l1, l2) + delta, exponent)

    def _bmm(self, W: torch.Tensor, exponent: float = -1) -> torch.Tensor:
        """``self ** exponent @ W`` row-wise (matrix.py:406-456) on ``lk_gemm_f32``: per Kronecker block the slice of
        every row is rotated into the eigenbasis, weighted by ``(l1 (x) l2 + delta) ** exponent`` in the epilogue of the
        second product and rotated back, operands addressed in place.  Differentiable operands (marginal-likelihood
        training through the predictive) and non-fp32 / CPU tensors take the same algebra through torch."""
        assert W.ndim == 3
        B, K_, P = W.shape
        K = get_kernels()
        if hasattr(K, "kron_sandwich") and W.dtype == torch.float32 and not (
                torch.is_grad_enabled() and (W.requires_grad or self.deltas.requires_grad)):
            Wc = W.reshape(B * K_, P).contiguous()
            out = torch.empty_like(Wc)
            cur = 0
            for ls, Qs, delta in zip(self.eigenvalues, self.eigenvectors, self.deltas):
                if len(ls) == 1:
                    lam = K.kron_pow(ls[0].contiguous(), None, delta, exponent)
                    K.kron_sandwich(Wc, cur, P, B * K_, Qs[0].contiguous(), None, lam, out)
                    cur += len(ls[0])
                else:
                    lam = K.kron_pow(ls[0].contiguous(), ls[1].contiguous(), delta, exponent, self.damping)
                    K.kron_sandwich(Wc, cur, P, B * K_, Qs[0].contiguous(), Qs[1].contiguous(), lam, out)
                    cur += len(ls[0]) * len(ls[1])
            return out.reshape(B, K_, P)
        # differentiable torch route (eigenvalues / prior with requires_grad, CPU tensors): in the eigenbasis of block b
        # the operator is diagonal, so  S W_b = Q ((Q^T W_b) . lam) Q^T  with Q = Q_1 (x) Q_2 applied factor by factor
        rows = W.reshape(B * K_, P)
        out = torch.empty_like(rows)
        off = 0
        for ls, Qs, delta in zip(self.eigenvalues, self.eigenvectors, self.deltas):
            dims = [len(l) for l in ls]
            size = 1
            for d in dims:
                size *= d
            lam = self._block_pow(ls, delta, exponent).reshape(1, *dims)
            X = rows[:, off:off + size].reshape(rows.shape[0], *dims)
            for i, Q in enumerate(Qs):                       # into the eigenbasis: contract axis i with Q^T
                X = torch.movedim(torch.tensordot(X, Q, dims=([i + 1], [0])), -1, i + 1)
            X = X * lam
            for i, Q in enumerate(Qs):                       # and back: contract axis i with Q
                X = torch.movedim(torch.tensordot(X, Q, dims=([i + 1], [1])), -1, i + 1)
            out[:, off:off + size] = X.reshape(rows.shape[0], size)
            off += size
        return out.reshape(B, K_, P)

    def inv_square_form(self, W: torch.Tensor) -> torch.Tensor:
        SW = self._bmm(W, exponent=-1)
        return torch.bmm(W, SW.transpose(1, 2))

    def bmm(self, W: torch.Tensor, exponent: float = -1) -> torch.Tensor:
        if W.ndim not in (1, 2, 3):
            raise ValueError("Invalid shape for W")
        lead = {1: (1, 1), 2: (W.shape[0], 1), 3: tuple(W.shape[:2])}[W.ndim]
        res = self._bmm(W.reshape(*lead, W.shape[-1]), exponent)
        return res if W.ndim == 3 else res.squeeze()

    def diag(self, exponent: float = 1) -> torch.Tensor:
        # diag(Q diag(lam) Q^T)_p = sum_e Q[p, e]^2 lam[e]; for Q = Q_1 (x) Q_2 the squares factorise as well
        parts = []
        for Qs, ls, delta in zip(self.eigenvectors, self.eigenvalues, self.deltas):
            lam = self._block_pow(ls, delta, exponent)
            if len(ls) == 1:
                parts.append(torch.einsum("pe,e->p", Qs[0].square(), lam.reshape(-1)))
            else:
                parts.append(torch.einsum("ae,ef,bf->ab", Qs[0].square(), lam, Qs[1].square()).reshape(-1))
        return torch.cat(parts)

    def to_matrix(self, exponent: float = 1) -> torch.Tensor:
        blocks = []
        for Qs, ls, delta in zip(self.eigenvectors, self.eigenvalues, self.deltas):
            lam = self._block_pow(ls, delta, exponent).reshape(-1)
            Q = Qs[0] if len(ls) == 1 else _kron2(Qs[0], Qs[1])
            blocks.append(torch.einsum("pe,e,qe->pq", Q, lam, Q))
        return torch.block_diag(*blocks)
