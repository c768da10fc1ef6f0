"""``HipGGN`` / ``HipEF`` — the MI355X-native curvature backend (drop-in ``CurvatureInterface``).

Usage with the unmodified reference::

    from laplace import Laplace
    from laplace_amd import HipGGN
    la = Laplace(model, "classification", "all", "kron", backend=HipGGN)
    la.fit(train_loader)

Boundary (laplace/baselaplace.py:179-194): the class is instantiated lazily as
``cls(model, likelihood, dict_key_x=..., dict_key_y=..., **backend_kwargs)``; Laplace then calls
``kron(X, y, N=N, **kw)`` (:1770), ``diag`` (:2069), ``full`` (:1623), ``jacobians`` (:1327) and
``last_layer_jacobians`` (laplace/lllaplace.py:219-231).  Class names deliberately avoid the
substrings ``backpack`` / ``asdl`` / ``asdfghjkl`` that baselaplace.py:142-149,943-952,1313-1325
sniff for.

What runs where: for convolutional networks the whole minibatch step runs on our own kernels — the forward (implicit-GEMM
convolution on split-fp16 operands, fused BatchNorm / add / activation) and the seed-batched reverse sweep on NHWC split
tensors (:mod:`laplace_amd.sweep_nhwc`: backward-data with the element-wise VJP in its epilogue, all seeds in one pass);
graphs that path has no rule for take the NCHW sweep of :mod:`laplace_amd.sweep` (library convolutions, our element-wise
kernels), models ``torch.fx`` cannot trace the autograd tape of :mod:`laplace_amd.capture`.  Everything named in
BASELINE.json's north_star — likelihood-Hessian root + loss, the VJPs of the sweep, A/G factor accumulation, diagonal /
dense GGN, per-sample Jacobian assembly — is a HIP entry point of ``include/laplace_hip.h`` (through
:mod:`laplace_amd._lib`).  Only fp32 models on a ROCm device are accepted; there is no CPU path.
"""
from __future__ import annotations

import collections
import math
import threading
import os
from collections.abc import MutableMapping

import torch
import torch.nn.functional as F
from torch import nn

from laplace_amd._lib import SplitTensor, get_kernels, is_channels_last, keep_layout
from laplace_amd.capture import Tape
from laplace_amd.sweep import SeedBatchedSweep, SweepUnsupported
from laplace_amd.sweep_nhwc import SplitSweep
from laplace_amd.kron import HipKron
from laplace_amd.refapi import EFInterface, GGNInterface


#: ``False``: the Kron predictive's eigenbasis rotation of the unfolded inputs stays a library convolution
_OWN_ROTATION = True


def shared_operands(tap, g, B, C, Q1=None, Q2=None, bounds=None):
    """``u [B, C, Do, L]`` and ``v [B, Dk, L]`` (both position-contiguous) of a weight-sharing layer (Conv2d, or Linear
    along a sequence), whose per-sample Jacobian of output / seed ``c`` is ``sum_l u[n, c, :, l] v[n, :, l]^T``
    (:mod:`laplace_amd.predictive`), rotated into the eigenbases ``Q1`` / ``Q2`` if given; plus the position-summed
    output gradient ``[C, B, Do]`` for the bias.  ``bounds``: a dict that receives the FORM the operands come back in —
    ``planes``: both are SplitTensors (lk_conv_nhwc_f16x2_planes: ``u [C * B, Do, L]`` seed-major with one scale, ``v [B, Dk, L]``
    with one scale per sample), ``u_seed_major``: fp32 with ``u [C, B, Do, L]``."""
    m = tap.module
    a = tap.a.to(torch.float32)
    if tap.kind == "conv2d":
        from laplace_amd import conv as cv

        K = get_kernels()
        Do = m.out_channels
        Dk = m.weight[0].numel()
        rotated = False
        own_v = (Q2 is not None and _OWN_ROTATION and hasattr(K, "conv_nhwc_f16x2") and cv._geometry_ok(m)
                 and m.in_channels % 32 == 0 and Dk % 8 == 0)
        if (isinstance(g, SplitTensor) and Q1 is not None and bounds is not None and not tap.has_bias and _OWN_ROTATION and own_v
                and getattr(K, "use_quad_planes", False) and Do % 32 == 0 and (g.shape[1] * g.shape[2]) % 16 == 0
                and C <= K.quadform_shared_max_outputs
                # (the planes kernel indexes u with 32 bits: a predictive batch beyond that takes the chunked fp32 route below
                #  instead of failing behind both rotations)
                and C * B * g.shape[1] * g.shape[2] * Do < (1 << 31)):
            # both rotations on our convolution kernel with SPLIT, position-contiguous outputs (lk_conv_nhwc_f16x2_planes):
            # u = Q1^T g seed-major [C * B, Do, L] with one scale, v = the unfolded inputs in Q2's basis [B, Dk, L] with one
            # scale per sample — what lk_kron_quadform_shared_planes_f16x2 stages without splitting anything
            u = cv.rotate_channels(g, Q1, Q1, planes=True)
            filt = Q2.T.reshape(Dk, *m.weight.shape[1:])
            v = cv.conv_forward_filters(m, a, filt, Q2, xs=getattr(tap, "a_split", None), planes=True)
            bounds["planes"] = True
            return u, v, None
        if (isinstance(g, SplitTensor) and Q1 is not None and bounds is not None and not tap.has_bias and _OWN_ROTATION
                and Do % 32 == 0 and (g.shape[1] * g.shape[2]) % 4 == 0 and C <= K.quadform_shared_max_outputs
                and hasattr(K, "conv_nhwc_f16x2")):
            # rotation into the G factor's eigenbasis as a 1x1 convolution over the split cotangent, position-contiguous
            # output: u stays SEED-major [C, B, Do, L] (the quadratic-form kernel takes it as it is)
            L = g.shape[1] * g.shape[2]
            u = cv.rotate_channels(g, Q1, Q1).reshape(C, B, Do, L)
            gsum = None
            bounds["u_seed_major"] = True
            rotated = True
        elif isinstance(g, SplitTensor):    # NHWC split cotangent [C*B, H, W, Do] straight from the sweep: one pass
            u = K.unsplit_transpose(g, C, B)                           # [B, C, Do, L]
            L = u.shape[-1]
            gsum = u.sum(-1).permute(1, 0, 2) if tap.has_bias else None  # [C, B, Do]: only a bias block reads it
        else:
            g4 = g.reshape(C, B, Do, -1)                               # [C, B, Do, L]
            L = g4.shape[-1]
            gsum = g4.sum(-1)
            u = g4.permute(1, 0, 2, 3)                                 # [B, C, Do, L]
        if Q2 is None:
            v = F.unfold(a, m.kernel_size, m.dilation, m.padding, m.stride)          # [B, Dk, L]
        else:
            # unfolded patches (x) Q2 = one convolution whose filters are the eigenvectors (rows of the A factor
            # follow F.unfold's (c_in, kh, kw) order = the weight layout); its position-contiguous output IS [B, Dk, L]
            filt = Q2.T.reshape(Dk, *m.weight.shape[1:])
            if (_OWN_ROTATION and hasattr(K, "conv_nhwc_f16x2") and cv._geometry_ok(m) and m.in_channels % 32 == 0 and Dk % 8 == 0
                    and L % 4 == 0):
                # our implicit-GEMM convolution (fp32-level products on the fp16 matrix cores), eigenvector filters
                # kept as split planes per decomposition
                v = cv.conv_forward_filters(m, a, filt, Q2, xs=getattr(tap, "a_split", None)).reshape(B, Dk, L)
            elif Dk <= 32:
                # a very thin layer (the 3 x 3 RGB stem: Dk = 27): unfold + ONE small batched GEMM.  The library convolution
                # runs this shape as im2col + GEMM PER IMAGE (256 launches of ~6 us for a minibatch of 128: 1.5 ms of a
                # 30 ms call); wider thin layers (LeNet's 5 x 5 stem, Dk = 75) are faster through the library (measured)
                v = torch.matmul(Q2.T, F.unfold(a, m.kernel_size, m.dilation, m.padding, m.stride))
            else:
                v = F.conv2d(a, filt, None, m.stride, m.padding, m.dilation).reshape(B, Dk, L)
        if Q1 is not None and not rotated:
            u = torch.matmul(Q1.T, u)
    else:                                                              # Linear over [B, ..., Di]
        Do = m.out_features
        v = a.reshape(B, -1, a.shape[-1])
        L = v.shape[1]
        gt = g.reshape(C, B, L, Do)
        gsum = gt.sum(2)
        u = gt.permute(1, 0, 2, 3)                                     # [B, C, L, Do]
        if Q2 is not None:
            v = v @ Q2
        if Q1 is not None:
            u = u @ Q1
        v = v.transpose(1, 2)
        u = u.transpose(2, 3)
    return u.contiguous(), v.contiguous(), gsum


class CachedFeatures:
    """Head input ``phi [B, D]`` and output ``f [B, C]`` of one batch; accepted wherever a last-layer backend takes ``x``."""

    __slots__ = ("f", "phi")

    def __init__(self, f: torch.Tensor, phi: torch.Tensor):
        self.f, self.phi = f, phi

    def __len__(self):
        return self.f.shape[0]


class _HipCurvatureMixin:
    """Shared machinery of :class:`HipGGN` and :class:`HipEF`."""

    # ---- forward / taps -------------------------------------------------------------------------
    def _tape(self) -> Tape:
        tape = getattr(self, "_tape_cache", None)
        if tape is None or tape.model is not self._model or tape.n_params != sum(p.numel() for p in self.params):
            tape = Tape(self._model, self.params)
            self._tape_cache = tape
        return tape

    def _split_sweep_state(self):
        """Does the split-fp16 NHWC sweep (one power-of-two scale per tensor) serve this model's minibatches?  ``True`` /
        ``False`` once a forward pass has built the sweep, ``None`` before that."""
        tape = getattr(self, "_tape_cache", None)
        sweep = getattr(tape, "sweep", None) if tape is not None else None
        if sweep is None:
            return None
        return bool(isinstance(sweep, SplitSweep) and sweep.split_reason is None)

    def _check_dtype(self, f: torch.Tensor):
        if f.dtype != torch.float32:
            raise TypeError(f"HIP curvature backend computes in float32; model output is {f.dtype}")

    # ---- models in another floating dtype ---------------------------------------------------------------------------
    # The reference computes in the model's dtype and its tests run fp64 / fp16 models through every backend
    # (tests/test_baselaplace.py:895-934: H, marginal likelihood and predictive keep the dtype).  The kernels here are
    # fp32 (split-fp16 products, fp32 accumulation — DESIGN section 3a), so a model in another floating dtype is served by
    # an fp32 TWIN of this backend on an fp32 copy of the model (re-synchronised whenever a parameter or buffer changes):
    # inputs go in as fp32, every result comes back in the model's dtype.  fp64 containers, fp32 accuracy — said in
    # INTEGRATION.md; round 3 raised TypeError here.
    def _twin(self):
        """``(fp32 twin backend, model dtype)`` for a model that is not fp32, else ``(None, torch.float32)``"""
        p0 = next((p for p in self.model.parameters() if p.is_floating_point()), None)
        dt = p0.dtype if p0 is not None else torch.float32
        if dt == torch.float32 or getattr(self, "_is_twin", False):
            return None, torch.float32
        ts = list(self.model.parameters()) + list(self.model.buffers())
        sig = tuple((t.data_ptr(), t._version) for t in ts)
        cur = self.__dict__.get("_twin_state")
        if cur is None:
            import copy

            m32 = copy.deepcopy(self.model).float()
            twin = object.__new__(type(self))
            twin.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_tape_cache", "_twin_state")})
            twin._is_twin = True
            twin.model = m32
            twin.params = [p for p in twin._model.parameters() if p.requires_grad]
            twin.params_dict = {k: v for k, v in twin._model.named_parameters() if v.requires_grad}
            twin.buffers_dict = dict(twin.model.named_buffers())
            cur = self.__dict__["_twin_state"] = [sig, twin]
        elif cur[0] != sig:
            with torch.no_grad():
                for a, b in zip(list(cur[1].model.parameters()) + list(cur[1].model.buffers()), ts):
                    a.copy_(b)  # (casts)
            cur[0] = sig
        twin = cur[1]
        for k in ("use_sweep", "use_split_sweep", "generator"):
            if k in self.__dict__:
                setattr(twin, k, self.__dict__[k])
        return twin, dt

    @staticmethod
    def _to32(x):
        if torch.is_tensor(x):
            return x.float() if x.is_floating_point() and x.dtype != torch.float32 else x
        if isinstance(x, (dict, MutableMapping)):  # dict-style inputs (HuggingFace BatchEncoding, UserDict): same container type
            # (a NEW container built from a plain dict: `copy.copy` of a mapping without `__copy__` shares its inner storage,
            #  and assigning into the copy rewrote the caller's batch in place)
            out = {k: _HipCurvatureMixin._to32(x[k]) for k in list(x.keys())}
            if type(x) is dict:
                return out
            try:
                return type(x)(out)
            except Exception:
                return out
        if isinstance(x, tuple) and hasattr(x, "_fields"):  # namedtuple: positional constructor
            return type(x)(*(_HipCurvatureMixin._to32(v) for v in x))
        if isinstance(x, (tuple, list)):
            return type(x)(_HipCurvatureMixin._to32(v) for v in x)
        return x

    @staticmethod
    def _cast(out, dt):
        if torch.is_tensor(out):
            return out.to(dt) if out.is_floating_point() else out
        if isinstance(out, HipKron):
            return HipKron([[t.to(dt) for t in F] for F in out.kfacs])
        if isinstance(out, tuple):
            return tuple(_HipCurvatureMixin._cast(o, dt) for o in out)
        return out

    def _forward(self, x, keep_tap_splits: bool = False):
        """Returns (f [B,C] detached, tape, grad_fn) where grad_fn(seeds[S,B,C]) -> per-tap [S,B,...].
        ``keep_tap_splits``: the NHWC sweep also keeps the split copies of the tapped inputs (``tap.a_split``)."""
        tape = self._tape()
        if self.last_layer:
            # f = last_layer(phi): the gradient w.r.t. the head's output IS the seed -> no reverse pass
            swept = (x.f, x.phi) if isinstance(x, CachedFeatures) else self._features_swept(x, tape)
            if swept is not None:
                f, phi = swept
            else:
                with torch.no_grad():
                    f, phi = self.model.forward_with_features(x)
            if len(tape.taps) != 1 or tape.taps[0].kind != "linear":
                raise NotImplementedError("last-layer mode needs an nn.Linear head")
            tape.taps[0].a = phi.detach()
            self._check_dtype(f)
            B = phi.shape[0]
            f = f.detach().reshape(B, -1).contiguous()
            return f, tape, lambda seeds, stack=True: [seeds.contiguous()]
        swept = self._forward_swept(x, tape, keep_tap_splits)
        if swept is not None:
            return swept
        f = tape.forward(x)
        self._check_dtype(f)
        if f.ndim == 1:
            f = f.unsqueeze(-1)
        if f.ndim != 2:
            raise NotImplementedError(f"model output must be [batch, outputs]; got {tuple(f.shape)}")
        return (f.detach().contiguous(), tape,
                lambda seeds, stack=True, f_graph=f: tape.output_grads(f_graph, seeds, stack=stack))

    #: ``False`` forces the autograd tape (one reverse pass per seed).  (Path selectors are plain attributes — of the class
    #: for a process-wide default, of an object for one backend; nothing here reads the environment.)
    use_sweep = True
    #: ``False`` keeps the reverse sweep on NCHW fp32 cotangents and the library's backward-data
    use_split_sweep = True
    #: largest batch (seeds x samples) of one reverse sweep and the memory its cotangents may take (4 live tensors of
    #: the largest activation are assumed); more seeds are processed in chunks
    sweep_max_rows = 8192
    sweep_mem_bytes = 16 << 30

    def cache_features(self, x) -> "CachedFeatures":
        """One feature pass of a last-layer flavour, kept for repeated predictives on the same batch (the prior
        gridsearch evaluates 100 posteriors on every validation batch; the backbone output does not depend on them)."""
        if not self.last_layer:
            raise NotImplementedError("feature caching is for the last-layer flavours")
        f, tape, _ = self._forward(x)
        phi = tape.taps[0].a
        tape.taps[0].a = None
        return CachedFeatures(f, phi)

    def _features_swept(self, x, tape):
        """Feature pass of the last-layer flavours through the sweep's own forward (fused eval-BatchNorm / residual /
        ReLU kernels) when the wrapped model is fx-traceable; ``None`` -> ``forward_with_features`` of the extractor."""
        inner = getattr(self.model, "model", None)
        name = getattr(self.model, "_last_layer_name", None)
        if (not self.use_sweep or not torch.is_tensor(x) or not isinstance(inner, nn.Module) or name is None
                or not x.is_floating_point()):
            return None
        fs = getattr(tape, "feature_sweep", None)
        if fs is None:
            try:
                fs = SeedBatchedSweep(inner, {name: dict(inner.named_modules())[name]}, kernels=get_kernels)
            except (SweepUnsupported, KeyError):
                fs = False
            tape.feature_sweep = fs
        if fs is False:
            return None
        try:
            f = fs.forward(x, need_vjp=False)
        except SweepUnsupported:
            return None
        phi = fs.taps[name]["a"]
        fs.release()
        return f, phi

    def _forward_swept(self, x, tape, keep_tap_splits: bool = False):
        """Seed-batched reverse sweep (laplace_amd/sweep.py) when the model is fx-traceable and built from
        modules with a closed-form VJP; ``None`` -> caller uses the autograd tape."""
        if not self.use_sweep or not torch.is_tensor(x) or not tape.taps:
            return None
        sweep = getattr(tape, "sweep", None)
        if sweep is None:
            try:
                # NHWC split-fp16 sweep (own convolution kernels) where the graph allows it, else the NCHW sweep
                cls = SplitSweep if self.use_split_sweep else SeedBatchedSweep
                sweep = cls(self._model, {t.name: t.module for t in tape.taps}, kernels=get_kernels)
            except SweepUnsupported as e:
                sweep = False
                tape.sweep_reason = str(e)
            tape.sweep = sweep
        if sweep is False:
            return None
        try:
            if isinstance(sweep, SplitSweep):
                sweep.act_sink = getattr(self, "_act_sink", None)
            f = sweep.forward(x, keep_tap_splits=True) if keep_tap_splits and isinstance(sweep, SplitSweep) else sweep.forward(x)
        except SweepUnsupported as e:  # e.g. the model is in training mode for this call
            tape.sweep_reason = str(e)
            return None
        finally:
            if isinstance(sweep, SplitSweep):
                sweep.act_sink = None
        self._check_dtype(f)
        if f.ndim == 1:
            f = f.unsqueeze(-1)
        if f.ndim != 2:
            raise NotImplementedError(f"model output must be [batch, outputs]; got {tuple(f.shape)}")
        for t in tape.taps:
            t.a = sweep.taps[t.name]["a"]
            t.a_split = getattr(sweep, "tap_splits", {}).get(t.name)  # NHWC SplitTensor of the same activation, if any

        def grad_fn(seeds, stack=True, on_tap=None, defer_bn_scale=False, keep_split=False):
            """All seeds in one sweep while ``S*B`` stays below ``sweep_max_rows`` images; many-output models
            (C = 1000 -> 999 seeds) go through in seed chunks of that size.  With ``on_tap`` every chunk's gradients
            are handed over layer by layer (additive consumers such as the KFAC accumulator) and nothing is returned."""
            seeds = seeds.reshape(seeds.shape[0], seeds.shape[1], *sweep.out_shape)
            S, B = seeds.shape[0], seeds.shape[1]
            # a few cotangents of the largest activation are alive at once: keep them inside the memory budget
            rows = min(int(self.sweep_max_rows), int(self.sweep_mem_bytes) // (16 * max(sweep.max_act_numel, 1)))
            chunk = max(1, rows // max(B, 1))
            if S <= chunk:
                if keep_split and isinstance(sweep, SplitSweep):  # NHWC SplitTensors for consumers that take them
                    grads = sweep.backward(seeds, on_tap=on_tap, defer_bn_scale=defer_bn_scale, keep_split=True)
                else:
                    grads = sweep.backward(seeds, on_tap=on_tap, defer_bn_scale=defer_bn_scale)
                return [grads[t.name] for t in tape.taps]
            parts = []
            for s0 in range(0, S, chunk):
                grads = sweep.backward(seeds[s0:s0 + chunk].contiguous(), on_tap=on_tap, defer_bn_scale=defer_bn_scale)
                if on_tap is None:
                    parts.append([grads[t.name] for t in tape.taps])
            if on_tap is not None:
                return None
            return [torch.cat([p[i] for p in parts]) for i in range(len(tape.taps))]

        grad_fn.streams_taps = True  # accepts on_tap: gradients are delivered layer by layer
        grad_fn.accepts_keep_split = True  # can hand conv-tap gradients back as NHWC SplitTensors
        grad_fn.grad_scale = lambda: sweep.grad_scale  # name -> per-channel scale owed by the caller (deferred BN)

        return f.detach().reshape(f.shape[0], -1).contiguous(), tape, grad_fn

    # ---- seeds ----------------------------------------------------------------------------------
    def _mc_functional_grads(self, f):
        """``num_samples`` draws of the functional gradient of the log-likelihood at the model's own predictive,
        ``[S, B, C]`` (laplace/curvature/curvature.py:341-364): regression ``f - y~ = -eps``, ``y~ ~ N(f, 1)``;
        classification ``softmax(f) - onehot(y~)``, ``y~ ~ Cat(softmax(f))``.  Device RNG = torch's Philox stream
        (``self.generator`` if set)."""
        S = int(self.num_samples)
        gen = getattr(self, "generator", None)
        B, C = f.shape
        if self.likelihood == "regression":
            return -torch.randn(S, B, C, generator=gen, device=f.device, dtype=f.dtype)
        p = torch.softmax(f, dim=-1)
        idx = torch.multinomial(p, S, replacement=True, generator=gen)  # [B, S]
        g = p.unsqueeze(0).repeat(S, 1, 1)
        g.scatter_add_(2, idx.t().unsqueeze(-1), torch.full((S, B, 1), -1.0, device=f.device, dtype=f.dtype))
        return g

    def _mc_seeds(self, f, y, loss):
        """MC-Fisher seeds: ``sum_s seeds_s seeds_s^T = 1/S sum_s g_s g_s^T`` (the middle matrix of
        ``_get_mc_functional_fisher``).  KFAC follows curvlinops' ``FisherType.MC`` for ``MSELoss``: the sampled
        gradient of ``sum (f-y)^2`` has covariance ``2I``, hence the hessian_scale 2 as in the exact case."""
        K = get_kernels()
        B, C = f.shape
        if self.likelihood == "regression":
            if y is not None:
                K.sq_err_sum(f, y.reshape(B, C).to(torch.float32).contiguous(), self.factor, loss)
            hs = 2.0
        else:
            if y is not None:
                K.softmax_hess_sqrt(f, y.reshape(B).to(torch.int64).contiguous(), loss)  # CE loss only
            hs = 1.0
        g = self._mc_functional_grads(f)
        return (g / math.sqrt(g.shape[0])).contiguous(), hs

    def _ggn_seeds(self, f, y, loss):
        """Columns of a root of the loss Hessian w.r.t. f, laid out ``[C, B, C]``; accumulates
        ``factor * loss`` into ``loss``.  Returns (seeds, hessian_scale)."""
        if getattr(self, "stochastic", False):
            return self._mc_seeds(f, y, loss)
        K = get_kernels()
        B, C = f.shape
        if self.likelihood == "regression":
            if y is not None:
                K.sq_err_sum(f, y.reshape(B, C).to(torch.float32).contiguous(), self.factor, loss)
            eye = torch.eye(C, dtype=f.dtype, device=f.device)
            return eye[:, None, :].expand(C, B, C).contiguous(), 2.0  # d2/df2 MSELoss(sum) = 2 I
        yy = None if y is None else y.reshape(B).to(torch.int64).contiguous()
        # rank-revealing root: C-1 seeds (one reverse pass fewer); G / diag / full are root-invariant
        return K.softmax_hess_sqrt(f, yy, loss if y is not None else None, cholesky=True), 1.0

    def _ef_seed(self, f, y, loss):
        """Gradient of the (unscaled, summed) torch loss w.r.t. f, ``[1, B, C]``; accumulates
        ``factor * loss``."""
        K = get_kernels()
        B, C = f.shape
        if self.likelihood == "regression":
            yy = y.reshape(B, C).to(torch.float32).contiguous()
            K.sq_err_sum(f, yy, self.factor, loss)
            return (2.0 * (f - yy)).unsqueeze(0).contiguous()
        yy = y.reshape(B).to(torch.int64).contiguous()
        K.softmax_hess_sqrt(f, yy, loss)  # loss only; the root itself is not needed for the EF
        p = torch.softmax(f, dim=-1)
        p[torch.arange(B, device=f.device), yy] -= 1.0
        return p.unsqueeze(0).contiguous()

    # ---- per-layer building blocks ----------------------------------------------------------------
    @staticmethod
    def _positions(tap, a) -> int:
        if tap.kind == "conv2d":
            m = tap.module
            H, W = a.shape[-2:]
            oh = (H + 2 * m.padding[0] - m.dilation[0] * (m.kernel_size[0] - 1) - 1) // m.stride[0] + 1
            ow = (W + 2 * m.padding[1] - m.dilation[1] * (m.kernel_size[1] - 1) - 1) // m.stride[1] + 1
            return oh * ow
        return int(a[0].numel() // a.shape[-1])

    def _factor_shapes(self, tap):
        m = tap.module
        if tap.kind == "linear":
            return m.out_features, m.in_features
        return m.out_channels, m.in_channels * m.kernel_size[0] * m.kernel_size[1]

    def _factor_A(self, tap, N, alpha_a_scale, kfac_approx, A, fused=False):
        """``A += alpha/(N L) * sum a a^T`` (input side; needs only the forward activations)."""
        K = get_kernels()
        a = tap.a.to(torch.float32)
        m = tap.module
        L = self._positions(tap, a)
        if tap.kind == "linear":
            Di = m.in_features
            if kfac_approx == "expand" or L == 1:
                K.gram_tn(a.reshape(-1, Di).contiguous(), alpha_a_scale / (N * L), A, upper_only=fused)
            else:  # 'reduce': average inputs over the weight-sharing positions
                K.gram_tn(a.reshape(a.shape[0], L, Di).mean(1).contiguous(), alpha_a_scale / N, A, upper_only=fused)
            return A
        if kfac_approx == "expand":
            ak = keep_layout(a)
            n = A.shape[0]
            if (fused and getattr(K, "use_gram_conv16", False) and hasattr(K, "im2col_split") and m.kernel_size[0] * m.kernel_size[1] > 1
                    and tuple(m.dilation) == (1, 1) and m.stride[0] == m.stride[1] and m.padding[0] == m.padding[1]
                    and m.groups == 1 and isinstance(m.padding[0], int) and ak.is_cuda == A.is_cuda
                    and (n + 127) // 128 * 128 <= 4096 and ak.shape[0] * L * ((n + 127) // 128 * 128) < (1 << 33)):  # (the Gram engine's widest matrix)
                # the patch matrix as split planes (one pass: im2col + split), then the split-fp16 Gram engine — the strided and
                # stem convolutions of c4 took 170 - 190 us each on the exact-fp32 MFMA kernel, 67 - 70 + the pass this way.
                # The engine takes 64 or a multiple of 128 columns: other widths are zero padded and the block copied out.
                Kp = n if (n == 64 or n % 128 == 0) else (64 if n < 64 else (n + 127) // 128 * 128)
                pm = K.im2col_split(ak, m.kernel_size, m.stride[0], m.padding[0], Kp)
                if Kp == n:
                    K.gram_tn_f16x2(pm, alpha_a_scale / (N * L), A)
                else:
                    Ap = torch.zeros(Kp, Kp, dtype=torch.float32, device=A.device)
                    K.gram_tn_f16x2(pm, alpha_a_scale / (N * L), Ap)
                    A += Ap[:n, :n]  # (upper tiles of the padded Gram; the lower triangle of A is mirrored once per fit)
                return A
            K.gram_conv(ak, m.kernel_size, m.stride, m.padding, m.dilation, alpha_a_scale / (N * L), A,
                        upper_only=fused, native=fused)
        else:
            cols = torch.nn.functional.unfold(a, m.kernel_size, dilation=m.dilation, padding=m.padding, stride=m.stride)
            if fused:  # keep the accumulator's native column order: (kh, kw, ci)
                KK = m.kernel_size[0] * m.kernel_size[1]
                cols = cols.reshape(cols.shape[0], m.in_channels, KK, -1).transpose(1, 2).reshape(cols.shape)
            K.gram_tn(cols.mean(2).contiguous(), alpha_a_scale / N, A, upper_only=fused)
        return A

    def _factor_G(self, tap, g, alpha_g, kfac_approx, G, fused=False, persist=None):
        """``G += alpha * sum g g^T`` (output side; ``g`` is ``[S, B, ...]`` or the unstacked per-seed list)."""
        K = get_kernels()
        m = tap.module
        if isinstance(g, SplitTensor):  # NHWC split cotangent [S*B, H, W, Do] of the split-fp16 sweep
            Do = g.shape[-1]
            if kfac_approx == "expand" and (Do == 64 or Do % 128 == 0):
                K.gram_tn_f16x2(g, alpha_g, G)  # upper 32x32 tiles; mirrored by the caller (symmetrize)
                if not fused:
                    K.symmetrize(G)
                return G
            gf = g.float()
            rows = gf.reshape(-1, Do) if kfac_approx == "expand" else gf.sum((1, 2))
            K.gram_tn(rows.contiguous(), alpha_g, G, upper_only=fused)
            return G
        L = self._positions(tap, tap.a)
        if isinstance(g, (list, tuple)):  # conv tap, per-seed gradients left unstacked
            S, B = len(g), g[0].shape[0]
        else:
            S, B = g.shape[0], g.shape[1]
        if tap.kind == "linear":
            Do = m.out_features
            if kfac_approx == "expand" or L == 1:
                K.gram_tn(g.reshape(-1, Do).contiguous(), alpha_g, G, upper_only=fused)
            else:
                K.gram_tn(g.reshape(S, B, L, Do).sum(2).reshape(S * B, Do).contiguous(), alpha_g, G, upper_only=fused)
            return G
        Do = m.out_channels
        if kfac_approx == "expand":
            if isinstance(g, (list, tuple)):
                K.gram_nt([gs.reshape(B, Do, L) for gs in g], alpha_g, G, upper_only=fused)
            else:
                K.gram_nt(g.reshape(S * B, Do, L).contiguous(), alpha_g, G, upper_only=fused, persist=persist)
        else:
            if isinstance(g, (list, tuple)):
                g = torch.stack(g)
            K.gram_tn(g.reshape(S * B, Do, L).sum(2).contiguous(), alpha_g, G, upper_only=fused)
        return G

    @staticmethod
    def _conv_view(tap, a, g):
        """``(a, g, kernel_size, stride, padding, dilation)`` of a weight-sharing tap as a convolution: a Conv2d as it
        is; an ``nn.Linear`` applied along extra dims (``a [B, ..., Di]``, ``g [S, B, ..., Do]``) is the 1x1
        convolution over its ``T`` shared positions, whose ``[Do, Di, 1, 1]`` weight flattens like the Linear's."""
        m = tap.module
        if tap.kind == "conv2d":
            return a.contiguous(), g.contiguous(), m.kernel_size, m.stride, m.padding, m.dilation
        B, S = a.shape[0], g.shape[0]
        a4 = a.reshape(B, -1, a.shape[-1]).transpose(1, 2).unsqueeze(-1).contiguous()          # [B, Di, T, 1]
        g5 = g.reshape(S, B, -1, g.shape[-1]).transpose(2, 3).unsqueeze(-1).contiguous()       # [S, B, Do, T, 1]
        return a4, g5, (1, 1), (1, 1), (0, 0), (1, 1)

    def _layer_jacobian(self, tap, g, Js):
        """Writes this module's columns of ``Js[B, S, P]``; ``g`` is ``[S, B, ...]``."""
        K = get_kernels()
        a = tap.a.to(torch.float32)
        if tap.kind == "linear" and a.ndim == 2:
            K.jac_linear(a.contiguous(), g.contiguous(), Js, tap.w_off, tap.b_off)
        else:
            a4, g5, ks, st, pd, dl = self._conv_view(tap, a, g)
            K.jac_conv(a4, g5, ks, st, pd, dl, Js, tap.w_off, tap.b_off)

    def _rows(self, x, seeds_fn):
        """``Z[B, S, P]`` = seed-contracted per-sample Jacobians (all tracked params must be covered)."""
        f, tape, grad_fn = self._forward(x)
        if tape.uncovered:
            raise NotImplementedError("parameters outside nn.Linear / nn.Conv2d are not covered by the HIP kernels")
        seeds = seeds_fn(f)
        grads = grad_fn(seeds)
        B, S = f.shape[0], seeds.shape[0]
        Z = torch.zeros(B, S, tape.n_params, dtype=torch.float32, device=f.device)
        for tap, g in zip(tape.taps, grads):
            self._layer_jacobian(tap, g, Z)
        tape.release()
        return Z, f

    def _supported(self) -> bool:
        return not self._tape().uncovered

    # ---- shared implementations ------------------------------------------------------------------
    def _kron_impl(self, x, y, N, seeds_fn, hess_scale_fn, kfac_approx):
        """One minibatch's ``(loss, Kron)`` — what the reference's literal loop ``self.H += backend.kron(X, y, N)``
        (laplace/baselaplace.py:969-985) consumes.  Same kernel schedule as a one-batch :class:`KronAccumulator`
        (A factors on a side stream under the reverse sweep, G factors streamed layer by layer), followed by the
        per-batch symmetrise / permute into the reference's layout; the pixel-pair forms, which pay off only when
        their assembly is amortised over a whole fit, are left to :meth:`kron_accumulator`."""
        if kfac_approx not in ("expand", "reduce"):
            raise ValueError(f"kfac_approx must be 'expand' or 'reduce', got {kfac_approx!r}")
        acc = KronAccumulator(self, N, kfac_approx, overlap=True)
        # a lazily handed-over minibatch leaves the banded pixel-pair products of its 3x3 A factors to the running sum that
        # absorbs it (KronAccumulator.defer_pix); an eager one computes every factor here, the per-minibatch way
        acc.defer_pix = bool(self.lazy_kron and self.lazy_pixpair and acc.use_pixgram and kfac_approx == "expand")
        if not acc.defer_pix:
            acc.use_pixgram = False
        acc._persist_slabs = False
        acc.lanes = 1  # (one minibatch: nothing to run beside it)
        acc.coalesce = False  # (... and nothing to stack it with: its raw form is read right away)
        try:
            acc.add_batch(x, y)
        except NotImplementedError as e:
            if "KFAC supports" in str(e):
                raise NotImplementedError(
                    "KFAC supports nn.Linear / nn.Conv2d parameters only (as the reference, docs/index.md:364-366); "
                    "freeze the others (requires_grad=False)") from e
            raise
        if self.lazy_kron:
            # the minibatch stays in the accumulator's raw form inside the returned Kron: `H += ` merges raw forms, the
            # symmetrise / permute into the public layout runs once, when somebody reads `kfacs`
            return acc.loss[0].clone(), HipKron(None, pending=acc)
        return acc.finalize()

    #: ``False``: ``kron`` returns its minibatch already in the reference's layout
    lazy_kron = True
    #: ``False``: a lazily handed-over minibatch computes its 3x3 A factors itself
    lazy_pixpair = True

    def _diag_impl(self, x, y, seeds_fn, alpha):
        K = get_kernels()
        f, tape, grad_fn = self._forward(x)
        if tape.uncovered:
            return None
        loss = torch.zeros(1, dtype=torch.float32, device=f.device)
        seeds, _ = seeds_fn(f, y, loss)
        grads = grad_fn(seeds)
        h = torch.zeros(tape.n_params, dtype=torch.float32, device=f.device)
        B, S = f.shape[0], seeds.shape[0]
        for tap, g in zip(tape.taps, grads):
            a = tap.a.to(torch.float32)
            m = tap.module
            if tap.kind == "linear" and a.ndim == 2:
                n_w = m.out_features * m.in_features
                K.diag_ggn_linear(a.contiguous(), g.contiguous(), alpha, h[tap.w_off:tap.w_off + n_w],
                                  h[tap.b_off:tap.b_off + m.out_features] if tap.has_bias else None)
            else:
                # exact diagonal of a weight-sharing layer = squared per-sample weight Jacobian, summed over
                # (sample, seed)
                width = m.weight.numel()
                n_out = m.weight.shape[0]
                if n_out >= 32 and width // n_out >= 64:
                    # MFMA tile GEMM of the per-sample Jacobian, squares summed in registers
                    smax = K.quadform_shared_max_outputs
                    u, v, gsum = shared_operands(tap, g, B, S)
                    for s0 in range(0, S, smax):  # the sum over seeds is additive: any S goes through in chunks
                        us = u if S <= smax else u[:, s0:s0 + smax].contiguous()
                        K.diag_ggn_shared(us, v, alpha, h[tap.w_off:tap.w_off + width])
                    if tap.has_bias:
                        h[tap.b_off:tap.b_off + n_out] += alpha * (gsum * gsum).sum((0, 1))
                else:
                    # narrow layers (LeNet's 6- and 16-channel convs) would leave the 32-row MFMA tiles mostly empty:
                    # their per-sample Jacobian block is small, write it and square-sum its columns
                    Jl = torch.zeros(B, S, width + (n_out if tap.has_bias else 0), dtype=torch.float32, device=f.device)
                    a4, g5, ks, st, pd, dl = self._conv_view(tap, a, g)
                    K.jac_conv(a4, g5, ks, st, pd, dl, Jl, 0, width if tap.has_bias else -1)
                    K.sq_colsum(Jl, 0, width, alpha, h[tap.w_off:tap.w_off + width])
                    if tap.has_bias:
                        K.sq_colsum(Jl, width, n_out, alpha, h[tap.b_off:tap.b_off + n_out])
        tape.release()
        if self.subnetwork_indices is not None:
            h = h[self.subnetwork_indices]
        return loss[0], h

    def _full_from_rows(self, Z, alpha):
        K = get_kernels()
        Z2 = Z.reshape(-1, Z.shape[-1])
        if self.subnetwork_indices is not None:
            Z2 = Z2[:, self.subnetwork_indices]
        Z2 = Z2.contiguous()
        H = torch.zeros(Z2.shape[1], Z2.shape[1], dtype=torch.float32, device=Z.device)
        return K.gram_tn(Z2, alpha, H)


class CurvatureExchange(list):
    """What `KronAccumulator.tensors` hands to `allreduce_curvature`: the tensors of the exchange and the accumulator they
    belong to."""

    owner = None


_PROCESS_STREAMS: dict = {}
_PROCESS_STREAMS_LOCK = threading.Lock()  # (fits from different threads share the queues: work of both merely serialises)


def _process_streams(key, make):
    """The HIP streams of the fit (lanes, their side streams, the flush streams) exist ONCE per process and device, not once
    per backend object: PyTorch's caching allocator keeps a pool of freed blocks per stream, so every backend with streams
    of its own reserved another ~100 GB for the same c4 fit (measured, `tools/fuse_ab.py`) — a loop that builds a new
    `Laplace` object per epoch (marglik training) ran the device out of memory after two of them and the allocator into
    freeing and re-allocating every step (7 -> 49 ms).  Streams are only queues: accumulators that share them are ordered
    by the same waits / events as before, work of different fits on one stream merely serialises."""
    with _PROCESS_STREAMS_LOCK:
        return _process_streams_locked(key, make)


def _process_streams_locked(key, make):
    hit = _PROCESS_STREAMS.get(key)
    if hit is None:
        hit = _PROCESS_STREAMS[key] = make()
    return hit


class KronAccumulator:
    """Running KFAC factors of one ``fit`` kept in the kernels' own form.

    The reference accumulates with ``self.H += H_batch`` (laplace/baselaplace.py:985): a fresh set of
    factors per minibatch (376 MB for ResNet-18), mirrored and permuted to the public layout, then
    added.  Here every minibatch accumulates *in place* (``C += alpha X^T X`` is what the Gram kernels
    do anyway), touching only the upper block triangle and leaving conv A factors in the native
    (kh, kw, ci) order; ``finalize`` mirrors / permutes ONCE and hands back an ordinary
    :class:`HipKron` (same values as the sum of per-batch ``kron()`` results; covered by
    tests/test_laplace_e2e.py and tests/test_gpu_backend.py).

    Scheduling on the device (all of it invisible in the results beyond fp32 addition order): the A- and G-factor
    kernels of a minibatch run on a side stream under its reverse sweep and may run on into the next minibatch
    (``lag_join``); the banded pixel-pair products of 3x3 A factors are stacked over ``pix_group`` minibatches per launch;
    consecutive minibatches alternate between ``lanes`` sub-accumulators with streams of their own, folded when the fit is
    read.  ``overlap=False`` switches all of it off (the serial schedule the tests compare against).
    """

    # ---- path selectors (class attributes; an instance may override them before its first minibatch; each is exercised
    #      against the default path by tests/test_gpu_switches.py).  The one environment variable the package reads is
    #      LK_LIB (_lib.py: which build of the shared library to load — A/B builds of a development session).
    #: ``False``: A factors of 3x3 / stride-1 convolutions per minibatch instead of through the pixel-pair accumulators
    use_pixgram = True
    #: ``False``: the BatchNorm scale of a shortcut branch is applied to every minibatch's cotangent instead of once per fit
    _defer_bn = True
    #: ``False``: the split-K slabs of the NCHW route's G factors are reduced per launch instead of once per fit
    _persist_slabs = True
    #: minibatches stacked per pixel-pair launch: the kernel is bound by the read-modify-write of its blocks, which
    #: happens once per LAUNCH, so stacking the NHWC inputs of consecutive minibatches divides that traffic
    pix_group = 8
    #: ``False``: the main stream waits for the factor kernels at the end of every minibatch
    lag_join = True

    def __init__(self, backend, N: int, kfac_approx: str = "expand", overlap: bool = True):
        # a model in another floating dtype is served by the backend's fp32 twin (see `_twin`): minibatches go in as fp32,
        # the factors come back in the model's dtype — like `backend.kron`, which the reference's literal loop calls
        twin, dt = backend._twin() if hasattr(backend, "_twin") else (None, torch.float32)
        self._caller, self._out_dtype = (backend, dt) if twin is not None else (None, torch.float32)
        if twin is not None:
            backend = twin
        self.backend, self.N, self.kfac_approx = backend, N, kfac_approx
        self.overlap = overlap
        self._side = None
        self._side_done = None  # event at the end of the previous minibatch's side-stream work (lagged join)
        self._side_older = []   # ... of the minibatches before that which the main stream has not waited for yet
        #: minibatches whose factor kernels may still be running when the next one starts (measured with the lanes' streams
        #: at high priority: 2 or 3 gain 1 % in the steady state and lose 3 - 5 % on a 20-minibatch fit, whose tail grows)
        self.lag_depth = 1
        self.factors = None  # per tap: [G, A]
        self.loss = None
        self._taps_meta = None
        #: ``True`` (set by ``backend.kron`` for the minibatches of the reference's literal loop): the banded pixel-pair
        #: products of the 3x3 A factors are NOT computed here — the minibatch keeps its NHWC inputs (`_pix_inputs`) and
        #: the running sum that absorbs it (`merge_`) stacks them and runs the grouped kernel into ITS accumulators, as
        #: the fused accumulator does.  A minibatch nobody absorbs computes them the per-minibatch way when it is read.
        self.defer_pix = False
        self._pix_inputs = {}  # tap index -> list of (geometry, alpha, NHWC fp32 tensor [B, H, W, C], module)
        self._pix_early = set()  # taps whose pixel-pair blocks were folded into this accumulator's A factor mid-fit
        #: ``True`` (`default_coalesce`, or `kron_accumulator(N, coalesce=False)`): consecutive SMALL minibatches of a small model are stacked and swept
        #: together.  The curvature is a sum over samples with per-sample terms that do not depend on the minibatch they
        #: arrive in (curvlinops.py:77-108: G sums over samples, A carries 1/N with the GLOBAL N), so minibatch boundaries
        #: are the caller's choice, not part of the result — and a 151-parameter MLP at batch 100 (BASELINE config c1) is
        #: ~50 launches of a few microseconds each per minibatch, i.e. bound by the host's enqueue rate, not by the device
        #: ("batch into one launch").  Stacking keeps copies of the inputs (2 small launches per
        #: minibatch) and sweeps `coalesce_target` samples at a time; models whose minibatch fills the chip never stack.
        self.coalesce = bool(type(self).default_coalesce)
        self._stash, self._stash_n = [], 0
        self._dispatched = 0   # minibatches swept so far (the first one is never stacked)
        self._act_numel = 0    # largest per-sample activation of this model, measured by the first forward pass
        #: minibatches in flight on the device (`default_lanes`): with 2, consecutive minibatches go alternately to two
        #: sub-accumulators, each with its own stream (and side stream) and its own factor buffers, summed when the fit
        #: is read — the forward pass of one minibatch (small grids at batch 128) then runs beside the reverse sweep of
        #: the one before it.  The sum over minibatches is linear: same factors up to the order of fp32 additions.
        self.lanes = max(1, int(type(self).default_lanes))
        self._lane_accs, self._lane_next, self._lane_id, self._lane_stream = None, 0, 0, None
        self._lane_sig = None
        self._ahead = collections.deque()  # events behind the minibatches the host has enqueued and not waited for (`max_ahead`)
        self._a_done = None  # event on the side stream behind the A-side work of the latest minibatch
        self._lanes_anywhere = False  # (tests: the lanes' host logic on the CPU emulation of the kernels, without streams)

    def _alloc(self, tape, dev):
        """Zeroed factors of every tap, carved out of TWO flat buffers (one fill each instead of one per factor — a fit
        starts with ~90 fewer launches per lane): the factors that keep their storage when the fit is read (G, Linear A,
        1x1-conv A) and the conv A factors in the kernels' native column order, whose storage is dropped by `finalize`
        (it writes the permuted copies)."""
        self.factors, self._taps_meta = [], []
        shapes = []
        for tap in tape.taps:
            m = tap.module
            if tap.kind == "linear":
                do, di = m.out_features, m.in_features
                native = None
            else:
                do, di = m.out_channels, m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                native = (m.in_channels, m.kernel_size[0] * m.kernel_size[1])
            shapes.append((do, di, native is not None and native[1] > 1))
            self._taps_meta.append((tap.has_bias, native))
        pad = lambda n: (n + 63) // 64 * 64  # (256-byte aligned factors)
        kept = sum(pad(do * do) + (0 if nat else pad(di * di)) for do, di, nat in shapes) + 64
        natv = sum(pad(di * di) for do, di, nat in shapes if nat)
        flat_k = torch.zeros(kept, dtype=torch.float32, device=dev)
        flat_n = torch.zeros(natv, dtype=torch.float32, device=dev) if natv else None
        ok, on = 0, 0
        for do, di, nat in shapes:
            G = flat_k[ok:ok + do * do].view(do, do)
            ok += pad(do * do)
            if nat:
                A = flat_n[on:on + di * di].view(di, di)
                on += pad(di * di)
            else:
                A = flat_k[ok:ok + di * di].view(di, di)
                ok += pad(di * di)
            self.factors.append([G, A])
        self.loss = flat_k[ok:ok + 1]
        self._pix = {}  # tap index -> (geometry, buffer): pixel-pair accumulators of 3x3 convs
        self._pix_pending = {}  # tap index -> minibatches stacked for the next pixel-pair launch
        self._gscale = {}  # tap index -> deferred BatchNorm scale owed to the G accumulator
        self._gslabs = {}  # tap index -> (persistent split-K slabs, n, L, alpha) of a conv G factor
        self._tap_index = {tap.name: i for i, tap in enumerate(tape.taps)}

    def ensure_allocated(self, device):
        """Zero factors for a rank that saw no minibatch (empty shard of a data-parallel fit): shapes come from the
        modules alone, so the rank can still take part in the all-reduce."""
        self._fold_lanes()
        if self.factors is None:
            self._alloc(self.backend._tape(), device)

    def _pix_geometry(self, tap):
        """How this tap's A factor is accumulated over the fit (3x3 / stride 1 / pad 1 convs, ``expand``):

        * ``("pair", H, W, Cin, plan)``: banded pixel-pair blocks (any map size, ``Cin % 64 == 0``) —
          ``lk_conv3x3_pixpair_*``;
        * ``("dense", H, W, Cin)``: the full pixel-pair Gram of maps of at most ``pixgram_max_hw`` pixels —
          ``lk_gram_tn_f32`` + ``lk_conv3x3_pixgram_assemble_f32``;
        * ``None``: the per-minibatch kernels (shift correlation / implicit im2col).

        Both pixel-pair forms are linear in the data: one small product per minibatch, the 81 patch blocks are
        assembled once per fit."""
        m = tap.module
        if tap.kind != "conv2d" or self.kfac_approx != "expand" or not self.use_pixgram:
            return None
        if (tuple(m.kernel_size), tuple(m.stride), tuple(m.padding), tuple(m.dilation)) != ((3, 3), (1, 1), (1, 1), (1, 1)):
            return None
        K = get_kernels()
        H, W = (int(v) for v in tap.a.shape[-2:])
        Cin = int(m.in_channels)
        if H * W > K.pixgram_max_hw or Cin % 64 == 0:
            plan = K.pixpair_plan(H, W, Cin, tap.a.device)
            if plan is not None and plan[0] * Cin * Cin * 4 <= (1 << 30):
                return ("pair", H, W, Cin, plan)
        if H * W <= K.pixgram_max_hw and H * W * Cin <= 16384:
            return ("dense", H, W, Cin)
        return None

    def _accumulate_A(self, idx, tap, F, rt):
        K = get_kernels()
        if self.defer_pix:
            geo = self._pix_geometry(tap)
            if geo is not None and geo[0] == "pair" and tap.a.dtype == torch.float32:
                t = K.nchw_to_nhwc(keep_layout(tap.a))  # (a copy: the sweep releases the activation)
                self._pix_inputs.setdefault(idx, []).append((geo, rt / (self.N * geo[1] * geo[2]), t, tap.module))
                return
            self.backend._factor_A(tap, self.N, rt, self.kfac_approx, F[1], fused=True)
            return
        acc = self._pix.get(idx)
        if acc is None:
            self.backend._factor_A(tap, self.N, rt, self.kfac_approx, F[1], fused=True)
            return
        geo, buf = acc
        alpha = rt / (self.N * geo[1] * geo[2])
        a = keep_layout(tap.a.to(torch.float32))
        if geo[0] != "pair":
            K.pixgram_accumulate(a, alpha, buf)
        elif self.pix_group <= 1:
            K.pixpair_accumulate(a, alpha, buf, geo[4])
        else:
            self._push_pix_input(idx, geo, alpha, a, nhwc=False)

    #: ``True``: the forward pass writes the input activation of a 3x3 convolution straight into its slot of the pixel-pair stack
    #: (`SplitSweep.act_sink`) instead of into a tensor of its own that is then copied there: 13 copies of an activation per c4
    #: minibatch less (0.2 ms of kernel time per step).  The stack is then measured when it is split (one pass per group).
    direct_stack = True

    def _pix_sink(self, name, shape, device):
        """(forward pass of the next minibatch, calling stream) the slot of tap ``name``'s stack that :meth:`_push_pix_input` will
        fill next, if the activation of that shape belongs there; else None"""
        idx = self._tap_index.get(name)
        pend = self._pix_pending.get(idx) if idx is not None else None
        if pend is None or pend["n"] >= self.pix_group:
            return None
        B = pend["B"]
        stack = pend["stack"]
        if tuple(shape) != (B,) + tuple(stack.shape[1:]) or stack.device != device:
            return None
        cur = torch.cuda.current_stream(device)
        free = pend.pop("free", None)
        if free is not None:
            cur.wait_event(free)  # (the split of the previous group has read the stack)
        stack.record_stream(cur)  # (allocated on the side stream, written here)
        return stack[pend["n"] * B:(pend["n"] + 1) * B]

    def _push_pix_input(self, idx, geo, alpha, a, nhwc: bool):
        """NHWC copy of one minibatch into its slot of the group buffer; launch when the group is full.  ``a``: the
        activation (logical NCHW) or, ``nhwc=True``, an NHWC fp32 tensor a deferred minibatch kept"""
        K = get_kernels()
        B = a.shape[0]
        pend = self._pix_pending.get(idx)
        if pend is not None and (pend["B"] != B or pend["alpha"] != alpha):
            self._drain_pixpair(idx)  # ragged last batch / changed scale: flush what is stacked
            pend = None
        if pend is None:
            stack = torch.empty(self.pix_group * B, geo[1], geo[2], geo[3], dtype=torch.float32, device=a.device)
            # (the planes the stacked images are split into, allocated WITH the stack: a fit shorter than one group would
            # otherwise leave the first full-size request to the middle of the next fit)
            planes = (torch.empty((2,) + tuple(stack.shape), dtype=torch.float16, device=a.device)
                      if getattr(K, "use_pixpair16", False) and stack.numel() % 8 == 0 else None)
            # (max|.| of what is stacked, measured by the copies into the stack: the split then needs no pass of its own)
            word = torch.zeros(1, dtype=torch.float32, device=a.device) if planes is not None and getattr(K, "use_copy_absmax", False) else None
            pend = self._pix_pending[idx] = {"B": B, "alpha": alpha, "stack": stack, "n": 0, "planes": planes, "amax": word}
        slot = pend["stack"][pend["n"] * B:(pend["n"] + 1) * B]
        src = a if nhwc else (a.permute(0, 2, 3, 1) if K.is_channels_last(a) else None)
        if src is not None and src.data_ptr() == slot.data_ptr() and tuple(src.shape) == tuple(slot.shape) and src.is_contiguous():
            pend["amax"] = None  # (the forward pass wrote it here: `_pix_sink`; this group is measured when it is split)
        elif (pend["amax"] is not None and src is not None and src.is_contiguous() and src.dtype == torch.float32
                and src.numel() % 4 == 0 and src.data_ptr() % 16 == 0 and slot.data_ptr() % 16 == 0):
            K.copy_absmax(src, slot, pend["amax"])
        else:
            pend["amax"] = None  # (this group is measured when it is split)
            if nhwc:
                slot.copy_(a)
            else:
                K.nchw_to_nhwc(a, out=slot)
        pend["n"] += 1
        if pend["n"] == self.pix_group:
            self._drain_pixpair(idx, keep=True)

    def _adopt_pix_inputs(self, other):
        """(merge_, calling stream) take over the NHWC inputs a deferred minibatch kept: once ``pix_group`` of them have
        come together for a tap, this accumulator allocates that tap's pixel-pair blocks and from then on stacks /
        launches like the fused accumulator; until then they stay deferred here as well.  Returns what
        :meth:`_push_adopted` has to stack (possibly on the side stream)"""
        todo = []
        for idx, items in other._pix_inputs.items():
            if idx not in self._pix:
                mine = self._pix_inputs.setdefault(idx, [])
                mine.extend(items)
                if len(mine) < max(self.pix_group, 2):
                    continue
                geo = mine[0][0]
                self._pix[idx] = (geo, torch.zeros(geo[4][0] * geo[3] * geo[3], dtype=torch.float32, device=mine[0][2].device))
                items = self._pix_inputs.pop(idx)
            todo.append((idx, items))
        return todo

    def _push_adopted(self, todo):
        for idx, items in todo:
            for geo, alpha, t, _ in items:
                self._push_pix_input(idx, geo, alpha, t, nhwc=True)

    def _resolve_pix_inputs(self):
        """(finalize) deferred inputs that never met a running sum: the per-minibatch A-factor kernel on each"""
        if not self._pix_inputs:
            return
        K = get_kernels()
        self._join_side()
        for idx, items in self._pix_inputs.items():
            for _, alpha, t, m in items:  # (alpha = sqrt(factor) / (N L), L = H W for these convs)
                K.gram_conv(t.permute(0, 3, 1, 2), m.kernel_size, m.stride, m.padding, m.dilation, alpha, self.factors[idx][1],
                            upper_only=True, native=True)
        self._pix_inputs = {}

    def _drain_pixpair(self, idx, keep=False):
        """launch the pixel-pair product over the minibatches stacked so far"""
        pend = self._pix_pending.get(idx)
        if pend is None:
            return
        if pend["n"]:
            geo, buf = self._pix[idx]
            xh = pend["stack"][:pend["n"] * pend["B"]]
            cur = torch.cuda.current_stream(xh.device) if xh.is_cuda else None
            K = get_kernels()
            if getattr(K, "use_pixpair16", False) and xh.numel() % 8 == 0:
                # one split of the stacked images (scale from their measured max), then the fp16 MFMA kernel
                ws = pend.get("planes")
                xs = K.split_f16x2(xh, amax=pend.get("amax"), out=None if ws is None else ws[:, :xh.shape[0]])
                if cur is not None and keep:  # (the stack has been read: the next group's first slot may be written — `_pix_sink`)
                    pend["free"] = torch.cuda.Event()
                    pend["free"].record(cur)
                K.pixpair_accumulate_split(xs, pend["alpha"], buf, geo[4])
            else:
                K.pixpair_accumulate_nhwc(xh, pend["alpha"], buf, geo[4])
                if cur is not None and keep:
                    pend["free"] = torch.cuda.Event()
                    pend["free"].record(cur)
            if cur is not None:
                pend["stack"].record_stream(cur)  # filled on the side stream, possibly consumed on another one
                if pend.get("planes") is not None:
                    pend["planes"].record_stream(cur)
                if pend.get("amax") is not None:
                    pend["amax"].record_stream(cur)
        if keep:
            pend["n"] = 0
            if pend.get("planes") is not None and getattr(get_kernels(), "use_copy_absmax", False):
                pend["amax"] = torch.zeros(1, dtype=torch.float32, device=pend["stack"].device)  # the next group's word
        else:
            del self._pix_pending[idx]

    def _ensure_pixgrams(self, tape):
        """allocate the pixel-pair accumulators on the CALLING stream (they are consumed there at the end of the fit); the
        first minibatch of a fit carves all of them out of one zeroed buffer (one fill instead of one per layer)"""
        todo = []
        for idx, tap in enumerate(tape.taps):
            if idx in self._pix:
                old = self._pix[idx][0]
                if (old[1], old[2]) == tuple(int(v) for v in tap.a.shape[-2:]):
                    continue
                self._flush_pixgrams(only=idx)  # the input size changed within the fit: fold what there is, start anew
                self._pix_early.add(idx)  # (this accumulator's own A factor of the tap is no longer untouched: `_fold_lanes`)
            geo = self._pix_geometry(tap)
            if geo is None:
                continue
            if geo[0] == "pair":
                todo.append((idx, geo, (geo[4][0] * geo[3] * geo[3],), tap.a.device))
            else:
                npix = geo[1] * geo[2] * geo[3]
                todo.append((idx, geo, (npix, npix), tap.a.device))
        if not todo:
            return
        pad = lambda n: (n + 63) // 64 * 64
        numel = [math.prod(shape) for _, _, shape, _ in todo]
        flat = torch.zeros(sum(pad(n) for n in numel), dtype=torch.float32, device=todo[0][3])
        off = 0
        for (idx, geo, shape, _), n in zip(todo, numel):
            self._pix[idx] = (geo, flat[off:off + n].view(shape))
            off += pad(n)

    #: streams the per-tap work of `_flush_pixgrams` is dealt to (1 = the calling stream only)
    flush_streams = 3
    #: queue priority of the lanes' streams (the device's range is 0 .. -1; 0: all streams alike)
    lane_priority = -1

    def _flush_pixgrams(self, only=None):
        """fold the pixel-pair accumulators into the (native-order) A factors; idempotent.  The taps are independent of
        each other and their kernels (the last partly filled group's products, the 81-block assembly) are small next to
        the chip: with several of them to do they are dealt, largest first, to a few streams that the calling stream
        joins at the end — this is once-per-fit work, but a 20-minibatch fit (the driver's bench) spends 5 % of its time
        in it."""
        K = get_kernels()
        self._join_side()
        todo = [only] if only is not None else sorted(self._pix, key=lambda i: -self._pix[i][1].numel())

        def one(idx):
            self._drain_pixpair(idx)
            geo, buf = self._pix.pop(idx)
            if geo[0] == "pair":
                K.pixpair_assemble(buf, geo[4], geo[1], geo[2], geo[3], 1.0, self.factors[idx][1], upper_only=True)
            else:
                K.pixgram_assemble(buf, geo[1], geo[2], geo[3], 1.0, self.factors[idx][1])
            return buf

        dev = self.loss.device if self.loss is not None else None
        if len(todo) < 3 or self.flush_streams < 2 or dev is None or dev.type != "cuda" or not self.overlap:
            for idx in todo:
                one(idx)
            return
        streams = _process_streams(("flush", dev, self.flush_streams), lambda: [torch.cuda.Stream(dev) for _ in range(self.flush_streams)])
        cur = torch.cuda.current_stream(dev)
        for st in streams:
            st.wait_stream(cur)
        for j, idx in enumerate(todo):
            st = streams[j % len(streams)]
            with torch.cuda.stream(st):
                buf = one(idx)
            buf.record_stream(st)  # (dropped here, still read by the assembly on `st`)
            self.factors[idx][1].record_stream(st)
        for st in streams:
            cur.wait_stream(st)

    def _lane_add_batch(self, x, y):
        dev = x.device
        on_device = x.is_cuda
        cur = torch.cuda.current_stream(dev) if on_device else None
        if self._lane_accs is None:
            streams = [None] * self.lanes
            if on_device:
                # the lanes' streams (forward + reverse sweep: the critical path) get a higher queue priority than the streams
                # the factor kernels run on (`lane_priority = 0`: all alike; measured 7.35 -> 6.96 ms per step)
                prio = int(self.lane_priority)
                streams = _process_streams(("lanes", dev, self.lanes, prio),
                                           lambda: [torch.cuda.Stream(dev, priority=prio) for _ in range(self.lanes)])
            self._lane_accs = []
            for k in range(self.lanes):
                sub = KronAccumulator(self.backend, self.N, self.kfac_approx, self.overlap)
                sub.lanes, sub._lane_id, sub._lane_stream = 1, k, streams[k]
                sub.coalesce = False  # (the parent stacks)
                sub.use_pixgram, sub.pix_group, sub.lag_join = self.use_pixgram, self.pix_group, self.lag_join
                sub.lag_depth, sub.direct_stack = self.lag_depth, self.direct_stack
                sub._defer_bn, sub._persist_slabs = self._defer_bn, self._persist_slabs
                self._lane_accs.append(sub)
        k = self._lane_next
        self._lane_next = (k + 1) % self.lanes
        sub = self._lane_accs[k]
        if not on_device:
            sub.add_batch(x, y)
            return
        st = sub._lane_stream
        # The lane works on PRIVATE copies of the minibatch, taken on the calling stream: the caller's stream does not wait for
        # the lane, so a caller who refills the same device buffers for the next minibatch would otherwise overwrite them
        # under the lane's forward pass.  (Making the calling stream wait until the inputs are consumed was measured instead:
        # it locks the lanes in phase and the gain of running them is gone — 9.52 ms per step against 9.07-9.18.)
        x = x.clone()
        y = y.clone() if torch.is_tensor(y) and y.is_cuda else y
        st.wait_stream(cur)
        if sub.factors is None and k > 0 and self.backend.__dict__.get("_lanes_warm") != self._shared_signature(x):
            # first minibatch of this lane: whatever lane 0's first minibatch builds lazily and everybody shares from then
            # on (split weight planes, BatchNorm scale words, pixel-pair tables) must exist before it is read here.  Only
            # the first fit after the model (or the input geometry) changed pays for this: `_fold_lanes` notes for which
            # parameter versions the shared state is known to be complete.
            first = self._lane_accs[0]
            st.wait_stream(first._lane_stream)
            if first._side is not None:
                st.wait_stream(first._side)
        if self._lane_sig is None:
            self._lane_sig = self._shared_signature(x)
        with torch.cuda.stream(st):
            sub.add_batch(x, y)
        for t in (x, y):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(st)
        self._act_numel = max(self._act_numel, sub._act_numel)
        sub.max_ahead = self.max_ahead
        sub._throttle(st)

    def _shared_signature(self, x):
        """what the lazily built state shared by the lanes depends on: every parameter / buffer of the model (storage and
        version counter) and the geometry of a sample"""
        m = self.backend.model
        ts = list(m.parameters()) + list(m.buffers())
        return (tuple((t.data_ptr(), t._version) for t in ts), tuple(x.shape[1:]), x.dtype, x.device)

    #: ``False``: the A-side work of reading a fit waits for the lanes' reverse sweeps
    early_flush = True
    #: the HOST waits for the lanes' streams before it enqueues the once-per-fit work.  The host runs up to `max_ahead`
    #: minibatches per lane ahead of the device; enqueued at that moment, the flush streams' and the calling stream's waits sit
    #: unsatisfied in their queues for ~60 ms beside the lanes' ~210 launches per minibatch — measured (tools/finalize_variants.py,
    #: profiles/r06_finalize_variants.log): a 20-minibatch fit 7.34 ms per step that way, 6.91 with the host waiting first and
    #: enqueueing onto a drained device (the finalize work itself is 7 ms either way).  ``False``: enqueue at once.
    drain_before_fold = True

    def _fold_lanes(self):
        """bring the lanes' partial sums together on the calling stream (before anything reads the accumulated state).

        What reading a fit costs besides its minibatches matters for short fits (the driver's bench is 20 minibatches: round
        3 spent 10 ms here, half a minibatch per step).  The A side — the last, partly filled pixel-pair groups and the
        81-block assembly of every 3x3 layer — depends only on the FORWARD passes, so it is dealt to the flush streams
        behind the event at the end of each lane's latest A-side work and runs UNDER the reverse sweeps that are still in
        flight; the lanes' blocks are summed inside the assembly (`blocks2`) instead of by a pass of their own.  Only the G
        side (one multi-tensor add) waits for the sweeps."""
        self._flush_stash()  # (stacked small minibatches that have not been swept yet)
        subs, self._lane_accs = self._lane_accs, None
        if not subs:
            return
        if self._lane_sig is not None and any(sub.factors is not None for sub in subs):
            self.backend.__dict__["_lanes_warm"] = self._lane_sig  # (the calling stream waits for the lanes just below)
        K = get_kernels()
        on_device = subs[0]._lane_stream is not None
        dev = subs[0]._lane_stream.device if on_device else None
        cur = torch.cuda.current_stream(dev) if on_device else None
        live = [sub for sub in subs if sub.factors is not None]
        if on_device and self.drain_before_fold:
            for sub in subs:
                sub._lane_stream.synchronize()
                if sub._side is not None:
                    sub._side.synchronize()
        early = (on_device and self.overlap and self.early_flush and bool(live) and all(sub._a_done is not None for sub in live)
                 and any(sub._pix for sub in live))
        fstreams = None
        if early:
            nfs = max(self.flush_streams, 1)
            fstreams = _process_streams(("flush", dev, nfs), lambda: [torch.cuda.Stream(dev) for _ in range(nfs)])
            for st in fstreams:
                st.wait_stream(cur)
                for sub in live:
                    st.wait_event(sub._a_done)
        else:
            for sub in subs:
                if on_device:
                    cur.wait_stream(sub._lane_stream)
            for sub in live:
                sub._join_side()
        # Pixel-pair state first: the lanes' blocks (and what is still stacked for their next launch) are brought together
        # in the first lane, so that the 81-block assembly — the expensive part of reading a fit — runs ONCE, not per lane
        # (a 20-minibatch fit spent 18 ms here with two lanes against 8.5 ms with one).
        merged = set()  # taps whose A factor is assembled into the first lane's (the other lanes' stay zero)
        if live:
            base = live[0]

            def one(idx, st):
                geo, buf = base._pix[idx]
                others = []
                for sub in live[1:]:
                    o = sub._pix.get(idx)
                    if o is None or o[0][:4] != geo[:4] or (geo[0] == "pair" and o[0][4] is not geo[4]):
                        continue  # (different geometry: this lane assembles its own)
                    pend = sub._pix_pending.pop(idx, None)
                    if pend is not None and pend["n"]:
                        bp = base._pix_pending.get(idx)
                        B = pend["B"]
                        if (bp is not None and bp["B"] == B and bp["alpha"] == pend["alpha"]
                                and bp["n"] + pend["n"] <= bp["stack"].shape[0] // B):
                            bp["stack"][bp["n"] * B:(bp["n"] + pend["n"]) * B].copy_(pend["stack"][:pend["n"] * B])
                            bp["n"] += pend["n"]
                            if bp.get("amax") is not None:  # (words of non-negative floats: the float maximum is theirs)
                                if pend.get("amax") is not None:
                                    torch.maximum(bp["amax"], pend["amax"], out=bp["amax"])
                                else:
                                    bp["amax"] = None
                            if st is not None:
                                pend["stack"].record_stream(st)
                        else:
                            sub._pix_pending[idx] = pend
                            sub._drain_pixpair(idx)
                    others.append(o[1])
                    del sub._pix[idx]
                if not others:
                    return False
                base._drain_pixpair(idx)
                del base._pix[idx]
                A = base.factors[idx][1]
                for o in others[1:]:
                    buf.add_(o)
                if geo[0] == "pair":
                    K.pixpair_assemble(buf, geo[4], geo[1], geo[2], geo[3], 1.0, A, blocks2=others[0], upper_only=True)
                else:
                    buf.add_(others[0])
                    K.pixgram_assemble(buf, geo[1], geo[2], geo[3], 1.0, A)
                if st is not None:
                    for t in [buf, A] + others:  # allocated on a lane's stream, read here
                        t.record_stream(st)
                return True

            todo = sorted(base._pix, key=lambda i: -base._pix[i][1].numel())
            for j, idx in enumerate(todo):
                if fstreams is not None:
                    st = fstreams[j % len(fstreams)]
                    with torch.cuda.stream(st):
                        if one(idx, st):
                            merged.add(idx)
                elif one(idx, None):
                    merged.add(idx)
        if early:
            for sub in subs:
                cur.wait_stream(sub._lane_stream)
            for sub in live:
                sub._join_side()
            for st in fstreams:
                cur.wait_stream(st)
        first = True
        for sub in subs:
            if sub.factors is None:
                continue
            sub._flush_pixgrams()
            sub._flush_g_slabs()
            if self.factors is None:
                self.factors, self.loss, self._taps_meta = sub.factors, sub.loss, sub._taps_meta
                self._pix, self._pix_pending, self._gslabs = {}, {}, {}
                self._gscale, self._tap_index = dict(sub._gscale), dict(sub._tap_index)
            else:
                if set(self._gscale) != set(sub._gscale):
                    raise RuntimeError("the lanes of a fit disagree about the deferred BatchNorm scales")
                # (those A factors were never written in the other lanes: zeros — unless the input size changed within the
                #  fit and the lane folded its blocks early into its OWN factor: that partial sum must be added)
                skip = (merged - sub._pix_early) if not first else set()
                mine = [t for i, F in enumerate(self.factors) for j, t in enumerate(F) if not (j == 1 and i in skip)] + [self.loss]
                theirs = [t for i, F in enumerate(sub.factors) for j, t in enumerate(F) if not (j == 1 and i in skip)] + [sub.loss]
                torch._foreach_add_(mine, theirs)
            first = False
            if on_device:
                for t in sub._raw_tensors():  # allocated on the lane's stream, read (and from now on owned) here
                    t.record_stream(cur)

    #: class-level defaults of the per-accumulator attributes `coalesce` / `lanes`
    default_coalesce = True
    default_lanes = 2
    #: stacked sweeps hold at most this many activation floats (forward activations, masks and the seed-batched cotangents of
    #: a sweep all scale with it): 2^28 floats = 1 GiB per fp32 copy
    coalesce_act_floats = 1 << 28
    #: ... and at most this many of the caller's minibatches: the batch size is the caller's memory knob
    coalesce_max_batches = 32

    def coalesce_target(self, x) -> int:
        """samples per stacked sweep for minibatches shaped like ``x`` (0: this model / minibatch does not stack): work per
        sample grows with the parameter count; 2^27 parameter-samples per sweep keeps a sweep around a millisecond (LeNet-5:
        eight loader batches of 256; ResNet-18 at batch 128: never).  Bounded by memory as well: by the largest per-sample
        ACTIVATION the first (never stacked) minibatch's forward pass measured — a few-parameter convolutional model on large
        images has small parameter and input counts and large maps —, by the input size, and by a multiple of the loader's
        batch.  The fit's first minibatch is always swept alone: unsupported-model and shape errors surface on the first
        `add_batch`, as they do without stacking."""
        b = self.backend
        if (not self.coalesce or not torch.is_tensor(x) or not x.is_floating_point() or x.dim() < 2
                or getattr(b, "stochastic", False) or b.last_layer or not self._dispatched):
            return 0
        n_params = sum(p.numel() for p in b.params)
        act = max(int(self._act_numel), x[0].numel(), 1)  # (per sample, times the seeds of its reverse sweep)
        target = min(8192, (1 << 27) // max(n_params, 1), (1 << 24) // max(x[0].numel(), 1),
                     int(self.coalesce_act_floats) // act, int(self.coalesce_max_batches) * x.shape[0])
        return target if target >= 2 * x.shape[0] else 0

    def _flush_stash(self):
        if not self._stash:
            return
        stash, self._stash, self._stash_n = self._stash, [], 0
        if len(stash) == 1:
            self._dispatch(*stash[0])
        else:
            self._dispatch(torch.cat([x for x, _ in stash]), torch.cat([y for _, y in stash]))

    def add_batch(self, x, y):
        if self._caller is not None:
            twin, _ = self._caller._twin()  # (re-synchronises the fp32 copy if a parameter or buffer changed)
            x, y = twin._to32(x), twin._to32(y)
        target = self.coalesce_target(x) if torch.is_tensor(x) and torch.is_tensor(y) and y.shape[:1] == x.shape[:1] else 0
        if target:
            if self._stash and (self._stash[0][0].shape[1:] != x.shape[1:] or self._stash[0][0].dtype != x.dtype
                                or self._stash[0][1].shape[1:] != y.shape[1:] or self._stash[0][1].dtype != y.dtype
                                or self._stash[0][0].device != x.device or self._stash_n + x.shape[0] > target):
                self._flush_stash()
            self._stash.append((x.clone(), y.clone()))  # (private copies: the caller may refill its buffers)
            self._stash_n += x.shape[0]
            if self._stash_n >= target:
                self._flush_stash()
            return
        self._flush_stash()
        self._dispatch(x, y)

    #: minibatches per stream the host may run ahead of the device (0: unbounded).  The host enqueues a ResNet-18 step in
    #: 4.3 ms, the device works 6.8 ms on it: left alone the host runs ~150 steps ahead (until the runtime's queues push
    #: back), and every tensor a side stream still has to read (`record_stream`) stays unavailable to the allocator until
    #: the device gets there — 160 GiB reserved for a fit whose live tensors are 12 GiB.  Waiting for the minibatch
    #: enqueued `max_ahead` steps ago costs nothing (the device still has that many steps queued per lane: same 6.7 ms per
    #: step) and bounds the lead: 42 / 87 GiB reserved at 4 / 12 (`tools/session_age.py`, `profiles/r05_box_session_age.log`;
    #: on a box whose later processes ran a 400-step loop at 13 ms per step it also brought them back to 6.8 - 8.0).
    #: Round 6 measured the lead itself (300 steps, `tools/session_age.py`, profiles/r06_host_lead.log): 6.56 - 6.63 ms per
    #: step at 1, 2 and 8 alike, 29 / 32 - 34 / 44 GiB reserved — two steps per lane keep every queue fed (the device works on
    #: one, the next is already there) and that is all a lead is for.
    max_ahead = 2

    def _throttle(self, stream):
        if not self.max_ahead or stream is None:
            return
        ev = torch.cuda.Event()
        ev.record(stream)
        self._ahead.append(ev)
        if len(self._ahead) > self.max_ahead:
            self._ahead.popleft().synchronize()

    def _dispatch(self, x, y):
        b = self.backend
        self._dispatched += 1
        if self.lanes > 1 and self.overlap and not self.defer_pix and torch.is_tensor(x) and (x.is_cuda or self._lanes_anywhere):
            return self._lane_add_batch(x, y)
        self._add_batch(x, y)
        if torch.is_tensor(x) and x.is_cuda and self._lane_stream is None:
            self._throttle(torch.cuda.current_stream(x.device))

    def _add_batch(self, x, y):
        b = self.backend
        b._act_sink = self._pix_sink if (self.direct_stack and self.overlap and not self.defer_pix and getattr(self, "_pix_pending", None)
                                         and torch.is_tensor(x) and x.is_cuda) else None
        try:
            f, tape, grad_fn = b._forward(x)
        finally:
            b._act_sink = None
        if tape.uncovered:
            raise NotImplementedError("KFAC supports nn.Linear / nn.Conv2d parameters only")
        if self.factors is None:
            self._alloc(tape, f.device)
        if not self._act_numel:  # (what bounds a stacked sweep's memory: `coalesce_target`)
            nb = max(int(f.shape[0]), 1)
            acts = [int(t.a.numel()) // nb for t in tape.taps if torch.is_tensor(getattr(t, "a", None))]
            self._act_numel = max(acts + [1]) * max(int(f.shape[-1]) - 1, 1)
        rt = math.sqrt(float(b.factor))
        if not self.defer_pix:
            self._ensure_pixgrams(tape)
        # The A factors need only the forward activations: enqueue them on a side stream so that they
        # overlap the C reverse passes (whose late, small-spatial conv kernels do not fill the chip).
        side = None
        if self.overlap and f.is_cuda:
            if self._side is None:  # one side stream per device and lane, shared by all accumulators of the process
                self._side = _process_streams(("side", f.device, self._lane_id), lambda: torch.cuda.Stream(f.device))
            side = self._side
            main = torch.cuda.current_stream(f.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for i, (tap, F) in enumerate(zip(tape.taps, self.factors)):
                    self._accumulate_A(i, tap, F, rt)
            # everything the A factors of this fit need from this minibatch has been enqueued: whoever reads the fit
            # (`_fold_lanes`) starts the A-side flush behind THIS event, under the reverse sweep that is still to come
            self._a_done = torch.cuda.Event()
            self._a_done.record(side)
        else:
            for i, (tap, F) in enumerate(zip(tape.taps, self.factors)):
                self._accumulate_A(i, tap, F, rt)
        seeds, hs = b._kron_seeds(f, y, self.loss)
        defer = getattr(grad_fn, "streams_taps", False) and self._defer_bn
        if getattr(grad_fn, "streams_taps", False):
            # The sweep hands over each layer's gradient the moment it is complete (and, for many-output models, seed
            # chunk by seed chunk).  With a side stream the G-factor kernel goes there at once: the MFMA-bound Gram
            # kernels then overlap MIOpen's backward-data kernels of the earlier layers.
            by_name = {tap.name: (tap, F) for tap, F in zip(tape.taps, self.factors)}

            def on_tap(name, g):
                tap, F = by_name[name]
                persist = None if isinstance(g, SplitTensor) else self._g_slabs(tap, g, rt * hs)
                if side is None:
                    b._factor_G(tap, g, rt * hs, self.kfac_approx, F[0], fused=True, persist=persist)
                    return
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    b._factor_G(tap, g, rt * hs, self.kfac_approx, F[0], fused=True, persist=persist)
                (g.planes if isinstance(g, SplitTensor) else g).record_stream(side)  # allocated on main, read on side
                if isinstance(g, SplitTensor) and torch.is_tensor(g.sexp):
                    g.sexp.record_stream(side)

            grad_fn(seeds, stack=False, on_tap=on_tap, defer_bn_scale=defer)
        else:
            grads = grad_fn(seeds, stack=False)
            for tap, g, F in zip(tape.taps, grads, self.factors):
                b._factor_G(tap, g, rt * hs, self.kfac_approx, F[0], fused=True)
        if defer:
            self._note_grad_scales(tape, grad_fn.grad_scale())
        if side is not None:
            if self.lag_join:
                # The factor kernels of this minibatch may run on into the next one: its forward pass (small grids at
                # batch 128) leaves most of the chip idle.  The main stream waits only for the minibatch BEFORE this one
                # (at most one step of lag, so the memory held for the side stream stays bounded); everything the side
                # stream reads that was allocated on the main stream is marked, so the allocator keeps it until then.
                for tap in tape.taps:
                    if torch.is_tensor(tap.a) and tap.a.is_cuda:
                        tap.a.record_stream(side)
                if self._side_done is not None:
                    self._side_older.append(self._side_done)
                while len(self._side_older) >= self.lag_depth and self._side_older:
                    main.wait_event(self._side_older.pop(0))
                self._side_done = torch.cuda.Event()
                self._side_done.record(side)
            else:
                main.wait_stream(side)
        tape.release()

    def _g_slabs(self, tap, g, alpha):
        """Persistent split-K slabs of a conv tap's G factor (``expand``): the partial tiles of every minibatch are
        accumulated in place (LK_GRAM_SLABS_PERSIST) and reduced ONCE per fit — the sum over minibatches is linear, so
        the two reduce launches per tap and minibatch disappear.  Allocated on the calling stream; ``None`` = plain
        per-launch reduction (Linear taps, `reduce`, the literal one-batch path)."""
        if not self._persist_slabs or tap.kind != "conv2d" or self.kfac_approx != "expand" or g.dim() != 5:
            return None
        K = get_kernels()
        S, B, Do = g.shape[0], g.shape[1], g.shape[2]
        L = g.shape[3] * g.shape[4]
        need = K.gram_nt_slab_bytes(S * B, Do, L)
        idx = self._tap_index[tap.name]
        cur = self._gslabs.get(idx)
        if cur is not None and (cur[0].numel() < need or cur[2] != L or cur[3] != alpha):
            self._flush_g_slabs(only=idx)  # geometry changed within the fit: fold what there is
            cur = None
        if cur is None:
            cur = (torch.zeros(need, dtype=torch.uint8, device=g.device), Do, L, alpha)
            self._gslabs[idx] = cur
        return cur[0]

    # ---- raw-form algebra (HipKron keeps the minibatches of the reference's literal loop in this form) --------------------
    def _raw_tensors(self):
        return [t for F in self.factors for t in F] + [self.loss]

    def _raw_compatible(self, other) -> bool:
        return (self.factors is not None and other.factors is not None and not other._pix
                and not other._pix_pending and not self._gslabs and not other._gslabs
                and (not (self._pix or self._pix_pending) or all(g[0] == "pair" for g, _ in self._pix.values()))
                and self.N == other.N and self.kfac_approx == other.kfac_approx and self.backend is other.backend
                and self._taps_meta == other._taps_meta and len(self.factors) == len(other.factors)
                and all(a.shape == b.shape for a, b in zip(self._raw_tensors(), other._raw_tensors()))
                and set(self._gscale) == set(other._gscale)
                and all(self._gscale[k] is other._gscale[k] or torch.equal(self._gscale[k], other._gscale[k])
                        for k in self._gscale))

    def clone(self) -> "KronAccumulator":
        """an independent copy of the accumulated state (factors in the accumulator's raw form)"""
        self._fold_lanes()
        self._join_side()
        new = KronAccumulator(self.backend, self.N, self.kfac_approx, self.overlap)
        new._caller, new._out_dtype, new.coalesce = self._caller, self._out_dtype, self.coalesce
        new.use_pixgram, new._persist_slabs = self.use_pixgram, self._persist_slabs
        new.defer_pix, new.pix_group = self.defer_pix, self.pix_group
        new._pix_inputs = {k: list(v) for k, v in self._pix_inputs.items()}  # (the tensors themselves are never written)
        if self.factors is not None:
            if self._gslabs or any(g[0] != "pair" for g, _ in self._pix.values()):
                raise RuntimeError("clone() of an accumulator with dense pixel-pair / slab state is not supported")
            for idx in list(self._pix_pending):
                self._drain_pixpair(idx)  # (a partly filled group: launch it, so the blocks below are the whole state)
            new.factors = [[t.clone() for t in F] for F in self.factors]
            new.loss = self.loss.clone()
            new._taps_meta = list(self._taps_meta)
            new._gscale = dict(self._gscale)
            new._tap_index = dict(self._tap_index)
            new._pix = {idx: (geo, buf.clone()) for idx, (geo, buf) in self._pix.items()}
            new._pix_pending, new._gslabs = {}, {}
            new._side = self._side
        return new

    def merge_(self, other: "KronAccumulator") -> bool:
        """``self += other`` on the raw forms (one multi-tensor add); False if the two do not have the same structure.

        With a side stream the add is enqueued THERE, behind ``other``'s factor kernels (same stream, in order): the
        calling stream does not wait for them and goes on with the next minibatch, at most one merge ahead — the schedule
        the fused accumulator has with ``lag_join``.  Readers (`finalize`, `tensors`, `clone`) join the side stream."""
        self._fold_lanes()
        other._fold_lanes()
        if not self._raw_compatible(other):
            return False
        skip = set(other._pix_inputs)  # A factors `other` left to this sum: identically zero there, nothing to add
        mine = [t for i, F in enumerate(self.factors) for j, t in enumerate(F) if not (j == 1 and i in skip)] + [self.loss]
        theirs = [t for i, F in enumerate(other.factors) for j, t in enumerate(F) if not (j == 1 and i in skip)] + [other.loss]
        side = self._side or other._side
        if side is None or not self.lag_join or not self.loss.is_cuda:
            self._join_side()
            other._join_side()
            torch._foreach_add_(mine, theirs)
            self._push_adopted(self._adopt_pix_inputs(other))
        else:
            self._side = side
            main = torch.cuda.current_stream(self.loss.device)
            if self._side_done is not None:
                main.wait_event(self._side_done)  # (bounded lag: the merge before this one has finished)
            todo = self._adopt_pix_inputs(other)  # (allocates pixel-pair blocks on the calling stream: read there at the end)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                torch._foreach_add_(mine, theirs)
                self._push_adopted(todo)
            for t in theirs:  # allocated on the calling stream, read on the side stream: keep them until it is done
                t.record_stream(side)
            for items in other._pix_inputs.values():
                for _, _, t, _ in items:
                    t.record_stream(side)
            self._side_done = torch.cuda.Event()
            self._side_done.record(side)
        return True

    def _join_side(self):
        """the calling stream waits for everything the factor kernels have been asked to do so far"""
        if self._side is not None:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)
            self._side_done = None
            self._side_older = []

    def _flush_g_slabs(self, only=None):
        K = get_kernels()
        if self._gslabs:
            self._join_side()
        for idx in ([only] if only is not None else list(self._gslabs)):
            buf, n, L, alpha = self._gslabs.pop(idx)
            K.gram_slabs_reduce(buf, n, L, alpha, self.factors[idx][0], upper_only=True)

    def _note_grad_scales(self, tape, scales):
        """Deferred BatchNorm scales (laplace_amd/sweep.py): the G accumulators of these taps hold sums of UNSCALED
        gradient products; ``finalize`` applies ``diag(s) G diag(s)`` once.  ``s`` is a constant of an eval-mode model;
        should it change between minibatches, what has been accumulated is scaled now and deferral stops."""
        for idx, tap in enumerate(tape.taps):
            s_new = scales.get(tap.name)
            s_old = self._gscale.get(idx)
            if s_old is None:
                if s_new is not None:
                    self._gscale[idx] = s_new
            elif s_new is None or (s_new is not s_old and not torch.equal(s_new, s_old)):
                raise RuntimeError(f"{tap.name}: the BatchNorm scale changed during the fit (model not in eval mode?)")

    def _apply_grad_scales(self):
        self._join_side()
        for idx, s_ in self._gscale.items():
            G = self.factors[idx][0]
            G.mul_(s_.reshape(-1, 1) * s_.reshape(1, -1))
        self._gscale = {}

    def tensors(self) -> list[torch.Tensor]:
        """Everything a data-parallel fit has to all-reduce (upper triangles are what counts).  The deferred BatchNorm
        scales are applied HERE, before the exchange: ``diag(s) G diag(s)`` is linear in G, so scaled factors add
        exactly, and a rank with an empty shard — which never learned a scale and contributes zeros — needs none."""
        self._fold_lanes()
        self._resolve_pix_inputs()
        self._flush_pixgrams()  # the assembled factors are what is exchanged, not the larger pixel-pair Grams
        self._flush_g_slabs()
        self._apply_grad_scales()
        out = CurvatureExchange([t for F in self.factors for t in F] + [self.loss])
        out.owner = self
        return out

    def finalize(self):
        """-> (loss, HipKron) in the reference's layout (laplace/curvature/curvlinops.py:55-75)."""
        K = get_kernels()
        rt = math.sqrt(float(self.backend.factor))
        self._fold_lanes()
        self._resolve_pix_inputs()
        self._flush_pixgrams()
        self._flush_g_slabs()
        self._join_side()
        # ONE launch for the layout pass of every factor: mirror the upper triangles, bring the conv A factors from the
        # kernels' (kh, kw, ci) column order into F.unfold's, apply the deferred BatchNorm scales diag(s) G diag(s)
        # (round 3: two or three small launches per factor, 1.7 ms of launch gaps at the end of every fit)
        items, done = [], []
        for idx, ((G, A), (has_bias, native)) in enumerate(zip(self.factors, self._taps_meta)):
            s_ = self._gscale.get(idx)
            if s_ is not None:
                s_ = s_.detach().to(torch.float32).reshape(-1).contiguous()
            items.append((G, None, s_, 0, 1))
            if native is not None and native[1] > 1:
                A_out = torch.empty_like(A)
                items.append((A, A_out, None, native[0], native[1]))
                A = A_out
            else:
                items.append((A, None, None, 0, 1))
            done.append((G, A, has_bias))
        K.finalize_factors(items)
        self._gscale = {}
        kfacs = []
        for G, A, has_bias in done:
            if G.numel() == 1 and A.numel() == 1 and not has_bias:
                kfacs.append([G * A])
            else:
                kfacs.append([G, A])
            if has_bias:
                kfacs.append([G * rt])
        self.factors = None
        if self._out_dtype != torch.float32:
            return _HipCurvatureMixin._cast((self.loss[0], HipKron(kfacs)), self._out_dtype)
        return self.loss[0], HipKron(kfacs)


class HipGGN(_HipCurvatureMixin, GGNInterface):
    """Generalised Gauss-Newton on HIP — replaces GGNInterface (laplace/curvature/curvature.py:293-433) and
    CurvlinopsGGN's KFAC (curvlinops.py:150-164).  ``stochastic=True`` is the MC Fisher with ``num_samples`` draws
    per data point (curvature.py:341-364; KFAC: ``FisherType.MC`` with ``mc_samples``): the same accumulation
    kernels, ``num_samples`` seed columns instead of ``C - 1``.  ``self.generator`` (optional ``torch.Generator``
    on the model's device) makes the draws reproducible."""

    def __init__(self, model, likelihood, last_layer=False, subnetwork_indices=None,
                 dict_key_x="input_ids", dict_key_y="labels", stochastic=False, num_samples=1):
        if stochastic and int(num_samples) < 1:
            raise ValueError("num_samples must be >= 1")
        super().__init__(model, likelihood, last_layer, subnetwork_indices, dict_key_x, dict_key_y,
                         stochastic=stochastic, num_samples=num_samples)
        self.generator = None

    _kron_seeds = _HipCurvatureMixin._ggn_seeds

    def kron_accumulator(self, N: int, **kwargs) -> KronAccumulator:
        """Fused-accumulation form of :meth:`kron` for a whole fit (see :class:`KronAccumulator`)."""
        acc = KronAccumulator(self, N, kwargs.get("kfac_approx", "expand"), kwargs.get("overlap", True))
        if kwargs.get("coalesce") is not None:  # (``coalesce=False``: every minibatch is swept as it arrives)
            acc.coalesce = bool(kwargs["coalesce"])
        return acc

    # KFAC — replaces CurvlinopsInterface.kron (laplace/curvature/curvlinops.py:77-108)
    def kron(self, x, y, N, **kwargs):
        twin, dt = self._twin()
        if twin is not None:
            return self._cast(twin.kron(self._to32(x), self._to32(y), N, **kwargs), dt)
        kfac_approx = kwargs.get("kfac_approx", "expand")
        return self._kron_impl(x, y, N, self._ggn_seeds, None, kfac_approx)

    # diag GGN — replaces GGNInterface.diag (laplace/curvature/curvature.py:413-433)
    def diag(self, x, y, **kwargs):
        twin, dt = self._twin()
        if twin is not None:
            return self._cast(twin.diag(self._to32(x), self._to32(y), **kwargs), dt)
        out = self._diag_impl(x, y, self._ggn_seeds, 1.0)
        if out is None:  # layers without a kernel: the reference's generic path on our Jacobians
            return super().diag(x, y, **kwargs)
        return out

    # dense GGN — replaces GGNInterface.full (laplace/curvature/curvature.py:375-411)
    def full(self, x, y, **kwargs):
        twin, dt = self._twin()
        if twin is not None:
            return self._cast(twin.full(self._to32(x), self._to32(y), **kwargs), dt)
        K = get_kernels()
        if self.last_layer and self.subnetwork_indices is None and not self.stochastic:
            f, tape, _ = self._forward(x)
            phi = tape.taps[0].a.to(torch.float32).contiguous()
            B, C = f.shape
            has_bias = tape.taps[0].has_bias
            loss = torch.zeros(1, dtype=torch.float32, device=f.device)
            if self.likelihood == "regression":
                K.sq_err_sum(f, y.reshape(B, C).to(torch.float32).contiguous(), self.factor, loss)
                probs = None
            else:
                K.softmax_hess_sqrt(f, y.reshape(B).to(torch.int64).contiguous(), loss)
                probs = torch.softmax(f, dim=-1).contiguous()
            P = tape.n_params
            H = torch.zeros(P, P, dtype=torch.float32, device=f.device)
            K.ll_ggn_full(phi, probs, has_bias, 1.0, H)
            tape.release()
            return loss[0], H
        if not self._supported():
            return super().full(x, y, **kwargs)
        loss = None

        def seeds_fn(f):
            nonlocal loss
            loss = torch.zeros(1, dtype=torch.float32, device=f.device)
            return self._ggn_seeds(f, y, loss)[0]

        Z, _ = self._rows(x, seeds_fn)
        return loss[0], self._full_from_rows(Z, 1.0)

    # last-layer Jacobians — replaces CurvatureInterface.last_layer_jacobians (curvature.py:131-167) for a Linear head:
    # the feature pass of the backend + one store-stream kernel; with enable_backprop the autograd form is needed
    def last_layer_jacobians(self, x, enable_backprop: bool = False):
        twin, dt = self._twin()
        if twin is not None and not enable_backprop:  # (a differentiable result must stay on the caller's own model)
            return self._cast(twin.last_layer_jacobians(self._to32(x), False), dt)
        if enable_backprop or not self._supported() or not self.last_layer:
            return super().last_layer_jacobians(x, enable_backprop)
        try:
            f, tape, _ = self._forward(x)
        except NotImplementedError:
            return super().last_layer_jacobians(x, enable_backprop)
        tap = tape.taps[0]
        phi = tap.a.reshape(tap.a.shape[0], -1).contiguous()
        tap.a = None
        C = f.shape[1]
        Js = get_kernels().jac_last_layer(phi, C, tap.module.bias is not None)
        return Js, f

    # per-sample output Jacobians — replaces CurvatureInterface.jacobians (curvature.py:88-129)
    def jacobians(self, x, enable_backprop: bool = False):
        twin, dt = self._twin()
        if twin is not None and not enable_backprop:  # (a differentiable result must stay on the caller's own model)
            return self._cast(twin.jacobians(self._to32(x), False), dt)
        if enable_backprop or self.last_layer or not self._supported():
            return super().jacobians(x, enable_backprop)

        def seeds_fn(f):
            B, C = f.shape
            eye = torch.eye(C, dtype=f.dtype, device=f.device)
            return eye[:, None, :].expand(C, B, C).contiguous()

        Js, f = self._rows(x, seeds_fn)
        if self.subnetwork_indices is not None:
            Js = Js[:, :, self.subnetwork_indices]
        return Js, f


class HipEF(_HipCurvatureMixin, EFInterface):
    """Empirical Fisher on HIP — replaces EFInterface (laplace/curvature/curvature.py:436-505) and
    CurvlinopsEF's KFAC (curvlinops.py:167-176, FisherType.EMPIRICAL)."""

    def _ef_seeds(self, f, y, loss):
        return self._ef_seed(f, y, loss), 1.0

    _kron_seeds = _ef_seeds

    def kron_accumulator(self, N: int, **kwargs) -> KronAccumulator:
        acc = KronAccumulator(self, N, kwargs.get("kfac_approx", "expand"))
        if kwargs.get("coalesce") is not None:
            acc.coalesce = bool(kwargs["coalesce"])
        return acc

    def kron(self, x, y, N, **kwargs):
        twin, dt = self._twin()
        if twin is not None:
            return self._cast(twin.kron(self._to32(x), self._to32(y), N, **kwargs), dt)
        return self._kron_impl(x, y, N, self._ef_seeds, None, kwargs.get("kfac_approx", "expand"))

    def diag(self, x, y, **kwargs):
        twin, dt = self._twin()
        if twin is not None:
            return self._cast(twin.diag(self._to32(x), self._to32(y), **kwargs), dt)
        out = self._diag_impl(x, y, self._ef_seeds, float(self.factor))
        if out is None:
            return super().diag(x, y, **kwargs)
        return out

    def full(self, x, y, **kwargs):
        twin, dt = self._twin()
        if twin is not None:
            return self._cast(twin.full(self._to32(x), self._to32(y), **kwargs), dt)
        if not self._supported() or self.last_layer:
            return super().full(x, y, **kwargs)
        loss = None

        def seeds_fn(f):
            nonlocal loss
            loss = torch.zeros(1, dtype=torch.float32, device=f.device)
            return self._ef_seed(f, y, loss)

        Z, _ = self._rows(x, seeds_fn)
        return loss[0], self._full_from_rows(Z, float(self.factor))

    def gradients(self, x, y):
        twin, dt = self._twin()
        if twin is not None:
            return self._cast(twin.gradients(self._to32(x), self._to32(y)), dt)
        if not self._supported() or self.last_layer or isinstance(x, MutableMapping):
            return super().gradients(x, y)
        loss = None

        def seeds_fn(f):
            nonlocal loss
            loss = torch.zeros(1, dtype=torch.float32, device=f.device)
            return self._ef_seed(f, y, loss)

        Z, _ = self._rows(x, seeds_fn)
        Gs = Z[:, 0, :]
        if self.subnetwork_indices is not None:
            Gs = Gs[:, self.subnetwork_indices]
        return Gs, loss[0] / float(self.factor)
