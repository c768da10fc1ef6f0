"""The reverse sweep on NHWC split-fp16 cotangents: our own convolution kernels instead of MIOpen's.

:class:`laplace_amd.sweep.SeedBatchedSweep` walks the traced graph backwards with ONE cotangent of batch ``S*B`` and
hands every convolution to the library's fp32 backward-data (60 % of a ResNet-18 KFAC step).  This subclass keeps the
graph walk and replaces the data path (SURVEY.md §8f-1 "our own layer-local backward"; the reverse passes of
laplace/curvature/curvlinops.py:87-100):

* cotangents of feature maps live in HBM as NHWC *split tensors* — two fp16 planes and a power-of-two scale
  (csrc/lk_conv.hip) — produced by the element-wise VJP kernel ``lk_vjp_nhwc_split_f16x2`` (activation mask x folded
  BatchNorm scale x residual add, all seeds in one pass);
* a convolution's backward-data is ``lk_conv_nhwc_f16x2`` (implicit GEMM, three fp16 MFMAs per fp32 product block),
  writing fp32 NHWC plus max|.| for the next producer's scale; a strided 1x1 down-sampling branch ACCUMULATES into
  the main branch's result instead of materialising its three-quarters-zero cotangent;
* the G factor of a convolution layer is the Gram of the split tensor itself (``lk_gram_tn_f16x2``).

Graphs with nodes this path has no rule for (pooling other than the global average, convolutions whose channel counts
are not multiples of 32, ...) run through the parent class unchanged.
"""
from __future__ import annotations

import operator
import os

import weakref

import torch
import torch.fx as fx
import torch.nn.functional as F
from torch import nn

from laplace_amd import conv as cv
from laplace_amd._lib import SplitTensor
from laplace_amd.sweep import SeedBatchedSweep, SweepUnsupported


class _F32:
    """fp32 NHWC cotangent ``[S*B, H, W, C]`` + device word with the bit pattern of max|.|"""

    __slots__ = ("t", "amax")

    def __init__(self, t, amax):
        self.t, self.amax = t, amax


class _PendingConv:
    """A forward convolution that has not run yet: its only consumer is an eval-mode BatchNorm, which runs it with the
    per-channel affine map, the residual add and the ReLU in the convolution's epilogue (lk_conv_bn_act_nhwc_f16x2) —
    or, where that launch does not apply, materialises it as before.  Quacks like its fp32 ``[N, C, H, W]`` result as far
    as the traced forward looks at it (``shape``, ``dtype``, ``dim()``)."""

    __slots__ = ("prep", "xs", "shape", "dtype", "device", "_run")

    def __init__(self, prep, xs, shape, run):
        self.prep, self.xs, self.shape, self._run = prep, xs, torch.Size(shape), run
        self.dtype, self.device = torch.float32, xs.planes.device

    def dim(self) -> int:
        return len(self.shape)

    def materialize(self) -> torch.Tensor:
        return self._run()


class _LazyConv:
    """A stride-1 convolution's backward-data that has not run yet: whoever consumes the cotangent decides whether it is
    materialised as an fp32 tensor (``f32()``) or comes out of the convolution kernel already multiplied / joined / split
    (``fused(...)``, lk_conv_nhwc_f16x2_vjp) — the element-wise VJP kernel and the fp32 tensor's round trip then vanish."""

    __slots__ = ("_f32", "_fused", "_done")

    def __init__(self, f32, fused):
        self._f32, self._fused, self._done = f32, fused, None

    def f32(self) -> "_F32":
        if self._done is None:
            self._done = self._f32()
        return self._done

    def fused(self, **kw) -> SplitTensor:
        assert self._done is None
        return self._fused(**kw)


class _LazyStrided(_LazyConv):
    """A STRIDED convolution's backward-data that has not run yet (``desc = (prep, g, cscale)``, input ``hw``): one or two
    of them reaching the same node (the 3 x 3 main branch and the 1 x 1 shortcut of a residual down-sampling block) leave
    ``_to_split`` as ONE launch over all residue classes with the element-wise VJP fused
    (lk_conv_nhwc_f16x2_vjp_strided); ``f32(into)`` is the fallback: class by class into an fp32 tensor."""

    __slots__ = ("desc", "hw", "_run")

    def __init__(self, run, desc, hw):
        super().__init__(None, None)
        self._run, self.desc, self.hw = run, desc, hw

    def f32(self, into=None) -> "_F32":
        if self._done is None:
            self._done = self._run(into)
        return self._done


def _materialize(parts):
    out, first = [], None
    for p in parts:  # (plain tensors first: a pending strided convolution then adds into one instead of making its own)
        if isinstance(p, _F32) and first is None:
            first = p
    for p in parts:
        if isinstance(p, _LazyStrided):
            if p._done is None and first is not None:
                p.f32(first)  # accumulated into the tensor that is already there
                continue
            p = p.f32()
            first = first if first is not None else p
        elif isinstance(p, _LazyConv):
            p = p.f32()
            first = first if first is not None else p
        out.append(p)
    return out


class SplitSweep(SeedBatchedSweep):
    """Seed-batched reverse sweep whose feature-map cotangents are NHWC split tensors (needs the HIP kernels)."""

    def __init__(self, model, tap_modules, kernels=None):
        super().__init__(model, tap_modules, kernels)
        self._prep: dict[str, cv.PreparedConv] = {}
        self._amax_cache: dict = {}
        self.split_reason = self._split_eligible()
        self.split_ok = self.split_reason is None

    #: callable ``(tap name, NHWC shape, device) -> fp32 tensor or None``, set around a forward by the KFAC accumulator: where the
    #: activation that a tapped convolution reads is to be written (see `_run_bn_act`)
    act_sink = None
    #: ``False``: every backward-data writes fp32 and the element-wise VJP kernel runs on it
    fuse_vjp = True
    #: ``False``: strided convolutions run class by class into an fp32 tensor (one launch per
    #: residue class and branch) instead of the strided fused launch
    fuse_strided = True

    def _consumes_lazily(self, node) -> bool:
        """nodes whose rule hands all incoming cotangent parts to ``_to_split`` (which can fuse a pending convolution)"""
        if node.op == "call_module":
            m = self.modules[node.target]
            return isinstance(m, (nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.Tanh, nn.Sigmoid, nn.Identity, nn.Dropout)
                              + self._GENERIC_ACT_MODULES)
        if node.op == "call_function":
            return node.target in self._ELEMENTWISE_FN or node.target in self._GENERIC_ACT_FN
        if node.op == "call_method":
            return node.target in ("relu", "tanh", "sigmoid", "contiguous")
        return False

    # ---- static eligibility ---------------------------------------------------------------------------------------
    def _split_eligible(self):
        if self.kernels is None or not hasattr(self.kernels(), "conv_nhwc_f16x2"):
            return "kernels without the split-fp16 convolution"
        n_conv = 0
        for node in self.gm.graph.nodes:
            if node.op == "call_module":
                m = self.modules[node.target]
                if isinstance(m, nn.Conv2d):
                    src = node.args[0]
                    first = isinstance(src, fx.Node) and src.op == "placeholder"
                    if not cv.supported(m) and not (first and node.target in self.tap_names and m.out_channels % 8 == 0):
                        return f"{node.target}: convolution outside the implicit-GEMM kernel's coverage"
                    n_conv += 1
                elif isinstance(m, (nn.MaxPool2d, nn.AvgPool2d, nn.BatchNorm1d)):
                    return f"{node.target}: {type(m).__name__} has no NHWC rule"
                elif isinstance(m, nn.AdaptiveAvgPool2d) and tuple(self._pair2(m.output_size)) != (1, 1):
                    return f"{node.target}: adaptive pooling to more than one cell"
            elif node.op == "call_function":
                if node.target in (F.max_pool2d, F.avg_pool2d, torch.mean, operator.getitem):
                    return f"{getattr(node.target, '__name__', node.target)} has no NHWC rule"
            elif node.op == "call_method" and node.target in ("mean", "size"):
                return f"method {node.target} has no NHWC rule"
        if not n_conv:
            return "no convolution in the graph"
        return self._region_check()

    # The reverse sweep must not fail half-way (`on_tap` has already added G factors by then): every structural
    # condition `backward` would raise on is checked here, on the graph alone, so that such a model runs through the
    # parent class's NCHW sweep from the start.
    def _is_reshape(self, node) -> bool:
        if node.op == "call_module":
            return isinstance(self.modules[node.target], nn.Flatten)
        if node.op == "call_function":
            return node.target is torch.flatten
        return node.op == "call_method" and node.target in ("view", "reshape", "flatten")

    def _is_global_pool(self, node) -> bool:
        if node.op == "call_module":
            return isinstance(self.modules[node.target], nn.AdaptiveAvgPool2d)
        return node.op == "call_function" and node.target is F.adaptive_avg_pool2d

    def _is_feature_op(self, node) -> bool:
        """produces / consumes NHWC feature maps"""
        if node.op == "call_module" and isinstance(self.modules[node.target], (nn.Conv2d, nn.BatchNorm2d)):
            return True
        return self._is_global_pool(node)

    def _passes_through(self, node) -> bool:
        """shape-preserving nodes: the cotangent keeps the representation it arrives in"""
        if node.op == "call_module":
            return isinstance(self.modules[node.target], (nn.ReLU, nn.Tanh, nn.Sigmoid, nn.Identity, nn.Dropout)
                              + self._GENERIC_ACT_MODULES)
        if node.op == "call_function":
            return (node.target in self._ELEMENTWISE_FN or node.target in self._GENERIC_ACT_FN
                    or node.target in (operator.add, torch.add, operator.iadd))
        return node.op == "call_method" and node.target in ("relu", "tanh", "sigmoid", "contiguous")

    def _walk(self, start, downstream: bool):
        """nodes reachable from ``start`` through shape-preserving nodes (users if ``downstream`` else inputs)"""
        seen, todo, out = set(), [start], []
        while todo:
            n = todo.pop()
            nxt = list(n.users) if downstream else [a for a in n.all_input_nodes]
            for m in nxt:
                if m in seen:
                    continue
                seen.add(m)
                out.append(m)
                if self._passes_through(m):
                    todo.append(m)
        return out

    def _region_check(self):
        for node in self.gm.graph.nodes:
            if self._is_reshape(node) and any(self._is_feature_op(u) for u in self._walk(node, True)):
                return f"{node.name}: view / reshape / flatten whose result is used as a feature map"
            if self._is_global_pool(node):
                users = list(node.users)
                if len(users) != 1 or not self._is_reshape(users[0]):
                    return f"{node.name}: pooled tensor with a consumer other than one flatten"
            if node.op == "call_module" and isinstance(self.modules[node.target], nn.Linear):
                if any(self._is_feature_op(a) and not self._is_global_pool(a) or a.op == "placeholder"
                       for a in self._walk(node, False)):
                    return f"{node.target}: Linear layer applied to a feature map"
        return None

    # ---- forward: own convolution + fused BatchNorm/add/activation kernels on NHWC -----------------------------------
    #: ``False``: the forward stays on the library's NCHW convolutions
    nhwc_forward = True

    @torch.no_grad()
    def forward(self, x, need_vjp: bool = True, keep_tap_splits: bool = False):
        """``keep_tap_splits``: keep the NHWC split copy of every tapped convolution's input (``tap_splits``) until
        ``release()`` — only the Kron predictive's eigenbasis rotation reads it; a fit would hold a second
        activation-sized copy through the whole reverse sweep for nothing."""
        self._keep_tap_splits = keep_tap_splits
        # data_ptr of a feature map produced here -> {"in_amax": [B] words, "mul": word, "add": word | None} (a convolution's
        # output: a per-image bound without a pass over it) / {"split": SplitTensor, "bound": [B] words} (an activation)
        self._aux = {}
        self.tap_splits = {}  # tap name -> NHWC SplitTensor of the tap's input (what its forward convolution consumed)
        self._fwd_words = None
        try:
            return super().forward(x, need_vjp)
        finally:
            self._aux = {}  # (the split copies of the activations are only needed while the forward runs)

    def _aux_put(self, t, entry):
        """register what is known about the feature map ``t`` (keyed by address, OWNED by the tensor object: a map that was
        freed — a materialised convolution output behind its BatchNorm launch — may hand its address to a later tensor that
        registers nothing, which must not inherit a stale per-image bound: too small a bound saturates fp16 planes silently)"""
        entry["_owner"] = weakref.ref(t)
        self._aux[t.data_ptr()] = entry

    def _aux_get(self, t):
        entry = self._aux.get(t.data_ptr())
        if entry is not None and entry["_owner"]() is not t:
            del self._aux[t.data_ptr()]
            return None
        return entry

    def _fwd_word(self, dev, n=1):
        """``n`` zeroed device words out of a per-forward pool (one fill launch per pool)"""
        if self._fwd_words is None or self._fwd_words[1] + n > self._fwd_words[0].numel():
            self._fwd_words = [torch.zeros(max(64, 32 * n), dtype=torch.float32, device=dev), 0]
        w = self._fwd_words[0][self._fwd_words[1]:self._fwd_words[1] + n]
        self._fwd_words[1] += n
        return w

    def _use_nhwc_forward(self, t) -> bool:
        return self.split_ok and self.nhwc_forward and torch.is_tensor(t) and t.dim() == 4 and t.dtype == torch.float32 \
            and t.shape[0] > 0

    def _split_input(self, inp, pad_to=None):
        aux = self._aux_get(inp)
        if aux is not None and "split" in aux and aux["split"] is not None and pad_to is None:
            return aux["split"]
        K = self.kernels()
        xh = inp.permute(0, 2, 3, 1).contiguous()  # (a view when inp is already NHWC in memory)
        if pad_to is not None and xh.shape[-1] != pad_to:
            xp = xh.new_zeros(*xh.shape[:3], pad_to)
            xp[..., :xh.shape[-1]] = xh
            xh = xp
        return K.split_images_f16x2(xh)  # one scale per image (lk_split_images_f16x2)

    def _run_conv(self, node, m, inp):
        if not (self._use_nhwc_forward(inp) and cv.forward_supported(m)):
            return m(inp)
        prep = self._prep.get(node.target)
        if prep is None:
            prep = self._prep[node.target] = cv.PreparedConv(m)
        xs = self._split_input(inp, pad_to=prep.padded_in if prep.padded_in != m.in_channels else None)
        if self._keep_tap_splits and node.target in self.tap_names and prep.padded_in == m.in_channels:
            # consumers of the tap's input that run our convolution on it again (the Kron predictive's eigenbasis
            # rotation) take the split copy instead of measuring and splitting the activation a second time
            self.tap_splits[node.target] = xs

        def run():
            out = cv.conv_forward(prep, xs)
            if m.bias is not None:
                out += m.bias
            y = out.permute(0, 3, 1, 2)  # logical [B, C, H, W] over NHWC memory
            if xs.amax is not None:
                # per-image bound of the output without a pass over it: measured max of the input image * l1(W) + max|bias|
                l1, bmax = prep.forward_l1()
                self._aux_put(y, {"in_amax": xs.amax, "mul": l1, "add": bmax})
            return y

        if self._bn_takes_conv(node, m, xs):
            s_, (ph, pw), (KH, KW) = m.stride[0], m.padding, m.kernel_size
            Ho, Wo = (xs.shape[1] + 2 * ph - KH) // s_ + 1, (xs.shape[2] + 2 * pw - KW) // s_ + 1
            return _PendingConv(prep, xs, (xs.shape[0], m.out_channels, Ho, Wo), run)
        return run()

    #: ``False``: a convolution and the BatchNorm / add / ReLU behind it stay two launches
    fuse_conv_bn = True

    def _bn_takes_conv(self, node, m, xs) -> bool:
        """does this convolution's output go to exactly one consumer, an eval-mode BatchNorm2d that the traced forward
        hands to ``_run_bn_act`` (laplace_amd/sweep.py: the conditions of its BatchNorm branch), so that both run as one
        launch?"""
        K = self.kernels()
        if not (self.fuse_conv_bn and getattr(K, "use_conv_bn_act", False) and m.bias is None and xs.amax is not None
                and xs.shape[0] <= getattr(K, "MAX_IMAGES_PER_LAUNCH", 65535) and m.out_channels % 8 == 0
                and len(node.users) == 1):
            return False
        nxt = next(iter(node.users))
        if not (nxt.op == "call_module" and len(nxt.args) == 1 and nxt.args[0] is node and not nxt.kwargs):
            return False
        bn = self.modules.get(nxt.target)
        return isinstance(bn, nn.BatchNorm2d) and bn.running_var is not None and nxt.target not in self.tap_names

    def _run_bn_act(self, node, inp, scale, shift, relu, addend, want_mask):
        K = self.kernels()
        if isinstance(inp, _PendingConv):
            if addend is None or (torch.is_tensor(addend) and addend.dtype == torch.float32 and K.is_channels_last(addend)
                                  and addend.shape == inp.shape):
                a_h = a_bound = None
                if addend is not None:
                    a_h = addend.permute(0, 2, 3, 1)
                    a_aux = self._aux_get(addend)
                    a_bound = a_aux["bound"] if a_aux is not None and "bound" in a_aux else K.absmax(a_h)
                scale = scale.to(torch.float32).contiguous()
                shift = shift.to(torch.float32).contiguous()
                # `act_sink` (the KFAC accumulator): a tapped 3x3 convolution that reads this output may want it written straight
                # into its pixel-pair stack — the copy it would otherwise make of every activation (13 per c4 minibatch)
                y_out = None
                sink, fin = self.act_sink, getattr(self, "_group_out_node", None)
                if sink is not None and fin is not None:
                    for u in fin.users:
                        if u.op == "call_module" and u.target in self.tap_names and u.args and u.args[0] is fin:
                            y_out = sink(u.target, (inp.shape[0], inp.shape[2], inp.shape[3], inp.shape[1]), inp.device)
                            if y_out is not None:
                                break
                y, mask, split, bound = cv.conv_forward_bn_act(
                    inp.prep, inp.xs, scale, shift, self._amax_of((node.target, "s"), scale),
                    self._amax_of((node.target, "t"), shift), 1 if relu else 0, addend=a_h, addend_bound=a_bound,
                    want_mask=want_mask, amax_words=self._fwd_word(inp.device, inp.shape[0]), y_out=y_out)
                out = y.permute(0, 3, 1, 2)
                self._aux_put(out, {"split": split, "bound": split.amax if split is not None else bound})
                if mask is not None:
                    mask = mask.view(torch.bool).permute(0, 3, 1, 2)
                return out, mask
            inp = inp.materialize()
        if not (self._use_nhwc_forward(inp) and K.is_channels_last(inp) and inp.shape[1] % 8 == 0
                and (addend is None or K.is_channels_last(addend))):
            return super()._run_bn_act(node, inp, scale, shift, relu, addend, want_mask)
        aux = self._aux_get(inp)
        xh = inp.permute(0, 2, 3, 1)
        x_mul = x_add = None
        if aux is not None and "in_amax" in aux:
            x_amax, x_mul, x_add = aux["in_amax"], aux["mul"], aux["add"]
        else:
            x_amax = K.absmax(xh)  # (one bound for every image: coarser scales, still guaranteed)
        a_h = a_bound = None
        if addend is not None:
            a_h = addend.permute(0, 2, 3, 1)
            a_aux = self._aux_get(addend)
            a_bound = a_aux["bound"] if a_aux is not None and "bound" in a_aux else K.absmax(a_h)
        scale = scale.to(torch.float32).contiguous()
        shift = shift.to(torch.float32).contiguous()
        y, mask, split, bound = K.bn_act_forward_nhwc(xh, x_amax, scale, shift, self._amax_of((node.target, "s"), scale),
                                                      self._amax_of((node.target, "t"), shift), 1 if relu else 0,
                                                      addend=a_h, addend_bound=a_bound, want_mask=want_mask, x_mul=x_mul,
                                                      x_add=x_add, amax_words=self._fwd_word(inp.device, inp.shape[0]))
        out = y.permute(0, 3, 1, 2)
        # (the MEASURED per-image maxima are the bound a later residual join adds: tighter than the guaranteed one)
        self._aux_put(out, {"split": split, "bound": split.amax if split is not None else bound})
        if mask is not None:
            mask = mask.view(torch.bool).permute(0, 3, 1, 2)  # logical NCHW view of the NHWC mask bytes
        return out, mask

    # ---- helpers ----------------------------------------------------------------------------------------------------
    def _amax_of(self, key, t):
        """device word with max|t| of a per-model constant (BatchNorm scale), cached until it changes"""
        k = (t._version, t.data_ptr())
        hit = self._amax_cache.get(key)
        if hit is None or hit[0] != k:
            hit = (k, self.kernels().absmax(t.contiguous()))
            self._amax_cache[key] = hit
        return hit[1]

    def _nhwc_mult(self, node, kind):
        """per-sample multiplier of an activation as an NHWC tensor (+ its max| | word for generic derivatives)"""
        hit = self._mult_cache.get(node)
        if hit is None:
            saved = self.saved[node]
            mult = self._act_mult(kind, saved)
            if mult.dim() != 4:
                raise SweepUnsupported("activation on a non-feature-map tensor inside the NHWC region")
            amax = None
            if mult.dtype == torch.bool:
                mult = mult.permute(0, 2, 3, 1).contiguous().view(torch.uint8)
            else:
                mult = mult.to(torch.float32).permute(0, 2, 3, 1).contiguous()
                generic = isinstance(kind, self._GENERIC_ACT_MODULES) or (callable(kind) and kind in self._GENERIC_ACT_FN)
                if generic:  # tanh' and sigmoid' are bounded by 1; anything else is measured
                    amax = self.kernels().absmax(mult)
            hit = (mult, amax)
            self._mult_cache[node] = hit
        return hit

    def _to_split(self, parts, S, mult=None, mult_amax=None, scale=None, scale_amax=None):
        """sum of cotangent parts (x multiplier x channel scale) -> one SplitTensor"""
        K = self.kernels()
        lazy = [p for p in parts if isinstance(p, _LazyConv)]
        if lazy:
            rest = [p for p in parts if not isinstance(p, _LazyConv)]
            strided = [p for p in lazy if isinstance(p, _LazyStrided)]
            if (strided and len(strided) == len(lazy) <= 2 and all(p._done is None and p.hw == strided[0].hw for p in strided)
                    and len(rest) <= 1 and all(isinstance(p, SplitTensor) for p in rest)
                    and cv.strided_taps([p.desc for p in strided], strided[0].hw) is not None):
                return cv.conv_backward_data_vjp_strided(
                    [p.desc for p in strided], strided[0].hw, add=rest[0] if rest else None, mult=mult, mult_amax=mult_amax,
                    scale=scale, scale_amax=scale_amax, amax_word=self._new_word() if self._new_word is not None else None)
            if len(lazy) == 1 and not strided and lazy[0]._done is None and len(rest) <= 1 and all(isinstance(p, SplitTensor) for p in rest):
                return lazy[0].fused(add=rest[0] if rest else None, mult=mult, mult_amax=mult_amax, scale=scale,
                                     scale_amax=scale_amax,
                                     amax_word=self._new_word() if self._new_word is not None else None)
            parts = _materialize(parts)
        f32 = [p for p in parts if isinstance(p, _F32)]
        spl = [p for p in parts if isinstance(p, SplitTensor)]
        if len(f32) > 1:  # (no graph of the supported families gets here: two un-activated conv branches joining)
            t = f32[0].t
            for p in f32[1:]:
                t = t + p.t
            f32 = [_F32(t, K.absmax(t))]
        if len(spl) > 1:  # several split addends (rare: joins of more than two branches): fold them through fp32
            t = spl[0].float()
            for p in spl[1:]:
                t = t + p.float()
            if f32:
                t = t + f32[0].t
            t = t.contiguous()
            f32, spl = [_F32(t, K.absmax(t))], []
        g = f32[0] if f32 else None
        g2 = spl[0] if spl else None
        if g is None and mult is None and scale is None:
            return g2
        shape = g.t.shape if g is not None else g2.shape
        return K.vjp_nhwc_split(None if g is None else g.t, None if g is None else g.amax, g2, mult, mult_amax, scale,
                                scale_amax, S, tuple(shape))

    # ---- reverse sweep ------------------------------------------------------------------------------------------------
    _new_word = None  # zeroed device words of the running sweep (one fill per 64 of them)

    @torch.no_grad()
    def backward(self, seeds, on_tap=None, defer_bn_scale: bool = False, keep_split: bool = False):
        """``keep_split``: hand conv-tap gradients back as NHWC SplitTensors (consumers with their own kernels for
        them) instead of converting to ``[S, B, C, H, W]`` fp32."""
        if not self.split_ok:
            return super().backward(seeds, on_tap=on_tap, defer_bn_scale=defer_bn_scale)
        K = self.kernels()
        self.grad_scale = {}
        self._mult_cache = {}
        pending_scale: dict[fx.Node, torch.Tensor] = {}
        deferred: dict[fx.Node, list] = {}
        S, B = seeds.shape[0], seeds.shape[1]
        SB = S * B
        cot: dict[fx.Node, list] = {self.out_node: [seeds.reshape(SB, *seeds.shape[2:])]}
        grads: dict = {}
        remaining = set(self.tap_names)
        words = torch.zeros(64, dtype=torch.float32, device=seeds.device)  # max|.| words of this sweep's conv outputs
        word_i = [0]

        def new_word():
            nonlocal words
            if word_i[0] == words.numel():
                words = torch.zeros(64, dtype=torch.float32, device=seeds.device)
                word_i[0] = 0
            w = words[word_i[0]:word_i[0] + 1]
            word_i[0] += 1
            return w

        self._new_word = new_word

        def push(n, part):
            if not isinstance(n, fx.Node) or n.op == "placeholder":
                return
            if isinstance(part, _LazyConv) and n in deferred:
                part = part.f32()
            if isinstance(part, _F32) and n in deferred:
                for fn in deferred.pop(n):  # strided 1x1 branches waiting for the main branch's tensor: add into it
                    fn(part)
            cot.setdefault(n, []).append(part)

        def feature_f32(t4_nchw_like):
            """fp32 [S*B, C, H, W]-shaped tensor -> NHWC _F32"""
            t = t4_nchw_like.permute(0, 2, 3, 1).contiguous()
            return _F32(t, K.absmax(t))

        def is4(node):
            shp = self.saved.get(node)
            return shp is not None

        for node in reversed(list(self.gm.graph.nodes)):
            if node.op in ("placeholder", "output"):
                continue
            if node in deferred and node not in cot:
                # nothing else reached this node: the deferred branches produce the cotangent on their own
                fns = deferred.pop(node)
                part = fns[0](None)
                for fn in fns[1:]:
                    fn(part)
                cot[node] = [part]
            if node not in cot:
                continue
            parts = cot.pop(node)
            if node in deferred or not self._consumes_lazily(node):
                parts = _materialize(parts)
            if node in deferred:
                f32p = [p for p in parts if isinstance(p, _F32)]
                fns = deferred.pop(node)
                if f32p:
                    for fn in fns:
                        fn(f32p[0])
                else:
                    part = fns[0](None)
                    for fn in fns[1:]:
                        fn(part)
                    parts.append(part)
            flat = all(torch.is_tensor(p) for p in parts)  # still in the head (Linear / flatten) region
            if node.op == "call_module":
                m = self.modules[node.target]
                src = node.args[0]
                if isinstance(m, nn.Conv2d):
                    g = self._to_split(parts, S)
                    if node.target in self.tap_names:
                        grads[node.target] = g
                        if node in pending_scale:
                            self.grad_scale[node.target] = pending_scale[node]
                        if on_tap is not None:
                            on_tap(node.target, g)
                        remaining.discard(node.target)
                        if not remaining:
                            break
                    if not (isinstance(src, fx.Node) and src.op != "placeholder"):
                        continue
                    prep = self._prep.get(node.target)
                    if prep is None:
                        prep = self._prep[node.target] = cv.PreparedConv(m)
                    in_shape = self.saved[node]  # [B, Cin, Hin, Win]
                    hw = (int(in_shape[2]), int(in_shape[3]))
                    cscale = pending_scale.get(node)
                    sparse = any(not p[4] for p in cv.backward_plan(m, *hw))

                    def run(into, g=g, prep=prep, hw=hw, cscale=cscale):
                        if into is None:
                            w = new_word()
                            out = cv.conv_backward_data(prep, g, hw, cscale=cscale, amax_out=w)
                            return _F32(out, w)
                        cv.conv_backward_data(prep, g, hw, cscale=cscale, out=into.t, accumulate=True, amax_out=into.amax)
                        return into

                    if (self.fuse_vjp and self.fuse_strided and cv.strided_fused_ok(m, hw) and src not in deferred
                            and hasattr(K, "conv_nhwc_f16x2_vjp_strided")
                            and all(isinstance(p, (_LazyStrided, SplitTensor)) for p in cot.get(src, []))):
                        # (the consumer of the cotangent decides: alone or with the block's other strided branch in one
                        # fused launch, or — something else joined — class by class into an fp32 tensor)
                        push(src, _LazyStrided(run, (prep, g, cscale), hw))
                        continue
                    if src in cot:
                        cot[src] = _materialize(cot[src])
                    existing = [p for p in cot.get(src, []) if isinstance(p, _F32)]
                    if existing:
                        run(existing[0])
                    elif sparse and len(src.users) > 1:
                        deferred.setdefault(src, []).append(run)  # wait for the dense branch, then add into it
                    elif self.fuse_vjp and cv.fused_backward_ok(m) and hasattr(K, "conv_nhwc_f16x2_vjp"):
                        def run_fused(g=g, prep=prep, hw=hw, cscale=cscale, **kw):
                            return cv.conv_backward_data_vjp(prep, g, hw, cscale=cscale, **kw)

                        push(src, _LazyConv(lambda run=run: run(None), run_fused))
                    else:
                        push(src, run(None))
                elif isinstance(m, nn.Linear):
                    g = parts[0] if len(parts) == 1 else sum(parts[1:], parts[0])
                    if not torch.is_tensor(g):
                        raise SweepUnsupported("Linear layer inside the NHWC region")
                    if node.target in self.tap_names:
                        grads[node.target] = g.reshape(S, B, *g.shape[1:])
                        if on_tap is not None:
                            on_tap(node.target, grads[node.target])
                        remaining.discard(node.target)
                        if not remaining:
                            break
                    push(src, g @ m.weight)
                elif isinstance(m, nn.BatchNorm2d):
                    scale = self._bn_scale(node.target, m)
                    if (defer_bn_scale and isinstance(src, fx.Node) and src.op == "call_module" and len(src.users) == 1
                            and isinstance(self.modules[src.target], nn.Conv2d) and src.target in self.tap_names
                            and src not in cot and len(parts) == 1 and isinstance(parts[0], SplitTensor)):
                        pending_scale[src] = scale
                        push(src, parts[0])
                    else:
                        push(src, self._to_split(parts, S, scale=scale, scale_amax=self._amax_of(node.target, scale)))
                elif isinstance(m, (nn.ReLU, nn.Tanh, nn.Sigmoid) + self._GENERIC_ACT_MODULES):
                    self._activation(node, m, src, parts, S, push, flat)
                elif isinstance(m, (nn.Identity, nn.Dropout)):
                    for p in parts:
                        push(src, p)
                elif isinstance(m, nn.Flatten):
                    self._unflatten(node, src, parts, SB, push, feature_f32)
                elif isinstance(m, nn.AdaptiveAvgPool2d):
                    self._global_pool(node, src, parts, SB, push, K)
                else:
                    raise SweepUnsupported(f"no NHWC rule for {type(m).__name__}")
            elif node.op == "call_function":
                t = node.target
                if t in (operator.add, torch.add, operator.iadd):
                    for a in node.args[:2]:
                        for p in parts:
                            push(a, p)
                elif t in self._ELEMENTWISE_FN or t in self._GENERIC_ACT_FN:
                    self._activation(node, t, node.args[0], parts, S, push, flat)
                elif t is torch.flatten:
                    self._unflatten(node, node.args[0], parts, SB, push, feature_f32)
                elif t is F.adaptive_avg_pool2d:
                    self._global_pool(node, node.args[0], parts, SB, push, K)
                else:
                    raise SweepUnsupported(f"no NHWC rule for {getattr(t, '__name__', t)}")
            elif node.op == "call_method":
                t = node.target
                if t in ("relu", "tanh", "sigmoid"):
                    self._activation(node, t, node.args[0], parts, S, push, flat)
                elif t in ("view", "reshape", "flatten"):
                    self._unflatten(node, node.args[0], parts, SB, push, feature_f32)
                elif t == "contiguous":
                    for p in parts:
                        push(node.args[0], p)
                else:
                    raise SweepUnsupported(f"no NHWC rule for method {t}")
        self._new_word = None
        if remaining:
            raise SweepUnsupported(f"no cotangent reached {sorted(remaining)}")
        if on_tap is None and not keep_split:
            # consumers that read the gradients as tensors (Jacobians, diagonal, predictive): [S, B, C, H, W] fp32
            for name, g in list(grads.items()):
                if isinstance(g, SplitTensor):
                    grads[name] = g.float().reshape(S, B, *g.shape[1:]).permute(0, 1, 4, 2, 3).contiguous()
        return grads

    # ---- node rules ---------------------------------------------------------------------------------------------------
    def _activation(self, node, kind, src, parts, S, push, flat):
        if flat:  # activation in the head region (MLP head after the flatten): parent-class math on plain tensors
            g = parts[0] if len(parts) == 1 else sum(parts[1:], parts[0])
            scale, dst = self._fold_bn(src)
            push(dst, self._scale_mask(g, S, self._act_mult(kind, self.saved[node]), scale))
            return
        mult, mult_amax = self._nhwc_mult(node, kind)
        scale, dst = self._fold_bn(src)
        scale_amax = None
        if scale is not None:
            scale_amax = self._amax_of(src.target, scale)
        push(dst, self._to_split(parts, S, mult=mult, mult_amax=mult_amax, scale=scale, scale_amax=scale_amax))

    def _unflatten(self, node, src, parts, SB, push, feature_f32):
        g = parts[0] if len(parts) == 1 else sum(parts[1:], parts[0])
        if not torch.is_tensor(g):
            raise SweepUnsupported("reshape inside the NHWC region")
        shp = tuple(self.saved[node])
        g = g.reshape((SB,) + shp[1:])
        push(src, feature_f32(g) if g.dim() == 4 else g)

    def _global_pool(self, node, src, parts, SB, push, K):
        shp = self.saved[node]  # [B, C, H, W]
        p = parts[0]
        if len(parts) != 1 or not isinstance(p, _F32):
            raise SweepUnsupported("global average pooling expects one fp32 cotangent")
        H, W = int(shp[-2]), int(shp[-1])
        t = (p.t / (H * W)).expand(SB, H, W, p.t.shape[-1]).contiguous()
        push(src, _F32(t, K.absmax(t)))
