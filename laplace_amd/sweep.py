"""Seed-batched reverse sweep: all C-1 likelihood-Hessian seeds in ONE pass with batch ``S*B``.

Stock autograd gives the per-seed output gradients only one seed at a time on ROCm (functorch has
no fused batching rule for MIOpen's convolution backward, so ``is_grads_batched`` loops and then
concatenates).  For a ResNet-18 minibatch that is ~120 small conv launches and ~550 tiny
element-wise kernels per step.  This module is the "fused host-side extraction" of SURVEY.md §8f:

* ``torch.fx`` traces the model once into a graph of modules / functions;
* the forward is executed node by node on the ``B`` samples, keeping only what a vector-Jacobian
  product needs (ReLU masks, pooling indices, BatchNorm-eval scales, ...);
* the reverse sweep pushes a cotangent of batch ``S*B`` (seed-major) through closed-form VJP rules:
  one MIOpen backward-data call per conv for *all* seeds, one element-wise kernel per activation.

It produces exactly what :class:`laplace_amd.capture.Tape` produces (layer inputs ``a`` and output
gradients ``g`` per tapped module), so everything downstream is unchanged.  Unsupported graphs
(untraceable control flow, modules in training mode, ops without a rule here) raise
:class:`SweepUnsupported` and the caller falls back to the autograd tape.  The model still runs on
stock PyTorch-ROCm kernels — this is host-side plumbing, not a replacement for them.
"""
from __future__ import annotations

import operator
from typing import Any

import torch
import torch.fx as fx
import torch.nn.functional as F
from torch import nn


class SweepUnsupported(RuntimeError):
    pass


class SeedBatchedSweep:
    """Forward + seed-batched reverse sweep over an fx-traced module."""

    _ELEMENTWISE_FN = {torch.relu, F.relu, torch.tanh, F.tanh, torch.sigmoid, F.sigmoid}
    # any other element-wise activation: its per-sample derivative is taken ONCE from autograd on the [B, ...]
    # forward value (1/S of the sweep's work) and then applied to all seeds by the same fused kernel
    _GENERIC_ACT_MODULES = (nn.GELU, nn.SiLU, nn.LeakyReLU, nn.ELU, nn.Softplus, nn.Hardtanh, nn.ReLU6, nn.Mish,
                            nn.Hardswish, nn.Hardsigmoid, nn.SELU, nn.CELU, nn.Softsign, nn.LogSigmoid)
    _GENERIC_ACT_FN = {F.gelu, F.silu, F.leaky_relu, F.elu, F.softplus, F.hardtanh, F.relu6, F.mish, F.hardswish,
                       F.hardsigmoid, F.selu, F.celu, F.softsign, F.logsigmoid}

    _POOL_FN = {F.max_pool2d, F.avg_pool2d, F.adaptive_avg_pool2d}

    @staticmethod
    def _with_derivative(fn, inp):
        """(fn(inp), d fn / d inp) for an element-wise ``fn``"""
        with torch.enable_grad():
            xin = inp.detach().requires_grad_(True)
            out = fn(xin)
            (d,) = torch.autograd.grad(out.sum(), xin)
        return out.detach(), d

    def __init__(self, model: nn.Module, tap_modules: dict[str, nn.Module], kernels=None):
        """``kernels``: callable returning the kernel object (``laplace_amd._lib.get_kernels``) whose
        ``vjp_scale_mask`` applies the element-wise VJPs to all seeds in one launch (lk_vjp.hip); ``None``
        = plain torch math (reference implementation of the same rule, used by the CPU tests)."""
        self.kernels = kernels
        self._bn_cache: dict[str, tuple] = {}
        self._w_cache: dict[str, tuple] = {}
        try:
            self.gm = fx.symbolic_trace(model)
        except Exception as e:  # data-dependent control flow, non-tensor inputs, ...
            raise SweepUnsupported(f"torch.fx cannot trace the model: {e}") from e
        self.modules = dict(self.gm.named_modules())
        self.tap_names = set(tap_modules)
        self._check_graph()

    # ---- static checks -------------------------------------------------------------------------------
    def _check_graph(self):
        n_inputs = 0
        for node in self.gm.graph.nodes:
            if node.op == "placeholder":
                n_inputs += 1
            elif node.op == "call_module":
                m = self.modules[node.target]
                if isinstance(m, self._GENERIC_ACT_MODULES):
                    continue
                if not isinstance(m, (nn.Conv2d, nn.Linear, nn.BatchNorm2d, nn.BatchNorm1d, nn.ReLU, nn.Tanh,
                                      nn.Sigmoid, nn.Identity, nn.Dropout, nn.Flatten, nn.AdaptiveAvgPool2d,
                                      nn.MaxPool2d, nn.AvgPool2d, nn.Sequential)):
                    raise SweepUnsupported(f"no VJP rule for module {type(m).__name__} ({node.target})")
                if isinstance(m, nn.Conv2d) and (m.groups != 1 or isinstance(m.padding, str) or m.padding_mode != "zeros"):
                    raise SweepUnsupported(f"{node.target}: unsupported convolution variant")
            elif node.op == "call_function":
                if node.target not in (self._ELEMENTWISE_FN | self._GENERIC_ACT_FN | self._POOL_FN
                                       | {operator.add, torch.add, torch.flatten, operator.iadd, operator.getitem,
                                          torch.mean}):
                    raise SweepUnsupported(f"no VJP rule for function {getattr(node.target, '__name__', node.target)}")
                if node.target in (operator.add, torch.add, operator.iadd) and node.kwargs.get("alpha", 1) != 1:
                    raise SweepUnsupported("add with alpha")  # (checked here: backward() must not fail half-way)
            elif node.op == "call_method":
                if node.target not in ("view", "reshape", "flatten", "relu", "tanh", "sigmoid", "contiguous", "size",
                                       "mean"):
                    raise SweepUnsupported(f"no VJP rule for method {node.target}")
            elif node.op == "get_attr":
                raise SweepUnsupported("graph reads attributes directly")
        if n_inputs != 1:
            raise SweepUnsupported("models with one tensor input only")

    # ---- forward ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, need_vjp: bool = True):
        """Returns ``f``; fills ``self.saved`` (per node: what the VJP needs) and ``self.taps[name]['a']``.
        ``need_vjp=False``: inference only (the feature pass of the last-layer flavours) — nothing is kept for a
        reverse sweep, only the tapped inputs."""
        mode_mods = self.__dict__.get("_mode_mods")
        if mode_mods is None:  # (the modules whose VJP rules assume eval mode; the graph does not change after tracing)
            mode_mods = self._mode_mods = [m for m in self.gm.modules() if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d, nn.Dropout))]
        if self.gm.training or any(m.training for m in mode_mods):
            raise SweepUnsupported("model must be in eval mode (BatchNorm / Dropout VJPs assume it)")
        env: dict[fx.Node, Any] = {}
        self.saved: dict[fx.Node, Any] = {}
        self.taps: dict[str, dict] = {}
        self.out_node = None
        fused_relu: dict[fx.Node, tuple] = {}  # ReLU node -> (output, mask) already produced by the BatchNorm kernel
        self.max_act_numel = 1  # largest per-sample activation: bounds the memory of a seed-batched cotangent
        for node in self.gm.graph.nodes:
            if node in fused_relu:
                env[node], keep = fused_relu.pop(node)
                if keep is not None:
                    self.saved[node] = keep
                continue
            if node.op == "placeholder":
                env[node] = x
            elif node.op == "output":
                self.out_node = node.args[0]
                if not isinstance(self.out_node, fx.Node):
                    raise SweepUnsupported("model must return a single tensor")
            elif node.op == "call_module":
                m = self.modules[node.target]
                inp = env[node.args[0]]
                if isinstance(m, nn.MaxPool2d):
                    out, idx = F.max_pool2d(inp, m.kernel_size, m.stride, m.padding, m.dilation, m.ceil_mode, True)
                    self.saved[node] = (idx, inp.shape)
                elif isinstance(m, self._GENERIC_ACT_MODULES):
                    out, self.saved[node] = self._with_derivative(m, inp) if need_vjp else (m(inp), None)
                elif (isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)) and self.kernels is not None and inp.dim() >= 2
                      and inp.dtype == torch.float32 and m.running_var is not None):
                    # eval-mode BatchNorm = per-channel affine map: one fused launch, with the ReLU that follows it
                    # (and the mask its VJP needs) when the BatchNorm output has no other consumer
                    scale, shift = self._bn_scale(node.target, m), self._bn_shift(node.target, m)
                    nxt = next(iter(node.users)) if len(node.users) == 1 else None
                    addend, add_node = None, None
                    if nxt is not None and self._is_plain_add(nxt):
                        # residual join `bn(..) + other`: folded in when the other branch is already available
                        other = nxt.args[1] if nxt.args[0] is node else nxt.args[0]
                        if isinstance(other, fx.Node) and other in env and torch.is_tensor(env[other]) \
                                and env[other].shape == inp.shape and env[other].dtype == inp.dtype:
                            addend, add_node = env[other], nxt
                            nxt = next(iter(add_node.users)) if len(add_node.users) == 1 else None
                    relu = nxt is not None and self._is_plain_relu(nxt) and nxt.args[0] is (add_node or node)
                    self._group_out_node = nxt if relu else (add_node or node)  # (whose users read this launch's output)
                    out, mask = self._run_bn_act(node, inp, scale, shift, relu, addend, need_vjp)
                    if add_node is not None:
                        fused_relu[add_node] = (out, None)  # (the add node itself keeps nothing for its VJP)
                    if relu:
                        fused_relu[nxt] = (out, mask)
                else:
                    out = self._run_conv(node, m, inp) if isinstance(m, nn.Conv2d) else m(inp)
                    if isinstance(m, (nn.ReLU,)):
                        self.saved[node] = (out > 0) if need_vjp else None
                    elif isinstance(m, (nn.Tanh, nn.Sigmoid)):
                        self.saved[node] = out
                    elif isinstance(m, (nn.AdaptiveAvgPool2d, nn.AvgPool2d, nn.Flatten)):
                        self.saved[node] = inp.shape
                    elif isinstance(m, nn.Conv2d):
                        self.saved[node] = inp.shape
                if node.target in self.tap_names:
                    if node.target in self.taps:
                        raise SweepUnsupported(f"{node.target}: module is applied more than once per forward")
                    self.taps[node.target] = {"a": inp, "node": node}
                env[node] = out
            elif node.op == "call_function":
                args = list(fx.node.map_arg(node.args, lambda n: env[n]))
                kwargs = dict(fx.node.map_arg(node.kwargs, lambda n: env[n]))
                if node.target is operator.getitem and torch.is_tensor(args[0]):
                    raise SweepUnsupported("tensor indexing")
                if node.target is F.max_pool2d:
                    p = self._bind(node, ("kernel_size", "stride", "padding", "dilation", "ceil_mode", "return_indices"),
                                   (None, None, 0, 1, False, False))
                    if p["return_indices"]:
                        raise SweepUnsupported("max_pool2d(return_indices=True)")
                    out, idx = F.max_pool2d(args[0], p["kernel_size"], p["stride"], p["padding"], p["dilation"],
                                            p["ceil_mode"], True)
                    self.saved[node] = (idx, args[0].shape)
                    env[node] = out
                    continue
                if node.target in (F.avg_pool2d, F.adaptive_avg_pool2d):
                    self.saved[node] = args[0].shape
                if node.target is torch.mean:
                    dim = kwargs.get("dim", args[1] if len(args) > 1 else None)
                    self.saved[node] = (args[0].shape, bool(kwargs.get("keepdim", args[2] if len(args) > 2 else False)),
                                        self._mean_dims(dim, args[0].dim()))
                if node.target is operator.iadd:
                    out = args[0] + args[1]
                elif node.target in self._GENERIC_ACT_FN:
                    kwargs.pop("inplace", None)
                    if need_vjp:
                        out, self.saved[node] = self._with_derivative(lambda t: node.target(t, *args[1:], **kwargs), args[0])
                    else:
                        out = node.target(*args, **kwargs)
                else:
                    out = node.target(*args, **kwargs)
                if node.target in (torch.relu, F.relu):
                    self.saved[node] = (out > 0) if need_vjp else None
                elif node.target in (torch.tanh, F.tanh, torch.sigmoid, F.sigmoid):
                    self.saved[node] = out
                elif node.target is torch.flatten:
                    self.saved[node] = args[0].shape
                env[node] = out
            elif node.op == "call_method":
                self_t = env[node.args[0]]
                args = list(fx.node.map_arg(node.args[1:], lambda n: env[n]))
                kwargs = dict(fx.node.map_arg(node.kwargs, lambda n: env[n]))
                out = getattr(self_t, node.target)(*args, **kwargs)
                if node.target == "mean":
                    dim = kwargs.get("dim", args[0] if args else None)
                    self.saved[node] = (self_t.shape, bool(kwargs.get("keepdim", args[1] if len(args) > 1 else False)),
                                        self._mean_dims(dim, self_t.dim()))
                if node.target == "relu":
                    self.saved[node] = (out > 0) if need_vjp else None
                elif node.target in ("tanh", "sigmoid"):
                    self.saved[node] = out
                elif node.target in ("view", "reshape", "flatten"):
                    self.saved[node] = self_t.shape
                env[node] = out
        if not need_vjp:
            self.saved = {}
        nb = max(int(x.shape[0]), 1)
        for v in env.values():
            if torch.is_tensor(v):
                self.max_act_numel = max(self.max_act_numel, v.numel() // nb)
        missing = self.tap_names - set(self.taps)
        if missing:
            raise SweepUnsupported(f"tapped modules not reached by the traced forward: {sorted(missing)}")
        out = env[self.out_node]
        self.out_shape = tuple(out.shape[1:])
        return out

    # ---- forward hooks (the NHWC sweep replaces them with its own kernels) -----------------------------------------
    def _run_conv(self, node, m, inp):
        return m(inp)

    def _run_bn_act(self, node, inp, scale, shift, relu, addend, want_mask):
        return self.kernels().bn_act_forward(inp.contiguous(), scale, shift, relu,
                                             None if addend is None else addend.contiguous(), want_mask=want_mask)

    @staticmethod
    def _bind(node, names, defaults):
        """positional / keyword arguments of a functional call -> dict (input excluded)"""
        vals = dict(zip(names, defaults))
        for n, a in zip(names, node.args[1:]):
            vals[n] = a
        for k, v in node.kwargs.items():
            if k in vals:
                vals[k] = v
        if any(isinstance(v, fx.Node) for v in vals.values()):
            raise SweepUnsupported("data-dependent pooling arguments")
        return vals

    @staticmethod
    def _pair2(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)

    @staticmethod
    def _avgpool_vjp(g, in_shape, kernel, stride, padding, SB, ceil_mode=False, count_include_pad=True,
                     divisor_override=None):
        """VJP of ``avg_pool2d`` for the whole seed batch: non-overlapping, unpadded windows that tile the input are an
        upsampling; every other geometry (padding, overlap, ragged edges, ceil_mode) goes through the pooling
        operator's own backward, which needs the input only for its shape."""
        k, st = SeedBatchedSweep._pair2(kernel), SeedBatchedSweep._pair2(kernel if stride in (None, []) else stride)
        if (SeedBatchedSweep._pair2(padding) == (0, 0) and k == st and in_shape[-2] % k[0] == 0
                and in_shape[-1] % k[1] == 0 and divisor_override is None):
            return g.repeat_interleave(k[0], -2).repeat_interleave(k[1], -1) / (k[0] * k[1])
        dummy = g.new_empty((SB,) + tuple(in_shape[1:]))
        return torch.ops.aten.avg_pool2d_backward(g.contiguous(), dummy, list(k), list(st),
                                                  list(SeedBatchedSweep._pair2(padding)), bool(ceil_mode),
                                                  bool(count_include_pad), divisor_override)

    @staticmethod
    def _adaptive_avgpool_vjp(g, in_shape, SB):
        if tuple(g.shape[-2:]) == (1, 1):
            return (g / (in_shape[-1] * in_shape[-2])).expand(SB, *in_shape[1:])
        return torch.ops.aten._adaptive_avg_pool2d_backward(g.contiguous(), g.new_empty((SB,) + tuple(in_shape[1:])))

    @staticmethod
    def _maxpool_vjp(g, idx, in_shape, S, B):
        idx_s = idx.unsqueeze(0).expand(S, *idx.shape).reshape(S * B, *idx.shape[1:])
        out = torch.zeros(S * B, in_shape[1], in_shape[2] * in_shape[3], dtype=g.dtype, device=g.device)
        out.scatter_add_(2, idx_s.reshape(S * B, in_shape[1], -1), g.reshape(S * B, in_shape[1], -1))
        return out.reshape(S * B, *in_shape[1:])

    @staticmethod
    def _mean_dims(dim, ndim):
        """sorted non-batch dims a ``mean`` reduces (spatial pooling of a conv map, pooling over the positions of a
        sequence, ...); the batch dim must survive"""
        if dim is None:
            raise SweepUnsupported("mean over all dims (the batch dim included)")
        dims = sorted({d % ndim for d in (dim if isinstance(dim, (tuple, list)) else (dim,))})
        if 0 in dims:
            raise SweepUnsupported("mean over the batch dim")
        return dims

    @staticmethod
    def _mean_vjp(g, saved, SB):
        shp, keep, dims = saved
        if not keep:
            for d in dims:
                g = g.unsqueeze(d)
        count = 1
        for d in dims:
            count *= shp[d]
        return (g / count).expand(SB, *shp[1:])

    def _scaled_weight(self, name, m, scale):
        """``scale[co] * W`` for a deferred BatchNorm scale (cached until the weight or the scale changes)"""
        if scale is None:
            return m.weight
        key = (m.weight._version, m.weight.data_ptr(), id(scale))
        hit = self._w_cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, (m.weight.detach() * scale.reshape(-1, 1, 1, 1)).contiguous(), scale)
            self._w_cache[name] = hit
        return hit[1]

    @staticmethod
    def _conv_input_grad(in_shape, m, g, weight=None):
        """Backward-data of a convolution for the whole seed batch.  ``torch.nn.grad.conv2d_input`` hands the op a
        stride-0 dummy input, from which PyTorch infers a channels-last result that then has to be copied back to
        NCHW (20 copies of [S*B, C, H, W] per ResNet-18 step); a contiguous, never-read dummy keeps it NCHW."""
        dummy = g.new_empty(in_shape)
        return torch.ops.aten.convolution_backward(g, dummy, m.weight if weight is None else weight, None, m.stride, m.padding, m.dilation, False,
                                                   [0] * len(m.stride), m.groups, [True, False, False])[0]

    # ---- element-wise VJPs ----------------------------------------------------------------------------------
    def _bn_scale(self, name: str, m) -> torch.Tensor:
        """gamma / sqrt(running_var + eps), cached until the module's buffers change."""
        key = (m.running_var._version, None if m.weight is None else m.weight._version, m.running_var.data_ptr())
        hit = self._bn_cache.get(name)
        if hit is None or hit[0] != key:
            scale = torch.rsqrt(m.running_var + m.eps)
            if m.weight is not None:
                scale = scale * m.weight.detach()
            hit = (key, scale)
            self._bn_cache[name] = hit
        return hit[1]

    def _is_activation(self, node) -> bool:
        if node.op == "call_module":
            return isinstance(self.modules[node.target], (nn.ReLU, nn.Tanh, nn.Sigmoid) + self._GENERIC_ACT_MODULES)
        if node.op == "call_function":
            return node.target in self._ELEMENTWISE_FN or node.target in self._GENERIC_ACT_FN
        return node.op == "call_method" and node.target in ("relu", "tanh", "sigmoid")

    def _bn_shift(self, name: str, m) -> torch.Tensor:
        """beta - running_mean * scale (cached like the scale)"""
        key = (m.running_mean._version, m.running_var._version, None if m.weight is None else m.weight._version,
               None if m.bias is None else m.bias._version, m.running_mean.data_ptr())
        hit = self._bn_cache.get(name + "/shift")
        if hit is None or hit[0] != key:
            shift = -m.running_mean * self._bn_scale(name, m)
            if m.bias is not None:
                shift = shift + m.bias.detach()
            hit = (key, shift)
            self._bn_cache[name + "/shift"] = hit
        return hit[1]

    @staticmethod
    def _is_plain_add(node) -> bool:
        return (node.op == "call_function" and node.target in (operator.add, torch.add, operator.iadd) and len(node.args) == 2
                and all(isinstance(a, fx.Node) for a in node.args) and node.kwargs.get("alpha", 1) == 1)

    def _is_plain_relu(self, node) -> bool:
        if node.op == "call_module":
            return isinstance(self.modules[node.target], nn.ReLU)
        if node.op == "call_function":
            return node.target in (torch.relu, F.relu)
        return node.op == "call_method" and node.target == "relu"

    def _scale_mask(self, g, S, mult, scale, g2=None):
        """``(g[s] + g2[s]) * mult * scale[channel]`` for all seeds (``g``: [S*B, C, ...], ``mult``: [B, C, ...])."""
        if mult is None and scale is None:
            return g if g2 is None else g + g2
        hw = 1
        for d in g.shape[2:]:
            hw *= d
        if self.kernels is not None:
            return self.kernels().vjp_scale_mask(g.contiguous(), S, mult, scale, hw,
                                                 None if g2 is None else g2.contiguous())
        out = g if g2 is None else g + g2
        if mult is not None:
            out = (out.reshape(S, *mult.shape) * mult).reshape(g.shape)
        if scale is not None:
            out = out * scale.reshape((1, -1) + (1,) * (g.dim() - 2))
        return out

    @classmethod
    def _act_mult(cls, kind, saved):
        """per-sample derivative of the activation from what the forward kept (ReLU: the mask itself)"""
        if isinstance(kind, cls._GENERIC_ACT_MODULES) or (callable(kind) and kind in cls._GENERIC_ACT_FN):
            return saved  # the derivative itself
        if kind in (torch.relu, F.relu, "relu") or isinstance(kind, nn.ReLU):
            return saved
        if kind in (torch.tanh, F.tanh, "tanh") or isinstance(kind, nn.Tanh):
            return 1 - saved * saved
        return saved * (1 - saved)  # sigmoid

    def _fold_bn(self, src):
        """If the activation's input is an eval-mode BatchNorm used only by it, its scale folds into the
        activation's VJP: returns (scale, node to push the cotangent to)."""
        if (isinstance(src, fx.Node) and src.op == "call_module" and len(src.users) == 1
                and isinstance(self.modules[src.target], (nn.BatchNorm2d, nn.BatchNorm1d))):
            return self._bn_scale(src.target, self.modules[src.target]), src.args[0]
        return None, src

    # ---- reverse sweep -----------------------------------------------------------------------------------
    @torch.no_grad()
    def backward(self, seeds: torch.Tensor, on_tap=None, defer_bn_scale: bool = False) -> dict[str, torch.Tensor]:
        """``seeds``: ``[S, B, C]`` cotangents of the output.  Returns per tapped module the gradient w.r.t.
        its output, ``[S, B, ...]`` (a view of the ``S*B``-batched cotangent).  ``on_tap(name, g)`` is called the
        moment a tapped module's gradient is complete — the accumulator uses it to start that layer's G-factor
        kernel on a side stream while the sweep goes on through the earlier layers.

        ``defer_bn_scale``: a conv whose output feeds ONLY an eval-mode BatchNorm that is not followed by an
        activation (``bn2`` / the down-sampling branch of a residual block) normally costs one full pass over the
        cotangent just to multiply by the per-channel scale ``s``.  Deferred, that pass disappears: the conv's
        backward-data uses the pre-scaled weights ``s[co] * W`` and the tap receives the UNSCALED gradient ``g`` with
        ``self.grad_scale[name] = s`` — the caller owes ``G <- diag(s) G diag(s)``, which a KFAC accumulator applies
        once per fit because ``s`` is constant."""
        self.grad_scale: dict[str, torch.Tensor] = {}
        pending_scale: dict[fx.Node, torch.Tensor] = {}
        S, B = seeds.shape[0], seeds.shape[1]
        # per node: the pending addends of its output cotangent (summed lazily, so that an activation can fold
        # the residual-branch addition into its own kernel)
        cot: dict[fx.Node, list] = {self.out_node: [seeds.reshape(S * B, *seeds.shape[2:])]}
        grads: dict[str, torch.Tensor] = {}
        remaining = set(self.tap_names)

        def push(n, g):
            if not isinstance(n, fx.Node) or n.op == "placeholder":
                return
            cot.setdefault(n, []).append(g)

        for node in reversed(list(self.gm.graph.nodes)):
            if node not in cot or node.op in ("placeholder", "output"):
                continue
            parts, g2 = cot.pop(node), None
            g = parts[0]
            if len(parts) > 1:
                if self._is_activation(node):
                    g2 = parts[1] if len(parts) == 2 else sum(parts[2:], parts[1])
                else:
                    g = sum(parts[1:], parts[0])
            if node.op == "call_module":
                m = self.modules[node.target]
                if node.target in self.tap_names:
                    grads[node.target] = g.reshape(S, B, *g.shape[1:])
                    if node in pending_scale:
                        self.grad_scale[node.target] = pending_scale[node]
                    if on_tap is not None:
                        on_tap(node.target, grads[node.target])
                    remaining.discard(node.target)
                    if not remaining:
                        break  # nothing upstream of the first tapped module is needed
                src = node.args[0]
                if isinstance(m, nn.Conv2d):
                    in_shape = (S * B,) + tuple(self.saved[node][1:])
                    push(src, self._conv_input_grad(in_shape, m, g, self._scaled_weight(node.target, m, pending_scale.get(node))))
                elif isinstance(m, nn.Linear):
                    push(src, g @ m.weight)
                elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                    scale = self._bn_scale(node.target, m)
                    if (defer_bn_scale and isinstance(src, fx.Node) and src.op == "call_module" and len(src.users) == 1
                            and isinstance(self.modules[src.target], nn.Conv2d) and src.target in self.tap_names
                            and src not in cot):
                        pending_scale[src] = scale  # the conv sees the unscaled cotangent (see the docstring)
                        push(src, g)
                    else:
                        push(src, self._scale_mask(g, S, None, scale))
                elif isinstance(m, (nn.ReLU, nn.Tanh, nn.Sigmoid) + self._GENERIC_ACT_MODULES):
                    scale, dst = self._fold_bn(src)
                    push(dst, self._scale_mask(g, S, self._act_mult(m, self.saved[node]), scale, g2))
                elif isinstance(m, (nn.Identity, nn.Dropout)):
                    push(src, g)
                elif isinstance(m, nn.Flatten):
                    push(src, g.reshape((S * B,) + tuple(self.saved[node][1:])))
                elif isinstance(m, nn.AdaptiveAvgPool2d):
                    push(src, self._adaptive_avgpool_vjp(g, self.saved[node], S * B))
                elif isinstance(m, nn.AvgPool2d):
                    push(src, self._avgpool_vjp(g, self.saved[node], m.kernel_size, m.stride, m.padding, S * B,
                                                m.ceil_mode, m.count_include_pad, m.divisor_override))
                elif isinstance(m, nn.MaxPool2d):
                    idx, shp = self.saved[node]
                    push(src, self._maxpool_vjp(g, idx, shp, S, B))
                elif isinstance(m, nn.Sequential):
                    raise SweepUnsupported("nested Sequential was not inlined by the tracer")
            elif node.op == "call_function":
                t = node.target
                if t in (operator.add, torch.add, operator.iadd):
                    if node.kwargs.get("alpha", 1) != 1:
                        raise SweepUnsupported("add with alpha")
                    for a in node.args[:2]:
                        push(a, g)
                elif t in self._ELEMENTWISE_FN or t in self._GENERIC_ACT_FN:
                    scale, dst = self._fold_bn(node.args[0])
                    push(dst, self._scale_mask(g, S, self._act_mult(t, self.saved[node]), scale, g2))
                elif t is torch.flatten:
                    push(node.args[0], g.reshape((S * B,) + tuple(self.saved[node][1:])))
                elif t is F.max_pool2d:
                    idx, shp = self.saved[node]
                    push(node.args[0], self._maxpool_vjp(g, idx, shp, S, B))
                elif t is F.avg_pool2d:
                    p = self._bind(node, ("kernel_size", "stride", "padding", "ceil_mode", "count_include_pad",
                                          "divisor_override"), (None, None, 0, False, True, None))
                    push(node.args[0], self._avgpool_vjp(g, self.saved[node], p["kernel_size"], p["stride"], p["padding"],
                                                         S * B, p["ceil_mode"], p["count_include_pad"],
                                                         p["divisor_override"]))
                elif t is F.adaptive_avg_pool2d:
                    push(node.args[0], self._adaptive_avgpool_vjp(g, self.saved[node], S * B))
                elif t is torch.mean:
                    push(node.args[0], self._mean_vjp(g, self.saved[node], S * B))
            elif node.op == "call_method":
                t = node.target
                if t in ("relu", "tanh", "sigmoid"):
                    scale, dst = self._fold_bn(node.args[0])
                    push(dst, self._scale_mask(g, S, self._act_mult(t, self.saved[node]), scale, g2))
                elif t in ("view", "reshape", "flatten"):
                    push(node.args[0], g.reshape((S * B,) + tuple(self.saved[node][1:])))
                elif t == "contiguous":
                    push(node.args[0], g)
                elif t == "mean":
                    push(node.args[0], self._mean_vjp(g, self.saved[node], S * B))
        if remaining:
            raise SweepUnsupported(f"no cotangent reached {sorted(remaining)}")
        return grads

    def release(self):
        self.saved = {}
        self.taps = {}
        self.tap_splits = {}
