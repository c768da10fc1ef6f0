"""Extraction of per-layer inputs ``a`` and output gradients ``g`` from stock autograd.

Host-side plumbing around the model (the model forward/backward stays PyTorch-ROCm): one forward
with hooks on the supported modules, then reverse passes that deliver, for every seed (a column of
the likelihood-Hessian root), the gradient w.r.t. every tapped module's *output* — no weight
gradients are ever formed and there is no second forward for the loss (cf. the reference's default
backend, laplace/curvature/curvlinops.py:87-106, and the ``jacrev`` materialisation of
laplace/curvature/curvature.py:88-129).  Pure-Linear models use one vmapped pass
(``is_grads_batched``); conv models run one pass per seed and hand the per-seed gradients to the
Gram kernel unstacked.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import torch
from torch import nn

SUPPORTED = (nn.Linear, nn.Conv2d)


@dataclass
class Tap:
    name: str
    module: nn.Module
    kind: str  # 'linear' | 'conv2d'
    w_off: int  # column offset of the weight in the flattened parameter vector
    b_off: int  # column offset of the bias, -1 if the module has no (tracked) bias
    a: torch.Tensor | None = None  # module input (detached)
    a_split: object | None = None  # the same input as an NHWC SplitTensor, when the forward pass produced one
    out: object | None = None  # gradient edge of the module output in the autograd graph

    @property
    def has_bias(self) -> bool:
        return self.b_off >= 0


def _conv_checks(m: nn.Conv2d, name: str):
    if m.groups != 1:
        raise NotImplementedError(f"{name}: grouped convolutions are not supported by the KFAC path")
    if isinstance(m.padding, str):
        raise NotImplementedError(f"{name}: string padding ('{m.padding}') not supported")
    if m.padding_mode != "zeros":
        raise NotImplementedError(f"{name}: padding_mode={m.padding_mode!r} not supported")


class Tape:
    """Finds the supported modules whose weight is Laplace-tracked (``named_modules`` order, as
    laplace/curvature/curvlinops.py:55-75) and records their inputs/outputs during a forward."""

    def __init__(self, model: nn.Module, params: Sequence[nn.Parameter]):
        self.model = model
        offsets, off = {}, 0
        for p in params:
            offsets[id(p)] = off
            off += p.numel()
        self.n_params = off
        self.taps: list[Tap] = []
        covered = set()
        for name, mod in model.named_modules():
            if not isinstance(mod, SUPPORTED) or id(mod.weight) not in offsets:
                continue
            if isinstance(mod, nn.Conv2d):
                _conv_checks(mod, name)
            b_off = -1
            if mod.bias is not None and id(mod.bias) in offsets:
                b_off = offsets[id(mod.bias)]
                covered.add(id(mod.bias))
            covered.add(id(mod.weight))
            self.taps.append(Tap(name, mod, "linear" if isinstance(mod, nn.Linear) else "conv2d",
                                 offsets[id(mod.weight)], b_off))
        # tracked parameters that no supported module owns (norm layers, embeddings, lone biases ...)
        self.uncovered = [p for p in params if id(p) not in covered]

    def forward(self, x):
        """Run ``model(x)`` with hooks; returns ``f`` (attached to the graph)."""
        handles, seen = [], set()
        for tap in self.taps:
            def hook(m, inp, out, tap=tap):
                if id(m) in seen:
                    raise NotImplementedError(f"{tap.name}: module is applied more than once per forward")
                seen.add(id(m))
                tap.a = inp[0].detach()
                # the gradient EDGE of the output as it is now: models that go on to modify the tensor in place
                # (torchvision's `out += identity; relu_(out)`) would otherwise hand back the gradient w.r.t. the
                # mutated tensor — the reference's curvlinops hooks see the pre-mutation gradient as well
                tap.out = torch.autograd.graph.get_gradient_edge(out) if out.requires_grad else out
            handles.append(tap.module.register_forward_hook(hook))
        try:
            with torch.enable_grad():
                f = self.model(x)
        finally:
            for h in handles:
                h.remove()
        for tap in self.taps:
            if tap.out is None:
                raise RuntimeError(f"{tap.name}: module did not run in the forward pass")
        return f

    def output_grads(self, f: torch.Tensor, seeds: torch.Tensor, stack: bool = True):
        """``seeds[s]`` is a cotangent of ``f``; returns, per tap, the gradients w.r.t. the module output
        for every seed: a ``[S, *out.shape]`` tensor, or — for conv taps when ``stack=False`` — the
        list of the ``S`` per-seed tensors exactly as autograd produced them (the Gram kernel reads
        them through a pointer table, so the multi-GB ``[S, B, C, H, W]`` stack is never written).

        Pure-Linear models use ONE vmapped reverse pass (``is_grads_batched``); conv models run one
        reverse pass per seed because functorch has no fused batching rule for MIOpen's conv backward
        (it loops internally and then pays an extra concatenation).
        """
        outs = [t.out for t in self.taps]
        if any(torch.is_tensor(o) for o in outs):
            raise RuntimeError("a tapped module's output does not require grad (frozen parameters upstream and "
                               "downstream?)")
        S = seeds.shape[0]
        has_conv = any(t.kind == "conv2d" for t in self.taps)
        if S == 1:
            return [g.unsqueeze(0).contiguous() for g in torch.autograd.grad(f, outs, grad_outputs=seeds[0])]
        if not has_conv:
            try:
                grads = torch.autograd.grad(f, outs, grad_outputs=seeds, is_grads_batched=True, retain_graph=True)
                return [g.contiguous() for g in grads]
            except RuntimeError:
                pass  # an op without a batching rule: fall through to one reverse pass per seed
        per_seed = [torch.autograd.grad(f, outs, grad_outputs=seeds[s], retain_graph=(s + 1 < S)) for s in range(S)]
        result = []
        for i, tap in enumerate(self.taps):
            gs = [ps[i].contiguous() for ps in per_seed]
            if tap.kind == "conv2d" and not stack:
                result.append(gs)
            else:
                result.append(torch.stack(gs))
        return result

    def release(self):
        for t in self.taps:
            t.a = None
            t.a_split = None
            t.out = None
        sweep = getattr(self, "sweep", None)
        if sweep:
            sweep.release()
