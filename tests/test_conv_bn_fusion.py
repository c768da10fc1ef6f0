"""Host logic of the forward's fused convolution + BatchNorm / add / ReLU launches (laplace_amd/sweep_nhwc.py:
`_PendingConv`, `_bn_takes_conv`) on the kernel emulation: which convolutions are taken over by their BatchNorm, which
stay two launches, and that the traced forward and the KFAC factors do not depend on it.  The kernel itself:
tests/test_gpu_conv_bn_act.py.

Reference behaviour matched: the model's forward inside the curvature backends (laplace/curvature/curvature.py:309-311)."""
import pytest
import torch
from torch import nn


class Counting:
    """wraps the emulation and counts the launches of the forward"""

    def __init__(self, inner):
        self._inner, self.calls = inner, {"conv_bn_act_nhwc": 0, "conv_nhwc_f16x2": 0, "bn_act_forward_nhwc": 0}

    def __getattr__(self, name):
        v = getattr(self._inner, name)
        if name in self.calls:
            def counted(*a, **kw):
                self.calls[name] += 1
                return v(*a, **kw)

            return counted
        return v


@pytest.fixture
def kernels():
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    k = Counting(EmulatedKernels())
    prev = _lib.set_kernels_for_testing(k)
    yield k
    _lib.set_kernels_for_testing(prev)


def _sweep(model):
    from laplace_amd._lib import get_kernels
    from laplace_amd.sweep_nhwc import SplitSweep

    taps = {nm: mod for nm, mod in model.named_modules() if isinstance(mod, (nn.Conv2d, nn.Linear))}
    return SplitSweep(model, taps, kernels=get_kernels)


def test_every_conv_of_resnet18_is_taken_over_by_its_batchnorm(kernels, monkeypatch):
    from laplace_amd.nets import ResNet18
    from laplace_amd.sweep_nhwc import SplitSweep

    torch.manual_seed(0)
    model = ResNet18(10).eval()
    x = torch.randn(2, 3, 16, 16)
    sw = _sweep(model)
    f = sw.forward(x)
    # conv1 + 8 blocks x 2 + 3 down-sampling shortcuts: twenty launches, none of them followed by a BatchNorm launch
    # (the emulation's fused op calls its own convolution and BatchNorm directly, not through the counter)
    assert kernels.calls == {"conv_bn_act_nhwc": 20, "conv_nhwc_f16x2": 0, "bn_act_forward_nhwc": 0}
    monkeypatch.setattr(SplitSweep, "fuse_conv_bn", False)
    kernels.calls.update(dict.fromkeys(kernels.calls, 0))
    sw2 = _sweep(model)
    f2 = sw2.forward(x)
    assert kernels.calls == {"conv_bn_act_nhwc": 0, "conv_nhwc_f16x2": 20, "bn_act_forward_nhwc": 20}
    assert torch.equal(f, f2)
    for nm in sw.taps:
        assert torch.equal(sw.taps[nm]["a"], sw2.taps[nm]["a"])
    for node, keep in sw.saved.items():
        other = sw2.saved[[n for n in sw2.saved if n.name == node.name][0]]
        if torch.is_tensor(keep):
            assert torch.equal(keep, other)


class Odd(nn.Module):
    """convolutions the BatchNorm must NOT take over: one with a bias, one whose output has a second consumer"""

    def __init__(self):
        super().__init__()
        self.c0 = nn.Conv2d(3, 32, 3, 1, 1, bias=False)
        self.b0 = nn.BatchNorm2d(32)
        self.c1 = nn.Conv2d(32, 32, 3, 1, 1, bias=True)
        self.b1 = nn.BatchNorm2d(32)
        self.c2 = nn.Conv2d(32, 32, 3, 1, 1, bias=False)
        self.b2 = nn.BatchNorm2d(32)
        self.c3 = nn.Conv2d(32, 32, 1, bias=False)
        self.b3 = nn.BatchNorm2d(32)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(32, 4)

    def forward(self, x):
        x = torch.relu(self.b0(self.c0(x)))
        x = torch.relu(self.b1(self.c1(x)))
        y = self.c2(x)
        x = torch.relu(self.b2(y) + y)  # (the convolution's raw output is used twice)
        x = self.b3(self.c3(x)) + x     # (add of a BatchNorm'ed shortcut, no activation)
        return self.fc(torch.flatten(self.pool(x), 1))


def test_biased_and_shared_convolutions_stay_two_launches(kernels):
    torch.manual_seed(1)
    model = Odd().eval()
    for bn in (model.b0, model.b1, model.b2, model.b3):
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(3, 3, 6, 6)
    sw = _sweep(model)
    assert sw.split_ok, sw.split_reason
    f = sw.forward(x)
    assert kernels.calls["conv_bn_act_nhwc"] == 2  # c0 -> b0 -> relu and c3 -> b3 (+ x)
    assert torch.allclose(f, model(x), rtol=1e-4, atol=1e-5)


def test_factors_do_not_depend_on_the_fusion(kernels, monkeypatch):
    from laplace_amd import HipGGN
    from laplace_amd.nets import ResNet18
    from laplace_amd.sweep_nhwc import SplitSweep

    torch.manual_seed(2)
    model = ResNet18(10).eval()
    X, y = torch.randn(3, 3, 8, 8), torch.randint(0, 10, (3,))
    out = {}
    for fuse in (True, False):
        monkeypatch.setattr(SplitSweep, "fuse_conv_bn", fuse)
        loss, H = HipGGN(model, "classification").kron(X, y, N=3)
        out[fuse] = (loss, H)
    assert torch.equal(out[True][0], out[False][0])
    for Fa, Fb in zip(out[True][1].kfacs, out[False][1].kfacs):
        for a, b in zip(Fa, Fb):
            assert torch.equal(a, b)
