"""Host logic of the NHWC split-fp16 reverse sweep (laplace_amd/sweep_nhwc.py) on the CPU emulation of its kernels:
graph walk, residual joins (identity and strided 1x1 down-sampling with accumulate-into), deferred BatchNorm scales,
the guaranteed scale bounds of the element-wise VJP, and the hand-over of split tensors to the consumers — against one
stock autograd pass per seed.  (The kernels themselves: tests/test_gpu_conv.py, tests/test_gpu_sweep_nhwc.py.)"""
import pytest
import torch
from torch import nn

from laplace_amd import _lib
from laplace_amd._lib import SplitTensor, get_kernels
from laplace_amd.nets import BasicBlock
from laplace_amd.sweep_nhwc import SplitSweep
from tests.emulated_kernels import EmulatedKernels


@pytest.fixture(autouse=True)
def _emulated():
    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    yield
    _lib.set_kernels_for_testing(prev)


class TinyResNet(nn.Module):
    def __init__(self, act=torch.relu, width=32, num_classes=5):
        super().__init__()
        self.act = act
        self.conv1 = nn.Conv2d(3, width, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.layers = nn.Sequential(BasicBlock(width, width, 1, act), BasicBlock(width, 2 * width, 2, act),
                                    BasicBlock(2 * width, 2 * width, 1, act))
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2 * width, num_classes)

    def forward(self, x):
        x = self.act(self.bn1(self.conv1(x)))
        x = self.layers(x)
        return self.fc(torch.flatten(self.pool(x), 1))


def _model(act):
    torch.manual_seed(3)
    m = TinyResNet(act).eval()
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):  # non-trivial eval-mode statistics and affine maps
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
            mod.weight.requires_grad_(False), mod.bias.requires_grad_(False)
    return m


def _autograd_tap_grads(model, taps, x, seeds):
    outs = {}
    hs = [m.register_forward_hook(lambda m_, i, o, n=n: outs.__setitem__(n, o)) for n, m in taps.items()]
    f = model(x)
    for h in hs:
        h.remove()
    res = {n: [] for n in taps}
    for s in range(seeds.shape[0]):
        grads = torch.autograd.grad(f, [outs[n] for n in taps], grad_outputs=seeds[s], retain_graph=True)
        for n, g in zip(taps, grads):
            res[n].append(g)
    return f.detach(), {n: torch.stack(v) for n, v in res.items()}


@pytest.mark.parametrize("act", [torch.relu, torch.tanh])
@pytest.mark.parametrize("defer", [False, True])
@pytest.mark.parametrize("fuse", [True, False, "stride-1 only"])
def test_split_sweep_matches_per_seed_autograd(act, defer, fuse):
    model = _model(act)
    taps = {n: m for n, m in model.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))}
    sw = SplitSweep(model, taps, kernels=get_kernels)
    sw.fuse_vjp = bool(fuse)
    sw.fuse_strided = fuse is True
    assert sw.split_ok, sw.split_reason
    K = get_kernels()
    calls = {"vjp": 0, "fused": 0, "strided": 0}
    for name, key in (("vjp_nhwc_split", "vjp"), ("conv_nhwc_f16x2_vjp", "fused"), ("conv_nhwc_f16x2_vjp_strided", "strided")):
        orig = getattr(K, name)
        setattr(K, name, lambda *a, _o=orig, _k=key, **k: (calls.__setitem__(_k, calls[_k] + 1), _o(*a, **k))[1])
    torch.manual_seed(0)
    x = torch.randn(4, 3, 8, 8)
    seeds = torch.randn(3, 4, 5)
    f = sw.forward(x)
    # the forward ran on the NHWC kernels: every tapped conv input but the RGB stem's is NHWC in memory
    from laplace_amd._lib import is_channels_last

    assert all(is_channels_last(sw.taps[n]["a"]) for n in taps if n not in ("conv1", "fc"))
    seen = []
    got = sw.backward(seeds, on_tap=lambda n, g: seen.append(n), defer_bn_scale=defer)
    f_ref, want = _autograd_tap_grads(model, taps, x, seeds)
    assert torch.allclose(f, f_ref, atol=1e-5)
    assert set(seen) == set(taps)
    for n in taps:
        g = got[n]
        if isinstance(g, SplitTensor):
            g = g.float().reshape(3, 4, *g.shape[1:]).permute(0, 1, 4, 2, 3)
        if defer and n in sw.grad_scale:  # the caller owes diag(s) G diag(s): the tap holds the unscaled gradient
            g = g * sw.grad_scale[n].reshape(1, 1, -1, 1, 1)
        err = (g - want[n]).abs().max() / want[n].abs().max()
        assert err < 2e-5, (n, float(err))
    if defer:
        assert sw.grad_scale, "no BatchNorm scale was deferred"
    # stride-1 backward-data passes hand their result over already multiplied / joined / split: five of the seven
    # convolutions of the three blocks; the strided block's two branches (3 x 3 and the 1 x 1 shortcut) leave as ONE
    # strided fused launch, or — switched off — keep the fp32 route and its accumulate-into
    if fuse is True:
        assert calls["fused"] == 5 and calls["strided"] == 1 and calls["vjp"] < 6, calls
    elif fuse:
        assert calls["fused"] == 5 and calls["strided"] == 0 and calls["vjp"] < 7, calls
    else:
        assert calls["fused"] == 0 and calls["strided"] == 0
    # consumers that want plain tensors get [S, B, C, H, W] fp32
    got2 = sw.backward(seeds)
    for n in taps:
        assert torch.is_tensor(got2[n]) and got2[n].shape == want[n].shape
        assert (got2[n] - want[n]).abs().max() / want[n].abs().max() < 2e-5


def test_unsupported_graphs_fall_back_to_the_parent_sweep():
    from laplace_amd.nets import lenet5

    model = lenet5().eval()  # 6- and 16-channel convs, max pooling
    taps = {n: m for n, m in model.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))}
    sw = SplitSweep(model, taps, kernels=get_kernels)
    assert not sw.split_ok and "convolution" in sw.split_reason
    x = torch.randn(3, 3, 32, 32)
    seeds = torch.randn(2, 3, 10)
    sw.forward(x)
    got = sw.backward(seeds)
    _, want = _autograd_tap_grads(model, taps, x, seeds)
    for n in taps:
        assert torch.allclose(got[n], want[n], atol=1e-5)


def test_kron_predictive_through_the_nhwc_rotations():
    """Full-network KFAC posterior of the tiny ResNet: the GLM predictive with both eigenbasis rotations on our
    convolution kernel (inputs: the forward's split activations; cotangents: a 1x1 convolution whose position-contiguous
    output stays seed-major for the quadratic-form kernel) == the same with library rotations and the transposed copy."""
    import laplace_amd.backend as be
    from laplace_amd.laplace import HipLaplace

    model = _model(torch.relu)
    torch.manual_seed(1)
    X, y = torch.randn(12, 3, 8, 8), torch.randint(5, (12,))
    la = HipLaplace(model, "classification", "all", "kron", prior_precision=0.7)

    class L(list):
        dataset = list(range(12))

    la.fit(L([(X[:6], y[:6]), (X[6:], y[6:])]))
    K = get_kernels()
    seen = {"seed_major": 0, "other": 0, "planes": 0}
    orig, orig_p = K.kron_quadform_shared, K.kron_quadform_shared_planes

    def counting(*a, seed_major=False, **k):
        seen["seed_major" if seed_major else "other"] += 1
        return orig(*a, seed_major=seed_major, **k)

    def counting_p(*a, **k):
        seen["planes"] += 1
        return orig_p(*a, **k)

    K.kron_quadform_shared, K.kron_quadform_shared_planes = counting, counting_p
    try:
        f1, v1 = la._glm_predictive_distribution(X[:5])
        n_fast = dict(seen)
        prev = be._OWN_ROTATION
        be._OWN_ROTATION = False
        try:
            f2, v2 = la._glm_predictive_distribution(X[:5])
        finally:
            be._OWN_ROTATION = prev
    finally:
        del K.kron_quadform_shared, K.kron_quadform_shared_planes
    # every 32- / 64-channel convolution of the three blocks: both rotations on our convolution kernel, operands handed to the
    # quadratic-form kernel as split planes (maps whose positions are no multiple of 16 keep the fp32 seed-major form)
    assert n_fast["planes"] + n_fast["seed_major"] >= 6 and n_fast["planes"] >= 3, n_fast
    assert torch.allclose(f1, f2, atol=1e-6)
    assert (v1 - v2).abs().max() / v2.abs().max() < 2e-5


class _ReshapeBetweenConvs(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1, self.c2 = nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.Conv2d(32, 32, 3, 1, 1, bias=False)
        self.pool, self.fc = nn.AdaptiveAvgPool2d(1), nn.Linear(32, 3)

    def forward(self, x):
        h = torch.relu(self.c1(x))
        h = h.reshape(-1, 32, 4, 4).contiguous()
        return self.fc(torch.flatten(self.pool(torch.relu(self.c2(h))), 1))


class _PoolWithTwoConsumers(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1, self.pool = nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.AdaptiveAvgPool2d(1)
        self.fc1, self.fc2 = nn.Linear(32, 3), nn.Linear(32, 3)

    def forward(self, x):
        p = self.pool(torch.relu(self.c1(x)))
        return self.fc1(torch.flatten(p, 1)) + self.fc2(torch.flatten(p, 1))


class _LinearOnFeatureMap(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1, self.lin, self.pool, self.fc = nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.Linear(4, 4), nn.AdaptiveAvgPool2d(1), nn.Linear(32, 3)

    def forward(self, x):
        h = self.lin(torch.relu(self.c1(x)))  # Linear along the last (W) axis of the map
        return self.fc(torch.flatten(self.pool(h), 1))


@pytest.mark.parametrize("cls", [_ReshapeBetweenConvs, _PoolWithTwoConsumers, _LinearOnFeatureMap])
def test_graphs_the_nhwc_walk_has_no_rule_for_are_rejected_before_any_gradient_is_handed_over(cls):
    """`backward` of the NHWC path would raise half-way on these (after `on_tap` has already added G factors); the
    static check sends them through the parent class's NCHW sweep instead, with the same results as autograd."""
    torch.manual_seed(0)
    model = cls().eval()
    taps = {n: m for n, m in model.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))}
    sweep = SplitSweep(model, taps, kernels=get_kernels)
    assert not sweep.split_ok and sweep.split_reason
    x = torch.randn(3, 32, 4, 4)
    f = sweep.forward(x)
    seeds = torch.eye(3)[:, None, :].expand(3, 3, 3).contiguous()
    seen = []
    grads = sweep.backward(seeds, on_tap=lambda n, g: seen.append(n))
    assert sorted(seen) == sorted(taps)
    f_ref, want = _autograd_tap_grads(model, taps, x, seeds)
    assert torch.allclose(f, f_ref, atol=1e-5)
    got = sweep.backward(seeds)
    for n in taps:
        assert torch.allclose(got[n].reshape(want[n].shape), want[n], atol=1e-5), n


class _MixedStrides(nn.Module):
    """one feature map read by a stride-1 AND a stride-2 convolution (their cotangents meet as a pending stride-1 part and a
    pending strided part: neither fused form applies, both must land in ONE fp32 tensor), next to a block whose two strided
    convolutions do meet alone"""

    def __init__(self, width=32):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv_a = nn.Conv2d(width, width, 3, 1, 1, bias=False)
        self.conv_c = nn.Conv2d(width, 2 * width, 3, 2, 1, bias=False)
        self.conv_b = nn.Conv2d(width, 2 * width, 3, 2, 1, bias=False)
        self.block = BasicBlock(2 * width, 4 * width, 2, torch.relu)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(4 * width, 5)

    def forward(self, x):
        x = torch.relu(self.bn1(self.conv1(x)))
        y = torch.relu(self.conv_c(torch.relu(self.conv_a(x))) + self.conv_b(x))
        return self.fc(torch.flatten(self.pool(self.block(y)), 1))


@pytest.mark.parametrize("strided", [True, False])
def test_a_strided_and_a_stride_1_consumer_of_the_same_map(strided):
    torch.manual_seed(5)
    model = _MixedStrides().eval()
    for mod in model.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_var.uniform_(0.5, 2.0), mod.weight.data.uniform_(0.5, 1.5)
            mod.weight.requires_grad_(False), mod.bias.requires_grad_(False)
    taps = {n: m for n, m in model.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))}
    sw = SplitSweep(model, taps, kernels=get_kernels)
    assert sw.split_ok, sw.split_reason
    sw.fuse_strided = strided
    K = get_kernels()
    calls = []
    orig = K.conv_nhwc_f16x2_vjp_strided
    K.conv_nhwc_f16x2_vjp_strided = lambda sources, *a, **k: (calls.append(len(sources)), orig(sources, *a, **k))[1]
    try:
        x = torch.randn(3, 3, 16, 16)
        seeds = torch.randn(2, 3, 5)
        f = sw.forward(x)
        got = sw.backward(seeds)
    finally:
        del K.conv_nhwc_f16x2_vjp_strided
    f_ref, want = _autograd_tap_grads(model, taps, x, seeds)
    assert torch.allclose(f, f_ref, atol=1e-5)
    for n in taps:
        err = (got[n] - want[n]).abs().max() / want[n].abs().max()
        assert err < 2e-5, (n, float(err))
    # the block's pair leaves as one launch of two sources, conv_c (alone at its node) as one launch of one source; conv_b
    # meets the stride-1 conv_a at `x` and goes class by class into conv_a's tensor
    assert sorted(calls) == ([1, 2] if strided else [])


@pytest.mark.parametrize("k,p,H,W", [(3, 1, 8, 8), (3, 1, 6, 10), (1, 0, 8, 8), (2, 0, 8, 6), (3, 1, 2, 2)])
def test_strided_launch_tables_against_conv2d_input(k, p, H, W):
    """`conv.strided_taps` + the (emulated) strided fused launch == the input gradient of the strided convolutions (one
    and two sources, with the folded channel scale of the second), for the geometries `strided_fused_ok` admits"""
    from laplace_amd import conv as cv

    torch.manual_seed(k * 10 + p + H)
    cin, cout, N = 32, 32, 3
    m1 = nn.Conv2d(cin, cout, k, 2, p, bias=False)
    m2 = nn.Conv2d(cin, cout, 1, 2, 0, bias=False)
    assert cv.strided_fused_ok(m2, (H, W))
    if not cv.strided_fused_ok(m1, (H, W)):
        pytest.skip("geometry outside the strided form (output grid is not H/2 x W/2)")
    K = get_kernels()
    Ho, Wo = H // 2, W // 2
    g1, g2 = torch.randn(N, cout, Ho, Wo), 5.0 * torch.randn(N, cout, Ho, Wo)
    cs = torch.rand(cout) + 0.5
    p1, p2 = cv.PreparedConv(m1), cv.PreparedConv(m2)
    s1, s2 = K.split_f16x2(g1.permute(0, 2, 3, 1).contiguous()), K.split_f16x2(g2.permute(0, 2, 3, 1).contiguous())
    want1 = torch.nn.grad.conv2d_input((N, cin, H, W), m1.weight.double(), g1.double(), stride=2, padding=p)
    want2 = torch.nn.grad.conv2d_input((N, cin, H, W), m2.weight.double() * cs.double().reshape(-1, 1, 1, 1), g2.double(), stride=2)
    rows1 = cv.strided_taps([(p1, s1, None)], (H, W))
    if rows1 is None:  # (a kernel that leaves residue classes untouched, e.g. 1 x 1: only valid next to one that covers them)
        assert k == 1
    else:
        got = cv.conv_backward_data_vjp_strided([(p1, s1, None)], (H, W))
        assert (got.float().permute(0, 3, 1, 2).double() - want1).abs().max() < 2e-5 * want1.abs().max()
    if k != 1:
        got = cv.conv_backward_data_vjp_strided([(p1, s1, None), (p2, s2, cs)], (H, W))
        want = want1 + want2
        assert (got.float().permute(0, 3, 1, 2).double() - want).abs().max() < 2e-5 * want.abs().max()
