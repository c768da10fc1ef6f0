"""lk_conv_bn_act_nhwc_f16x2 — the forward convolution with the eval-mode BatchNorm, the residual add and the ReLU in its
epilogue — on the device.  Its contract is "the same bits as lk_conv_nhwc_f16x2 followed by lk_bn_act_fwd_nhwc_f16x2"
(one launch instead of two, no fp32 round trip of the convolution's output), so that is what is asserted: y, the mask,
both planes, the per-image scales, the measured maxima EQUAL, over every tile shape of the generic kernel, position-major
tiles of small maps, strides, a thin stem, ragged tile edges and images of very different magnitudes; plus against fp64.
Whole model: ResNet-18's forward and every KFAC factor with the fused launches against the two-launch forward.
Host logic (which convolutions are taken over by their BatchNorm): tests/test_conv_bn_fusion.py.  -m gpu only.

Reference behaviour matched: the model's forward inside the curvature backends (laplace/curvature/curvature.py:309-311,
curvlinops.py:77-108) — conv -> bn -> (+ identity) -> relu blocks in eval mode."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
DEV = __import__("os").environ.get("LK_TEST_DEVICE", "cuda")


@pytest.fixture(autouse=True)
def _kernels():
    if DEV != "cpu":
        yield
        return
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    yield
    _lib.set_kernels_for_testing(prev)


def _both(m, x, scale, shift, act, addend, config=None):
    """(fused, two launches) on the same split input"""
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    prep = cv.PreparedConv(m)
    xh = x.permute(0, 2, 3, 1).contiguous()
    if prep.padded_in != m.in_channels:
        xp = xh.new_zeros(*xh.shape[:3], prep.padded_in)
        xp[..., :xh.shape[-1]] = xh
        xh = xp
    xs = K.split_images_f16x2(xh)
    planes, sexp = prep.forward_planes()
    l1, bmax = prep.forward_l1()
    assert bmax is None
    s, (ph, pw), (KH, KW) = m.stride[0], m.padding, m.kernel_size
    N, Hin, Win, _ = xs.shape
    Ho, Wo = (Hin + 2 * ph - KH) // s + 1, (Win + 2 * pw - KW) // s + 1
    taps = [(kh - ph, kw - pw, kh * KW + kw) for kh in range(KH) for kw in range(KW)]
    s_amax, t_amax = K.absmax(scale), K.absmax(shift)
    a_bound = None if addend is None else addend.abs().reshape(N, -1).amax(1).contiguous()
    cfg = K.conv_config if config is None else (K.conv_config | config)
    fused = K.conv_bn_act_nhwc(xs, planes, sexp, l1, Ho, Wo, s, taps, scale, shift, s_amax, t_amax, act, addend=addend,
                               addend_bound=a_bound, config=cfg)
    out = torch.empty(N, Ho, Wo, m.out_channels, dtype=torch.float32, device=x.device)
    K.conv_nhwc_f16x2(xs, planes, sexp, Ho, Wo, s, out, 1, 0, 0, taps, config=cfg)
    two = K.bn_act_forward_nhwc(out, xs.amax, scale, shift, s_amax, t_amax, act, addend=addend, addend_bound=a_bound, x_mul=l1)
    return fused, two


def _same(fused, two, act):
    y, mask, split, bound = fused
    y2, mask2, split2, bound2 = two
    assert torch.equal(y, y2)
    if act == 1:
        assert torch.equal(mask, mask2)
    else:
        assert mask is None and mask2 is None
    assert torch.equal(split.sexp, split2.sexp)
    assert torch.equal(split.planes, split2.planes)
    assert torch.equal(split.amax, split2.amax)
    assert torch.equal(bound, bound2)


def _images(N, C, H, W, seed, decades=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g)
    if decades and N > 1:
        e = torch.rand(N, generator=g) * 10.0 - 5.0
        e[0], e[-1] = -5.0, 5.0
        x = x * (10.0 ** e).reshape(N, 1, 1, 1)
        if N > 2:
            x[1] = 0.0  # (a dead image: scale, bound and maximum of nothing)
    return x.to(DEV)


# (N, Cin, Cout, H, W, kernel, stride, padding): the c4 families, a thin stem, a 1 x 1 down-sampling shortcut, ragged edges,
# position-major tiles (small maps at N >= 64), Cout that is no multiple of a tile
SHAPES = [
    (6, 64, 64, 16, 16, 3, 1, 1),
    (3, 3, 64, 32, 32, 3, 1, 1),
    (5, 64, 128, 16, 16, 3, 2, 1),
    (5, 64, 128, 16, 16, 1, 2, 0),
    (128, 256, 256, 4, 4, 3, 1, 1),
    (70, 128, 136, 2, 2, 3, 1, 1),
    (7, 32, 72, 5, 7, 3, 1, 1),
    (2, 32, 8, 9, 9, 3, 1, 0),
    (66, 64, 64, 1, 1, 1, 1, 0),
]


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("with_addend", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_one_launch_equals_the_two_it_replaces(shape, with_addend, act):
    N, Ci, Co, H, W, k, s, p = shape
    torch.manual_seed(3)
    m = nn.Conv2d(Ci, Co, k, s, p, bias=False).to(DEV)
    x = _images(N, Ci, H, W, seed=5)
    scale = (torch.rand(Co, device=DEV) + 0.5) * (torch.randint(0, 2, (Co,), device=DEV) * 2 - 1)
    shift = torch.randn(Co, device=DEV) * 0.1
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    addend = None
    if with_addend:
        addend = _images(N, Co, Ho, Wo, seed=7).permute(0, 2, 3, 1).contiguous()
    fused, two = _both(m, x, scale, shift, act, addend)
    _same(fused, two, act)
    # ... and the pair is right: against fp64 on every image's own scale
    from tests.parity_log import record_error

    want = torch.nn.functional.conv2d(x.double(), m.weight.double(), None, s, p).permute(0, 2, 3, 1) * scale.double() + shift.double()
    if with_addend:
        want = want + addend.double()
    if act == 1:
        want = want.clamp_min(0)
    y = fused[0].double()
    # (scale of an image's error: what went INTO its sums, not what cancellation left of them)
    ref = (torch.nn.functional.conv2d(x.double().abs(), m.weight.double().abs(), None, s, p).permute(0, 2, 3, 1) * scale.double().abs()
           + shift.double().abs() + (addend.double().abs() if with_addend else 0.0)).reshape(N, -1).amax(1)
    err = ((y - want).abs().reshape(N, -1).amax(1) / (ref + 1e-300)).max().item()
    assert record_error(err) < 2e-6
    split = fused[2]
    live = split.amax > 0
    if bool(live.any()):
        d = (split.float().double() - y).abs().reshape(N, -1).amax(1)[live] / split.amax.double()[live]
        assert float(d.max()) < 2.0 ** -20  # the planes carry every image at fp32 level on ITS OWN scale


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("shape", [(9, 64, 64, 16, 16, 3, 1, 1), (130, 64, 192, 4, 4, 3, 1, 1), (4, 32, 136, 12, 10, 3, 2, 1)])
def test_every_tile_shape(shape, tile):
    """config bits 12..14 walk through the five tile shapes of the generic kernel (each has its own forward-epilogue
    instantiation: staging image, per-image slots, lanes per row)"""
    if DEV == "cpu":
        pytest.skip("tile shapes exist on the device only")
    N, Ci, Co, H, W, k, s, p = shape
    torch.manual_seed(11)
    m = nn.Conv2d(Ci, Co, k, s, p, bias=False).to(DEV)
    x = _images(N, Ci, H, W, seed=13)
    scale, shift = torch.rand(Co, device=DEV) + 0.5, torch.randn(Co, device=DEV)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    addend = _images(N, Co, Ho, Wo, seed=17).permute(0, 2, 3, 1).contiguous()
    fused, two = _both(m, x, scale, shift, 1, addend, config=tile << 12)
    _same(fused, two, 1)


@pytest.mark.parametrize("act", ["relu", "tanh"])
def test_resnet18_forward_and_factors_with_and_without_the_fused_launches(act, monkeypatch):
    """the whole pipeline on the timed model: logits and every KFAC factor of two minibatches, the convolutions taken over
    by their BatchNorms against the two-launch forward — the forward is the same to the bit, so the factors are"""
    from laplace_amd import HipGGN
    from laplace_amd.nets import ResNet18
    from laplace_amd.sweep_nhwc import SplitSweep

    torch.manual_seed(19)
    model = ResNet18(10, act=torch.relu if act == "relu" else torch.tanh).to(DEV).eval()
    n = 32 if DEV != "cpu" else 2
    data = [(_images(n, 3, 32, 32, seed=23 + i, decades=False), torch.randint(0, 10, (n,), device=DEV)) for i in range(2)]
    res = {}
    for fuse in (True, False):
        monkeypatch.setattr(SplitSweep, "fuse_conv_bn", fuse)
        b = HipGGN(model, "classification")
        f = b._forward(data[0][0])[0]
        acc = b.kron_accumulator(2 * n)
        acc.lanes = 1
        for X, y in data:
            acc.add_batch(X, y)
        loss, H = acc.finalize()
        res[fuse] = (f, loss, H)
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])
    for Fa, Fb in zip(res[True][2].kfacs, res[False][2].kfacs):
        for a, b_ in zip(Fa, Fb):
            assert torch.equal(a, b_)
