"""The kernels behind one scale PER IMAGE on the forward's split tensors, on the device against fp64:
lk_split_images_f16x2, lk_bn_act_fwd_nhwc_f16x2 (per-image guaranteed bound, measured per-image maxima),
lk_conv_nhwc_f16x2 with in_nsexp = N (row-by-row un-scaling in the plain, position-contiguous and position-major
epilogues).  Every image is compared with ITS OWN maximum over twelve to thirteen decades in one tensor — what one scale
per tensor resolved to 2^-39 of the LARGEST image only (round 4: tests/test_gpu_dynamic_range.py pinned an image 1e-9
below its neighbour at 2^-14 .. 2^-6 relative).  Host logic: tests/test_per_image_scales.py.  -m gpu only.

Reference behaviour matched: every sample is computed in fp32 whatever else is in its minibatch
(laplace/curvature/curvature.py:375-433, curvlinops.py:77-108)."""
import math

import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu
# LK_TEST_DEVICE=cpu: self-check of this file's host logic on the kernel emulation (GPU-less box)
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


def rel_rows(a, b):
    from tests.parity_log import record_error

    a, b = a.double().cpu().flatten(1), b.double().cpu().flatten(1)
    return record_error(((a - b).abs().amax(1) / (b.abs().amax(1) + 1e-300)).max().item())


def _decades(n, lo, hi, seed=0):
    g = torch.Generator().manual_seed(seed)
    e = torch.rand(n, generator=g) * (hi - lo) + lo
    e[0], e[1] = lo, hi
    return (10.0 ** e).to(DEV)


@pytest.mark.parametrize("shape", [(8, 8, 8, 64), (130, 4, 4, 32), (3, 32, 32, 32), (1, 2, 2, 8), (70000, 1, 1, 8)])
def test_split_images(shape):
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    torch.manual_seed(0)
    N = shape[0]
    x = torch.randn(*shape, device=DEV) * _decades(N, -9.0, 6.0).reshape(N, 1, 1, 1) if N > 1 else torch.randn(*shape, device=DEV)
    if N > 2:
        x[2] = 0.0
    st = K.split_images_f16x2(x.contiguous())
    assert st.per_image == (N > 1) and st.sexp.numel() == N and st.amax.numel() == N
    assert torch.equal(st.amax.cpu(), x.abs().reshape(N, -1).amax(1).cpu())
    live = [i for i in range(N) if float(x[i].abs().max()) > 0]
    assert rel_rows(st.float()[live], x[live]) < 2.0 ** -21
    dead = [i for i in range(N) if i not in live]
    if dead:
        assert float(st.float()[dead].abs().max()) == 0.0
    top = x.abs().reshape(N, -1).amax(1)[live] * torch.exp2(st.sexp.float()[live])
    assert bool(((top >= 2.0 ** 14) & (top < 2.0 ** 15)).all())


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("with_addend", [False, True])
@pytest.mark.parametrize("shape", [(16, 8, 8, 64), (5, 3, 3, 24), (128, 4, 4, 512), (66000, 1, 1, 8)])
def test_bn_act_forward_nhwc_per_image(shape, act, with_addend):
    """y, the ReLU mask, the per-image planes (against every image's own maximum), the measured maxima, the guaranteed
    bounds — with the bound of the input given the way the sweep gives it: maxima of the producing convolution's INPUT
    images x an l1 factor + a bias bound (loose by 2^6 here: it must only cost fixed-point range)"""
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    torch.manual_seed(1)
    N, H, W, C = shape
    # (tanh: images up to O(1) only — the sum of two addends of 1e6 is resolved to 0.1 in fp32, and tanh of THAT is not a test
    #  of this kernel)
    sc = _decades(N, -6.0, 6.0 if act != 2 else 0.0, seed=2).reshape(N, 1, 1, 1)
    x = (torch.randn(N, H, W, C, device=DEV) * sc).contiguous()
    scale = (torch.rand(C, device=DEV) + 0.5)
    shift = torch.zeros(C, device=DEV) if act != 2 else torch.randn(C, device=DEV) * 0.1  # (homogeneous: the image scale survives)
    addend = (torch.randn(N, H, W, C, device=DEV) * sc).contiguous() if with_addend else None
    if N > 1000:  # (8 elements per image: keep the sum free of cancellation, or "the image's own maximum" is itself rounding)
        x = x.abs_()
        addend = None if addend is None else addend.abs_()
    in_amax = (x.abs().reshape(N, -1).amax(1) / 48.0).contiguous()      # "the convolution's input maxima"
    x_mul = torch.tensor([64.0], device=DEV)                              # its l1 norm: in_amax * 64 >= max|x_n| (slack 4/3 .. )
    x_add = torch.tensor([0.0], device=DEV)
    a_bound = addend.abs().reshape(N, -1).amax(1).contiguous() if with_addend else None
    y, mask, split, bound = K.bn_act_forward_nhwc(x, in_amax, scale, shift, K.absmax(scale), K.absmax(shift), act, addend=addend,
                                                  addend_bound=a_bound, x_mul=x_mul, x_add=x_add)
    want = x.double() * scale.double() + shift.double()
    if with_addend:
        want = want + addend.double()
    if act == 1:
        want = want.clamp_min(0)
    elif act == 2:
        want = torch.tanh(want)
    assert rel_rows(y, want) < 2e-6
    if act == 1:
        assert torch.equal(mask.view(torch.bool).cpu(), (y > 0).cpu())
    else:
        assert mask is None
    assert split.per_image and tuple(split.sexp.shape) == (N,)
    assert rel_rows(split.float(), y) < 2.0 ** -20          # against every image's OWN maximum
    assert torch.equal(split.amax.cpu(), y.abs().reshape(N, -1).amax(1).cpu())
    assert bool((bound.cpu() * (1 + 1e-6) >= split.amax.cpu()).all())
    # a single bound word for all images (a producer that is not one of our convolutions) stays legal
    y1, _, split1, _ = K.bn_act_forward_nhwc(x, K.absmax(x), scale, shift, K.absmax(scale), K.absmax(shift), act,
                                             addend=addend, addend_bound=None if addend is None else K.absmax(addend))
    big = int(y.abs().reshape(N, -1).amax(1).argmax())
    assert torch.equal(y1, y) and rel_rows(split1.float()[big:big + 1], y[big:big + 1]) < 2.0 ** -20  # (the largest image keeps everything)


CONVS = [(64, 64, 3, 1, 1, 32, 16), (128, 128, 3, 1, 1, 16, 16), (512, 512, 3, 1, 1, 4, 128), (64, 128, 3, 2, 1, 32, 16),
         (256, 512, 1, 2, 0, 8, 64), (32, 64, 3, 1, 1, 7, 9)]


@pytest.mark.parametrize("cfg", CONVS, ids=[f"{c[0]}-{c[1]}-k{c[2]}s{c[3]}-{c[5]}x{c[5]}-b{c[6]}" for c in CONVS])
def test_forward_convolution_on_a_per_image_operand_over_thirteen_decades(cfg):
    """plain NHWC epilogue (incl. the position-major tiles of small maps at >= 64 images, strided grids, a ragged 7x7
    map) and the position-contiguous epilogue of the predictive's rotation: every image against its own maximum"""
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    cin, cout, k, s, p, H, B = cfg
    torch.manual_seed(cin + cout + k)
    m = nn.Conv2d(cin, cout, k, s, p, bias=False).to(DEV)
    x = torch.randn(B, cin, H, H, device=DEV).relu_() * _decades(B, -7.0, 6.0, seed=3).reshape(B, 1, 1, 1)
    want = F.conv2d(x.double().cpu(), m.weight.double().cpu(), None, s, p)
    xs = K.split_images_f16x2(x.permute(0, 2, 3, 1).contiguous())
    prep = cv.PreparedConv(m)
    y = cv.conv_forward(prep, xs).permute(0, 3, 1, 2)
    r = rel_rows(y, want)
    assert r < 1e-5, f"per-image error {r:.2e}"
    # the same through one scale per tensor: what the per-image operand is for (reported, and pinned to be far worse)
    y1 = cv.conv_forward(prep, K.split_f16x2(x.permute(0, 2, 3, 1).contiguous())).permute(0, 3, 1, 2)
    d1, w1 = y1.double().cpu().flatten(1), want.double().flatten(1)   # (not through rel_rows: this is no parity number)
    assert ((d1 - w1).abs().amax(1) / (w1.abs().amax(1) + 1e-300)).max().item() > 1e-3
    Ho = want.shape[-1]
    if (Ho * Ho) % 4 == 0:  # the predictive's rotation: arbitrary filter bank, position-contiguous output
        filt = torch.randn(96, cin, k, k, device=DEV)
        out = cv.conv_forward_filters(m, x, filt, filt, xs=xs)
        want_f = F.conv2d(x.double().cpu(), filt.double().cpu(), None, s, p)
        assert rel_rows(out, want_f) < 1e-5
        out2 = cv.conv_forward_filters(m, x, filt, filt)  # (splits the activation itself: one scale per image as well)
        assert rel_rows(out2, want_f) < 1e-5


def test_fused_launches_refuse_a_per_image_operand():
    from laplace_amd import conv as cv
    from laplace_amd._lib import LaplaceHipError, get_kernels

    K = get_kernels()
    m = nn.Conv2d(64, 64, 3, 1, 1, bias=False).to(DEV)
    g = K.split_images_f16x2(torch.randn(18, 8, 8, 64, device=DEV))
    with pytest.raises(LaplaceHipError, match="one scale per image"):
        cv.conv_backward_data_vjp(cv.PreparedConv(m), g, (8, 8))
    G = torch.zeros(64, 64, device=DEV)
    with pytest.raises(LaplaceHipError, match="one scale per image"):
        K.gram_tn_f16x2(g, 1.0, G)


def test_forward_of_the_sweep_per_image_on_c4_shapes():
    """ResNet-18 (ReLU: positively homogeneous, the image scale survives to the last layer) on a minibatch spanning
    twelve decades: the activation every convolution consumes, image by image against the fp64 forward, and the masks
    against fp64 where fp64 is not within fp32 rounding of zero"""
    import copy

    from laplace_amd._lib import get_kernels
    from laplace_amd.nets import ResNet18
    from laplace_amd.sweep_nhwc import SplitSweep

    torch.manual_seed(7)
    m = ResNet18(10).eval()
    m64 = copy.deepcopy(m).double()
    m = m.to(DEV)
    X = torch.randn(8, 3, 32, 32) * torch.tensor([1e-6, 1e6, 1.0, 1e-3, 30.0, 1e-6, 1e3, 0.3]).reshape(8, 1, 1, 1)
    ins64 = {}
    hs = [mod.register_forward_hook(lambda m_, i, o, n=n: ins64.__setitem__(n, i[0].detach()))
          for n, mod in m64.named_modules() if isinstance(mod, (nn.Conv2d, nn.Linear))]
    f64 = m64(X.double())
    taps = {n: mod for n, mod in m.named_modules() if isinstance(mod, (nn.Conv2d, nn.Linear))}
    sw = SplitSweep(m, taps, kernels=get_kernels)
    assert sw.split_ok, sw.split_reason
    f = sw.forward(X.to(DEV), keep_tap_splits=True)
    assert rel_rows(f, f64) < 1e-5
    assert len(sw.tap_splits) >= 19
    for n, s in sw.tap_splits.items():
        want = ins64[n].permute(0, 2, 3, 1)
        assert s.per_image
        r = rel_rows(s.float()[..., :want.shape[-1]], want)
        assert r < 1e-5, f"{n}: per-image error of the split activation {r:.2e}"
