"""The split-fp16 implicit-GEMM convolution (csrc/lk_conv.hip, laplace_amd/conv.py) against fp64 convolutions of the
same operands: backward-data on the ten convolution shapes of ResNet-18 (config c4: 3x3 stride 1 and 2, 1x1 stride 2,
64...512 channels, 32x32...4x4 maps), the forward form, the residual accumulate mode, and the split itself.
Tolerance: 1e-4 of the largest element (BASELINE.json); the measured errors are ~1e-6.  -m gpu only."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (Cin, Cout, k, stride, pad, H) of the input map: every conv of ResNet-18 (CIFAR stem) that needs a backward-data pass
C4_SHAPES = [
    (64, 64, 3, 1, 1, 32), (64, 128, 3, 2, 1, 32), (128, 128, 3, 1, 1, 16), (64, 128, 1, 2, 0, 32),
    (128, 256, 3, 2, 1, 16), (256, 256, 3, 1, 1, 8), (128, 256, 1, 2, 0, 16), (256, 512, 3, 2, 1, 8),
    (512, 512, 3, 1, 1, 4), (256, 512, 1, 2, 0, 8),
]


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    from tests.parity_log import record_error

    return record_error((a - b).abs().max().item() / (b.abs().max().item() + 1e-300))


def _conv(cin, cout, k, s, p):
    torch.manual_seed(cin * 7 + cout + k + s)
    return nn.Conv2d(cin, cout, k, s, p, bias=False).to(DEV)


def test_split_reconstructs_to_22_bits_and_survives_wide_dynamic_range():
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 8, 8, 32, generator=g) * torch.exp(4 * torch.randn(64, 1, 1, 1, generator=g))  # e^±12 spread
    x[0, 0, 0, :8] = 0.0
    xs = K.split_f16x2(x.to(DEV).contiguous())
    back = xs.float().cpu()
    amax = x.abs().max()
    err = (back - x).abs()
    # elements within 2^-17 of the largest: 22-bit relative accuracy; everything else: 2^-39 of the largest
    assert (err <= torch.maximum(x.abs() * 2.0 ** -21, amax * 2.0 ** -38)).all()
    assert torch.isfinite(xs.planes.float()).all()
    assert xs.planes[0].abs().max().item() < 2.0 ** 15
    # bound_mul: a looser (guaranteed) bound only costs fixed-point range
    xs2 = K.split_f16x2(x.to(DEV).contiguous(), bound_mul=64.0)
    assert int(xs2.sexp.item()) == int(xs.sexp.item()) - 6
    assert ((xs2.float().cpu() - x).abs() <= torch.maximum(x.abs() * 2.0 ** -21, amax * 2.0 ** -32)).all()
    z = K.split_f16x2(torch.zeros(16, 32, device=DEV))
    assert torch.equal(z.float(), torch.zeros(16, 32, device=DEV))


@pytest.mark.parametrize("shape", C4_SHAPES, ids=[f"{c[0]}-{c[1]}-k{c[2]}s{c[3]}-{c[5]}x{c[5]}" for c in C4_SHAPES])
@pytest.mark.parametrize("config", [2] + [2 | (t << 12) for t in range(1, 6)])  # bits 12..14: an explicit tile shape instead of the occupancy rule
def test_backward_data_on_the_c4_layer_shapes(shape, config):
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    cin, cout, k, s, p, H = shape
    m = _conv(cin, cout, k, s, p)
    Ho = (H + 2 * p - k) // s + 1
    N = 19  # odd: the last pixel tile is ragged
    g = torch.randn(N, cout, Ho, Ho, device=DEV)
    want = torch.nn.grad.conv2d_input((N, cin, H, H), m.weight.double().cpu(), g.double().cpu(), stride=s, padding=p)
    gs = K.split_f16x2(g.permute(0, 2, 3, 1).contiguous())
    prep = cv.PreparedConv(m)
    amax = torch.zeros(1, dtype=torch.float32, device=DEV)
    prev = K.conv_config
    K.conv_config = config
    try:
        dx = cv.conv_backward_data(prep, gs, (H, H), amax_out=amax)
    finally:
        K.conv_config = prev
    got = dx.permute(0, 3, 1, 2)
    assert rel(got, want) < 1e-5, rel(got, want)
    assert abs(amax.item() - got.abs().max().item()) <= 1e-6 * amax.item()
    # deferred BatchNorm scale folded into the weights: backward-data of `scale[co] * W`
    sc = (torch.rand(cout, device=DEV) + 0.5).contiguous()
    want2 = torch.nn.grad.conv2d_input((N, cin, H, H), (m.weight * sc.reshape(-1, 1, 1, 1)).double().cpu(),
                                       g.double().cpu(), stride=s, padding=p)
    dx2 = cv.conv_backward_data(prep, gs, (H, H), cscale=sc)
    assert rel(dx2.permute(0, 3, 1, 2), want2) < 1e-5
    # residual accumulate: out += backward-data (the down-sampling branch adds into the main branch's cotangent)
    base = torch.randn_like(dx)
    acc = base.clone()
    cv.conv_backward_data(prep, gs, (H, H), out=acc, accumulate=True)
    assert rel(acc - base, dx) < 1e-5


@pytest.mark.parametrize("shape", [(64, 64, 3, 1, 1, 32), (64, 128, 3, 2, 1, 32), (64, 128, 1, 2, 0, 32), (512, 512, 3, 1, 1, 4),
                                   (96, 160, 3, 1, 1, 7), (32, 32, 2, 1, 0, 9), (32, 64, 3, 3, 1, 11)])
def test_forward_and_odd_geometries(shape):
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    cin, cout, k, s, p, H = shape
    m = _conv(cin, cout, k, s, p)
    N = 5
    x = torch.randn(N, cin, H, H + 1, device=DEV)
    want = F.conv2d(x.double().cpu(), m.weight.double().cpu(), None, s, p)
    prep = cv.PreparedConv(m)
    y = cv.conv_forward(prep, K.split_f16x2(x.permute(0, 2, 3, 1).contiguous()))
    assert rel(y.permute(0, 3, 1, 2), want) < 1e-5
    # and the matching backward-data (non-square map, stride that does not divide the size)
    g = torch.randn_like(y)
    wantb = torch.nn.grad.conv2d_input((N, cin, H, H + 1), m.weight.double().cpu(), g.permute(0, 3, 1, 2).double().cpu(),
                                       stride=s, padding=p)
    dx = cv.conv_backward_data(prep, K.split_f16x2(g), (H, H + 1))
    assert rel(dx.permute(0, 3, 1, 2), wantb) < 1e-5


def test_tiny_and_huge_magnitudes_keep_fp32_level_accuracy():
    """cotangents of 1e-20 and weights of 1e+6: the power-of-two scales keep both inside fp16's range"""
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    m = _conv(64, 64, 3, 1, 1)
    with torch.no_grad():
        m.weight.mul_(1e6)
    g = torch.randn(3, 64, 8, 8, device=DEV) * 1e-20
    want = torch.nn.grad.conv2d_input((3, 64, 8, 8), m.weight.double().cpu(), g.double().cpu(), stride=1, padding=1)
    dx = cv.conv_backward_data(cv.PreparedConv(m), K.split_f16x2(g.permute(0, 2, 3, 1).contiguous()), (8, 8))
    assert rel(dx.permute(0, 3, 1, 2), want) < 1e-5


@pytest.mark.parametrize("shape", [(64, 3, 1, 1, 32), (128, 3, 1, 1, 16), (64, 3, 2, 1, 32), (512, 3, 1, 1, 4), (64, 1, 2, 0, 32)])
def test_rotation_convolution_with_position_contiguous_output(shape):
    """conv2d with an arbitrary filter bank (the eigenvectors of an A factor) and [B, Dk, L] output — the Kron
    predictive's rotation of the unfolded inputs (matrix.py:406-456) on the implicit-GEMM kernel."""
    from laplace_amd import conv as cv

    cin, k, s, p, H = shape
    m = _conv(cin, 32, k, s, p)
    Dk = cin * k * k
    torch.manual_seed(1)
    Q2 = torch.linalg.qr(torch.randn(Dk, Dk, device=DEV))[0].contiguous()
    filt = Q2.T.reshape(Dk, cin, k, k)
    a = torch.randn(6, cin, H, H, device=DEV)
    want = F.conv2d(a.double().cpu(), filt.double().cpu(), None, s, p)
    for layout in ("nchw", "nhwc"):
        x = a if layout == "nchw" else a.to(memory_format=torch.channels_last)
        got = cv.conv_forward_filters(m, x, filt, Q2)
        assert got.shape == want.shape and got.is_contiguous()
        assert rel(got, want) < 1e-5


def test_unsplit_transpose_of_a_seed_batched_cotangent():
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    S, B, H, W, C = 3, 5, 6, 7, 40
    x = torch.randn(S * B, H, W, C, device=DEV)
    u = K.unsplit_transpose(K.split_f16x2(x), S, B)
    want = x.reshape(S, B, H * W, C).permute(1, 0, 3, 2)
    assert u.shape == (B, S, C, H * W) and rel(u, want) < 1e-6


FUSED_SHAPES = [(64, 64, 3, 1, 1, 32), (128, 128, 3, 1, 1, 16), (256, 256, 3, 1, 1, 8), (512, 512, 3, 1, 1, 4),
                (96, 160, 3, 1, 1, 7), (64, 32, 1, 1, 0, 5)]


@pytest.mark.parametrize("shape", FUSED_SHAPES, ids=[f"{c[0]}-{c[1]}-k{c[2]}-{c[5]}x{c[5]}" for c in FUSED_SHAPES])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("variant", ["mask+scale", "mask+add", "float-mult+add+scale", "plain"])
def test_backward_data_with_the_fused_vjp_epilogue(shape, tile, variant):
    """lk_conv_nhwc_f16x2_vjp against the fp64 composition it replaces — backward-data, residual add, activation
    multiplier, channel scale, split — on the stride-1 convolution shapes of c4 (and two odd ones), every tile shape."""
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    cin, cout, k, s, p, H = shape
    m = _conv(cin, cout, k, s, p)
    Ho = (H + 2 * p - k) // s + 1
    S, B = 3, 5  # three seeds of five samples: the multiplier is shared by the seeds; 15 images: ragged last tile
    N = S * B
    torch.manual_seed(11)
    g = torch.randn(N, cout, Ho, Ho, device=DEV) * 3e-3
    want = torch.nn.grad.conv2d_input((N, cin, H, H), m.weight.double().cpu(), g.double().cpu(), stride=s, padding=p)
    want = want.permute(0, 2, 3, 1)  # NHWC
    gs = K.split_f16x2(g.permute(0, 2, 3, 1).contiguous())
    kw = {}
    if "add" in variant:
        addend = torch.randn(N, H, H, cin, device=DEV) * 0.02
        kw["add"] = K.split_f16x2(addend)
        want = want + kw["add"].float().double().cpu()
    if variant.startswith("mask"):
        mask = (torch.rand(B, H, H, cin, device=DEV) > 0.4)
        kw["mult"] = mask.to(torch.uint8)
        want = (want.reshape(S, B, H, H, cin) * mask.double().cpu()).reshape(N, H, H, cin)
    elif variant.startswith("float-mult"):
        mult = (torch.rand(B, H, H, cin, device=DEV) * 0.9 + 0.05).contiguous()
        kw["mult"], kw["mult_amax"] = mult, K.absmax(mult)
        want = (want.reshape(S, B, H, H, cin) * mult.double().cpu()).reshape(N, H, H, cin)
    if "scale" in variant:
        sc = (torch.rand(cin, device=DEV) * 1.5 + 0.25).contiguous()
        kw["scale"], kw["scale_amax"] = sc, K.absmax(sc)
        want = want * sc.double().cpu()
    prep = cv.PreparedConv(m)
    prev = K.conv_config
    K.conv_config = 2 | (tile << 12)
    try:
        out = cv.conv_backward_data_vjp(prep, gs, (H, H), **kw)
    finally:
        K.conv_config = prev
    got = out.float()
    assert rel(got, want) < 1e-5, rel(got, want)
    # the measured max rides along (it is the next launch's bound), and the planes respect the fixed-point range
    assert abs(out.amax.item() - got.abs().max().item()) <= 1e-5 * out.amax.item()
    assert out.planes[0].float().abs().max().item() < 2.0 ** 15
    # chained: the fused result as the input of another fused launch (bound from the measured max, not 2^(15 - sexp))
    if cin == cout and k == 3:
        out2 = cv.conv_backward_data_vjp(prep, out, (H, H))
        want2 = torch.nn.grad.conv2d_input((N, cin, H, H), m.weight.double().cpu(), want.permute(0, 3, 1, 2).contiguous(),
                                           stride=s, padding=p).permute(0, 2, 3, 1)
        assert rel(out2.float(), want2) < 1e-5


SMALL_MAPS = [(512, 512, 3, 1, 1, 4), (256, 256, 3, 1, 1, 8), (256, 512, 3, 2, 1, 8), (256, 512, 1, 2, 0, 8), (64, 96, 3, 1, 1, 2),
              (64, 64, 3, 1, 1, 1)]


@pytest.mark.parametrize("shape", SMALL_MAPS, ids=[f"{c[0]}-{c[1]}-k{c[2]}s{c[3]}-{c[5]}x{c[5]}" for c in SMALL_MAPS])
@pytest.mark.parametrize("config", [2, 2 | 32768, 2 | (1 << 12), 2 | (4 << 12)])
def test_position_major_rows_on_small_maps(shape, config):
    """Maps of at most 64 pixels with at least 64 images run with GEMM rows ordered (pixel, image), and taps that reach
    no row of a tile leave its K loop (lk_conv.hip): backward-data, forward and the fused epilogue against fp64, with
    an image count that is no multiple of any tile (ragged tiles straddle pixel positions); bit 15 = the plain order."""
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    cin, cout, k, s, p, H = shape
    m = _conv(cin, cout, k, s, p)
    Ho = (H + 2 * p - k) // s + 1
    S, B = 2, 37
    N = S * B
    torch.manual_seed(5)
    g = torch.randn(N, cout, Ho, Ho, device=DEV)
    x = torch.randn(N, cin, H, H, device=DEV)
    want_b = torch.nn.grad.conv2d_input((N, cin, H, H), m.weight.double().cpu(), g.double().cpu(), stride=s, padding=p)
    want_f = F.conv2d(x.double().cpu(), m.weight.double().cpu(), stride=s, padding=p)
    gs = K.split_f16x2(g.permute(0, 2, 3, 1).contiguous())
    xs = K.split_f16x2(x.permute(0, 2, 3, 1).contiguous())
    prep = cv.PreparedConv(m)
    prev = K.conv_config
    K.conv_config = config
    try:
        dx = cv.conv_backward_data(prep, gs, (H, H))
        y = cv.conv_forward(prep, xs)
        fused = None
        if cv.fused_backward_ok(m):
            mask = (torch.rand(B, H, H, cin, device=DEV) > 0.3).to(torch.uint8)
            addend = K.split_f16x2(torch.randn(N, H, H, cin, device=DEV))
            fused = cv.conv_backward_data_vjp(prep, gs, (H, H), add=addend, mult=mask)
    finally:
        K.conv_config = prev
    assert rel(dx.permute(0, 3, 1, 2), want_b) < 1e-5
    assert rel(y.permute(0, 3, 1, 2), want_f) < 1e-5
    if fused is not None:
        want = (want_b.permute(0, 2, 3, 1) + addend.float().double().cpu()).reshape(S, B, H, H, cin) * mask.double().cpu()
        assert rel(fused.float(), want.reshape(N, H, H, cin)) < 1e-5


WINP_CASES = [(64, 64, 32, 131), (64, 64, 20, 131), (128, 64, 16, 290), (128, 128, 16, 37), (256, 256, 8, 300), (64, 128, 12, 75),
              (512, 512, 4, 200), (64, 64, 32, 1152), (64, 96, 9, 64),
              # the split tail (leftover tiles of the last round in K slices): the c4 launches, and counts that leave 40 / 88 tiles
              (128, 128, 16, 1152), (256, 256, 8, 1152), (512, 512, 4, 1152), (512, 512, 4, 1100), (128, 128, 16, 300)]


@pytest.mark.parametrize("cin,cout,H,n_img", WINP_CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}x{c[2]}-n{c[3]}" for c in WINP_CASES])
def test_persistent_window_form_of_the_fused_launch(cin, cout, H, n_img):
    """conv_winp_f16x2_kernel (lk_conv_nhwc_f16x2_vjp_wc: chunk-major weights, two persistent workgroups per CU, the next
    tile's operands requested under the epilogue) against fp64 and against the generic kernel: ragged last tiles, tile
    strides that are no whole number of images, several 64-channel column tiles, several tiles per workgroup (the 1152-image
    case is the c4 launch), with and without addend / mask / channel scale."""
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    m = _conv(cin, cout, 3, 1, 1)
    N = n_img
    torch.manual_seed(29)
    g = torch.randn(N, cout, H, H, device=DEV) * 1e-2
    gs = K.split_f16x2(g.permute(0, 2, 3, 1).contiguous())
    prep = cv.PreparedConv(m)
    assert K.conv_winp_eligible(N, H, H, cout, cin, 9)
    S = 3 if N % 3 == 0 else 1
    B = N // S
    mask = (torch.rand(B, H, H, cin, device=DEV) > 0.4).to(torch.uint8)
    addend = K.split_f16x2(torch.randn(N, H, H, cin, device=DEV) * 0.05)
    sc = (torch.rand(cin, device=DEV) * 1.5 + 0.25).contiguous()
    want_b = None
    if N * H * H * cin <= 40e6:  # (the fp64 reference on the host: small cases only)
        want_b = torch.nn.grad.conv2d_input((N, cin, H, H), m.weight.double().cpu(), g.double().cpu(), stride=1, padding=1)
    prev = K.conv_config
    try:
        for kw in ({}, {"add": addend}, {"mult": mask}, {"add": addend, "mult": mask, "scale": sc, "scale_amax": K.absmax(sc)}):
            K.conv_config = 2
            fused = cv.conv_backward_data_vjp(prep, gs, (H, H), **kw)
            K.conv_config = 2 | (1 << 25)  # (bit 25: the last round's leftover tiles in K slices, summed through slabs in slice order)
            split = cv.conv_backward_data_vjp(prep, gs, (H, H), **kw)
            K.conv_config = 2 | (1 << 26)  # (bit 26: one eight-wave workgroup per CU on 512-pixel tiles)
            wide = cv.conv_backward_data_vjp(prep, gs, (H, H), **kw)
            K.conv_config = 2 | (1 << 27)  # (bit 27: the persistent form off)
            ref = cv.conv_backward_data_vjp(prep, gs, (H, H), **kw)
            assert torch.equal(fused.sexp, ref.sexp) and rel(fused.float(), ref.float()) < 4e-6, sorted(kw)  # (other order of the K steps)
            assert torch.equal(split.sexp, ref.sexp) and rel(split.float(), ref.float()) < 4e-6, sorted(kw)
            assert torch.equal(wide.sexp, ref.sexp) and rel(wide.float(), ref.float()) < 4e-6, sorted(kw)
            assert abs(split.amax.item() - split.float().abs().max().item()) <= 1e-5 * split.amax.item()
            assert abs(fused.amax.item() - fused.float().abs().max().item()) <= 1e-5 * fused.amax.item()
            if want_b is not None:
                want = want_b.permute(0, 2, 3, 1)
                if "add" in kw:
                    want = want + addend.float().double().cpu()
                if "mult" in kw:
                    want = (want.reshape(S, B, H, H, cin) * mask.double().cpu()).reshape(N, H, H, cin)
                if "scale" in kw:
                    want = want * sc.double().cpu()
                assert rel(fused.float(), want) < 1e-5, sorted(kw)
    finally:
        K.conv_config = prev


STRIDED_CASES = [(64, 128, 32, 15, True), (64, 128, 32, 1152, True), (128, 256, 16, 75, True), (256, 512, 8, 300, True),
                 (128, 256, 16, 33, False), (32, 64, 8, 12, True), (64, 96, 12, 21, True), (192, 64, 6, 40, True)]


@pytest.mark.parametrize("cin,cout,H,n_img,shortcut", STRIDED_CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}x{c[2]}-n{c[3]}-{'pair' if c[4] else 'single'}" for c in STRIDED_CASES])
def test_strided_fused_launch_of_a_downsampling_block(cin, cout, H, n_img, shortcut):
    """conv_strided_f16x2_kernel (lk_conv_nhwc_f16x2_vjp_strided): the backward-data of the 3 x 3 / stride-2 convolution of a
    residual down-sampling block — four residue classes of 1 / 2 / 2 / 4 taps — together with the block's 1 x 1 / stride-2
    shortcut (its own cotangent, weights and fixed-point units), with and without addend / mask / channel scale, against
    fp64 and against the class-by-class route (five launches into an fp32 tensor + the element-wise VJP kernel); image
    counts that leave the last tile of every class ragged; the 1152-image case is the first down-sampling block of c4."""
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    m1, m2 = _conv(cin, cout, 3, 2, 1), _conv(cin, cout, 1, 2, 0)
    N, Ho = n_img, H // 2
    torch.manual_seed(31)
    g1 = torch.randn(N, cout, Ho, Ho, device=DEV) * 1e-2
    g2 = torch.randn(N, cout, Ho, Ho, device=DEV) * 3.0  # (another magnitude: the two sources carry different units)
    gs1 = K.split_f16x2(g1.permute(0, 2, 3, 1).contiguous())
    gs2 = K.split_f16x2(g2.permute(0, 2, 3, 1).contiguous())
    p1, p2 = cv.PreparedConv(m1), cv.PreparedConv(m2)
    assert cv.strided_fused_ok(m1, (H, H)) and cv.strided_fused_ok(m2, (H, H))
    cs2 = (torch.rand(cout, device=DEV) + 0.5).contiguous()  # a BatchNorm scale folded into the shortcut's weights
    descs = [(p1, gs1, None)] + ([(p2, gs2, cs2)] if shortcut else [])
    assert cv.strided_taps(descs, (H, H)) is not None and cv.strided_taps([(p2, gs2, None)], (H, H)) is None
    S = 3 if N % 3 == 0 else 1
    B = N // S
    mask = (torch.rand(B, H, H, cin, device=DEV) > 0.4).to(torch.uint8)
    fmult = (torch.rand(B, H, H, cin, device=DEV) * 2 - 0.5).contiguous()
    addend = K.split_f16x2(torch.randn(N, H, H, cin, device=DEV) * 0.05)
    sc = (torch.rand(cin, device=DEV) * 1.5 + 0.25).contiguous()
    want_b = None
    if N * H * H * cin <= 40e6:  # (the fp64 reference on the host: small cases only)
        want_b = torch.nn.grad.conv2d_input((N, cin, H, H), m1.weight.double().cpu(), g1.double().cpu(), stride=2, padding=1)
        if shortcut:
            w2 = m2.weight.double().cpu() * cs2.double().cpu().reshape(-1, 1, 1, 1)
            want_b = want_b + torch.nn.grad.conv2d_input((N, cin, H, H), w2, g2.double().cpu(), stride=2, padding=0)
    for kw in ({}, {"add": addend}, {"mult": mask}, {"mult": fmult, "mult_amax": K.absmax(fmult)},
               {"add": addend, "mult": mask, "scale": sc, "scale_amax": K.absmax(sc)}):
        fused = cv.conv_backward_data_vjp_strided(descs, (H, H), **kw)
        # the class-by-class route: fp32 tensor (the shortcut accumulates into it), then the element-wise VJP kernel
        w = torch.zeros(1, dtype=torch.float32, device=DEV)
        dx = cv.conv_backward_data(p1, gs1, (H, H), amax_out=w)
        if shortcut:
            cv.conv_backward_data(p2, gs2, (H, H), cscale=cs2, out=dx, accumulate=True, amax_out=w)
        ref = K.vjp_nhwc_split(dx, K.absmax(dx), kw.get("add"), kw.get("mult"), kw.get("mult_amax"), kw.get("scale"),
                               kw.get("scale_amax"), S, (N, H, H, cin))
        assert rel(fused.float(), ref.float()) < 4e-6, sorted(kw)
        assert abs(fused.amax.item() - fused.float().abs().max().item()) <= 1e-5 * fused.amax.item()
        assert fused.planes[0].abs().max().item() < 2.0 ** 15  # the guaranteed bound holds: no fixed-point overflow
        if want_b is not None:
            want = want_b.permute(0, 2, 3, 1)
            if "add" in kw:
                want = want + addend.float().double().cpu()
            if "mult" in kw:
                mm = kw["mult"].double().cpu()
                want = (want.reshape(S, B, H, H, cin) * mm).reshape(N, H, H, cin)
            if "scale" in kw:
                want = want * sc.double().cpu()
            assert rel(fused.float(), want) < 1e-5, sorted(kw)


@pytest.mark.parametrize("n,where", [(4096, 0), (4096, 4095), (4100, 2049), (4099, 4098), (1 << 22, 1234567), (3 * (1 << 20) + 8, 17)])
def test_absmax_vector_and_scalar_forms(n, where):
    """lk_absmax_f32 streams float4s when it can (16-byte aligned, n % 4 == 0, n >= 4096) and falls back to the scalar
    form otherwise: the planted extreme must be found wherever it sits, negative or not, and through an unaligned view."""
    from laplace_amd import _lib

    K = _lib.get_kernels()
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n + 1, generator=g).to(DEV)
    for sign in (1.0, -1.0):
        y = x.clone()
        y[where] = sign * 77.5
        assert float(K.absmax(y[:n])[0]) == 77.5  # (a leading slice of a 1-D tensor is contiguous and keeps the alignment)
        tail = y[1:]  # data pointer 4 bytes past a 16-byte boundary: the scalar form
        assert float(K.absmax(tail)[0]) == float(tail.abs().max())


IM2COL_CASES = [(64, 32, 3, 2, 1, 6), (128, 16, 3, 2, 1, 5), (256, 8, 3, 2, 1, 9), (3, 32, 3, 1, 1, 4), (64, 32, 1, 2, 0, 3), (24, 10, 3, 1, 0, 2),
                (16, 9, 5, 2, 2, 3)]


@pytest.mark.parametrize("cin,H,k,s,p,B", IM2COL_CASES, ids=[f"c{c[0]}-{c[1]}x{c[1]}-k{c[2]}s{c[3]}p{c[4]}-b{c[5]}" for c in IM2COL_CASES])
def test_patch_matrix_as_split_planes_and_its_gram(cin, H, k, s, p, B):
    """lk_im2col_split_f16x2 (the unfolded inputs of a convolution as split planes, native column order, zero padded) against
    F.unfold in fp64, and lk_gram_tn_f16x2 on it against the exact-fp32 MFMA kernel it replaces for the A factors of strided / stem
    convolutions (lk_gram_conv_nhwc_f32) and against the fp64 Gram (curvlinops.py:55-75: A = sum of patch outer products)."""
    from laplace_amd._lib import get_kernels, keep_layout

    K = get_kernels()
    torch.manual_seed(cin + H + k)
    x = (torch.randn(B, cin, H, H, device=DEV).relu_() * 3.0).contiguous(memory_format=torch.channels_last)
    n = cin * k * k
    Kp = n if (n == 64 or n % 128 == 0) else (64 if n < 64 else (n + 127) // 128 * 128)
    pm = K.im2col_split(keep_layout(x), (k, k), s, p, Kp)
    cols = F.unfold(x.double().cpu().contiguous(), k, 1, p, s)                    # [B, cin k k, L], (ci, kh, kw)
    L = cols.shape[2]
    want = cols.reshape(B, cin, k * k, L).permute(0, 3, 2, 1).reshape(B * L, n)   # rows (b, oh, ow), columns (kh, kw, ci)
    got = pm.float().double().cpu()
    assert tuple(got.shape) == (B * L, Kp) and not pm.per_image
    assert rel(got[:, :n], want) < 1e-6 and float(got[:, n:].abs().max() if Kp > n else 0.0) == 0.0
    top = got.abs().max() * 2.0 ** float(pm.sexp.item())
    assert 2.0 ** 13 < top < 2.0 ** 15   # the scale puts the largest entry under 2^15
    A16 = torch.zeros(Kp, Kp, device=DEV)
    K.gram_tn_f16x2(pm, 0.25, A16)
    G = 0.25 * (want.T @ want)
    blk = torch.arange(Kp) // 32
    upper = (blk[:, None] <= blk[None, :])[:n, :n]
    assert rel(A16.double().cpu()[:n, :n] * upper, G * upper) < 2e-6
    A32 = torch.zeros(n, n, device=DEV)
    K.gram_conv(keep_layout(x), (k, k), (s, s), (p, p), (1, 1), 0.25, A32, upper_only=True, native=True)
    tri = torch.triu(torch.ones(n, n, dtype=torch.bool))
    assert rel(A16.double().cpu()[:n, :n] * tri, A32.double().cpu() * tri) < 2e-6
