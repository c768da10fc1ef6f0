"""The Kron predictive's quadratic form on PRE-SPLIT operands (round 5): lk_conv_nhwc_f16x2_planes (the eigenbasis rotations
emit two fp16 planes, position-contiguous, scaled from a guaranteed bound) and lk_kron_quadform_shared_planes_f16x2 (stages
them as they are: three v_mfma_f32_32x32x16_f16 per product block, no in-flight splitting) against fp64 on the c4 layer
shapes, and the whole predictive through them against the fp32-operand route.  Replaces, for weight-sharing layers,
KronDecomposed._bmm / inv_square_form under KronLaplace.functional_variance (laplace/utils/matrix.py:406-461,
baselaplace.py:1834-1835).  -m gpu only (LK_TEST_DEVICE=cpu: self-check on the kernel emulation)."""
import pytest
import torch
import torch.nn.functional as F
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


def rel(a, b):
    from tests.parity_log import record_error

    a, b = a.double().cpu(), b.double().cpu()
    return record_error((a - b).abs().max().item() / (b.abs().max().item() + 1e-300))


def rel_rows(a, b):
    from tests.parity_log import record_error

    a, b = a.double().cpu().flatten(1), b.double().cpu().flatten(1)
    return record_error(((a - b).abs().amax(1) / (b.abs().amax(1) + 1e-300)).max().item())


CONVS = [(64, 64, 3, 1, 1, 32, 6), (128, 128, 3, 1, 1, 16, 6), (512, 512, 3, 1, 1, 4, 70), (64, 128, 3, 2, 1, 32, 4), (64, 128, 1, 2, 0, 32, 4)]


@pytest.mark.parametrize("cfg", CONVS, ids=[f"{c[0]}-{c[1]}-k{c[2]}s{c[3]}-{c[5]}x{c[5]}-b{c[6]}" for c in CONVS])
def test_rotation_convolutions_emit_split_planes(cfg):
    """v = the unfolded inputs in an eigenbasis (filter bank = eigenvectors) from a per-image operand, and u = Q^T g over a
    one-scale cotangent: values against fp64, image by image against the image's own maximum; scales from the bound"""
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    cin, cout, k, s, p, H, B = cfg
    torch.manual_seed(cin + k + s)
    m = nn.Conv2d(cin, cout, k, s, p, bias=False).to(DEV)
    scale = (10.0 ** torch.linspace(-5, 4, B)).reshape(B, 1, 1, 1).to(DEV)
    x = torch.randn(B, cin, H, H, device=DEV).relu_() * scale
    Dk = cin * k * k
    Q2 = torch.linalg.qr(torch.randn(Dk, Dk, device=DEV))[0].contiguous()
    filt = Q2.T.reshape(Dk, cin, k, k)
    want = F.conv2d(x.double().cpu(), filt.double().cpu(), None, s, p).flatten(2)
    Ho = (H + 2 * p - k) // s + 1
    if (Ho * Ho) % 4:
        pytest.skip("position-contiguous output needs Ho * Wo % 4 == 0")
    v = cv.conv_forward_filters(m, x, filt, Q2, planes=True)
    assert tuple(v.shape) == (B, Dk, Ho * Ho) and v.per_image and v.planes.dtype == torch.float16
    assert rel_rows(v.float(), want) < 1e-5
    top = v.float().abs().flatten(1).amax(1) * torch.exp2(v.sexp.float())
    assert bool((top < 2.0 ** 15).all())  # the bound holds: nothing overflows the planes
    # ... and the fp32 form of the same launch agrees to the last bits of the split
    v32 = cv.conv_forward_filters(m, x, filt, Q2).flatten(2)
    assert rel_rows(v.float(), v32) < 2e-6
    # u: 1x1 rotation of a split cotangent (ONE scale), [S * B, Ho, Ho, cout] -> [S * B, cout, L]
    S = 3
    g = torch.randn(S * B, Ho, Ho, cout, device=DEV)
    gs = K.split_f16x2(g.contiguous())
    Q1 = torch.linalg.qr(torch.randn(cout, cout, device=DEV))[0].contiguous()
    u = cv.rotate_channels(gs, Q1, Q1, planes=True)
    assert tuple(u.shape) == (S * B, cout, Ho * Ho) and not u.per_image
    want_u = torch.einsum("npc,cd->ndp", g.double().cpu().reshape(S * B, Ho * Ho, cout), Q1.double().cpu())
    assert rel(u.float(), want_u) < 2e-6


SHAPES = [(64, 576, 1024, 10, 3), (128, 1152, 256, 10, 4), (256, 2304, 64, 10, 8), (512, 4608, 16, 10, 16), (64, 27 * 8, 64, 3, 5),
          (32, 288, 16, 1, 9)]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"Do{s[0]}-Dk{s[1]}-L{s[2]}-C{s[3]}-B{s[4]}" for s in SHAPES])
def test_quadratic_form_on_split_planes(shape):
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    Do, Dk, L, C, B = shape
    torch.manual_seed(Do + L)
    u = torch.randn(C * B, Do, L, device=DEV) * 3e-3
    v = torch.randn(B, Dk, L, device=DEV) * (10.0 ** torch.linspace(-3, 2, B)).reshape(B, 1, 1).to(DEV)
    l1, l2 = torch.rand(Do, device=DEV) + 0.1, torch.rand(Dk, device=DEV) + 0.1
    delta = torch.tensor([0.7], device=DEV)
    us, vs = K.split_f16x2(u.contiguous()), K.split_images_f16x2(v.contiguous())
    fvar = torch.zeros(B, C, C, device=DEV)
    K.kron_quadform_shared_planes(us, vs, l1, l2, delta, fvar, C)
    u64 = u.double().cpu().reshape(C, B, Do, L).permute(1, 0, 2, 3)
    M = torch.einsum("ncol,nil->ncoi", u64, v.double().cpu())
    want = torch.einsum("ncoi,nkoi,oi->nck", M, M, 1.0 / (torch.outer(l1.double().cpu(), l2.double().cpu()) + 0.7))
    # every sample against its own largest variance (v spans five decades); up to 2.4 M fp32 terms per pair sum
    assert rel_rows(fvar, want) < 1e-4
    # accumulates into fvar; v with one scale for the tensor is legal too
    K.kron_quadform_shared_planes(us, K.split_f16x2(v.contiguous()), l1, l2, delta, fvar, C)
    assert rel(fvar[-1:], 2 * want[-1:]) < 1e-4
    # the fp32-operand kernel (in-flight three-piece bf16 split) agrees
    f32 = torch.zeros(B, C, C, device=DEV)
    K.kron_quadform_shared(u.reshape(C, B, Do, L).contiguous(), v.contiguous(), l1, l2, delta, f32, seed_major=True)
    assert rel_rows(f32, want) < 1e-4


def test_predictive_through_the_planes_route_equals_the_fp32_operand_route():
    from laplace_amd import HipGGN
    from laplace_amd import predictive as Pr
    from laplace_amd._lib import get_kernels
    from laplace_amd.nets import ResNet18

    K = get_kernels()
    torch.manual_seed(711)
    model = ResNet18(10, act=torch.tanh).to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    X = torch.randn(6, 3, 32, 32, generator=g).to(DEV)
    y = torch.randint(10, (6,), generator=g).to(DEV)
    b = HipGGN(model, "classification")
    acc = b.kron_accumulator(50_000)
    acc.add_batch(X, y)
    _, H = acc.finalize()
    post = H.decompose() * (50_000 / 6.0) + torch.tensor(1.0, device=DEV)
    calls = []
    orig = K.kron_quadform_shared_planes
    K.kron_quadform_shared_planes = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        mu1, var1 = Pr.glm_variance_kron(b, X[:4], post)
    finally:
        del K.kron_quadform_shared_planes
    assert len(calls) >= 16  # every 3x3 convolution of the four stages (and the strided ones whose map has L % 16 == 0)
    prev = K.use_quad_planes
    K.use_quad_planes = False
    try:
        mu2, var2 = Pr.glm_variance_kron(b, X[:4], post)
    finally:
        K.use_quad_planes = prev
    assert rel(mu1, mu2) < 1e-6 and rel_rows(var1, var2) < 1e-5
