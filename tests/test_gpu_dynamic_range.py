"""Dynamic range of the split-fp16 scheme (csrc/lk_conv.hip), PER IMAGE / PER SAMPLE / PER BLOCK — each compared with
its own largest magnitude, as the reference's element-wise assertions imply (tests/test_baselaplace.py:334-410,
``rtol=1e-4``) — on inputs built to stress the scales: minibatches mixing images of very different magnitudes, seed
columns of a saturated softmax (root columns of ~1e-6 next to ~0.5), ReLU.  Tolerance 1e-4 (BASELINE.json).

Round 5: the FORWARD's split tensors carry one scale per image (tests/test_gpu_per_image.py), so an image keeps 2^-22 of
its own maximum whatever else is in its minibatch; the reverse sweep's cotangents keep one scale per tensor (absolute
error <= 2^-39 of the tensor's largest element): what consumes them are sums over samples (the factors of a fit), or
per-sample rows whose seeds are identity columns (the predictive, the Jacobians), where the samples' cotangents are of
one magnitude by construction.  The `range_guard` of rounds 3 - 4 (refuse / sweep in magnitude groups) is gone: the
whole-model test below runs ONE sweep.  -m gpu only."""
import copy
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
TOL = 1e-4

# the stride-1 3x3 convolutions of c4 (fused epilogue) and the strided ones (plain epilogue)
SHAPES = [(64, 64, 3, 1, 1, 32), (128, 128, 3, 1, 1, 16), (512, 512, 3, 1, 1, 4), (64, 128, 3, 2, 1, 32), (256, 512, 1, 2, 0, 8)]


def rel_rows(a, b):
    """worst over the leading dim of max|a_n - b_n| / max|b_n|: every image / sample against ITS OWN maximum"""
    a, b = a.double().cpu().flatten(1), b.double().cpu().flatten(1)
    from tests.parity_log import record_error

    return record_error(((a - b).abs().amax(1) / (b.abs().amax(1) + 1e-300)).max().item())


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    from tests.parity_log import record_error

    return record_error((a - b).abs().max().item() / (b.abs().max().item() + 1e-300))


def _conv(cin, cout, k, s, p):
    torch.manual_seed(cin * 7 + cout + k + s)
    return nn.Conv2d(cin, cout, k, s, p, bias=False).to(DEV)


def _image_scales(n, lo=-3.0, hi=3.0, seed=0):
    """10^U(lo, hi) per image, with the two extremes present"""
    g = torch.Generator().manual_seed(seed)
    e = torch.rand(n, generator=g) * (hi - lo) + lo
    e[0], e[1] = lo, hi
    return (10.0 ** e).to(DEV)


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{c[0]}-{c[1]}-k{c[2]}s{c[3]}-{c[5]}x{c[5]}" for c in SHAPES])
def test_backward_data_per_image_over_the_range_one_sweep_may_span(shape):
    """identity-like seeds (the predictive / Jacobian sweeps) x images spread over 2^16: every image's cotangent against
    its own maximum, plain and fused epilogue, and chained through a second fused launch"""
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    cin, cout, k, s, p, H = shape
    m = _conv(cin, cout, k, s, p)
    Ho = (H + 2 * p - k) // s + 1
    S, B = 9, 16
    N = S * B
    torch.manual_seed(3)
    img = _image_scales(B, 0.0, math.log10(2.0 ** 16))            # 1 ... 65536, both ends present
    sc = img.repeat(S).reshape(N, 1, 1, 1)                        # seed-major: the same image scale under every seed
    g = torch.randn(N, cout, Ho, Ho, device=DEV) * sc
    want = torch.nn.grad.conv2d_input((N, cin, H, H), m.weight.double().cpu(), g.double().cpu(), stride=s, padding=p)
    gs = K.split_f16x2(g.permute(0, 2, 3, 1).contiguous())
    prep = cv.PreparedConv(m)
    dx = cv.conv_backward_data(prep, gs, (H, H)).permute(0, 3, 1, 2)
    r = rel_rows(dx, want)
    assert r < 1e-5, f"per-image error {r:.2e} (tensor-wide {rel(dx, want):.2e})"
    # the fused epilogue re-splits its result with a scale from a GUARANTEED bound (max|in| l1(W)), loose by a few bits:
    # those bits come off the smallest image's precision (measured 1-2e-5 at a spread of 2^16), inside the 1e-4 bar
    if cv.fused_backward_ok(m):
        mask = (torch.rand(B, H, H, cin, device=DEV) > 0.4)
        out = cv.conv_backward_data_vjp(prep, gs, (H, H), mult=mask.to(torch.uint8)).float()
        want_f = (want.permute(0, 2, 3, 1).reshape(S, B, H, H, cin) * mask.double().cpu()).reshape(N, H, H, cin)
        r = rel_rows(out, want_f)
        assert r < TOL, f"fused epilogue, per-image error {r:.2e}"
        if cin == cout:  # chained: the error of a small image must not compound with the split's floor
            out2 = cv.conv_backward_data_vjp(prep, cv.conv_backward_data_vjp(prep, gs, (H, H)), (H, H)).float()
            want2 = torch.nn.grad.conv2d_input((N, cin, H, H), m.weight.double().cpu(), want, stride=s, padding=p)
            assert rel_rows(out2, want2.permute(0, 2, 3, 1)) < TOL


@pytest.mark.parametrize("shape", SHAPES[:3], ids=[f"{c[0]}-{c[5]}x{c[5]}" for c in SHAPES[:3]])
def test_saturated_softmax_seed_columns_through_backward_data_and_gram(shape):
    """the sweep of a FIT: seed columns from 0.5 down to 1e-6 (a saturated softmax) x per-image scales over three
    decades.  What a fit consumes are sums over (seed, sample, pixel): the G factor of the layer below — the Gram of the
    fused launch's result — against fp64, relative to the block's own maximum."""
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    cin, cout, k, s, p, H = shape
    m = _conv(cin, cout, k, s, p)
    S, B = 9, 16
    N = S * B
    torch.manual_seed(4)
    seed_scale = torch.tensor([0.5, 0.3, 1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-6], device=DEV)
    sc = (seed_scale[:, None] * _image_scales(B, -1.5, 1.5)[None, :]).reshape(N, 1, 1, 1)
    g = torch.randn(N, cout, H, H, device=DEV) * sc
    want = torch.nn.grad.conv2d_input((N, cin, H, H), m.weight.double().cpu(), g.double().cpu(), stride=s, padding=p)
    gs = K.split_f16x2(g.permute(0, 2, 3, 1).contiguous())
    out = cv.conv_backward_data_vjp(cv.PreparedConv(m), gs, (H, H))
    assert rel(out.float().permute(0, 3, 1, 2), want) < 1e-5
    G = torch.zeros(cin, cin, device=DEV)
    K.gram_tn_f16x2(out, 1.0, G)
    K.symmetrize(G)
    rows = want.permute(0, 2, 3, 1).reshape(-1, cin)
    assert rel(G, rows.T @ rows) < 1e-5


@pytest.mark.parametrize("shape", SHAPES[:3], ids=[f"{c[0]}-{c[5]}x{c[5]}" for c in SHAPES[:3]])
def test_forward_per_image_over_the_range_one_sweep_may_span(shape):
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    cin, cout, k, s, p, H = shape
    m = _conv(cin, cout, k, s, p)
    B = 32
    x = torch.randn(B, cin, H, H, device=DEV).relu_() * _image_scales(B, 0.0, math.log10(2.0 ** 16)).reshape(B, 1, 1, 1)
    want = F.conv2d(x.double().cpu(), m.weight.double().cpu(), None, s, p)
    y = cv.conv_forward(cv.PreparedConv(m), K.split_f16x2(x.permute(0, 2, 3, 1).contiguous())).permute(0, 3, 1, 2)
    r = rel_rows(y, want)
    assert r < 1e-5, f"per-image error {r:.2e}"


def test_g_factor_gram_with_saturated_softmax_seed_columns():
    """the Gram of a cotangent whose seeds span six decades: every factor block against its own maximum"""
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    for C, L in ((64, 1024), (128, 256), (512, 16)):
        S, B = 9, 16
        seed_scale = torch.tensor([0.5, 0.3, 1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-6], device=DEV)
        hw = int(math.isqrt(L))
        g = torch.randn(S, B, hw, hw, C, device=DEV) * seed_scale.reshape(S, 1, 1, 1, 1)
        G = torch.zeros(C, C, device=DEV)
        K.gram_tn_f16x2(K.split_f16x2(g.reshape(S * B, hw, hw, C).contiguous()), 1.0, G)
        K.symmetrize(G)
        rows = g.double().reshape(-1, C)
        assert rel(G, rows.T @ rows) < 1e-5


def _adversarial_c4(act):
    from laplace_amd.nets import ResNet18

    torch.manual_seed(711)
    m = ResNet18(10, act=act)
    with torch.no_grad():
        m.fc.weight.mul_(20.0)  # logits x 20: saturated softmax (root columns of ~1e-6 next to ~0.5)
        m.fc.bias.mul_(20.0)
    return m


@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_c4_factors_and_predictive_per_sample_on_adversarial_inputs(act):
    """Whole model, config c4: logits x 20 (saturated softmax) and a minibatch mixing images scaled 1e-3 ... 1e+3.
    KFAC factors block by block against the fp64 oracle (curvlinops.py:77-108), GLM predictive variances SAMPLE BY SAMPLE
    against the oracle's Jacobians pushed through matrix.py:406-461 — a small-gradient test point in a batch with a
    large-gradient one is where a tensor-wide scale would show.

    ONE sweep, no guard (rounds 3 - 4 refused this minibatch by default and needed magnitude groups to meet the bar: with
    one scale per tensor the small images' activations were resolved to 2^-39 of the LARGEST image, and the ReLU masks
    decided on them moved one G block by 3e-3).  Bar: 1e-4 block by block, tanh AND ReLU.  ReLU additionally flips the
    pre-activations that sit within fp32 rounding of zero between any two executions; how much that moves a block is not
    asserted from an explanation but MEASURED here: the same oracle run in fp32 on the CPU against its fp64 run — the
    kernels must be within 2x of that where it exceeds the bar (it does not on this input: recorded in the parity log)."""
    from laplace_amd import HipGGN
    from laplace_amd import predictive as Pr
    from oracle import curvature_oracle as co
    from tests.parity_log import record_error

    m32 = _adversarial_c4(torch.relu if act == "relu" else torch.tanh)
    m64 = copy.deepcopy(m32).double().cpu().eval()
    m32cpu = copy.deepcopy(m32).cpu().eval()
    m32 = m32.to(DEV).eval()
    g = torch.Generator().manual_seed(11)
    B = 8
    X = torch.randn(B, 3, 32, 32, generator=g)
    X *= torch.tensor([1e-3, 1e3, 1.0, 1e-2, 30.0, 1e-3, 1e3, 0.3]).reshape(B, 1, 1, 1)
    y = torch.randint(10, (B,), generator=g)
    N = 50_000
    b = HipGGN(m32, "classification")
    assert not hasattr(b, "range_guard")
    acc = b.kron_accumulator(N)
    acc.add_batch(X.to(DEV), y.to(DEV))
    loss, kron = acc.finalize()                       # (rounds 3 - 4: RuntimeError here)
    loss_ref, kf_ref = co.kfac_ggn(m64, X.double(), y, N, "classification")
    _, kf_32 = co.kfac_ggn(m32cpu, X, y, N, "classification")   # the reference's arithmetic: fp32 on the CPU
    assert rel(loss, loss_ref) < TOL
    worst = worst32 = 0.0
    for i, (F_, G_, G32) in enumerate(zip(kron.kfacs, kf_ref, kf_32)):
        for j, (a_, w_, w32) in enumerate(zip(F_, G_, G32)):
            r = rel(a_, w_)
            r32 = (w32.double() - w_).abs().max().item() / (w_.abs().max().item() + 1e-300)
            worst, worst32 = max(worst, r), max(worst32, r32)
            assert r < max(TOL, 2.0 * r32), (f"{act}, block {i} factor {j} (n={a_.shape[0]}): rel to the block's own max "
                                             f"{r:.2e}; the fp32 oracle is at {r32:.2e}")
    record_error(worst32, "fp32-oracle")  # what fp32 arithmetic itself does to the worst block (its own log entry)
    print(f"adversarial c4 / {act}: worst factor block in one sweep {worst:.2e} (the fp32 CPU oracle: {worst32:.2e})")
    if act == "relu":
        return  # (the predictive half runs once, on the smooth network: the fp64 Jacobians of the oracle take a minute)
    # ONE posterior on both sides (the predictive kernels are what is compared here; the eigensolver has its own tests):
    # our decomposition of the ORACLE's factors, its eigenpairs handed to the oracle's matrix.py:406-461 in fp64
    from laplace_amd.kron import HipKron

    dec_ref = HipKron([[M.to(DEV).float() for M in F_] for F_ in kf_ref]).decompose()
    dec_ref.check_converged()
    hf = float(N) / B
    Qs = [[Q.double() for Q in blk] for blk in dec_ref.eigenvectors]
    ls = co.krondecomposed_scale([[l.double() for l in blk] for blk in dec_ref.eigenvalues], hf)
    # prior two decades below the largest curvature eigenvalue: the posterior precision's condition number stays ~1e2,
    # so that a 1e-4 bar on a variance measures the kernels and not the fp32 storage of H
    prior = 1e-2 * max(math.prod(float(l.max()) for l in blk) for blk in ls)
    post = dec_ref * hf + torch.tensor(prior, device=DEV, dtype=torch.float32)
    Xt = torch.stack([X[0] * 1e-4, X[1] * 1e3, X[2]])   # 1e-7, 1e+6, 1: thirteen decades in one call, ONE sweep
    f_mu, f_var = Pr.glm_variance_kron(b, Xt.to(DEV), post)
    Jt, ft = co.jacobians(m64, Xt.double())
    want = co.krondecomposed_inv_square_form_blocks(Qs, ls, prior, Jt.to(DEV))
    r_mu, r_var = rel_rows(f_mu, ft), rel_rows(f_var, want)
    print(f"adversarial c4 / {act}, thirteen decades in one predictive call: per-sample f_mu {r_mu:.2e}, f_var {r_var:.2e}; "
          f"variance maxima {[f'{v:.1e}' for v in want.abs().flatten(1).amax(1).tolist()]}")
    assert r_mu < TOL
    assert r_var < TOL, f"{act}: per-sample f_var error {r_var:.2e}"


def test_per_sample_jacobians_of_a_homogeneous_network_over_twelve_decades():
    """`backend.jacobians` (curvature.py:88-129) on a ReLU network — positively homogeneous: the Jacobian rows of the
    first layers scale with the image — for a minibatch spanning twelve decades in ONE sweep: every sample's Jacobian
    against the fp64 oracle relative to that sample's own largest entry, and through the reference's own quadratic form
    `J diag(v) J^T` sample by sample."""
    from laplace_amd import HipGGN
    from laplace_amd.nets import ResNet18
    from oracle import curvature_oracle as co

    torch.manual_seed(3)
    m32 = ResNet18(10)
    m64 = copy.deepcopy(m32).double().cpu().eval()
    m32 = m32.to(DEV).eval()
    X = torch.randn(4, 3, 32, 32) * torch.tensor([1e-6, 1e6, 1.0, 1e-3]).reshape(4, 1, 1, 1)
    Js, f = HipGGN(m32, "classification").jacobians(X.to(DEV))
    Jt, ft = co.jacobians(m64, X.double())
    assert rel_rows(f, ft) < TOL
    assert rel_rows(Js, Jt) < TOL


def test_one_scale_per_tensor_floor_and_what_the_per_image_split_makes_of_it():
    """2^-39 of the tensor's largest element is the floor of ONE scale per tensor: an image 1e-9 below the largest one in
    its minibatch keeps ~2^-9 — pinned, so that the limit of the reverse sweep's format is a documented number; the
    forward's per-image split (lk_split_images_f16x2) gives the same image 2^-21."""
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    x = torch.randn(2, 8, 8, 64, device=DEV)
    x[1] *= 1e-9
    back = K.split_f16x2(x.contiguous()).float()
    assert rel_rows(back[:1], x[:1]) < 2.0 ** -21
    floor = ((back[1:] - x[1:]).abs().max() / x[1:].abs().max()).item()  # (the pinned LIMIT, not a parity number: not logged)
    assert 2.0 ** -14 < floor < 2.0 ** -6
    back = K.split_images_f16x2(x.contiguous()).float()
    assert rel_rows(back, x) < 2.0 ** -21
