"""The split-fp16 scheme uses ONE power-of-two scale per tensor (all seeds x all samples x all pixels of a cotangent, a
whole minibatch of activations; csrc/lk_conv.hip).  A norm-wise tolerance over the whole tensor cannot see what that
costs an element far below the tensor's maximum, so these tests are PER IMAGE / PER SAMPLE / PER BLOCK — each compared
with its own largest magnitude, as the reference's element-wise assertions imply (tests/test_baselaplace.py:334-410,
``rtol=1e-4``) — on inputs built to stress the shared scale: minibatches mixing images of very different magnitudes,
seed columns of a saturated softmax (root columns of ~1e-6 next to ~0.5), ReLU.  Tolerance 1e-4 (BASELINE.json).

What the scheme guarantees: absolute error <= 2^-39 of the tensor's largest element (times the
slack of the producer's bound), i.e. a sample whose own maximum is r times the tensor's keeps a relative accuracy of
2^-39 / r.  The product therefore keeps the samples of one sweep within 2^16 of each other wherever a result is per
sample (laplace_amd/backend.py: range_groups — the predictive and the Jacobians sweep wider minibatches in magnitude
groups; a fit records the spread and refuses a minibatch outside it unless ``range_guard = "group"``), which leaves
2^-23 of every sample's own maximum.  Sums over samples (the factors of a fit) only need the tensor-wide bound.
-m gpu only."""
import copy
import math

import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda"
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

    The minibatch spans six decades, more than the 2^16 one sweep resolves per sample: by default a fit REFUSES it
    (asserted), with ``range_guard = "group"`` it is swept in magnitude groups and must meet the bar; the predictive
    groups by itself.  tanh pins the kernels; ReLU (what bench.py times) additionally flips the pre-activations that
    sit within fp32 rounding of zero between two executions, which at 8 samples — of which the
    saturated softmax leaves three that carry the G factors — moves a block by up to ~1e-3: looser bar, stated."""
    from laplace_amd import HipGGN
    from laplace_amd import predictive as Pr
    from oracle import curvature_oracle as co

    m32 = _adversarial_c4(torch.relu if act == "relu" else torch.tanh)
    m64 = copy.deepcopy(m32).double().cpu().eval()
    m32 = m32.to(DEV).eval()
    tol_fit = TOL if act == "tanh" else 2e-3
    g = torch.Generator().manual_seed(11)
    B = 8
    X = torch.randn(B, 3, 32, 32, generator=g)
    X *= torch.tensor([1e-3, 1e3, 1.0, 1e-2, 30.0, 1e-3, 1e3, 0.3]).reshape(B, 1, 1, 1)
    y = torch.randint(10, (B,), generator=g)
    N = 50_000
    b = HipGGN(m32, "classification")
    acc = b.kron_accumulator(N)
    acc.add_batch(X.to(DEV), y.to(DEV))
    with pytest.raises(RuntimeError, match="range_guard"):
        acc.finalize()
    loss_ref, kf_ref = co.kfac_ggn(m64, X.double(), y, N, "classification")
    worst = {}
    for mode in ("off", "group"):
        b.range_guard = mode
        acc = b.kron_accumulator(N)
        acc.add_batch(X.to(DEV), y.to(DEV))
        loss, kron = acc.finalize()
        assert rel(loss, loss_ref) < TOL
        worst[mode] = max(rel(a_, w_) for F_, G_ in zip(kron.kfacs, kf_ref) for a_, w_ in zip(F_, G_))
    print(f"adversarial c4 / {act}: worst factor block, one sweep {worst['off']:.2e}, magnitude groups {worst['group']:.2e}")
    for i, (F_, G_) in enumerate(zip(kron.kfacs, kf_ref)):  # (the "group" fit)
        for j, (a_, w_) in enumerate(zip(F_, G_)):
            r = rel(a_, w_)
            assert r < tol_fit, f"{act}, block {i} factor {j} (n={a_.shape[0]}): rel to the block's own max {r:.2e}"
    b.range_guard = "check"
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
    Xt = torch.stack([X[0] * 1e-4, X[1] * 1e3, X[2]])   # 1e-7, 1e+6, 1: thirteen decades in one call
    assert Pr._range_groups(Xt.to(DEV)) is not None
    assert Pr._range_groups(X[2:3].expand(4, -1, -1, -1).to(DEV)) is None
    f_mu, f_var = Pr.glm_variance_kron(b, Xt.to(DEV), post)
    Jt, ft = co.jacobians(m64, Xt.double())
    want = co.krondecomposed_inv_square_form_blocks(Qs, ls, prior, Jt.to(DEV))
    r_mu, r_var = rel_rows(f_mu, ft), rel_rows(f_var, want)
    print(f"adversarial c4 / {act}, thirteen decades in one predictive call: per-sample f_mu {r_mu:.2e}, f_var {r_var:.2e}; "
          f"variance maxima {[f'{v:.1e}' for v in want.abs().flatten(1).amax(1).tolist()]}")
    tol_pred = TOL if act == "tanh" else 1e-3
    assert r_mu < tol_pred
    assert r_var < tol_pred, f"{act}: per-sample f_var error {r_var:.2e}"
    # ... and what the shared scale alone would have made of it (reported, not asserted: it is why the guard exists)
    b.range_guard = "off"
    _, f_var_one = Pr.glm_variance_kron(b, Xt.to(DEV), post)
    b.range_guard = "check"
    print(f"adversarial c4 / {act}: the same call as ONE sweep: per-sample f_var error {rel_rows(f_var_one, want):.2e}")


def test_range_limit_of_the_shared_scale_is_where_the_design_says():
    """2^-39 of the tensor's largest element is the floor: an image 1e-9 below the largest one in its minibatch keeps
    ~2^-9 — outside the 1e-4 bar.  Pinned so that the limit is a documented number, not a surprise: minibatches beyond
    2^16 are swept in magnitude groups / refused (laplace_amd/backend.py: `range_groups`, `range_guard`)."""
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    x = torch.randn(2, 8, 8, 64, device=DEV)
    x[1] *= 1e-9
    back = K.split_f16x2(x.contiguous()).float()
    assert rel_rows(back[:1], x[:1]) < 2.0 ** -21
    assert 2.0 ** -14 < rel_rows(back[1:], x[1:]) < 2.0 ** -6
