"""Kernels of the NHWC split-fp16 reverse sweep on the MI355X (-m gpu): the element-wise VJP that emits split tensors
(lk_vjp_nhwc_split_f16x2), the Gram of a split tensor (lk_gram_tn_f16x2, transposing LDS reads), and the sweep as a
whole against one stock autograd pass per seed and against the NCHW sweep it replaces.  Tolerances: 1e-4 of the
largest element (BASELINE.json); measured ~1e-6."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    from tests.parity_log import record_error

    return record_error((a - b).abs().max().item() / (b.abs().max().item() + 1e-300))


def test_vjp_nhwc_split_all_operand_combinations():
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    torch.manual_seed(0)
    S, B, H, W, C = 3, 5, 6, 6, 64
    g = torch.randn(S * B, H, W, C, device=DEV) * 3.0
    g2f = torch.randn(S * B, H, W, C, device=DEV) * 0.01
    g2 = K.split_f16x2(g2f)
    mask = torch.rand(B, H, W, C, device=DEV) > 0.4
    multf = torch.randn(B, H, W, C, device=DEV) * 2.0
    scale = (torch.rand(C, device=DEV) + 0.5) * torch.where(torch.rand(C, device=DEV) > 0.5, 1.0, -1.0)
    g_amax, s_amax, m_amax = K.absmax(g), K.absmax(scale), K.absmax(multf)
    shape = (S * B, H, W, C)

    def want(gg, gg2, mult, sc):
        v = torch.zeros(shape, device=DEV, dtype=torch.float64)
        if gg is not None:
            v = v + gg.double()
        if gg2 is not None:
            v = v + gg2.double()
        if mult is not None:
            v = (v.reshape(S, B, H, W, C) * mult.double()).reshape(shape)
        if sc is not None:
            v = v * sc.double()
        return v

    cases = [
        (g, None, mask, scale), (g, g2, mask, scale), (None, g2, None, scale), (g, g2, None, None), (g, None, multf, None),
        (g, g2, multf, scale),
    ]
    for ci, (gg, gs, mult, sc) in enumerate(cases):
        out = K.vjp_nhwc_split(gg, None if gg is None else g_amax, gs, mult,
                               m_amax if (mult is not None and mult.dtype == torch.float32) else None, sc,
                               None if sc is None else s_amax, S, shape)
        ref = want(gg, None if gs is None else g2.float(), mult, sc)
        assert rel(out.float(), ref) < 2e-6, (ci, rel(out.float(), ref), int(out.sexp.item()))
        assert out.planes[0].abs().max().item() < 2.0 ** 15


@pytest.mark.parametrize("C,R", [(64, 1000), (64, 70001), (128, 4097), (256, 2500), (512, 1153), (384, 640)])
def test_gram_of_a_split_tensor(C, R):
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    torch.manual_seed(C + R)
    X = torch.randn(R, C, device=DEV) * torch.exp(torch.randn(R, 1, device=DEV))  # rows of different magnitude
    pad = (-R) % 8
    Xp = torch.cat([X, torch.zeros(pad, C, device=DEV)]) if pad else X
    xs = K.split_f16x2(Xp.contiguous())
    xs.planes = xs.planes[:, :R]  # ragged row count: the kernel zero-fills the last stage itself
    G0 = torch.randn(C, C, device=DEV)
    G = G0.clone()
    from laplace_amd._lib import SplitTensor

    K.gram_tn_f16x2(SplitTensor(xs.planes.contiguous(), xs.sexp), 0.5, G)
    want = G0.double() + 0.5 * (X.double().T @ X.double())
    idx = torch.arange(C, device=DEV) // 32
    upper = idx[:, None] <= idx[None, :]
    assert rel(torch.where(upper, G.double(), want), want) < 2e-6
    assert torch.equal(G[~upper], G0[~upper])  # tiles below the diagonal are not touched


@pytest.mark.parametrize("act", ["relu", "tanh"])
def test_split_sweep_against_autograd_on_a_small_resnet(act):
    from laplace_amd._lib import SplitTensor, get_kernels
    from laplace_amd.sweep_nhwc import SplitSweep
    from tests.test_sweep_nhwc import TinyResNet, _autograd_tap_grads

    torch.manual_seed(3)
    model = TinyResNet(torch.relu if act == "relu" else torch.tanh, width=64).to(DEV).eval()
    for mod in model.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.uniform_(0.5, 1.5)
    taps = {n: m for n, m in model.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))}
    sw = SplitSweep(model, taps, kernels=get_kernels)
    assert sw.split_ok, sw.split_reason
    x = torch.randn(6, 3, 16, 16, device=DEV)
    seeds = torch.randn(4, 6, 5, device=DEV)
    sw.forward(x)
    got = sw.backward(seeds, on_tap=lambda n, g: None, defer_bn_scale=True)
    # fp64 autograd on the host: independent of the device's convolution kernels
    m64 = TinyResNet(torch.relu if act == "relu" else torch.tanh, width=64).double().eval()
    m64.load_state_dict({k: v.double().cpu() for k, v in model.state_dict().items()})
    taps64 = {n: m for n, m in m64.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))}
    _, want = _autograd_tap_grads(m64, taps64, x.double().cpu(), seeds.double().cpu())
    for n in taps:
        g = got[n]
        if isinstance(g, SplitTensor):
            g = g.float().reshape(4, 6, *g.shape[1:]).permute(0, 1, 4, 2, 3)
        if n in sw.grad_scale:
            g = g * sw.grad_scale[n].reshape(1, 1, -1, 1, 1)
        # ReLU: a pre-activation within fp32 rounding of zero may flip between the fp32 device forward and the fp64 host
        # forward (see laplace_amd/nets.py); compare where it cannot (tanh) tightly, ReLU on the aggregate
        if act == "tanh":
            assert rel(g, want[n]) < 1e-5, n
        else:
            assert (g.double().cpu() - want[n]).norm() / want[n].norm() < 1e-4, n


def test_c4_split_sweep_factors_equal_the_nchw_sweep():
    """ResNet-18 (config c4), one minibatch through the fused accumulator: NHWC split-fp16 sweep (own convolution, VJP and
    Gram kernels) vs the NCHW sweep on the library's backward-data."""
    from laplace_amd import HipGGN
    from laplace_amd.nets import ResNet18

    torch.manual_seed(711)
    model = ResNet18(10, act=torch.tanh).to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    X = torch.randn(32, 3, 32, 32, generator=g).to(DEV)
    y = torch.randint(10, (32,), generator=g).to(DEV)
    res = []
    for split in (True, False):
        b = HipGGN(model, "classification")
        b.use_split_sweep = split
        acc = b.kron_accumulator(50_000)
        acc.add_batch(X, y)
        res.append(acc.finalize())
        sweep = b._tape().sweep
        assert (getattr(sweep, "split_ok", False)) == split
    (l1, k1), (l2, k2) = res
    assert rel(l1, l2) < 1e-6
    for F1, F2 in zip(k1.kfacs, k2.kfacs):
        for a, b_ in zip(F1, F2):
            assert rel(a, b_) < 1e-5
