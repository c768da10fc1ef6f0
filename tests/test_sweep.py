"""Seed-batched reverse sweep (laplace_amd/sweep.py) against stock autograd (laplace_amd/capture.py).

The sweep replaces `C-1` reverse passes by one pass of batch `S*B`; what it hands the kernels must be
what autograd hands them: same layer inputs, same output gradients (fp64 here, so the comparison is
exact up to summation order).  Unsupported graphs must fall back to the tape, not fail.
"""
import pytest
import torch
from torch import nn

from laplace_amd.capture import Tape
from laplace_amd.nets import ResNet18, lenet5
from laplace_amd.sweep import SeedBatchedSweep, SweepUnsupported
from oracle.fixtures import FIXTURES, build_model, input_shape


@pytest.fixture(autouse=True)
def _one_sweep_per_minibatch(monkeypatch):
    """these tests look at the accumulator's state minibatch by minibatch (lanes, pixel-pair groups, deferred scales) on
    models small enough for `KronAccumulator.coalesce` to stack their minibatches (tests/test_coalesce.py): off here"""
    from laplace_amd.backend import KronAccumulator

    monkeypatch.setattr(KronAccumulator, "default_coalesce", False)


class Residual(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 6, 3, padding=1)
        self.bn = nn.BatchNorm2d(6)
        self.c2 = nn.Conv2d(6, 6, 3, padding=1, bias=False)
        self.short = nn.Sequential(nn.Conv2d(3, 6, 1, bias=False), nn.BatchNorm2d(6))
        self.pool = nn.AvgPool2d(2)
        self.mp = nn.MaxPool2d(2)
        self.fc = nn.Linear(6 * 2 * 2, 4)

    def forward(self, x):
        h = torch.relu(self.bn(self.c1(x)))
        h = torch.sigmoid(self.c2(h))
        h = h + self.short(x)
        h = self.mp(self.pool(h.relu()))
        return self.fc(h.flatten(1))


class TorchvisionStyleBlock(nn.Module):
    """in-place ReLU module reused at two call sites + `out += identity`, as torchvision's BasicBlock"""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 8, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(8)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(8, 8, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(8)
        self.conv3 = nn.Conv2d(8, 8, 3, padding=1, bias=False)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(8, 5)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        identity = x
        out = self.relu(self.bn2(self.conv2(x)))
        out = self.conv3(out)
        out += identity
        out = self.relu(out)
        return self.fc(torch.flatten(self.pool(out), 1))


class SmoothActs(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 6, 3, padding=1)
        self.bn = nn.BatchNorm2d(6)
        self.act1 = nn.GELU()
        self.c2 = nn.Conv2d(6, 6, 3, stride=2, padding=1)
        self.act2 = nn.LeakyReLU(0.1)
        self.fc1 = nn.Linear(6 * 4 * 4, 12)
        self.act3 = nn.SiLU()
        self.fc2 = nn.Linear(12, 4)

    def forward(self, x):
        h = self.act1(self.bn(self.c1(x)))
        h = self.act2(self.c2(h))
        h = torch.nn.functional.elu(self.fc1(h.flatten(1)), alpha=0.7)
        return self.fc2(torch.nn.functional.softplus(self.act3(h), beta=2.0))


class FunctionalStyle(nn.Module):
    """the idioms of hand-written CNNs: functional pooling, `x.view(x.size(0), -1)`, spatial `mean`"""

    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 6, 3, padding=1)
        self.c2 = nn.Conv2d(6, 8, 3, padding=1)
        self.c3 = nn.Conv2d(8, 8, 3, padding=1)
        self.fc1 = nn.Linear(8 * 2 * 2, 10)
        self.fc2 = nn.Linear(10, 4)
        self.fc3 = nn.Linear(8, 4)

    def forward(self, x):
        h = torch.nn.functional.max_pool2d(torch.relu(self.c1(x)), 2)
        h = torch.nn.functional.avg_pool2d(torch.relu(self.c2(h)), kernel_size=2)
        h = torch.tanh(self.c3(h))
        a = torch.relu(self.fc1(h.view(h.size(0), -1)))
        b = h.mean((2, 3)) + torch.mean(h, dim=[-1, -2]) + torch.nn.functional.adaptive_avg_pool2d(h, 1).flatten(1)
        return self.fc2(a) + self.fc3(b)


def _models():
    yield "resnet18", ResNet18(), (3, 16, 16), 10
    yield "torchvision_block", TorchvisionStyleBlock(), (3, 8, 8), 5
    yield "smooth_acts", SmoothActs(), (3, 8, 8), 4
    yield "functional_style", FunctionalStyle(), (3, 8, 8), 4
    yield "lenet5", lenet5(), (3, 32, 32), 10
    yield "residual", Residual(), (3, 8, 8), 4
    for n in FIXTURES:
        m = build_model(n)
        with torch.no_grad():
            c = m(torch.zeros(1, *input_shape(n), dtype=next(m.parameters()).dtype)).shape[-1]
        yield n, m, input_shape(n), c


@pytest.mark.parametrize("name,model,shape,C", list(_models()), ids=lambda v: v if isinstance(v, str) else "")
def test_sweep_matches_autograd(name, model, shape, C):
    torch.manual_seed(3)
    model = model.double().eval()
    for m in model.modules():  # non-trivial BatchNorm statistics
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.normal_()
    B, S = 5, 4
    x = torch.randn(B, *shape, dtype=torch.float64)
    params = [p for p in model.parameters() if p.requires_grad]
    tape = Tape(model, params)
    sweep = SeedBatchedSweep(model, {t.name: t.module for t in tape.taps})
    f_ref = tape.forward(x)
    f = sweep.forward(x)
    assert torch.equal(f, f_ref.detach())
    seeds = torch.randn(S, B, C, dtype=torch.float64)
    want = tape.output_grads(f_ref, seeds)
    got = sweep.backward(seeds)
    for t, w in zip(tape.taps, want):
        assert torch.equal(sweep.taps[t.name]["a"], t.a), t.name
        g = got[t.name]
        assert g.shape == w.shape, t.name
        assert (g - w).abs().max() <= 1e-12 * (1 + w.abs().max()), t.name


class Gated(nn.Module):
    def __init__(self):
        super().__init__()
        self.a, self.b = nn.Linear(3, 5), nn.Linear(5, 2)

    def forward(self, x):
        return self.b(self.a(x) * 2.0)  # operator.mul has no rule here


def test_unsupported_graphs_are_refused():
    m = Gated()
    with pytest.raises(SweepUnsupported):
        SeedBatchedSweep(m, {"a": m.a, "b": m.b})
    sw = SeedBatchedSweep(ResNet18().train(), {})
    with pytest.raises(SweepUnsupported):
        sw.forward(torch.randn(2, 3, 8, 8))  # BatchNorm in training mode


def test_backend_falls_back_to_the_tape():
    """An untraceable / unsupported model still works (autograd tape) and gives the same factors."""
    from laplace_amd import _lib
    from laplace_amd.backend import HipGGN
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        torch.manual_seed(0)
        m = Gated()
        X, y = torch.randn(6, 3), torch.randint(0, 2, (6,))
        b = HipGGN(m, "classification")
        loss, H = b.kron(X, y, N=6)
        assert b._tape().sweep is False and "mul" in b._tape().sweep_reason
        # same model written with a supported graph: fold the factor 2 into the first layer
        m2 = nn.Sequential(nn.Linear(3, 5), nn.Linear(5, 2))
        m2[0].weight.data, m2[0].bias.data = 2 * m.a.weight.data, 2 * m.a.bias.data
        m2[1].weight.data, m2[1].bias.data = m.b.weight.data, m.b.bias.data
        b2 = HipGGN(m2, "classification")
        loss2, H2 = b2.kron(X, y, N=6)
        assert b2._tape().sweep not in (None, False)
        assert torch.allclose(loss, loss2, rtol=1e-5)
        # G factors coincide layer by layer scaled by 4 / 1 (first layer's output gradient is halved)
        # kfacs = [[G_a, A_a], [G_a], [G_b, A_b], [G_b]]
        assert torch.allclose(H.kfacs[2][0], H2.kfacs[2][0], rtol=1e-4, atol=1e-7)
        assert torch.allclose(H.kfacs[2][1], H2.kfacs[2][1], rtol=1e-4, atol=1e-7)
        assert torch.allclose(H.kfacs[0][0], 4 * H2.kfacs[0][0], rtol=1e-4, atol=1e-7)
    finally:
        _lib.set_kernels_for_testing(prev)


@pytest.mark.parametrize("use_sweep", [True, False])
def test_backend_sweep_and_tape_agree(use_sweep):
    from laplace_amd import _lib
    from laplace_amd.backend import HipGGN
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        torch.manual_seed(1)
        model = Residual().eval()
        for m in model.modules():  # the HIP Jacobian kernels cover nn.Linear / nn.Conv2d parameters
            if isinstance(m, nn.BatchNorm2d):
                m.weight.requires_grad_(False), m.bias.requires_grad_(False)
        X, y = torch.randn(6, 3, 8, 8), torch.randint(0, 4, (6,))
        ref = HipGGN(model, "classification")
        ref.use_sweep = False
        loss0, H0 = ref.kron(X, y, N=12)
        b = HipGGN(model, "classification")
        b.use_sweep = use_sweep
        loss1, H1 = b.kron(X, y, N=12)
        assert (b._tape().sweep not in (None, False)) == use_sweep if use_sweep else True
        assert torch.allclose(loss0, loss1)
        for F0, F1 in zip(H0.kfacs, H1.kfacs):
            for a, c in zip(F0, F1):
                assert torch.allclose(a, c, rtol=1e-4, atol=1e-7)
        if use_sweep:  # the BatchNorm of the shortcut branch is deferred: unscaled G sums + one scaling at finalize
            acc = b.kron_accumulator(12)
            acc.add_batch(X, y)
            assert len(acc._gscale) == 1
            _, Hacc = acc.finalize()
            for F0, F1 in zip(H0.kfacs, Hacc.kfacs):
                for a, c in zip(F0, F1):
                    assert torch.allclose(a, c, rtol=1e-4, atol=1e-7)
        _, d0 = ref.diag(X, y, N=12)
        _, d1 = b.diag(X, y, N=12)
        assert torch.allclose(d0, d1, rtol=1e-4, atol=1e-7)
        J0, f0 = ref.jacobians(X)
        J1, f1 = b.jacobians(X)
        assert torch.allclose(J0, J1, rtol=1e-4, atol=1e-6) and torch.allclose(f0, f1)
    finally:
        _lib.set_kernels_for_testing(prev)


def test_inplace_model_equals_its_out_of_place_twin():
    """torchvision-style `out += identity; relu_(out)`: the gradient w.r.t. a tapped output must be the one of the
    tensor as the module produced it, not of the tensor after the in-place updates (both extraction paths)."""
    torch.manual_seed(4)
    m_in = TorchvisionStyleBlock().double().eval()

    class Twin(TorchvisionStyleBlock):
        def forward(self, x):
            x = torch.relu(self.bn1(self.conv1(x)))
            out = torch.relu(self.bn2(self.conv2(x)))
            out = self.conv3(out) + x
            return self.fc(torch.flatten(self.pool(torch.relu(out)), 1))

    m_out = Twin().double().eval()
    m_out.load_state_dict(m_in.state_dict())
    x = torch.randn(3, 3, 8, 8, dtype=torch.float64)
    seeds = torch.randn(2, 3, 5, dtype=torch.float64)
    results = []
    for m in (m_in, m_out):
        params = [p for p in m.parameters() if p.requires_grad]
        tape = Tape(m, params)
        f = tape.forward(x.clone())
        results.append([g for g in tape.output_grads(f, seeds)])
        sweep = SeedBatchedSweep(m, {t.name: t.module for t in tape.taps})
        sweep.forward(x.clone())
        got = sweep.backward(seeds)
        results.append([got[t.name] for t in tape.taps])
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert (a - b).abs().max() <= 1e-12 * (1 + a.abs().max())


@pytest.mark.parametrize("pix_group", [1, 2, 3])
def test_fused_accumulator_pixel_pair_paths_equal_literal_loop(pix_group):
    """KronAccumulator keeps 3x3/s1/p1 conv A factors in pixel-pair form over the fit (banded blocks / dense small
    maps) and assembles once: same factors as summing per-minibatch kron() results -- whatever the number of
    minibatches stacked per pixel-pair launch, including a ragged batch in the middle of a group."""
    from laplace_amd import _lib
    from laplace_amd.backend import HipGGN
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        torch.manual_seed(2)
        model = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(),
                              nn.Conv2d(8, 4, 3, padding=1), nn.Flatten(), nn.Linear(4 * 6 * 6, 3)).eval()
        b = HipGGN(model, "classification")
        acc = b.kron_accumulator(40)
        acc.pix_group = pix_group
        H = None
        for seed, bs in ((1, 4), (2, 4), (3, 3), (4, 4), (5, 4)):
            g = torch.Generator().manual_seed(seed)
            X, y = torch.randn(bs, 3, 6, 6, generator=g), torch.randint(0, 3, (bs,), generator=g)
            acc.add_batch(X, y)
            _, Hb = b.kron(X, y, N=40)
            H = Hb if H is None else H + Hb
        kinds = sorted(geo[0] for geo, _ in acc._pix.values())
        assert kinds == ["pair", "pair"], kinds  # the two 8-channel convs; the 3-channel stem is not eligible
        # a minibatch of a different spatial size in the same fit (fully convolutional body): accumulators are
        # folded and restarted for the new geometry
        model2 = nn.Sequential(*list(model.children())[:5], nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(4, 3)).eval()
        b2 = HipGGN(model2, "classification")
        acc2, H2 = b2.kron_accumulator(40), None
        acc2.pix_group = pix_group
        for seed, hw in ((1, 6), (2, 5), (3, 6)):
            g = torch.Generator().manual_seed(seed)
            X, y = torch.randn(4, 3, hw, hw, generator=g), torch.randint(0, 3, (4,), generator=g)
            acc2.add_batch(X, y)
            _, Hb = b2.kron(X, y, N=40)
            H2 = Hb if H2 is None else H2 + Hb
        _, Hf2 = acc2.finalize()
        for F_, G_ in zip(Hf2.kfacs, H2.kfacs):
            for a, c in zip(F_, G_):
                assert torch.allclose(a, c, rtol=1e-4, atol=1e-7)
        _, Hf = acc.finalize()
        for F_, G_ in zip(Hf.kfacs, H.kfacs):
            for a, c in zip(F_, G_):
                assert torch.allclose(a, c, rtol=1e-4, atol=1e-7)
    finally:
        _lib.set_kernels_for_testing(prev)


def test_seed_chunking_for_many_output_models():
    """S*B above `sweep_max_rows`: the sweep runs in seed chunks; KFAC factors, diag and Jacobians are unchanged."""
    from laplace_amd import _lib
    from laplace_amd.backend import HipGGN
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        torch.manual_seed(6)
        model = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Flatten(),
                              nn.Linear(8 * 4 * 4, 12)).eval()
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.requires_grad_(False), m.bias.requires_grad_(False)
        X, y = torch.randn(5, 3, 4, 4), torch.randint(0, 12, (5,))
        ref = HipGGN(model, "classification")
        loss0, H0 = ref.kron(X, y, N=20)
        _, d0 = ref.diag(X, y)
        J0, _ = ref.jacobians(X)
        b = HipGGN(model, "classification")
        b.sweep_max_rows = 15  # 11 seeds x 5 samples -> chunks of 3 seeds
        loss1, H1 = b.kron(X, y, N=20)
        assert torch.allclose(loss0, loss1)
        for F0, F1 in zip(H0.kfacs, H1.kfacs):
            for a, c in zip(F0, F1):
                assert torch.allclose(a, c, rtol=1e-4, atol=1e-7)
        assert torch.allclose(b.diag(X, y)[1], d0, rtol=1e-4, atol=1e-7)
        assert torch.allclose(b.jacobians(X)[0], J0, rtol=1e-4, atol=1e-6)
    finally:
        _lib.set_kernels_for_testing(prev)


@pytest.mark.parametrize("pool", [nn.AvgPool2d(3, 2, 1), nn.AvgPool2d(2, 2, 0, ceil_mode=True), nn.AvgPool2d(3, 1),
                                  nn.AdaptiveAvgPool2d(2), nn.AvgPool2d(3, 2, 1, count_include_pad=False)])
def test_general_avgpool_geometries_go_through_the_sweep(pool):
    """Padded / overlapping / ragged average pooling (and adaptive pooling to more than one cell) used to raise
    SweepUnsupported from INSIDE backward(), i.e. after the forward had succeeded and with no fallback left."""
    torch.manual_seed(0)
    model = nn.Sequential(nn.Conv2d(2, 4, 3, padding=1), nn.ReLU(), pool, nn.Flatten(), nn.LazyLinear(3)).double().eval()
    x = torch.randn(5, 2, 7, 7, dtype=torch.float64)
    model(x)
    taps = {n: m for n, m in model.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))}
    sw = SeedBatchedSweep(model, taps)
    f = sw.forward(x)
    seeds = torch.randn(3, 5, 3, dtype=torch.float64)
    got = sw.backward(seeds)
    # reference: one autograd pass per seed
    outs = {}
    hs = [m.register_forward_hook(lambda m_, i, o, n=n: outs.__setitem__(n, o)) for n, m in taps.items()]
    f2 = model(x)
    for h in hs:
        h.remove()
    assert torch.allclose(f, f2)
    for s in range(3):
        grads = torch.autograd.grad(f2, [outs[n] for n in taps], grad_outputs=seeds[s], retain_graph=True)
        for n, g in zip(taps, grads):
            assert torch.allclose(got[n][s], g, atol=1e-12), n


@pytest.mark.parametrize("lanes", [2, 3])
def test_lanes_of_a_fit_sum_to_the_single_chain(lanes):
    """KronAccumulator.lanes: consecutive minibatches go alternately to sub-accumulators (on the device: own streams) that are
    folded when the fit is read.  Host logic on the kernel emulation: same loss / factors as one chain, for a ragged last
    minibatch, deferred BatchNorm scales, pixel-pair accumulators in every lane, `tensors()` (the all-reduce's view) and
    an accumulator that never saw a minibatch."""
    from laplace_amd import HipGGN, _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        torch.manual_seed(5)
        model = nn.Sequential(
            nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.Tanh(),
            nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 3)).eval()
        with torch.no_grad():
            model[1].weight.uniform_(0.5, 1.5), model[1].running_var.uniform_(0.5, 2.0)
        for p_ in model[1].parameters():
            p_.requires_grad_(False)
        g = torch.Generator().manual_seed(9)
        batches = [(torch.randn(b, 3, 6, 6, generator=g), torch.randint(3, (b,), generator=g)) for b in (4, 4, 4, 4, 3)]
        b = HipGGN(model, "classification")

        def fit(n_lanes, read):
            acc = b.kron_accumulator(19)
            acc.lanes, acc._lanes_anywhere, acc.pix_group = n_lanes, True, 2
            for X, y in batches:
                acc.add_batch(X, y)
            if n_lanes > 1:
                assert acc._lane_accs is not None and all(s.factors is not None for s in acc._lane_accs)
            if read == "tensors":
                ts = [t.clone() for t in acc.tensors()]
                assert acc._lane_accs is None
            loss, H = acc.finalize()
            return loss, [t for F in H.kfacs for t in F]

        loss1, want = fit(1, "finalize")
        for read in ("finalize", "tensors"):
            loss, got = fit(lanes, read)
            assert torch.allclose(loss, loss1, rtol=1e-6)
            for a, w in zip(got, want):
                assert (a - w).abs().max() <= 1e-6 * w.abs().max() + 1e-12
        empty = b.kron_accumulator(19)
        empty.lanes, empty._lanes_anywhere = lanes, True
        empty.ensure_allocated(torch.device("cpu"))
        assert all(float(t.abs().max()) == 0.0 for t in empty.tensors())
    finally:
        _lib.set_kernels_for_testing(prev)
