"""Stacking of consecutive SMALL minibatches in the fused accumulator (KronAccumulator.coalesce): the curvature is a sum
over samples whose terms do not depend on the minibatch they arrive in (curvlinops.py:77-108: G sums over samples, A
carries 1/N with the global N; baselaplace.py:969-985 adds the minibatches up), so a fit may sweep several loader batches at
once — what takes BASELINE config c1 (151-parameter MLP, batch 100) off the host's enqueue rate.  Host logic on the CPU
emulation of the kernels; the result is compared with the un-stacked fit and with the fp64 oracle."""
import pytest
import torch
from torch import nn

from laplace_amd import HipGGN, _lib
from oracle import curvature_oracle as co
from tests.emulated_kernels import EmulatedKernels


@pytest.fixture(autouse=True)
def _emulated():
    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    yield
    _lib.set_kernels_for_testing(prev)


def rel(a, b):
    a, b = a.double(), b.double()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-300)


def _fit(model, lik, batches, N, coalesce):
    b = HipGGN(model, lik)
    sweeps = []
    orig = b._forward
    b._forward = lambda x, *a, **k: (sweeps.append(x.shape[0]), orig(x, *a, **k))[1]
    acc = b.kron_accumulator(N)
    acc.coalesce = coalesce
    for X, y in batches:
        acc.add_batch(X, y)
    loss, kron = acc.finalize()
    return loss, kron, sweeps


@pytest.mark.parametrize("lik", ["regression", "classification"])
def test_stacked_minibatches_give_the_unstacked_fit(lik):
    torch.manual_seed(0)
    C = 1 if lik == "regression" else 3
    model = nn.Sequential(nn.Linear(4, 50), nn.Tanh(), nn.Linear(50, C))
    X = torch.randn(1000, 4)
    y = torch.randn(1000, C) if lik == "regression" else torch.randint(C, (1000,))
    batches = [(X[i:i + 100], y[i:i + 100]) for i in range(0, 1000, 100)]
    l1, k1, s1 = _fit(model, lik, batches, 1000, True)
    l0, k0, s0 = _fit(model, lik, batches, 1000, False)
    assert s0 == [100] * 10 and s1 == [100, 900]  # (the fit's first minibatch is swept alone: errors surface on the first call)
    assert rel(l1, l0) < 1e-6
    for F1, F0 in zip(k1.kfacs, k0.kfacs):
        for a, b in zip(F1, F0):
            assert rel(a, b) < 1e-5
    loss_ref, kf_ref = co.kfac_ggn(model.double(), X.double(), y.double() if lik == "regression" else y, 1000, lik)
    model.float()
    for F1, G in zip(k1.kfacs, kf_ref):
        for a, w in zip(F1, G):
            assert rel(a, w) < 1e-5


def test_ragged_and_changing_minibatches_and_the_target():
    """a ragged last minibatch, a change of the input shape mid-fit (flushes what was stacked), targets by model size"""
    torch.manual_seed(1)
    model = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 4))
    mk = lambda n, h: (torch.randn(n, 3, h, h), torch.randint(4, (n,)))
    batches = [mk(16, 8), mk(16, 8), mk(5, 8), mk(16, 6), mk(16, 6)]
    l1, k1, s1 = _fit(model, "classification", batches, 69, True)
    l0, k0, s0 = _fit(model, "classification", batches, 69, False)
    assert s0 == [16, 16, 5, 16, 16] and s1 == [16, 21, 32]
    assert rel(l1, l0) < 1e-6
    for F1, F0 in zip(k1.kfacs, k0.kfacs):
        for a, b in zip(F1, F0):
            assert rel(a, b) < 1e-5
    acc = HipGGN(model, "classification").kron_accumulator(69)
    assert acc.coalesce_target(batches[0][0]) == 0  # (nothing swept yet)
    acc._dispatched = 1
    assert acc.coalesce_target(batches[0][0]) == 32 * 16 and acc.coalesce_target(torch.randn(5000, 3, 8, 8)) == 0  # (at most 32 loader batches)
    # a few-parameter model on large maps: the measured per-sample activation (times the seeds of its sweep) bounds the stack
    acc._act_numel = 3 * 8 * 256 * 256
    assert acc.coalesce_target(torch.randn(16, 3, 8, 8)) == (1 << 28) // (3 * 8 * 256 * 256)
    assert HipGGN(model, "classification").kron_accumulator(69, coalesce=False).coalesce is False
    from laplace_amd.nets import ResNet18

    big = HipGGN(ResNet18(10), "classification").kron_accumulator(50000)
    assert big.coalesce_target(torch.randn(128, 3, 32, 32)) == 0          # the benched configuration never stacks
    assert big.coalesce_target({"input_ids": torch.zeros(2, 3)}) == 0     # dict-style inputs neither


def test_the_callers_buffers_may_be_refilled_between_minibatches():
    torch.manual_seed(2)
    model = nn.Sequential(nn.Linear(3, 6), nn.Tanh(), nn.Linear(6, 2))
    data = [(torch.randn(10, 3), torch.randn(10, 2)) for _ in range(4)]
    bufx, bufy = torch.empty(10, 3), torch.empty(10, 2)
    b = HipGGN(model, "regression")
    acc = b.kron_accumulator(40)
    for X, y in data:
        bufx.copy_(X), bufy.copy_(y)
        acc.add_batch(bufx, bufy)
    _, k1 = acc.finalize()
    _, k0, _ = _fit(model, "regression", data, 40, False)
    for F1, F0 in zip(k1.kfacs, k0.kfacs):
        for a, c in zip(F1, F0):
            assert rel(a, c) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["c1_mlp", "c2_lenet"])
def test_stacked_minibatches_on_the_device(which):
    """the BASELINE small configs on the real kernels: the stacked fit against the minibatch-by-minibatch fit (same kernels,
    other sweep sizes) and, for c1, against the fp64 oracle"""
    from laplace_amd import _lib
    from laplace_amd.nets import lenet5, mlp_1_50_1

    prev = _lib.set_kernels_for_testing(None)  # (this file's fixture installed the emulation: back to the HIP library)
    try:
        dev = "cuda"
        torch.manual_seed(711)
        if which == "c1_mlp":
            model, lik = mlp_1_50_1().to(dev), "regression"
            X = (8 * torch.rand(1000, 1)).to(dev)
            y = (torch.sin(X) + 0.3 * torch.randn_like(X)).to(dev)
            bs = 100
        else:
            model, lik = lenet5().to(dev), "classification"
            X = torch.randn(2560, 3, 32, 32, device=dev)
            y = torch.randint(0, 10, (2560,), device=dev)
            bs = 256
        batches = [(X[i:i + bs], y[i:i + bs]) for i in range(0, len(X), bs)]
        l1, k1, s1 = _fit(model, lik, batches, len(X), True)
        l0, k0, s0 = _fit(model, lik, batches, len(X), False)
        assert len(s1) < len(s0) and sum(s1) == sum(s0) == len(X)
        assert rel(l1.cpu(), l0.cpu()) < 1e-5
        for F1, F0 in zip(k1.kfacs, k0.kfacs):
            for a, b in zip(F1, F0):
                assert rel(a.cpu(), b.cpu()) < 1e-5
        if which == "c1_mlp":
            import copy

            m64 = copy.deepcopy(model).double().cpu()
            _, kf_ref = co.kfac_ggn(m64, X.double().cpu(), y.double().cpu(), len(X), lik)
            for F1, G in zip(k1.kfacs, kf_ref):
                for a, w in zip(F1, G):
                    assert rel(a.cpu(), w) < 1e-5
    finally:
        _lib.set_kernels_for_testing(prev)
