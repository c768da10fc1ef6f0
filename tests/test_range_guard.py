"""Minibatches that mix samples of very different magnitudes (laplace_amd/backend.py: range_groups, range_guard): the
split-fp16 sweep carries one scale per tensor, so per-sample results sweep such a minibatch in magnitude groups, and a
fit refuses it at the end (or, with ``range_guard = "group"``, sweeps it in groups too — the curvature is a sum over
samples, laplace/baselaplace.py:984-985, so that is exact).  Host logic on the CPU emulation; the numerics on the device:
tests/test_gpu_dynamic_range.py."""
import pytest
import torch

from laplace_amd import _lib
from tests.conftest import golden_model, load_golden
from tests.emulated_kernels import EmulatedKernels


@pytest.fixture(autouse=True)
def _emulated():
    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    yield
    _lib.set_kernels_for_testing(prev)


def _setup():
    """a model the split-fp16 NHWC sweep serves (the guard concerns that path only: one scale per tensor)"""
    from tests.test_sweep_nhwc import _model

    model = _model(torch.relu)
    torch.manual_seed(11)
    return model, torch.randn(10, 3, 8, 8), torch.randint(5, (10,))


def test_range_groups_partition_by_magnitude():
    from laplace_amd.backend import range_groups

    x = torch.randn(6, 3, 4, 4)
    assert range_groups(x) is None
    x = x * torch.tensor([1.0, 1e-6, 1e6, 2.0, 0.0, 3e-6]).reshape(6, 1, 1, 1)
    groups = range_groups(x)
    assert groups is not None
    flat = sorted(int(i) for idx in groups for i in idx)
    assert flat == list(range(6))                      # every sample in exactly one group (the all-zero one too)
    for idx in groups:
        a = x[idx].abs().flatten(1).amax(1)
        a = a[a > 0]
        assert a.numel() == 0 or float(a.max() / a.min()) <= 2.0 ** 17


def test_a_fit_refuses_a_minibatch_outside_the_range_and_group_mode_is_exact():
    from laplace_amd import HipGGN

    model, X, y = _setup()
    Xw = X.clone()
    Xw[0] *= 1e-4
    Xw[1] *= 1e4
    b = HipGGN(model, "classification")
    assert b.range_guard == "check"
    acc = b.kron_accumulator(10)
    acc.add_batch(X, y)            # fine
    assert b._split_sweep_state() is True
    acc.add_batch(Xw, y)           # spread 1e8 > 2^16: recorded on the device, no synchronisation here
    with pytest.raises(RuntimeError, match="range_guard"):
        acc.finalize()
    # group mode: the wide minibatch is swept in magnitude groups == the groups handed over one by one
    b.range_guard = "group"
    acc = b.kron_accumulator(10)
    acc.add_batch(Xw, y)
    loss, H = acc.finalize()
    b.range_guard = "off"
    ref = b.kron_accumulator(10)
    from laplace_amd.backend import range_groups

    groups = range_groups(Xw)
    assert groups is not None and len(groups) >= 2
    for idx in groups:
        ref.add_batch(Xw[idx].contiguous(), y[idx])
    loss_r, H_r = ref.finalize()
    assert torch.allclose(loss, loss_r, rtol=1e-6)
    for F1, F2 in zip(H.kfacs, H_r.kfacs):
        for a, b_ in zip(F1, F2):
            assert torch.allclose(a, b_, rtol=1e-5, atol=1e-7 * float(b_.abs().max()))


def test_per_sample_results_are_swept_in_magnitude_groups():
    from laplace_amd import HipGGN

    model, X, y = _setup()
    Xw = X.clone()
    Xw[0] *= 1e-5
    Xw[3] *= 1e5
    b = HipGGN(model, "classification")
    Js, f = b.jacobians(Xw)
    b.range_guard = "off"
    Js0, f0 = b.jacobians(Xw)      # (the emulation is exact either way: this checks the scatter back into batch order)
    assert torch.allclose(f, f0, rtol=1e-5, atol=1e-6) and Js.shape == Js0.shape
    for n in range(X.shape[0]):
        assert torch.allclose(Js[n], Js0[n], rtol=1e-4, atol=1e-6 * float(Js0[n].abs().max()))


def test_the_guard_leaves_models_outside_the_split_sweep_alone():
    """ADVICE (round 3): a 1-D regression set with one sample near zero raised at the end of the fit although no split-fp16
    tensor was involved (an MLP runs through the fp32 kernels, per element like the reference)."""
    from laplace_amd import HipGGN

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(1, 50), torch.nn.Tanh(), torch.nn.Linear(50, 1))
    X = torch.rand(64, 1) * 8
    X[3] = 5e-5
    y = torch.randn(64, 1)
    b = HipGGN(model, "regression")
    acc = b.kron_accumulator(64)
    acc.add_batch(X, y)
    loss, H = acc.finalize()                       # no RuntimeError
    assert b._split_sweep_state() is False
    loss2, H2 = b.kron(X, y, 64)                   # the literal loop's entry point
    _ = H2.kfacs
    from laplace_amd.backend import range_groups
    assert range_groups(torch.tensor([[0.0], [1.0], [2.0]])) is None   # an all-zero sample forces no extra group
