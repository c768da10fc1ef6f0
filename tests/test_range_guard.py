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
    g = load_golden("bnres", "classification")
    return golden_model("bnres", g, dtype=torch.float32)


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
    small, mid, big = torch.tensor([0]), torch.arange(2, 10), torch.tensor([1])
    for idx in (small, mid, big):
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
