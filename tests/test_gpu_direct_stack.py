"""The forward pass writes the input activation of a 3x3 convolution straight into its slot of the pixel-pair stack
(`KronAccumulator.direct_stack`, `SplitSweep.act_sink`) instead of into a tensor that is then copied there: same bytes in the
same place, so the factors are the SAME bits as with the copies — across group boundaries (the stack is reused: the next
group's first slot is written while the previous group's split may still be queued on the side stream), with one and two
minibatches in flight, and with a ragged last minibatch (-m gpu)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fit(direct, lanes, sizes, pix_group=2):
    from laplace_amd import HipGGN
    from laplace_amd._lib import get_kernels
    from laplace_amd.nets import ResNet18

    torch.manual_seed(3)
    model = ResNet18(10).to(DEV).eval()
    g = torch.Generator().manual_seed(11)
    batches = [(torch.randn(b, 3, 32, 32, generator=g).to(DEV), torch.randint(10, (b,), generator=g).to(DEV)) for b in sizes]
    K = get_kernels()
    calls = {"n": 0}
    orig = K.copy_absmax

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    K.copy_absmax = counting
    try:
        acc = HipGGN(model, "classification").kron_accumulator(1000)
        acc.lanes, acc.pix_group, acc.direct_stack, acc.coalesce = lanes, pix_group, direct, False
        for X, y in batches:
            acc.add_batch(X, y)
        loss, H = acc.finalize()
        torch.cuda.synchronize()
    finally:
        del K.copy_absmax  # (back to the class's method)
    return loss, H, calls["n"]


@pytest.mark.parametrize("lanes,sizes", [(1, (8,) * 7), (2, (8,) * 11), (2, (8,) * 6 + (5,)), (1, (8, 8, 8, 3))])
def test_activations_written_into_the_stack_give_the_same_bits(lanes, sizes):
    loss_c, H_c, copies_c = _fit(False, lanes, sizes)
    loss_d, H_d, copies_d = _fit(True, lanes, sizes)
    assert copies_d < copies_c  # (the mechanism is engaged: every lane's minibatches after its first skip their 13 copies)
    assert torch.equal(loss_c, loss_d)
    for Fc, Fd in zip(H_c.kfacs, H_d.kfacs):
        for a, b in zip(Fc, Fd):
            assert torch.equal(a, b)
