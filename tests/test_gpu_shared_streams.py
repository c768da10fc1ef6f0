"""The fit's HIP streams are per process, not per backend object (laplace_amd/backend.py: `_process_streams`).  PyTorch's
caching allocator keeps freed blocks per stream, so backends with lanes / side / flush streams of their own each reserved
another ~100 GB for the same ResNet-18 fit: a loop that builds a new `Laplace` object per epoch (the reference's marglik
training does, laplace/marglik_training.py:293-301) exhausted the device after two of them.  Here: three backends fit the same
minibatches one after the other — the same factors, and the third reserves (almost) nothing the second did not.  -m gpu only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_three_backends_share_the_streams_and_their_memory_pools():
    from laplace_amd import HipGGN
    from laplace_amd.backend import _PROCESS_STREAMS
    from laplace_amd.nets import ResNet18

    torch.manual_seed(3)
    model = ResNet18(10).cuda().eval()
    data = [(torch.randn(128, 3, 32, 32, device="cuda"), torch.randint(0, 10, (128,), device="cuda")) for _ in range(4)]
    reserved, streams, first = [], [], None
    for rep in range(3):
        b = HipGGN(model, "classification")
        acc = b.kron_accumulator(512)
        for X, y in data:
            acc.add_batch(X, y)
        loss, H = acc.finalize()
        torch.cuda.synchronize()
        reserved.append(torch.cuda.memory_reserved())
        streams.append(len(_PROCESS_STREAMS))
        if first is None:
            first = (loss.clone(), [[t.clone() for t in F] for F in H.kfacs])
        else:
            assert torch.allclose(loss, first[0], rtol=1e-6)
            for Fa, Fb in zip(H.kfacs, first[1]):
                for a, b_ in zip(Fa, Fb):
                    assert float((a - b_).abs().max()) <= 1e-6 * float(b_.abs().max())
        del acc, H, b
    assert streams[0] == streams[1] == streams[2] > 0
    gib = 2.0 ** 30
    print(f"reserved after each backend's fit: {[round(r / gib, 1) for r in reserved]} GiB")
    assert reserved[2] - reserved[1] < 8 * gib  # (own streams per backend: + 60 - 100 GiB each)


def test_a_bounded_lead_of_the_host_bounds_the_reserved_memory(monkeypatch):
    """`KronAccumulator.max_ahead`: the host may enqueue that many minibatches per lane ahead of the device and then waits
    for the oldest.  Same factors (the waits order nothing on the device), a fraction of the reserved memory of an
    unbounded lead (400 minibatches: 42 GiB at 4, 160 GiB unbounded, same 6.7 ms per step: profiles/r05_box_session_age.log)."""
    import gc

    from laplace_amd import HipGGN
    from laplace_amd.backend import KronAccumulator
    from laplace_amd.nets import ResNet18

    torch.manual_seed(5)
    model = ResNet18(10).cuda().eval()
    X, y = torch.randn(128, 3, 32, 32, device="cuda"), torch.randint(0, 10, (128,), device="cuda")
    b = HipGGN(model, "classification")
    out = {}
    for ahead in (2, 0):
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        base = torch.cuda.memory_reserved()
        monkeypatch.setattr(KronAccumulator, "max_ahead", ahead)
        acc = b.kron_accumulator(128 * 160)
        for _ in range(160):
            acc.add_batch(X, y)
        loss, H = acc.finalize()
        torch.cuda.synchronize()
        out[ahead] = (torch.cuda.memory_reserved() - base, loss, [[t.clone() for t in F] for F in H.kfacs])
        del acc, H
    gib = 2.0 ** 30
    print(f"reserved by a 160-minibatch fit: lead 2: {out[2][0] / gib:.0f} GiB, unbounded: {out[0][0] / gib:.0f} GiB")
    # an ABSOLUTE bound: what a lead of 2 (the default) reserves is (lead x lanes x one minibatch's buffers), the same on every
    # box (32 - 34 GiB over 300 minibatches, profiles/r06_host_lead.log; 42 - 45 GiB at 4, profiles/r05_box_session_age.log); what the unbounded lead reserves depends on how far
    # the host of that box gets ahead in 160 minibatches (68 - 160 GiB) — a ratio of the two passed or failed with the box
    assert out[2][0] < 40 * gib, f"lead 2 reserved {out[2][0] / gib:.0f} GiB"
    assert torch.allclose(out[2][1], out[0][1], rtol=1e-6)
    for Fa, Fb in zip(out[2][2], out[0][2]):
        for a, b_ in zip(Fa, Fb):
            assert float((a - b_).abs().max()) <= 1e-6 * float(b_.abs().max())
