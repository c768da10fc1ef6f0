"""``backend.kron`` hands its minibatch over in the accumulator's raw form (laplace_amd/kron.py: HipKron._pending): the
algebra of the reference's literal fit loop — ``H = Kron.init_from_model(..)``, ``H += backend.kron(X, y, N)[1]`` per
minibatch (laplace/baselaplace.py:969-985) — must give what the eager, public-layout objects give, and must not change
its operands.  Every test runs twice: on the CPU emulation of the kernels (host logic, `not gpu` tier) and — marked
``gpu`` — on the real library, where the raw forms are device buffers merged with a multi-tensor add."""
import pytest
import torch

from laplace_amd import _lib
from tests.conftest import golden_model, load_golden
from tests.emulated_kernels import EmulatedKernels

DEVICES = [pytest.param("cpu", id="emulated"), pytest.param("cuda", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=DEVICES)
def dev(request):
    if request.param == "cuda":
        assert isinstance(_lib.get_kernels(), _lib.HipKernels), "the real library, not the emulation"
        yield "cuda"
        return
    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    yield "cpu"
    _lib.set_kernels_for_testing(prev)


def _flat(K):
    return [t.clone() for F in K.kfacs for t in F]


def _close(a, b, tol=1e-6):
    return all((x - y).abs().max() <= tol * (y.abs().max() + 1e-30) for x, y in zip(a, b))


@pytest.mark.parametrize("name", ["conv", "bnres", "mlp"])
def test_lazy_minibatch_krons_equal_the_eager_ones(name, dev):
    from laplace_amd import HipGGN, HipKron

    g = load_golden(name, "classification")
    model, X, y = golden_model(name, g, dtype=torch.float32, device=dev)
    N = X.shape[0]
    eager = HipGGN(model, "classification")
    eager.lazy_kron = False
    lazy = HipGGN(model, "classification")
    assert lazy.lazy_kron
    parts = [slice(0, 4), slice(4, 7), slice(7, 10)]
    want = [_flat(eager.kron(X[s], y[s], N)[1]) for s in parts]
    total = [sum(w[i] for w in want) for i in range(len(want[0]))]

    ks = [lazy.kron(X[s], y[s], N) for s in parts]
    assert all(k._pending is not None for _, k in ks), "kron() should hand over the raw form"
    for (loss, _), s in zip(ks, parts):
        assert torch.allclose(loss, eager.kron(X[s], y[s], N)[0], rtol=1e-6)
    # out-of-place sum: operands untouched, result right
    s01 = ks[0][1] + ks[1][1]
    assert ks[0][1]._pending is not None and ks[1][1]._pending is not None
    assert _close(_flat(s01), [a + b for a, b in zip(want[0], want[1])])
    assert _close(_flat(ks[0][1]), want[0]) and _close(_flat(ks[1][1]), want[1])
    # the literal loop, starting from zeros in the public layout
    params = [p for p in model.parameters() if p.requires_grad]
    H = HipKron.init_from_model(params, X.device, torch.float32)
    for s in parts:
        H += lazy.kron(X[s], y[s], N)[1]
    assert H._pending is not None, "nothing should have been brought into the public layout yet"
    assert _close(_flat(H), total)
    assert H._pending is None
    # mixing: a materialised Kron plus a raw one, both orders; scalar multiplication reads the public layout
    k_raw = lazy.kron(X[parts[2]], y[parts[2]], N)[1]
    assert _close(_flat(s01 + k_raw), total) and _close(_flat(k_raw + s01), total)
    doubled = k_raw * 2.0
    for F2, F1 in zip(doubled.kfacs, k_raw.kfacs):
        for t2, t1 in zip(F2, F1):
            assert torch.allclose(t2, 2.0 ** (1 / len(F1)) * t1)
    assert len(k_raw) == len(k_raw.kfacs)


def test_lazy_kron_survives_deepcopy_and_pickle(dev):
    import copy
    import pickle

    from laplace_amd import HipGGN

    g = load_golden("conv", "classification")
    model, X, y = golden_model("conv", g, dtype=torch.float32, device=dev)
    k = HipGGN(model, "classification").kron(X[:5], y[:5], X.shape[0])[1]
    assert k._pending is not None
    c = copy.deepcopy(k)
    p = pickle.loads(pickle.dumps(k))
    for other in (c, p):
        assert other._pending is None and _close(_flat(other), _flat(k), 0.0)


@pytest.mark.parametrize("group", [1, 3, 8])
def test_literal_loop_leaves_the_pixel_pair_products_to_the_running_sum(dev, group, monkeypatch):
    """3x3 convs whose A factor has a banded pixel-pair form: a lazily handed-over minibatch keeps its NHWC input
    (KronAccumulator.defer_pix), the running sum stacks ``LK_PIX_GROUP`` of them per launch as the fused accumulator
    does.  Same factors as the eager per-minibatch kernels, whatever is read when, and no operand is changed."""
    from laplace_amd import HipGGN, HipKron

    from laplace_amd.backend import KronAccumulator

    monkeypatch.setattr(KronAccumulator, "pix_group", group)
    C = 64 if dev == "cuda" else 8
    torch.manual_seed(3)
    model = torch.nn.Sequential(
        torch.nn.Conv2d(3, C, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(C, C, 3, padding=1, bias=False), torch.nn.Tanh(),
        torch.nn.Conv2d(C, C, 3, padding=1), torch.nn.Tanh(), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(),
        torch.nn.Linear(C, 4)).to(dev).eval()
    X, y = torch.randn(20, 3, 6, 6, device=dev), torch.randint(0, 4, (20,), device=dev)
    N = X.shape[0]
    parts = [slice(i, min(i + 3, N)) for i in range(0, N, 3)]  # 7 minibatches, the last one ragged
    eager = HipGGN(model, "classification")
    eager.lazy_kron = False
    want = [_flat(eager.kron(X[s], y[s], N)[1]) for s in parts]
    total = [sum(w[i] for w in want) for i in range(len(want[0]))]
    lazy = HipGGN(model, "classification")
    first = lazy.kron(X[parts[0]], y[parts[0]], N)[1]
    assert first._pending.defer_pix and first._pending._pix_inputs, "the wide 3x3 convs should have been deferred"

    params = [p for p in model.parameters() if p.requires_grad]
    H = HipKron.init_from_model(params, X.device, torch.float32)
    H += first
    for i, s in enumerate(parts[1:], 1):
        H += lazy.kron(X[s], y[s], N)[1]
        if i == 4:  # out of place on a running sum that holds pixel-pair blocks: a copy, the sum goes on
            part = H + HipKron.init_from_model(params, X.device, torch.float32)
            assert _close(_flat(part), [sum(w[j] for w in want[:5]) for j in range(len(total))], 1e-5)
            assert H._pending is not None
    assert bool(H._pending._pix) == (group <= len(parts)), "blocks are allocated once a whole group has come together"
    assert _close(_flat(H), total, 1e-5)
    assert _close(_flat(first), want[0], 1e-5), "an absorbed minibatch still stands for itself"
    # the fused accumulator of the same fit
    acc = lazy.kron_accumulator(N)
    for s in parts:
        acc.add_batch(X[s], y[s])
    assert _close(_flat(acc.finalize()[1]), total, 1e-5)
    # switch off: every minibatch computes its own A factors
    lazy.lazy_pixpair = False
    k = lazy.kron(X[parts[1]], y[parts[1]], N)[1]
    assert not k._pending._pix_inputs and _close(_flat(k), want[1], 1e-5)
