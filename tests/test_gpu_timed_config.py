"""Parity of the configuration bench.py TIMES (-m gpu): >= 10 minibatches of 128 through ``KronAccumulator`` with its
DEFAULT settings — pixel-pair groups of 8 drained with buffer reuse (full group + ragged remainder), factor kernels
lagging one minibatch on the side stream, persistent accumulators, batch-128 tile selection — against

* fp64 Gram matrices of the SAME minibatches' activations / output gradients, taken from a stock-autograd tape of the
  model (library convolutions, one reverse pass per seed: nothing of the fused path), i.e. what
  laplace/curvature/curvlinops.py:77-108 + ``self.H += H_batch`` (laplace/baselaplace.py:984-985) accumulate;
* the same minibatches through the accumulator with every scheduling feature off (``LK_PIX_GROUP=1``,
  ``LK_LAG_JOIN=0``, no side stream).

Tolerance: 1e-4, each factor block relative to ITS OWN largest element (tests/test_curv_backends_curvlinops.py:207-305
of the reference compare element-wise at rtol 1e-4 / 1e-5; a block-wise bar is the strictest norm-wise form that the
K4 scalar split of utils/matrix.py:100-118 permits)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
# LK_TEST_DEVICE=cpu: self-check of this file's host logic on the kernel emulation (GPU-less box), tiny minibatches
DEV = os.environ.get("LK_TEST_DEVICE", "cuda")
N = 50_000
BATCH = 128 if DEV != "cpu" else 2


@pytest.fixture(autouse=True, scope="module")
def _kernels():
    if DEV != "cpu":
        yield
        return
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    yield
    _lib.set_kernels_for_testing(prev)


def rel(a, b):
    a, b = a.double(), b.double()
    from tests.parity_log import record_error

    return record_error((a - b).abs().max().item() / (b.abs().max().item() + 1e-300))


def _batch(i):
    g = torch.Generator().manual_seed(1000 + i)
    return torch.randn(BATCH, 3, 32, 32, generator=g).to(DEV), torch.randint(10, (BATCH,), generator=g).to(DEV)


@pytest.fixture(scope="module", params=["relu", "tanh"])
def resnet(request):
    """ReLU is what bench.py times.  Two separately executed forward passes of a ReLU network flip the few
    pre-activations that sit within fp32 rounding of zero, which moves individual gradients by
    O(1) and the factors — sums over 1408 x 10 x L outer products — by ~1e-5; tanh has no such effect and pins the
    kernels themselves."""
    from laplace_amd.nets import ResNet18

    torch.manual_seed(711)
    return ResNet18(10, act=torch.relu if request.param == "relu" else torch.tanh).to(DEV).eval()


class _TapeGrams:
    """per tap [G, A] in fp64 (A in F.unfold order, 1/(N L)) + the summed loss: stock autograd, fp64 products"""

    def __init__(self, model, backend):
        from laplace_amd.capture import Tape

        self.tape = Tape(model, backend.params)
        self.refs = [[None, None] for _ in self.tape.taps]
        self.has_bias = [t.has_bias for t in self.tape.taps]
        self.loss = torch.zeros((), dtype=torch.float64, device=DEV)

    def add(self, X, y):
        from laplace_amd._lib import get_kernels

        K, tape = get_kernels(), self.tape
        f = tape.forward(X)
        self.loss += F.cross_entropy(f.detach().double(), y, reduction="sum")
        S = K.softmax_hess_sqrt(f.detach().contiguous(), y, None)  # symmetric root, C seeds: G is root-invariant
        grads = tape.output_grads(f, S, stack=False)
        for i, (tap, g) in enumerate(zip(tape.taps, grads)):
            a = tap.a.double()
            if tap.kind == "conv2d":
                m = tap.module
                cols = F.unfold(a, m.kernel_size, dilation=m.dilation, padding=m.padding, stride=m.stride)
                L = cols.shape[-1]
                A = torch.einsum("bil,bjl->ij", cols, cols) / (N * L)
                g2 = torch.stack(g).double().reshape(-1, m.out_channels, L)
                G = torch.einsum("bil,bjl->ij", g2, g2)
                del cols, g2
            else:
                A = a.T @ a / N
                g2 = g.double().reshape(-1, tap.module.out_features)
                G = g2.T @ g2
            r = self.refs[i]
            r[0] = G if r[0] is None else r[0] + G
            r[1] = A if r[1] is None else r[1] + A
        del grads, f
        tape.release()


def test_eleven_minibatches_of_128_with_default_scheduling(resnet, monkeypatch):
    from laplace_amd import HipGGN

    n_batches = 11  # one full pixel-pair group of 8 (buffer reused) + a ragged remainder of 3
    batches = [_batch(i) for i in range(n_batches)]
    b = HipGGN(resnet, "classification")

    # (1) the timed configuration: defaults, side stream, lagged join.  Interleaved with unrelated main-stream work
    # (the tape's library kernels of the NEXT comparison) so that a missing stream dependency has something to race
    # and with (3), the fp64 Grams of the same minibatches from a stock-autograd tape
    acc = b.kron_accumulator(N)
    assert acc.pix_group == 8 and acc.lag_join and acc.overlap, "defaults changed: update bench.py's `check` too"
    ref = _TapeGrams(resnet, b)
    for X, y in batches:
        acc.add_batch(X, y)
        ref.add(X, y)
    loss, H = acc.finalize()

    # (2) every scheduling feature off
    from laplace_amd.backend import KronAccumulator

    monkeypatch.setattr(KronAccumulator, "pix_group", 1)
    monkeypatch.setattr(KronAccumulator, "lag_join", False)
    ser = b.kron_accumulator(N, overlap=False)
    assert ser.pix_group == 1 and not ser.lag_join and not ser.overlap
    for X, y in batches:
        ser.add_batch(X, y)
    loss_s, H_s = ser.finalize()
    assert rel(loss, loss_s) < 1e-6
    worst = 0.0
    for blk, (F_, S_) in enumerate(zip(H.kfacs, H_s.kfacs)):
        for j, (a_, s_) in enumerate(zip(F_, S_)):
            r = rel(a_, s_)
            worst = max(worst, r)
            assert r < 1e-5, f"default vs serial scheduling: block {blk} factor {j} (n={a_.shape[0]}) rel {r:.2e}"

    # (3)
    refs, has_bias, loss_ref = ref.refs, ref.has_bias, ref.loss
    if DEV == "cpu" and resnet.act is torch.relu:
        return  # 22 samples: a single ReLU flip between the two executions is 1e-3 of a factor; meaningful at 1408 only
    assert rel(loss, loss_ref) < 1e-5
    blk, worst_ref = 0, 0.0
    for (G_ref, A_ref), hb in zip(refs, has_bias):
        G, A = H.kfacs[blk]
        for name, got, want in (("G", G, G_ref), ("A", A, A_ref)):
            r = rel(got, want)
            worst_ref = max(worst_ref, r)
            assert r < 1e-4, f"tap {blk} {name} (n={got.shape[0]}): rel to the block's own max {r:.2e}"
        blk += 1
        if hb:
            assert rel(H.kfacs[blk][0], G_ref) < 1e-4
            blk += 1
    assert blk == len(H.kfacs) == 22
    print(f"timed configuration, {n_batches} x {BATCH}: default vs serial {worst:.2e}, vs fp64 tape Grams {worst_ref:.2e}")


def test_two_consecutive_fits_reuse_the_side_stream_and_buffers(resnet):
    """bench.py runs warm-up fit, timed fit, instrumented fit on ONE backend object: the second fit must not see
    anything of the first (pixel-pair accumulators, pending groups, deferred BatchNorm scales, lagged events)."""
    from laplace_amd import HipGGN

    b = HipGGN(resnet, "classification")
    batches = [_batch(20 + i) for i in range(3)]

    def fit(bs):
        acc = b.kron_accumulator(N)
        for X, y in bs:
            acc.add_batch(X, y)
        return acc.finalize()

    fit([_batch(40 + i) for i in range(9)])  # a different fit first; its last group is ragged (9 = 8 + 1)
    loss1, H1 = fit(batches)
    loss2, H2 = fit(batches)
    assert torch.equal(loss1, loss2)
    for F1, F2 in zip(H1.kfacs, H2.kfacs):
        for a_, b_ in zip(F1, F2):
            assert rel(a_, b_) < 1e-6  # (split-K partials are reduced in a fixed order: equal up to the launch geometry)


def test_a_ragged_last_minibatch_as_in_a_50k_fit(resnet, monkeypatch):
    """N = 50 000 = 390 x 128 + 80 (bench.py's fit_50k): the 80-sample minibatch ends the stacked pixel-pair group early
    and runs other tile shapes; default scheduling against the serial one."""
    from laplace_amd import HipGGN

    b = HipGGN(resnet, "classification")
    full = [_batch(60 + i) for i in range(3)]
    rest = 80 if BATCH == 128 else 1
    batches = full + [(full[0][0][:rest].contiguous(), full[0][1][:rest].contiguous())]
    acc = b.kron_accumulator(N)
    for X, y in batches:
        acc.add_batch(X, y)
    loss, H = acc.finalize()
    from laplace_amd.backend import KronAccumulator

    monkeypatch.setattr(KronAccumulator, "pix_group", 1)
    monkeypatch.setattr(KronAccumulator, "lag_join", False)
    ser = b.kron_accumulator(N, overlap=False)
    for X, y in batches:
        ser.add_batch(X, y)
    loss_s, H_s = ser.finalize()
    assert rel(loss, loss_s) < 1e-6
    for F_, S_ in zip(H.kfacs, H_s.kfacs):
        for a_, s_ in zip(F_, S_):
            assert rel(a_, s_) < 1e-5


@pytest.mark.skipif(DEV == "cpu", reason="streams: device only")
def test_a_caller_that_refills_one_device_buffer_per_minibatch():
    """With two minibatches in flight the forward of a minibatch runs on a lane stream while the caller's stream goes on:
    a loader that copies every minibatch into the SAME device tensors must still be safe (the caller's stream waits until
    the lane has consumed its inputs).  Same factors as with a fresh tensor per minibatch."""
    from laplace_amd import HipGGN
    from laplace_amd.nets import ResNet18

    torch.manual_seed(711)
    model = ResNet18(10, act=torch.tanh).to(DEV).eval()
    g = torch.Generator().manual_seed(21)
    host = [(torch.randn(128, 3, 32, 32, generator=g).pin_memory(), torch.randint(10, (128,), generator=g).pin_memory()) for _ in range(6)]
    b = HipGGN(model, "classification")
    acc = b.kron_accumulator(50_000)
    assert acc.lanes >= 2
    Xd, yd = torch.empty(128, 3, 32, 32, device=DEV), torch.empty(128, dtype=torch.long, device=DEV)
    for X, y in host:
        Xd.copy_(X, non_blocking=True)
        yd.copy_(y, non_blocking=True)
        acc.add_batch(Xd, yd)
    _, H = acc.finalize()
    ref = b.kron_accumulator(50_000)
    for X, y in host:
        ref.add_batch(X.to(DEV), y.to(DEV))
    _, Hr = ref.finalize()
    for F_, G_ in zip(H.kfacs, Hr.kfacs):
        for a, w in zip(F_, G_):
            assert (a - w).abs().max().item() <= 1e-5 * w.abs().max().item()


def _post_activation_maps(model, X):
    outs = {}
    hs = [mod.register_forward_hook(lambda m_, i, o, n=n: outs.__setitem__(n, i[0].detach()))
          for n, mod in model.named_modules() if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear))]
    with torch.no_grad():
        model(X)
    for h in hs:
        h.remove()
    return outs


@pytest.mark.parametrize("act", ["relu", "tanh"])
def test_where_the_relu_margin_comes_from(act):
    """The ReLU rows of this file sit at 4e-5 against fp64 products of a stock fp32 tape, tanh at 3e-7; the explanation —
    pre-activations within rounding of zero are decided differently by any two executions, and ONE flipped mask moves a
    factor of a small minibatch by 1e-4 — was asserted until round 5, not measured.  Measured here on 64 samples of the timed
    configuration, three executions against the fp64 forward (profiles/r05_mask_flips.log): the kernels', a stock fp32
    forward on the device (library convolutions), a stock fp32 forward on the CPU.

    * tanh (no masks): every factor block of the kernels within 1e-5 of the fp64 oracle (curvlinops.py:77-108 restated) —
      the arithmetic itself is not the margin.
    * ReLU: the kernels' post-activation maps are within 1 - 3.2 x of the stock device forward's error, and their masks differ from
      fp64's in as few places (measured: 5 of 45 M, the stock device forward: 5, the CPU: 0) — so the factor blocks below a
      flipped mask move by ~1e-4 at 64 samples (measured worst block 2.5e-4 where the CPU fp32 oracle, with no flip, is at
      1.6e-7) and by 1 / sqrt(samples) of that at the benched size (4.4e-5 over 1408 samples, the first test of this file).
      Asserted: the flip counts and the activation errors (within 6 x; measured 1 - 3.2 x) against the stock device forward, the
      blocks at 1e-3; the three
      numbers are logged separately (tests/parity_log.py)."""
    import copy

    from laplace_amd import HipGGN
    from laplace_amd._lib import get_kernels
    from laplace_amd.nets import ResNet18
    from laplace_amd.sweep_nhwc import SplitSweep
    from oracle import curvature_oracle as co
    from tests.parity_log import record_error

    torch.manual_seed(711)
    m32 = ResNet18(10, act=torch.relu if act == "relu" else torch.tanh).eval()
    m64 = copy.deepcopy(m32).double()
    n = 64 if DEV != "cpu" else 2
    X, y = _batch(0)
    X, y = X[:n].cpu(), y[:n].cpu()
    mdev = copy.deepcopy(m32).to(DEV)
    acc = HipGGN(mdev, "classification").kron_accumulator(N)
    acc.add_batch(X.to(DEV), y.to(DEV))
    loss, H = acc.finalize()
    _, kf64 = co.kfac_ggn(m64, X.double(), y, N, "classification")
    _, kf32 = co.kfac_ggn(m32, X, y, N, "classification")
    worst = worst32 = 0.0
    for F_, G64, G32 in zip(H.kfacs, kf64, kf32):
        for a, w, w32 in zip(F_, G64, G32):
            worst = max(worst, (a.double().cpu() - w).abs().max().item() / (w.abs().max().item() + 1e-300))
            worst32 = max(worst32, (w32.double() - w).abs().max().item() / (w.abs().max().item() + 1e-300))
    record_error(worst, "kernels-vs-fp64-oracle")
    record_error(worst32, "fp32-cpu-oracle-vs-fp64-oracle")
    print(f"timed configuration, {n} samples, {act}: worst block kernels {worst:.2e}, fp32 CPU oracle {worst32:.2e} (both against the fp64 oracle)")
    if act == "tanh":
        assert worst < 1e-5
        return
    assert worst < 1e-3
    # the forward that decides the masks: the kernels' against a stock fp32 forward on the same device, both against fp64
    ref = _post_activation_maps(m64, X.double())
    stock = _post_activation_maps(copy.deepcopy(m32).to(DEV), X.to(DEV))
    taps = {nm: mod for nm, mod in mdev.named_modules() if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear))}
    sw = SplitSweep(mdev, taps, kernels=get_kernels)
    sw.forward(X.to(DEV))
    flips = {"kernels": 0, "stock": 0}
    ratio = 0.0
    for nm in taps:
        if nm == "conv1" or sw.taps[nm].get("a") is None:
            continue
        w = ref[nm]
        errs = {}
        for tag, t in (("kernels", sw.taps[nm]["a"]), ("stock", stock[nm])):
            t = t.double().cpu().reshape(w.shape)
            flips[tag] += int(((t > 0) != (w > 0)).sum())
            errs[tag] = (t - w).abs().max().item()
        ratio = max(ratio, errs["kernels"] / (errs["stock"] + 1e-300))
    print(f"  masks differing from the fp64 forward: kernels {flips['kernels']}, stock fp32 on the device {flips['stock']}; "
          f"worst ratio of the activation errors {ratio:.1f}")
    assert ratio < 6.0 and flips["kernels"] <= 4 * flips["stock"] + 8  # (measured: 3.2, 5 vs 5)
    # ... and the FACTORS of that stock device forward (the autograd tape: library convolutions forward and backward, one
    # reverse pass per seed, `use_sweep = False`; the Gram kernels are exact-fp32 MFMA products) against the same fp64 oracle,
    # block by block: the 1e-3 above is earned if the kernels' blocks are no further from fp64 than twice what a stock fp32
    # execution of curvlinops.py:77-108 on this device is — the rest is the masks, which no fp32 execution gets right
    bs = HipGGN(copy.deepcopy(m32).to(DEV), "classification")
    bs.use_sweep = False
    accs = bs.kron_accumulator(N)
    accs.add_batch(X.to(DEV), y.to(DEV))
    _, Hs = accs.finalize()
    worst_stock = worst_ratio = 0.0
    moved = {"kernels": 0, "stock": 0}  # blocks further than 2e-5 from fp64: those below a flipped mask
    for F_, Fs, G64 in zip(H.kfacs, Hs.kfacs, kf64):
        for a, st, w in zip(F_, Fs, G64):
            scale = w.abs().max().item() + 1e-300
            ek = (a.double().cpu() - w).abs().max().item() / scale
            es = (st.double().cpu() - w).abs().max().item() / scale
            worst_stock = max(worst_stock, es)
            worst_ratio = max(worst_ratio, ek / max(es, 2e-6))  # (blocks where both sit at fp32 rounding: ratio of noise)
            moved["kernels"] += ek > 2e-5
            moved["stock"] += es > 2e-5
    record_error(worst_stock, "stock-device-tape-vs-fp64-oracle")
    record_error(worst_ratio, "worst-block-ratio-kernels-over-stock-device (the two executions flip DIFFERENT masks)")
    print(f"  factors of the stock fp32 device tape against fp64: worst block {worst_stock:.2e} (kernels {worst:.2e}); blocks beyond 2e-5: "
          f"kernels {moved['kernels']}, stock {moved['stock']}; worst per-block ratio kernels / stock {worst_ratio:.1f}")
    # Measured (round 6, profiles/r06_parity_errors_*.log): stock device tape 2.461e-4, kernels 2.456e-4 — a stock fp32
    # execution of the reference on this device lands where the kernels do.  Block by block the two differ by up to 68 x in
    # either direction: each execution flips its own five masks, and a block is at 1e-4 below a flip and at 1e-6 elsewhere.
    assert worst <= 2.0 * worst_stock and moved["kernels"] <= 2 * moved["stock"] + 4
