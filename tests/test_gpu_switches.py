"""Every path selector of the product (class / instance attributes; the case names are the environment switches most of
them were up to round 4 — round 6 keeps LK_LIB alone) gives
the same curvature and the same predictive as the default path: a c4-shaped KFAC fit (ResNet-18, two minibatches of 16,
one of 5) + Kron GLM predictive with ONE selector flipped at a time, against the default run.  A switch nobody tests is a
code path nobody knows to work (-m gpu)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
N = 50_000


def rel(a, b):
    a, b = a.double(), b.double()
    from tests.parity_log import record_error

    return record_error((a - b).abs().max().item() / (b.abs().max().item() + 1e-300))


@pytest.fixture(scope="module")
def setup():
    from laplace_amd.nets import ResNet18

    torch.manual_seed(711)
    model = ResNet18(10, act=torch.tanh).to(DEV).eval()  # (smooth: two executions of the forward are comparable)
    g = torch.Generator().manual_seed(5)
    batches = [(torch.randn(b, 3, 32, 32, generator=g).to(DEV), torch.randint(10, (b,), generator=g).to(DEV)) for b in (16, 16, 5)]
    return model, batches


def _run(model, batches, backend_attrs=None, acc_attrs=None, sweep_attrs=None, kernel_attrs=None, literal=False):
    from laplace_amd import HipGGN
    from laplace_amd import predictive as Pr
    from laplace_amd._lib import get_kernels
    from laplace_amd.sweep_nhwc import SplitSweep

    K = get_kernels()
    saved_k = {k: getattr(K, k) for k in (kernel_attrs or {})}
    saved_s = {k: getattr(SplitSweep, k) for k in (sweep_attrs or {})}
    try:
        for k, v in (kernel_attrs or {}).items():
            setattr(K, k, v)
        for k, v in (sweep_attrs or {}).items():
            setattr(SplitSweep, k, v)
        b = HipGGN(model, "classification")
        for k, v in (backend_attrs or {}).items():
            setattr(b, k, v)
        if literal:  # the reference's loop: H += backend.kron(X, y, N)
            from laplace_amd.kron import HipKron

            H = HipKron.init_from_model(b.params, DEV, torch.float32)
            loss = 0.0
            for X, y in batches:
                lb, Hb = b.kron(X, y, N=N)
                H += Hb
                loss = loss + lb
        else:
            acc = b.kron_accumulator(N)
            for k, v in (acc_attrs or {}).items():
                setattr(acc, k, v)
            for X, y in batches:
                acc.add_batch(X, y)
            loss, H = acc.finalize()
        dec = H.decompose()
        dec.check_converged()
        post = dec * (N / 37.0) + torch.tensor(1.0, device=DEV)
        f_mu, f_var = Pr.glm_variance_kron(b, batches[0][0][:4], post)
        return loss, H, f_mu, f_var
    finally:
        for k, v in saved_k.items():
            setattr(K, k, v)
        for k, v in saved_s.items():
            setattr(SplitSweep, k, v)


@pytest.fixture(scope="module")
def default(setup):
    return _run(*setup)


CASES = {
    # env switch                          what is flipped
    "use_sweep False": dict(backend_attrs={"use_sweep": False}),
    "use_split_sweep False": dict(backend_attrs={"use_split_sweep": False}),
    "LK_LAZY_KRON=0 (literal loop)": dict(backend_attrs={"lazy_kron": False}, literal=True),
    "LK_LAZY_KRON=1 (literal loop)": dict(literal=True),
    "LK_PIXGRAM=0": dict(acc_attrs={"use_pixgram": False}),
    "LK_DEFER_BN=0": dict(acc_attrs={"_defer_bn": False}),
    "LK_PERSIST_SLABS=0": dict(acc_attrs={"_persist_slabs": False}, backend_attrs={"use_split_sweep": False}),
    "LK_PIX_GROUP=1": dict(acc_attrs={"pix_group": 1}),
    "LK_PIX_GROUP=2": dict(acc_attrs={"pix_group": 2}),
    "LK_LAG_JOIN=0": dict(acc_attrs={"lag_join": False}),
    "activations copied into the pixel-pair stacks": dict(acc_attrs={"direct_stack": False}),
    "lag_depth=3": dict(acc_attrs={"lag_depth": 3}),
    "LK_FUSE_STRIDED=0": dict(sweep_attrs={"fuse_strided": False}),
    "copies + a separate absmax pass": dict(kernel_attrs={"use_copy_absmax": False}),
    "overlap=False": dict(acc_attrs={"overlap": False}),
    "lanes 1": dict(acc_attrs={"lanes": 1}),
    "lanes 3": dict(acc_attrs={"lanes": 3}),
    "LK_FLUSH_STREAMS=1": dict(acc_attrs={"flush_streams": 1}),
    "LK_FLUSH_STREAMS=5": dict(acc_attrs={"flush_streams": 5}),
    "LK_FUSE_VJP=0": dict(sweep_attrs={"fuse_vjp": False}),
    "LK_NHWC_FORWARD=0": dict(sweep_attrs={"nhwc_forward": False}),
    "LK_PIXPAIR16=0": dict(kernel_attrs={"use_pixpair16": False}),
    "64-channel pixel pairs: one workgroup per (pixel, shift)": dict(kernel_attrs={"use_pixpair13": False}),
    "LK_SHIFTCORR=0": dict(kernel_attrs={"use_shiftcorr": False}, acc_attrs={"use_pixgram": False}),
    "fp32-operand quadratic form": dict(kernel_attrs={"use_quad_planes": False}),
    "A factors of strided / stem convolutions on the exact-fp32 MFMA kernel": dict(kernel_attrs={"use_gram_conv16": False}),
    "LK_WINP=0 (generic fused launches)": dict(kernel_attrs={"use_winp": False}),
    "conv_config plain row order": dict(kernel_attrs={"conv_config": 2 | 32768}),
    "conv_config round-4 window staging": dict(kernel_attrs={"conv_config": 2 | (1 << 30)}),
    "conv_config two columns per XCD": dict(kernel_attrs={"conv_config": 2 | (1 << 28)}),
    "conv_config eight columns per XCD": dict(kernel_attrs={"conv_config": 2 | (3 << 28)}),
    "lane_priority=0": dict(acc_attrs={"lane_priority": 0}),
}


@pytest.mark.parametrize("name", list(CASES))
def test_switch_gives_the_default_results(setup, default, name):
    loss0, H0, mu0, var0 = default
    loss, H, mu, var = _run(*setup, **CASES[name])
    assert rel(loss, loss0) < 1e-5, name
    assert len(H.kfacs) == len(H0.kfacs)
    for i, (F_, G_) in enumerate(zip(H.kfacs, H0.kfacs)):
        for j, (a, b_) in enumerate(zip(F_, G_)):
            assert rel(a, b_) < 1e-4, f"{name}: block {i} factor {j} rel {rel(a, b_):.2e}"
    assert rel(mu, mu0) < 1e-4, name
    assert rel(var, var0) < 1e-3, f"{name}: predictive variance rel {rel(var, var0):.2e}"  # (through two eigendecompositions)


def test_the_rotation_switch(setup, default, monkeypatch):
    """LK_ROT_CONV=0: the predictive's eigenbasis rotations as library convolutions / GEMMs"""
    import laplace_amd.backend as be

    monkeypatch.setattr(be, "_OWN_ROTATION", False)
    _, _, mu, var = _run(*setup)
    assert rel(mu, default[2]) < 1e-4 and rel(var, default[3]) < 1e-3
