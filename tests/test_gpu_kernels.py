"""Parity of every C-ABI entry point against its CPU restatement (tests/emulated_kernels.py in
fp64), on seeded inputs incl. ragged / edge shapes.  Runs on the MI355X only (-m gpu).

Tolerance: fp32 results vs fp64 truth, 1e-4 relative to the largest magnitude of the expected
tensor (BASELINE.json north_star: "within 1e-4 rel fp32") — most ops land near 1e-6.
"""
import os

import numpy as np
import pytest
import torch

from tests.emulated_kernels import EmulatedKernels

pytestmark = pytest.mark.gpu

EMU = EmulatedKernels()
# LK_TEST_DEVICE=cpu runs the test bodies against the emulation itself (a self-check of this file on
# a GPU-less box); the real run uses the HIP library on cuda:0.
DEV = os.environ.get("LK_TEST_DEVICE", "cuda")


@pytest.fixture(scope="module")
def K():
    if DEV == "cpu":
        return EMU
    from laplace_amd._lib import HipKernels

    return HipKernels()


def _sync():
    if DEV != "cpu":
        torch.cuda.synchronize()


def relerr(got, want):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    scale = want.abs().max().item() + 1e-30
    from tests.parity_log import record_error

    return record_error((got - want).abs().max().item() / scale)


def assert_close(got, want, tol=1e-4, what=""):
    e = relerr(got, want)
    assert e < tol, f"{what}: rel err {e:.3e} >= {tol}"


def _sum_others(p):
    """sum_{k != i} p_k without forming 1 - p_i"""
    C = p.shape[1]
    mask = 1.0 - torch.eye(C, dtype=p.dtype)
    return p @ mask


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


# ---- Gram family ---------------------------------------------------------------------------------
@pytest.mark.parametrize("K_,n", [(1, 1), (7, 3), (100, 27), (1000, 64), (333, 65), (257, 128), (50, 150),
                                  (4096, 200), (130, 400), (2048, 576), (64, 1152), (5, 130)])
def test_gram_tn(K, K_, n):
    X = rnd(K_, n, seed=K_ + n)
    C0 = rnd(n, n, seed=1)
    C0 = C0 + C0.T
    want = EMU.gram_tn(X, 0.37, C0.clone())
    got = K.gram_tn(X.float().to(DEV), 0.37, C0.float().to(DEV))
    assert_close(got, want, what=f"gram_tn K={K_} n={n}")
    # symmetric output
    assert_close(got, got.T, tol=1e-6, what="symmetry")


def test_gram_tn_unaligned_rows(K):
    # ldx / pointer alignment that forces the scalar (VEC=1) loader: odd n
    X = rnd(300, 75, seed=3)
    want = EMU.gram_tn(X, 1.0, torch.zeros(75, 75, dtype=torch.float64))
    got = K.gram_tn(X.float().to(DEV), 1.0, torch.zeros(75, 75, device=DEV))
    assert_close(got, want, what="gram_tn odd n")


@pytest.mark.parametrize("nb,n,L", [(3, 4, 4), (20, 16, 100), (12, 64, 256), (5, 64, 1024), (40, 128, 64),
                                    (7, 256, 16), (30, 6, 784), (9, 130, 36), (2, 512, 16)])
def test_gram_nt(K, nb, n, L):
    X = rnd(nb, n, L, seed=nb + n + L)
    want = EMU.gram_nt(X, 1.7, torch.zeros(n, n, dtype=torch.float64))
    got = K.gram_nt(X.float().to(DEV), 1.7, torch.zeros(n, n, device=DEV))
    assert_close(got, want, what=f"gram_nt nb={nb} n={n} L={L}")


def test_gram_nt_segments(K):
    """per-seed gradients passed as a pointer table == the stacked tensor"""
    segs = [rnd(6, 128, 64, seed=s) for s in range(10)]
    want = EMU.gram_nt(torch.cat(segs), 0.3, torch.zeros(128, 128, dtype=torch.float64))
    got = K.gram_nt([s.float().to(DEV) for s in segs], 0.3, torch.zeros(128, 128, device=DEV))
    assert_close(got, want, what="gram_nt segments")
    segs = [rnd(3, 64, 100, seed=s) for s in range(5)]
    want = EMU.gram_nt(torch.cat(segs), 1.0, torch.zeros(64, 64, dtype=torch.float64))
    got = K.gram_nt([s.float().to(DEV) for s in segs], 1.0, torch.zeros(64, 64, device=DEV))
    assert_close(got, want, what="gram_nt segments small/unaligned")


CONV_CASES = [
    # B, Cin, H, W, k, stride, pad, dil
    (4, 3, 5, 5, 2, 2, 0, 1),      # reference "complex_model" conv
    (3, 2, 5, 5, 3, 1, 1, 1),
    (3, 4, 5, 5, 3, 2, 1, 1),
    (2, 3, 32, 32, 5, 1, 0, 1),    # LeNet conv1
    (2, 6, 14, 14, 5, 1, 0, 1),    # LeNet conv2
    (4, 64, 8, 8, 3, 1, 1, 1),     # ResNet 3x3
    (4, 64, 8, 8, 3, 2, 1, 1),     # ResNet strided 3x3
    (4, 64, 8, 8, 1, 2, 0, 1),     # ResNet 1x1 downsample
    (2, 8, 9, 7, (3, 2), (2, 1), (1, 0), (1, 2)),  # anisotropic everything
    (2, 128, 4, 4, 3, 1, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_gram_conv(K, case):
    B, Cin, H, W, k, s, p, d = case
    x = rnd(B, Cin, H, W, seed=B + Cin + H)
    kk = k if isinstance(k, tuple) else (k, k)
    n = Cin * kk[0] * kk[1]
    want = EMU.gram_conv(x, k, s, p, d, 0.5, torch.zeros(n, n, dtype=torch.float64))
    got = K.gram_conv(x.float().to(DEV), k, s, p, d, 0.5, torch.zeros(n, n, device=DEV))
    assert_close(got, want, what=f"gram_conv {case}")
    # fused mode: native order + upper only, then symmetrize + permute must give the same matrix
    nat = K.gram_conv(x.float().to(DEV), k, s, p, d, 0.5, torch.zeros(n, n, device=DEV), upper_only=True, native=True)
    K.symmetrize(nat)
    out = K.permute_native_to_unfold(nat, Cin, kk[0] * kk[1], torch.zeros(n, n, device=DEV))
    assert_close(out, want, what=f"gram_conv fused {case}")


@pytest.mark.parametrize("B,Cin,H,W", [(2, 8, 8, 8), (3, 64, 8, 8), (2, 64, 32, 32), (2, 5, 9, 11), (2, 128, 16, 16),
                                        (1, 256, 8, 8), (3, 4, 2, 40), (2, 12, 17, 3)])
def test_conv3x3_shift_correlation_path(K, B, Cin, H, W):
    """3x3 / stride 1 / pad 1 A factor through the shift-correlation identity == implicit-im2col Gram == unfold."""
    if DEV == "cpu":
        pytest.skip("kernel-level identity; the emulation has a single path")
    x = rnd(B, Cin, H, W, seed=B + Cin + H + W)
    n = Cin * 9
    want = EMU.gram_conv(x, 3, 1, 1, 1, 0.7, torch.zeros(n, n, dtype=torch.float64))
    xg = x.float().to(DEV)
    prev = K.use_shiftcorr
    try:
        K.use_shiftcorr = True
        nb = K.lib.lk_conv3x3_shiftcorr_workspace_bytes(B, H, W, Cin)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        nat = torch.zeros(n, n, device=DEV)
        xh = K.nchw_to_nhwc(xg)
        import ctypes
        rc = K.lib.lk_conv3x3_shiftcorr_f32(ctypes.c_void_p(xh.data_ptr()), B, H, W, Cin, 0.7, ctypes.c_void_p(nat.data_ptr()),
                                            ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, K.lib.lk_last_error()
        got_sc = K.permute_native_to_unfold(nat, Cin, 9, torch.zeros(n, n, device=DEV))
        K.use_shiftcorr = False
        got_direct = K.gram_conv(xg, 3, 1, 1, 1, 0.7, torch.zeros(n, n, device=DEV))
    finally:
        K.use_shiftcorr = prev
    assert_close(got_sc, want, what="shift-correlation")
    assert_close(got_direct, want, what="implicit im2col")
    assert_close(got_sc, got_direct, tol=1e-5, what="paths agree")


def test_gram_accumulates_and_is_additive(K):
    """sum over minibatches == one big batch (the sharding licence, baselaplace.py:984-985)."""
    X = rnd(900, 200, seed=9).float().to(DEV)
    full = K.gram_tn(X, 1.0, torch.zeros(200, 200, device=DEV))
    acc = torch.zeros(200, 200, device=DEV)
    for part in X.split(300):
        K.gram_tn(part.contiguous(), 1.0, acc)
    assert_close(acc, full, tol=1e-5, what="additivity")


@pytest.mark.parametrize("n", [1, 10, 27, 33, 40, 64, 70, 100, 130, 200])
def test_gram_direct_epilogue_stays_inside_the_matrix(K, n):
    """Single-split, upper-only launches accumulate into C straight from the accumulators.  The tile covers
    up to 192 x 192 entries; everything beyond n x n must not even be read-modified-written (a `+= 0` out of
    bounds is invisible in the values but faults at the end of an allocation).  Guard bands of -0.0 reveal it:
    -0.0 + 0.0 = +0.0 flips the sign bit."""
    pad = 64 * 1024
    for kind in ("tn", "nt", "conv"):
        buf = torch.full((2 * pad + n * n,), -0.0, device=DEV)
        out = buf[pad:pad + n * n].view(n, n)
        out.zero_()
        if kind == "tn":
            X = rnd(48, n, seed=n)
            want = EMU.gram_tn(X, 0.5, torch.zeros(n, n, dtype=torch.float64))
            K.gram_tn(X.float().to(DEV), 0.5, out, upper_only=True)
        elif kind == "nt":
            X = rnd(3, n, 16, seed=n)
            want = EMU.gram_nt(X, 0.5, torch.zeros(n, n, dtype=torch.float64))
            K.gram_nt(X.float().to(DEV), 0.5, out, upper_only=True)
        else:
            if n % 9:
                continue
            x = rnd(2, n // 9, 4, 4, seed=n)
            want = EMU.gram_conv(x, 3, 1, 1, 1, 0.5, torch.zeros(n, n, dtype=torch.float64))
            K.gram_conv(x.float().to(DEV), 3, 1, 1, 1, 0.5, out, upper_only=True, native=False)
        _sync()
        assert torch.signbit(buf[:pad]).all() and torch.signbit(buf[pad + n * n:]).all(), f"{kind}: wrote outside C"
        got = torch.triu(out.double().cpu())
        if kind == "conv":
            continue  # native/unfold ordering is covered by test_gram_conv; only the bounds matter here
        assert (got - torch.triu(want)).abs().max() / want.abs().max() < 1e-5, kind


def test_nchw_to_nhwc(K):
    x = rnd(3, 70, 9, 11).float().to(DEV)
    assert torch.equal(K.nchw_to_nhwc(x), x.permute(0, 2, 3, 1).contiguous())


# ---- likelihood ------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C", [(1, 2), (10, 2), (257, 10), (33, 100)])
def test_softmax_hess_sqrt(K, B, C):
    f = rnd(B, C, seed=B) * 3
    y = torch.randint(C, (B,), generator=torch.Generator().manual_seed(1))
    la = torch.zeros(1, dtype=torch.float64)
    want = EMU.softmax_hess_sqrt(f, y, la)
    lg = torch.zeros(1, device=DEV)
    got = K.softmax_hess_sqrt(f.float().to(DEV), y.to(DEV), lg)
    assert_close(got, want, tol=1e-5, what="S")
    assert_close(lg, la, tol=1e-5, what="CE loss")
    # S S^T == diag(p) - p p^T
    S = got.permute(1, 2, 0).double().cpu()  # [B, j, c]
    p = torch.softmax(f, -1)
    assert_close(S @ S.transpose(1, 2), torch.diag_embed(p) - p.unsqueeze(2) * p.unsqueeze(1), tol=1e-5)


@pytest.mark.parametrize("B,C", [(1, 2), (10, 2), (257, 10), (33, 100), (5, 3)])
def test_softmax_hess_cholesky_root(K, B, C):
    f = rnd(B, C, seed=B) * 4
    f[0, 0] = 60.0  # one nearly one-hot row: vanishing tail sums
    y = torch.randint(C, (B,), generator=torch.Generator().manual_seed(1))
    la = torch.zeros(1, dtype=torch.float64)
    want = EMU.softmax_hess_sqrt(f, y, la, cholesky=True)
    lg = torch.zeros(1, device=DEV)
    got = K.softmax_hess_sqrt(f.float().to(DEV), y.to(DEV), lg, cholesky=True)
    assert got.shape == (C - 1, B, C)
    assert_close(got, want, tol=1e-5, what="Cholesky root")
    assert_close(lg, la, tol=1e-5, what="CE loss")
    S = got.permute(1, 2, 0).double().cpu()
    p = torch.softmax(f, -1)
    # cancellation-free reference: Lambda_ii = p_i * sum_{k != i} p_k  (1 - p_i rounds to 0 for one-hot rows)
    others = p.sum(1, keepdim=True) - p
    Lam = -p.unsqueeze(2) * p.unsqueeze(1)
    Lam[:, torch.arange(C), torch.arange(C)] = p * torch.where(others > 1e-3, others, _sum_others(p))
    assert_close(S @ S.transpose(1, 2), Lam, tol=1e-5, what="L L^T")


def test_sq_err_sum(K):
    f, y = rnd(1000, 3, seed=1), rnd(1000, 3, seed=2)
    lg = torch.zeros(1, device=DEV)
    K.sq_err_sum(f.float().to(DEV), y.float().to(DEV), 0.5, lg)
    assert_close(lg, 0.5 * ((f - y) ** 2).sum().reshape(1), tol=1e-5)


# ---- diag / Jacobians --------------------------------------------------------------------------------
@pytest.mark.parametrize("B,S,Di,Do", [(10, 2, 3, 20), (100, 1, 1, 50), (130, 10, 400, 120), (64, 3, 17, 33)])
def test_diag_and_jac_linear(K, B, S, Di, Do):
    a, g = rnd(B, Di, seed=1), rnd(S, B, Do, seed=2)
    hw, hb = torch.zeros(Do * Di, dtype=torch.float64), torch.zeros(Do, dtype=torch.float64)
    EMU.diag_ggn_linear(a, g, 0.7, hw, hb)
    ghw, ghb = torch.zeros(Do * Di, device=DEV), torch.zeros(Do, device=DEV)
    K.diag_ggn_linear(a.float().to(DEV), g.float().to(DEV), 0.7, ghw, ghb)
    assert_close(ghw, hw, what="diag w")
    assert_close(ghb, hb, what="diag b")
    P = Do * Di + Do + 5
    Js = torch.zeros(B, S, P, dtype=torch.float64)
    EMU.jac_linear(a, g, Js, 2, 2 + Do * Di)
    gJ = torch.zeros(B, S, P, device=DEV)
    K.jac_linear(a.float().to(DEV), g.float().to(DEV), gJ, 2, 2 + Do * Di)
    assert_close(gJ, Js, tol=1e-6, what="jac_linear")


@pytest.mark.parametrize("case", CONV_CASES[:7])
def test_jac_conv_and_sq_colsum(K, case):
    B, Cin, H, W, k, s, p, d = case
    Do, S = 5, 3
    x = rnd(B, Cin, H, W, seed=5)
    kk = (k, k)
    OH = (H + 2 * p - d * (k - 1) - 1) // s + 1
    OW = (W + 2 * p - d * (k - 1) - 1) // s + 1
    g = rnd(S, B, Do, OH, OW, seed=6)
    Dk = Cin * k * k
    P = Do * Dk + Do
    Js = torch.zeros(B, S, P, dtype=torch.float64)
    EMU.jac_conv(x, g, kk, s, p, d, Js, 0, Do * Dk)
    gJ = torch.zeros(B, S, P, device=DEV)
    K.jac_conv(x.float().to(DEV), g.float().to(DEV), kk, s, p, d, gJ, 0, Do * Dk)
    assert_close(gJ, Js, tol=1e-5, what=f"jac_conv {case}")
    h = torch.zeros(Do * Dk, dtype=torch.float64)
    EMU.sq_colsum(Js, 0, Do * Dk, 1.3, h)
    gh = torch.zeros(Do * Dk, device=DEV)
    K.sq_colsum(gJ, 0, Do * Dk, 1.3, gh)
    assert_close(gh, h, what="sq_colsum")


# ---- dense last layer --------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,D,bias,cls", [(10, 2, 20, True, True), (64, 10, 33, True, True), (50, 3, 16, False, True),
                                            (40, 2, 20, True, False), (33, 1, 50, True, False), (200, 10, 130, True, True),
                                            (300, 3, 512, True, True), (129, 2, 257, False, False)])
def test_ll_ggn_full_and_quadform(K, B, C, D, bias, cls):
    phi = rnd(B, D, seed=B)
    probs = torch.softmax(rnd(B, C, seed=3), -1) if cls else None
    P = C * D + (C if bias else 0)
    H = torch.zeros(P, P, dtype=torch.float64)
    EMU.ll_ggn_full(phi, probs, bias, 0.9, H)
    gH = torch.zeros(P, P, device=DEV)
    K.ll_ggn_full(phi.float().to(DEV), None if probs is None else probs.float().to(DEV), bias, 0.9, gH)
    assert_close(gH, H, what="ll_ggn_full")
    Sigma = torch.linalg.inv(H + torch.eye(P, dtype=torch.float64))
    want = EMU.dense_quadform_ll(phi, Sigma, C, bias)
    got = K.dense_quadform_ll(phi.float().to(DEV), Sigma.float().to(DEV).contiguous(), C, bias)
    assert_close(got, want, what="dense_quadform_ll")


# ---- eigensolver ---------------------------------------------------------------------------------------
def _eig_checks(K, A64, tol_rec=5e-6, tol_orth=5e-6, tol_val=5e-6):  # (measured: <= 1.2e-6 up to n = 576, profiles/r04_parity_errors.log)
    n = A64.shape[0]
    w, Q, info = K.syevj(A64.float().to(DEV).contiguous())
    _sync()
    assert int(info[0].item()) == 0, "eigensolver did not converge"
    w64, Q64 = w.double().cpu(), Q.double().cpu()
    wref = torch.linalg.eigvalsh(A64).clamp(min=0)
    scale = wref.abs().max().item() + 1e-30
    assert torch.all(w64[1:] >= w64[:-1]), "eigenvalues not ascending"
    from tests.parity_log import record_error

    val = record_error((w64 - wref).abs().max().item() / scale)
    assert val < tol_val, f"eigenvalues off by {val:.2e}"
    orth = record_error((Q64.T @ Q64 - torch.eye(n, dtype=torch.float64)).abs().max().item())
    assert orth < tol_orth, f"orthogonality {orth:.2e}"
    rec = record_error(((Q64 * w64) @ Q64.T - A64).abs().max().item() / scale)
    assert rec < tol_rec, f"reconstruction {rec:.2e}"
    return w, Q


@pytest.mark.parametrize("n", [1, 2, 10, 27, 64, 65, 84, 120, 128, 150, 257, 400, 576])
def test_syevj_psd(K, n):
    X = rnd(3 * n + 5, n, seed=n)
    _eig_checks(K, X.T @ X / (3 * n))


@pytest.mark.parametrize("n,rank", [(50, 10), (130, 40), (300, 17)])
def test_syevj_rank_deficient(K, n, rank):
    """KFAC factors are low rank (rank <= batch x positions): zero eigenvalues must clamp cleanly
    (tests/test_utils.py:39-49 of the reference)."""
    X = rnd(rank, n, seed=n)
    w, _ = _eig_checks(K, X.T @ X)
    assert (w[: n - rank].abs().max() / w.max()).item() < 1e-5


def test_syevj_reads_upper_triangle_only(K):
    n = 70
    X = rnd(200, n, seed=4)
    A = X.T @ X / 200
    garbage = A.clone()
    garbage[np.tril_indices(n, -1)] = 123.0  # lower triangle must be ignored (UPLO="U")
    w1, _, _ = K.syevj(A.float().to(DEV).contiguous())
    w2, _, _ = K.syevj(garbage.float().to(DEV).contiguous())
    assert_close(w2, w1, tol=1e-6)


def test_syevj_clustered_and_diagonal(K):
    A = torch.diag(torch.tensor([1.0] * 40 + [2.0] * 40 + [1e-3] * 20, dtype=torch.float64))
    _eig_checks(K, A)
    Qr, _ = torch.linalg.qr(rnd(100, 100, seed=8))
    _eig_checks(K, Qr @ A @ Qr.T)


@pytest.mark.parametrize("nstreams", [1, 3, 8])
def test_syevj_batched(K, nstreams):
    """All factors of a decomposition in one scheduled call: same guarantees per matrix as the single solve, for
    ragged sizes, rank-deficient and already-diagonal members, with fewer / more streams than matrices."""
    sizes = [(300, None), (257, 17), (130, None), (128, None), (84, 5), (64, None), (27, None), (10, None), (1, None)]
    mats64 = []
    for n, rank in sizes:
        X = rnd(rank if rank else 3 * n + 5, n, seed=n)
        mats64.append(X.T @ X / X.shape[0])
    mats64.append(torch.diag(torch.arange(1, 41, dtype=torch.float64)))  # converges in the first sweep
    mats = [A.float().to(DEV).contiguous() for A in mats64]
    streams = None
    if DEV != "cpu":
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = K.syevj_batched(mats, clamp=True, streams=streams)
    if streams:
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    _sync()
    assert len(outs) == len(mats)
    for A64, (w, Q, info) in zip(mats64, outs):
        n = A64.shape[0]
        assert int(info[0].item()) == 0
        w64, Q64 = w.double().cpu(), Q.double().cpu()
        wref = torch.linalg.eigvalsh(A64).clamp(min=0)
        scale = wref.abs().max().item() + 1e-30
        assert torch.all(w64[1:] >= w64[:-1])
        assert (w64 - wref).abs().max().item() / scale < 2e-5
        assert (Q64.T @ Q64 - torch.eye(n, dtype=torch.float64)).abs().max().item() < 2e-5
        assert ((Q64 * w64) @ Q64.T - A64).abs().max().item() / scale < 2e-5
    # identical to the single-matrix entry point (same kernels, same order of operations)
    w1, Q1, _ = K.syevj(mats[0])
    _sync()
    assert torch.equal(w1, outs[0][0]) and torch.equal(Q1, outs[0][1])


@pytest.mark.parametrize("nstreams", [2, 3])
def test_syevj_batched_more_matrices_than_slots(K, nstreams):
    """A round serves at most 64 matrices per lane: with 150 of them the rest waits for freed slots (a converged matrix
    hands its slot on).  Every matrix must come out as from the single solve, bit for bit."""
    sizes = [3 + (37 * i) % 140 for i in range(150)]
    mats64 = []
    for i, n in enumerate(sizes):
        X = rnd(2 * n + 3, n, seed=1000 + i)
        mats64.append(X.T @ X / X.shape[0])
    order = sorted(range(150), key=lambda i: -sizes[i])  # (largest first, as decompose passes them)
    mats = [mats64[i].float().to(DEV).contiguous() for i in order]
    streams = [torch.cuda.Stream() for _ in range(nstreams)] if DEV != "cpu" else None
    outs = K.syevj_batched(mats, clamp=True, streams=streams)
    if streams:
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    _sync()
    for k, i in enumerate(order):
        w, Q, info = outs[k]
        assert int(info[0].item()) == 0
        wref = torch.linalg.eigvalsh(mats64[i]).clamp(min=0)
        assert (w.double().cpu() - wref).abs().max().item() / wref.max().item() < 2e-5, f"matrix {i} (n = {sizes[i]})"
    for k in (0, 63, 64, 65, 127, 128, 149):
        w1, Q1, _ = K.syevj(mats[k])
        _sync()
        assert torch.equal(w1, outs[k][0]) and torch.equal(Q1, outs[k][1]), f"position {k}"


# ---- logdet / predictive -----------------------------------------------------------------------------------
@pytest.mark.parametrize("n1,n2", [(1, 0), (20, 0), (2, 20), (64, 576), (10, 513), (130, 77)])
def test_kron_logdet(K, n1, n2):
    l1 = rnd(n1, seed=1).abs() + 0.01
    l2 = rnd(n2, seed=2).abs() + 0.01 if n2 else None
    delta = torch.tensor([0.3], dtype=torch.float64)
    want = EMU.kron_logdet(l1, l2, delta, False, True)
    got = K.kron_logdet(l1.float().to(DEV), None if l2 is None else l2.float().to(DEV), delta.float().to(DEV), False, True)
    for g_, w_, nm in zip(got, want, ("val", "d_l1", "d_l2", "d_delta")):
        if w_ is not None:
            assert_close(g_, w_, tol=1e-5, what=nm)
    if n2:
        wd = EMU.kron_logdet(l1, l2, delta, True)[0]
        gd = K.kron_logdet(l1.float().to(DEV), l2.float().to(DEV), delta.float().to(DEV), True)[0]
        assert_close(gd, wd, tol=1e-5, what="damped")


@pytest.mark.parametrize("B,C,Do,Dk,L", [(3, 2, 5, 27, 9), (4, 10, 64, 576, 64), (2, 10, 512, 4608, 16),
                                        (5, 3, 40, 150, 100), (2, 1, 7, 9, 1), (130, 4, 33, 129, 17), (3, 7, 96, 200, 33),
                                        (2, 9, 64, 64, 256), (600, 5, 16, 16, 4)])
def test_quadform_shared(K, B, C, Do, Dk, L):
    """weight-sharing predictive (conv / sequence Linear): ragged tiles, padded output counts (7 -> 8, 9 -> 10),
    one and many workgroups per sample, Kronecker and diagonal weights; against the fp64 einsum"""
    u, v = rnd(B, C, Do, L, seed=1), rnd(B, Dk, L, seed=2)
    l1, l2 = rnd(Do, seed=3).abs(), rnd(Dk, seed=4).abs()
    d = torch.tensor([0.3], dtype=torch.float64)
    f32 = lambda t: t.float().to(DEV).contiguous()
    base = rnd(B, C, C, seed=5)
    base = base + base.transpose(1, 2)
    want = EMU.kron_quadform_shared(u, v, l1, l2, d, base.clone())
    got = K.kron_quadform_shared(f32(u), f32(v), f32(l1), f32(l2), f32(d), f32(base))
    assert_close(got, want, what="kron_quadform_shared")
    assert_close(got, got.transpose(1, 2), tol=1e-6, what="symmetry")
    var = rnd(Do, Dk, seed=6).abs()
    want = EMU.diag_quadform_shared(u, v, var, torch.zeros(B, C, C, dtype=torch.float64))
    got = K.diag_quadform_shared(f32(u), f32(v), f32(var), torch.zeros(B, C, C, device=DEV))
    assert_close(got, want, what="diag_quadform_shared")
    again = K.diag_quadform_shared(f32(u), f32(v), f32(var), torch.zeros(B, C, C, device=DEV))
    assert torch.equal(again, got)  # fixed-order reduction


@pytest.mark.parametrize("B,S,Do,Dk,L", [(3, 2, 5, 27, 9), (4, 9, 64, 576, 64), (2, 10, 512, 1152, 16), (130, 4, 33, 129, 17),
                                        (7, 1, 96, 200, 33), (1, 3, 16, 16, 4)])
def test_diag_ggn_shared(K, B, S, Do, Dk, L):
    """exact GGN diagonal of a weight-sharing layer: sum over (sample, seed) of the squared per-sample Jacobian,
    accumulated on top of what h already holds; ragged tiles, one and many samples per workgroup"""
    u, v = rnd(B, S, Do, L, seed=1), rnd(B, Dk, L, seed=2)
    base = rnd(Do * Dk, seed=3)
    f32 = lambda t: t.float().to(DEV).contiguous()
    want = EMU.diag_ggn_shared(u, v, 0.7, base.clone())
    got = K.diag_ggn_shared(f32(u), f32(v), 0.7, f32(base))
    assert_close(got, want, what="diag_ggn_shared")
    again = K.diag_ggn_shared(f32(u), f32(v), 0.7, f32(base))
    assert torch.equal(again, got)


def test_quadform_shared_rejects_more_outputs_than_accumulators(K):
    if DEV == "cpu":
        pytest.skip("limit of the HIP kernel")
    from laplace_amd._lib import LaplaceHipError

    z = lambda *s: torch.zeros(*s, device=DEV)
    with pytest.raises(LaplaceHipError, match="more than 10 outputs"):
        K.kron_quadform_shared(z(1, 11, 4, 2), z(1, 4, 2), z(4) + 1, z(4) + 1, z(1) + 1, z(1, 11, 11))


@pytest.mark.parametrize("nblocks,with_scale", [(1, False), (5, True), (45, True), (70, False)])
def test_kron_logdet_blocks(K, nblocks, with_scale):
    """whole-posterior logdet: mixed one-/two-factor blocks, more blocks than one launch carries (32), per-block
    deltas, the pending `H * scalar`; value and both derivatives against fp64"""
    g = torch.Generator().manual_seed(nblocks)
    blocks = []
    for b in range(nblocks):
        n1 = int(torch.randint(1, 300, (1,), generator=g))
        n2 = 0 if b % 3 == 2 else int(torch.randint(1, 700, (1,), generator=g))
        l1 = torch.rand(n1, generator=g, dtype=torch.float64) + 1e-3
        blocks.append((l1,) if n2 == 0 else (l1, torch.rand(n2, generator=g, dtype=torch.float64) * 3))
    deltas = torch.rand(nblocks, generator=g, dtype=torch.float64) + 0.05
    scale = torch.tensor([0.37], dtype=torch.float64) if with_scale else None
    want = EMU.kron_logdet_blocks(blocks, deltas, scale, True)
    f32 = lambda t: t.float().to(DEV).contiguous()
    got = K.kron_logdet_blocks([tuple(f32(l) for l in ls) for ls in blocks], f32(deltas),
                               None if scale is None else f32(scale), True)
    for g_, w_, nm in zip(got, want, ("val", "d_delta", "d_scale")):
        assert (g_ is None) == (w_ is None), nm
        if w_ is not None:
            assert_close(g_, w_, tol=1e-5, what=nm)
    # bitwise reproducible (fixed-order reduction)
    again = K.kron_logdet_blocks([tuple(f32(l) for l in ls) for ls in blocks], f32(deltas),
                                 None if scale is None else f32(scale), True)
    assert torch.equal(again[0], got[0]) and torch.equal(again[1], got[1])


@pytest.mark.parametrize("B,C,Do,Di", [(10, 2, 2, 20), (33, 10, 10, 512), (7, 3, 120, 400), (5, 1, 1, 50)])
def test_quadforms(K, B, C, Do, Di):
    u, v = rnd(C, B, Do, seed=1), rnd(B, Di, seed=2)
    l1, l2, lb = rnd(Do, seed=3).abs(), rnd(Di, seed=4).abs(), rnd(Do, seed=5).abs()
    d = torch.tensor([0.5], dtype=torch.float64)
    want = EMU.kron_quadform_linear(u, v, l1, l2, d, torch.zeros(B, C, C, dtype=torch.float64), u * 0.5, lb, d)
    f32 = lambda t: t.float().to(DEV).contiguous()
    got = K.kron_quadform_linear(f32(u), f32(v), f32(l1), f32(l2), f32(d), torch.zeros(B, C, C, device=DEV), f32(u * 0.5),
                                 f32(lb), f32(d))
    assert_close(got, want, what="kron_quadform_linear")
    vw, vb = rnd(Do * Di, seed=6).abs(), rnd(Do, seed=7).abs()
    want = EMU.diag_quadform_linear(v, u, vw, vb, torch.zeros(B, C, C, dtype=torch.float64))
    got = K.diag_quadform_linear(f32(v), f32(u), f32(vw), f32(vb), torch.zeros(B, C, C, device=DEV))
    assert_close(got, want, what="diag_quadform_linear")
    Js, var = rnd(B, C, 301, seed=8), rnd(301, seed=9).abs()
    assert_close(K.diag_quadform_js(f32(Js), f32(var)), EMU.diag_quadform_js(Js, var), what="diag_quadform_js")


# ---- element-wise VJP of the seed-batched sweep -------------------------------------------------------------
@pytest.mark.parametrize("S,B,C,hw", [(9, 8, 16, 64), (3, 5, 7, 9), (1, 4, 6, 1), (4, 3, 5, 12), (9, 2, 3, 1024)])
@pytest.mark.parametrize("mult", ["bool", "float", None])
@pytest.mark.parametrize("scaled", [True, False])
@pytest.mark.parametrize("two", [False, True])
def test_vjp_scale_mask(K, S, B, C, hw, mult, scaled, two):
    gen = torch.Generator().manual_seed(S * 1000 + B * 100 + C * 10 + hw)
    g64 = torch.randn(S * B, C, hw, generator=gen, dtype=torch.float64)
    m64 = None
    if mult == "bool":
        m64 = torch.rand(B, C, hw, generator=gen) > 0.4
    elif mult == "float":
        m64 = torch.randn(B, C, hw, generator=gen, dtype=torch.float64)
    sc64 = torch.randn(C, generator=gen, dtype=torch.float64) if scaled else None
    h64 = torch.randn(S * B, C, hw, generator=gen, dtype=torch.float64) if two else None
    want = EMU.vjp_scale_mask(g64, S, m64, sc64, hw, h64)
    g = g64.float().to(DEV)
    h = None if h64 is None else h64.float().to(DEV)
    m = None if m64 is None else (m64.to(DEV) if m64.dtype == torch.bool else m64.float().to(DEV))
    sc = None if sc64 is None else sc64.float().to(DEV)
    got = K.vjp_scale_mask(g, S, m, sc, hw, h)
    _sync()
    assert got.shape == g.shape
    assert_close(got, want, 1e-6, "vjp_scale_mask")
    if mult == "bool":  # masked entries are exact zeros
        dead = (~m64).unsqueeze(0).expand(S, B, C, hw).reshape(S * B, C, hw)
        assert (got.cpu()[dead] == 0).all()


# ---- guard bands: no kernel writes (or read-modify-writes) outside its output -----------------------------------
def _banded(shape, fill=None):
    """an output tensor carved out of a larger buffer whose margins are -0.0 (a `+= 0` flips the sign bit, a store of
    anything else changes the value); returns (view, check)"""
    numel = 1
    for d in shape:
        numel *= d
    pad = 16 * 1024
    buf = torch.full((2 * pad + numel,), -0.0, device=DEV)
    view = buf[pad:pad + numel].view(*shape)
    view.fill_(0.0 if fill is None else fill)

    def check(what):
        lo, hi = buf[:pad], buf[pad + numel:]
        assert torch.signbit(lo).all() and (lo == 0).all(), f"{what}: wrote before the output"
        assert torch.signbit(hi).all() and (hi == 0).all(), f"{what}: wrote past the output"

    return view, check


@pytest.mark.parametrize("seed", range(6))
def test_outputs_stay_inside_their_buffers(K, seed):
    """Ragged shapes through every entry point that writes into a caller-provided output (multi-split Gram paths
    with slab reduction, Jacobian assembly, diagonal accumulation, predictive quadratic forms)."""
    gen = torch.Generator().manual_seed(100 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))  # noqa: E731
    f32 = lambda t: t.float().to(DEV).contiguous()  # noqa: E731
    # Gram family with K large enough for split-K + reduce, n ragged
    n, Kr = ri(1, 300), ri(200, 5000)
    out, chk = _banded((n, n))
    K.gram_tn(f32(rnd(Kr, n, seed=seed)), 0.3, out)
    _sync(); chk("gram_tn")
    n, L, nb = ri(1, 200), ri(1, 70), ri(1, 40)
    out, chk = _banded((n, n))
    K.gram_nt(f32(rnd(nb, n, L, seed=seed)), 0.3, out)
    _sync(); chk("gram_nt")
    Cin, H, W, B = ri(1, 40), ri(2, 20), ri(2, 20), ri(1, 6)
    out, chk = _banded((9 * Cin, 9 * Cin))
    K.gram_conv(f32(rnd(B, Cin, H, W, seed=seed)), 3, 1, 1, 1, 0.3, out)
    _sync(); chk("gram_conv 3x3")
    k, s_, p_ = ri(1, 4), ri(1, 3), ri(0, 2)
    if H + 2 * p_ >= k and W + 2 * p_ >= k:
        out, chk = _banded((k * k * Cin, k * k * Cin))
        K.gram_conv(f32(rnd(B, Cin, H, W, seed=seed + 1)), k, s_, p_, 1, 0.3, out, upper_only=True, native=True)
        _sync(); chk("gram_conv general, fused")
    # Jacobian assembly / diagonal
    Bq, C, Di, Do = ri(1, 9), ri(1, 7), ri(1, 70), ri(1, 40)
    P = Do * Di + Do + 5
    Js, chk = _banded((Bq, C, P))
    K.jac_linear(f32(rnd(Bq, Di, seed=seed)), f32(rnd(C, Bq, Do, seed=seed + 1)), Js, 3, 3 + Do * Di)
    _sync(); chk("jac_linear")
    hw, chk = _banded((Do * Di,))
    hb, chk2 = _banded((Do,))
    K.diag_ggn_linear(f32(rnd(Bq, Di, seed=seed)), f32(rnd(C, Bq, Do, seed=seed + 1)), 0.5, hw, hb)
    _sync(); chk("diag_ggn_linear w"); chk2("diag_ggn_linear b")
    Co, kk = ri(1, 9), ri(1, 3)
    Hc, Wc = ri(kk, 9), ri(kk, 9)
    oh, ow = Hc - kk + 1, Wc - kk + 1
    width = Co * Cin * kk * kk
    Jc, chk = _banded((Bq, C, width + Co))
    K.jac_conv(f32(rnd(Bq, Cin, Hc, Wc, seed=seed)), f32(rnd(C, Bq, Co, oh, ow, seed=seed + 2)), kk, 1, 0, 1, Jc, 0, width)
    _sync(); chk("jac_conv")
    h, chk = _banded((width,))
    K.sq_colsum(Jc, 0, width, 1.0, h)
    _sync(); chk("sq_colsum")
    # dense last-layer GGN and predictive quadratic forms
    D = ri(1, 40)
    Pll = C * D + C
    Hm, chk = _banded((Pll, Pll))
    probs = torch.softmax(rnd(Bq, C, seed=seed), -1) if C > 1 else None
    K.ll_ggn_full(f32(rnd(Bq, D, seed=seed)), None if probs is None else f32(probs), True, 1.0, Hm)
    _sync(); chk("ll_ggn_full")
    fv, chk = _banded((Bq, C, C))
    K.kron_quadform_linear(f32(rnd(C, Bq, Do, seed=seed)), f32(rnd(Bq, Di, seed=seed + 1)), f32(rnd(Do, seed=2).abs()),
                           f32(rnd(Di, seed=3).abs()), f32(torch.tensor([0.5], dtype=torch.float64)), fv)
    _sync(); chk("kron_quadform_linear")
    fv, chk = _banded((Bq, C, C))
    K.diag_quadform_linear(f32(rnd(Bq, Di, seed=seed)), f32(rnd(C, Bq, Do, seed=seed + 1)), f32(rnd(Do * Di, seed=4).abs()),
                           f32(rnd(Do, seed=5).abs()), fv)
    _sync(); chk("diag_quadform_linear")


@pytest.mark.parametrize("B,Cin,H,W", [(3, 4, 4, 4), (5, 8, 2, 3), (2, 64, 4, 4), (4, 7, 1, 1), (3, 5, 3, 4), (2, 130, 4, 4)])
def test_conv3x3_pixel_pair_form(K, B, Cin, H, W):
    """Small-map 3x3/s1/p1 A factor: pixel-pair Gram accumulated over TWO minibatches, assembled once ==
    the patch Gram of both minibatches (implicit im2col) == unfold."""
    x1, x2 = rnd(B, Cin, H, W, seed=B + Cin), rnd(B, Cin, H, W, seed=B + Cin + 1)
    n, npix = 9 * Cin, H * W * Cin
    want = torch.zeros(n, n, dtype=torch.float64)
    for x in (x1, x2):
        EMU.gram_conv(x, 3, 1, 1, 1, 0.5, want)
    Cp = torch.zeros(npix, npix, device=DEV)
    for x in (x1, x2):
        K.pixgram_accumulate(x.float().to(DEV), 0.5, Cp)
    nat = K.pixgram_assemble(Cp, H, W, Cin, 1.0, torch.zeros(n, n, device=DEV))
    got = K.permute_native_to_unfold(nat, Cin, 9, torch.zeros(n, n, device=DEV))
    assert_close(got, want, what="pixel-pair form")


@pytest.mark.parametrize("B,Cin,H,W", [(3, 64, 4, 4), (2, 64, 5, 7), (5, 128, 3, 3), (2, 256, 2, 5), (3, 64, 1, 1), (2, 128, 8, 8),
                                        (130, 64, 3, 2)])
def test_conv3x3_banded_pixel_pair_form(K, B, Cin, H, W):
    """Banded pixel-pair blocks accumulated over TWO minibatches, assembled once == patch Gram of both."""
    if DEV == "cpu":
        Cin = 8  # the emulation mirrors the structure at a small width
    x1, x2 = rnd(B, Cin, H, W, seed=B + Cin + H), rnd(B, Cin, H, W, seed=B + Cin + H + 1)
    n = 9 * Cin
    want = torch.zeros(n, n, dtype=torch.float64)
    for x in (x1, x2):
        EMU.gram_conv(x, 3, 1, 1, 1, 0.5, want)
    plan = K.pixpair_plan(H, W, Cin, torch.device(DEV))
    assert plan is not None
    blocks = torch.zeros(plan[0] * Cin * Cin, device=DEV)
    for x in (x1, x2):
        K.pixpair_accumulate(x.float().to(DEV), 0.5, blocks, plan)
    nat = K.pixpair_assemble(blocks, plan, H, W, Cin, 1.0, torch.zeros(n, n, device=DEV))
    got = K.permute_native_to_unfold(nat, Cin, 9, torch.zeros(n, n, device=DEV))
    assert_close(got, want, what="banded pixel-pair form")


@pytest.mark.parametrize("B,Cin,H,W", [(3, 64, 4, 4), (2, 64, 5, 7), (5, 128, 3, 3), (2, 256, 2, 5), (3, 64, 1, 1), (40, 128, 8, 8),
                                        (130, 64, 3, 2), (72, 64, 6, 6)])
def test_conv3x3_banded_pixel_pair_form_on_split_tensors(K, B, Cin, H, W):
    """lk_conv3x3_pixpair_accumulate_f16x2: the same blocks from the split images (fp16 MFMAs), two stacked launches of
    different scales, == patch Gram of both (fp64)."""
    if DEV == "cpu":
        Cin = 8
    x1, x2 = rnd(B, Cin, H, W, seed=B + Cin + H), 7.5 * rnd(B, Cin, H, W, seed=B + Cin + H + 1)
    n = 9 * Cin
    want = torch.zeros(n, n, dtype=torch.float64)
    for x in (x1, x2):
        EMU.gram_conv(x, 3, 1, 1, 1, 0.5, want)
    plan = K.pixpair_plan(H, W, Cin, torch.device(DEV))
    assert plan is not None
    blocks = torch.zeros(plan[0] * Cin * Cin, device=DEV)
    for x in (x1, x2):
        xs = K.split_f16x2(x.float().to(DEV).permute(0, 2, 3, 1).contiguous())
        K.pixpair_accumulate_split(xs, 0.5, blocks, plan)
    nat = K.pixpair_assemble(blocks, plan, H, W, Cin, 1.0, torch.zeros(n, n, device=DEV))
    got = K.permute_native_to_unfold(nat, Cin, 9, torch.zeros(n, n, device=DEV))
    assert_close(got, want, 1e-5, what="banded pixel-pair form, split tensors")


@pytest.mark.parametrize("B,Cin,H,W", [(3, 64, 4, 4), (2, 64, 5, 7), (2, 128, 3, 3), (2, 64, 32, 32), (3, 64, 1, 1)])
def test_conv3x3_banded_pixel_pair_form_two_accumulator_sets(K, B, Cin, H, W):
    """lk_conv3x3_pixpair_assemble2_f32: the blocks of the two lanes of a fit are summed inside the assembly (maps whose
    pixel count is not a multiple of the kernel's eight-deep request window included) == patch Gram of both minibatches,
    and == adding the sets first."""
    if DEV == "cpu":
        Cin = 8
    x1, x2 = rnd(B, Cin, H, W, seed=B + Cin + H), rnd(B, Cin, H, W, seed=B + Cin + H + 1)
    n = 9 * Cin
    want = torch.zeros(n, n, dtype=torch.float64)
    for x in (x1, x2):
        EMU.gram_conv(x, 3, 1, 1, 1, 0.5, want)
    plan = K.pixpair_plan(H, W, Cin, torch.device(DEV))
    sets = []
    for x in (x1, x2):
        blocks = torch.zeros(plan[0] * Cin * Cin, device=DEV)
        K.pixpair_accumulate(x.float().to(DEV), 0.5, blocks, plan)
        sets.append(blocks)
    nat = K.pixpair_assemble(sets[0], plan, H, W, Cin, 1.0, torch.zeros(n, n, device=DEV), blocks2=sets[1])
    got = K.permute_native_to_unfold(nat, Cin, 9, torch.zeros(n, n, device=DEV))
    assert_close(got, want, what="banded pixel-pair form, two accumulator sets")
    nat1 = K.pixpair_assemble(sets[0] + sets[1], plan, H, W, Cin, 1.0, torch.zeros(n, n, device=DEV))
    assert torch.equal(nat, nat1)  # the same additions in the same order
    # upper_only: the upper triangle is the same, bit for bit; nothing below the block diagonal is written on the device
    natu = K.pixpair_assemble(sets[0], plan, H, W, Cin, 1.0, torch.zeros(n, n, device=DEV), blocks2=sets[1], upper_only=True)
    assert torch.equal(torch.triu(natu), torch.triu(nat))
    if DEV != "cpu":
        assert float(torch.tril(natu, -Cin).abs().max()) == 0.0


def test_finalize_factors_is_symmetrize_scale_and_permute_in_one_launch(K):
    """lk_finalize_factors_f32 against the per-factor kernels it replaces (lk_symmetrize_f32, lk_permute_sym_f32, the
    deferred BatchNorm scale diag(s) G diag(s)); only the upper triangles of the inputs are valid."""
    g = torch.Generator().manual_seed(5)
    specs = [(1, 0, 1, False), (10, 0, 1, True), (64, 0, 1, False), (65, 0, 1, True), (130, 0, 1, False), (27, 3, 9, False),
             (576, 64, 9, False), (200, 0, 1, True), (128, 128, 1, False), (1152, 128, 9, False), (0, 0, 1, False)]
    specs = specs * 6  # more factors than one launch's descriptor table holds
    items, want = [], []
    for n, cin, kk, scaled in specs:
        M = torch.randn(n, n, generator=g)
        up = torch.triu(M) + torch.tril(torch.full((n, n), float("nan")), -1)  # the lower triangle must never be read
        full = torch.triu(M) + torch.triu(M, 1).T
        s_ = torch.rand(n, generator=g) + 0.5 if scaled else None
        src = up.to(DEV)
        if kk > 1:
            dst = torch.full((n, n), float("nan"), device=DEV)
            items.append((src, dst, None, cin, kk))
            want.append((dst, EMU.permute_native_to_unfold(full, cin, kk, torch.empty(n, n))))
        else:
            items.append((src, None, s_.to(DEV) if scaled else None, 0, 1))
            want.append((src, full * (s_.reshape(-1, 1) * s_.reshape(1, -1)) if scaled else full))
    K.finalize_factors(items)
    for (got, ref), (n, cin, kk, scaled) in zip(want, specs):
        assert got.shape == ref.shape
        if n:
            assert torch.equal(got.cpu(), ref) if not scaled else torch.allclose(got.cpu(), ref, rtol=1e-6, atol=0), (n, cin, kk, scaled)


@pytest.mark.parametrize("shape", [(4, 16, 8, 8), (3, 7, 5, 3), (5, 6), (2, 64, 32, 32), (3, 5, 9)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("with_addend", [False, True])
def test_bn_act_forward(K, shape, relu, with_addend):
    """eval-mode BatchNorm as an affine map (+ ReLU and its mask) == torch's batch_norm (+ relu) to fp32 rounding"""
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=gen, dtype=torch.float64)
    C = shape[1]
    w, b = torch.randn(C, generator=gen, dtype=torch.float64), torch.randn(C, generator=gen, dtype=torch.float64)
    rm, rv = torch.randn(C, generator=gen, dtype=torch.float64), torch.rand(C, generator=gen, dtype=torch.float64) + 0.5
    want = torch.nn.functional.batch_norm(x, rm, rv, w, b, False, 0.1, 1e-5)
    add = torch.randn(*shape, generator=gen, dtype=torch.float64) if with_addend else None
    if add is not None:
        want = want + add
    if relu:
        want = want.clamp_min(0)
    scale = w * torch.rsqrt(rv + 1e-5)
    shift = b - rm * scale
    y, mask = K.bn_act_forward(x.float().to(DEV).contiguous(), scale.float().to(DEV), shift.float().to(DEV), relu,
                               None if add is None else add.float().to(DEV).contiguous())
    _sync()
    assert_close(y, want, 2e-6, "bn_act_forward")
    if relu:
        assert mask.dtype == torch.bool and torch.equal(mask.cpu(), y.cpu() > 0)
    else:
        assert mask is None


@pytest.mark.parametrize("nb,n,L", [(40, 64, 256), (30, 128, 64), (24, 200, 16), (9, 512, 16), (6, 16, 100)])
def test_gram_nt_persistent_slabs(K, nb, n, L):
    """Split-K partial tiles accumulated across launches (LK_GRAM_SLABS_PERSIST) and reduced once == the same launches
    with their own reductions (the sum over minibatches is linear)."""
    xs = [rnd(nb, n, L, seed=nb + n + L + i) for i in range(3)]
    want = torch.zeros(n, n, dtype=torch.float64)
    for x in xs:
        EMU.gram_nt(x, 0.7, want)
    slabs = torch.zeros(max(K.gram_nt_slab_bytes(nb, n, L), 1), dtype=torch.uint8, device=DEV)
    out = torch.zeros(n, n, device=DEV)
    for x in xs:
        K.gram_nt(x.float().to(DEV), 0.7, out, upper_only=True, persist=slabs)
    _sync()
    assert (out == 0).all(), "persistent mode must not touch C"
    K.gram_slabs_reduce(slabs, n, L, 0.7, out, upper_only=True)
    if DEV != "cpu":
        K.symmetrize(out)
    _sync()
    got = out.double().cpu()
    assert_close(torch.triu(got), torch.triu(want), what="persistent slabs")
