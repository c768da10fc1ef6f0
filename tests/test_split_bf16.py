"""The numerical claim behind the split-bf16 products (MODE_NTB in lk_gram.hip, qc_tile_gemm_b6 in lk_quadconv.hip):
an fp32 value splits EXACTLY into three bf16 pieces by truncation, and the six retained piece products reproduce the
fp32 product to a few 2^-24 — restated in numpy bit operations (no GPU needed; the kernels themselves are checked
against fp64 in tests/test_gpu_kernels.py)."""
import numpy as np


def split3(x):
    xi = x.view(np.uint32)
    h = (xi & np.uint32(0xFFFF0000)).view(np.float32)
    r1 = x - h
    m = (r1.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    lo = r1 - m
    return h, m, lo


def _values(n, seed):
    rng = np.random.default_rng(seed)
    scale = np.float32(10.0) ** rng.integers(-8, 8, n).astype(np.float32)
    return (rng.standard_normal(n).astype(np.float32) * scale).astype(np.float32)


def test_three_bf16_pieces_represent_an_fp32_exactly():
    x = _values(200_000, 0)
    h, m, lo = split3(x)
    for piece in (h, m, lo):  # every piece is a bf16: its low 16 bits are zero
        assert not np.any(piece.view(np.uint32) & np.uint32(0xFFFF))
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))


def test_six_piece_products_are_fp32_accurate():
    x, y = _values(200_000, 1), _values(200_000, 2)
    (h, m, lo), (h2, m2, l2) = split3(x), split3(y)
    f = np.float64
    kept = h.astype(f) * h2 + h.astype(f) * m2 + m.astype(f) * h2 + m.astype(f) * m2 + h.astype(f) * l2 + lo.astype(f) * h2
    exact = x.astype(f) * y.astype(f)
    rel = np.abs(kept - exact) / np.abs(exact)
    assert rel.max() < 8 * 2.0**-24  # the three dropped terms: <= (2 * 2^-8 * 2^-16 + 2^-32) of the product, ~7.5 ulp worst case
    # and a K-term dot product is as good as an fp32 one
    a, b = _values(4096, 3) / 1e3, _values(4096, 4) / 1e3
    (h, m, lo), (h2, m2, l2) = split3(a), split3(b)
    acc = np.float32(0)
    for terms in ((lo, h2), (h, l2), (m, m2), (m, h2), (h, m2), (h, h2)):  # small terms first, fp32 accumulation
        acc = np.float32(acc + np.sum(terms[0].astype(f) * terms[1].astype(f)).astype(np.float32))
    ref = np.dot(a.astype(f), b.astype(f))
    fp32 = np.dot(a, b)
    scale = np.sqrt(np.sum((a.astype(f) * b.astype(f)) ** 2))
    assert abs(acc - ref) <= max(4 * abs(fp32 - ref), 1e-6 * scale)
