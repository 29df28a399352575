"""The arithmetic claim behind the dense kernels (csrc/gat_gemm_kernels.hip: split_pair / mfma_bf16), checked in numpy
on the CPU: an fp32 value is EXACTLY the sum of three bf16 pieces cut by truncation, and a product assembled from the
six piece products with (piece index of a) + (piece index of b) <= 2 differs from the exact product by < 2^-21 |a b|
(worst case; about one fp32 rounding, 2^-24, on average).
The kernels themselves are held to the same bound on the GPU (tests/test_dense_precision.py)."""
import numpy as np

MASK = np.uint32(0xFFFF0000)


def split3(x):
    """split_pair's arithmetic, one value at a time: hi = x & mask; mid = (x - hi) & mask; lo = (x - hi) - mid (all fp32)."""
    x = np.asarray(x, np.float32)
    hi = (x.view(np.uint32) & MASK).view(np.float32)
    r = (x - hi).astype(np.float32)
    mid = (r.view(np.uint32) & MASK).view(np.float32)
    q = (r - mid).astype(np.float32)
    lo = (q.view(np.uint32) & MASK).view(np.float32)             # what the kernel packs: the upper 16 bits of the remainder
    return hi, mid, lo


def _samples():
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(200000) * np.exp2(rng.integers(-60, 60, 200000))).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, 3.4028235e38, -3.4028235e38,
                     1.17549435e-38, 2.0 ** -100, 16777215.0, 0.1, 1.0 / 3.0], np.float32)
    return np.concatenate([x, edge])


def test_three_truncated_bf16_pieces_are_the_value_exactly():
    x = _samples()
    hi, mid, lo = split3(x)
    for p in (hi, mid, lo):                                       # every piece is a bf16 value: low 16 bits clear
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))
    s = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(s, x.astype(np.float64))
    # the remainders are exact in fp32 (that is why the pieces can be cut one after the other)
    assert np.array_equal((x - hi).astype(np.float64), x.astype(np.float64) - hi.astype(np.float64))
    # sizes: |mid| < 2^-7 |x|, |lo| < 2^-15 |x| (truncation keeps 8 significant bits per piece)
    nz = x != 0
    assert np.all(np.abs(mid[nz]) < np.abs(x[nz]) * 2.0 ** -7)
    assert np.all(np.abs(lo[nz]) < np.abs(x[nz]) * 2.0 ** -15)


def test_six_piece_products_are_an_fp32_accurate_product():
    rng = np.random.default_rng(6)
    a = (rng.standard_normal(300000) * np.exp2(rng.integers(-20, 20, 300000))).astype(np.float32)
    b = (rng.standard_normal(300000) * np.exp2(rng.integers(-20, 20, 300000))).astype(np.float32)
    pa, pb = split3(a), split3(b)
    kept = np.zeros(len(a), np.float64)
    for i in range(3):
        for j in range(3):
            if i + j <= 2:
                prod = pa[i].astype(np.float64) * pb[j].astype(np.float64)
                # a bf16 x bf16 product has <= 16 significant bits: exact in fp32
                assert np.array_equal(prod, (pa[i] * pb[j]).astype(np.float64))
                kept += prod
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(kept - exact) / np.abs(exact)
    # dropped: mid*lo + lo*mid + lo*lo < 2 * 2^-7 * 2^-15 + 2^-30; on average the size of ONE fp32 rounding (2^-24)
    assert rel.max() < 2.0 ** -21, rel.max()
    assert rel.mean() < 2.0 ** -24, rel.mean()
    # with the two 2^-22-class terms as well (-DGAT_X3_EIGHT_TERMS): < 2^-29
    eight = kept + pa[1].astype(np.float64) * pb[2] + pa[2].astype(np.float64) * pb[1]
    assert (np.abs(eight - exact) / np.abs(exact)).max() < 2.0 ** -29
    # three terms (hi*hi + hi*mid + mid*hi) would NOT be: that is the 2^-16-class "bf16x3" the kernels do not use
    three = (pa[0].astype(np.float64) * pb[0] + pa[0].astype(np.float64) * pb[1] + pa[1].astype(np.float64) * pb[0])
    assert (np.abs(three - exact) / np.abs(exact)).max() > 2.0 ** -17
