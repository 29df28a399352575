"""The arithmetic claim behind the dense kernels (csrc/gat_gemm_kernels.hip: split_pair / mfma_bf16), checked in numpy
on the CPU: an fp32 value is EXACTLY the sum of three bf16 pieces cut by rounding to nearest, and a product assembled
from the six piece products with (piece index of a) + (piece index of b) <= 2 differs from the exact product by
< 2^-24 |a b| — less than the rounding of an fp32 multiply — in the worst case (2^-28 on average).
The kernels themselves are held to the same bound on the GPU (tests/test_dense_precision.py)."""
import numpy as np

MASK = np.uint32(0xFFFF0000)


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), as v_cvt_pk_bf16_f32 does for finite values; returned as fp32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split3(x):
    """split_pair's arithmetic, one value at a time: hi = bf16(x); mid = bf16(x - hi); lo = bf16((x - hi) - mid)."""
    x = np.asarray(x, np.float32)
    hi = bf16_rne(x)
    r = (x - hi).astype(np.float32)
    mid = bf16_rne(r)
    q = (r - mid).astype(np.float32)
    lo = bf16_rne(q)
    return hi, mid, lo


def _samples():
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(200000) * np.exp2(rng.integers(-60, 60, 200000))).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, 3.4028235e38, -3.4028235e38,
                     1.17549435e-38, 2.0 ** -100, 16777215.0, 0.1, 1.0 / 3.0], np.float32)
    return np.concatenate([x, edge])


def test_three_rounded_bf16_pieces_are_the_value_exactly():
    x = _samples()
    with np.errstate(invalid="ignore", over="ignore"):
        hi, mid, lo = split3(x)
    for p in (hi, mid, lo):                                       # every piece is a bf16 value: low 16 bits clear
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))
    with np.errstate(invalid="ignore"):
        s = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    fin = np.isfinite(hi)
    assert np.array_equal(s[fin], x.astype(np.float64)[fin]) and fin.sum() >= len(x) - 2
    # the remainders are exact in fp32 (that is why the pieces can be cut one after the other)
    fin = np.isfinite(hi)                                          # |x| within half a bf16 ulp of FLT_MAX rounds to inf
    assert np.array_equal((x - hi).astype(np.float64)[fin], (x.astype(np.float64) - hi.astype(np.float64))[fin])
    # sizes: |mid| <= 2^-9 |hi|-ish, |lo| <= 2^-17: rounding halves what truncation would leave
    nz = (x != 0) & fin
    assert np.all(np.abs(mid[nz]) <= np.abs(x[nz]) * 2.0 ** -8)
    assert np.all(np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


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
    # dropped: mid*lo + lo*mid + lo*lo <= 2 * 2^-9 * 2^-17 (1 + ...) : below the rounding of ONE fp32 multiply (2^-24)
    assert rel.max() < 2.0 ** -24, rel.max()
    assert rel.mean() < 2.0 ** -27, rel.mean()
    # with the two 2^-26-class terms as well (-DGAT_X3_EIGHT_TERMS): < 2^-33
    eight = kept + pa[1].astype(np.float64) * pb[2] + pa[2].astype(np.float64) * pb[1]
    assert (np.abs(eight - exact) / np.abs(exact)).max() < 2.0 ** -33
    # three terms (hi*hi + hi*mid + mid*hi) would NOT be: that is the 2^-16-class "bf16x3" the kernels do not use
    three = (pa[0].astype(np.float64) * pb[0] + pa[0].astype(np.float64) * pb[1] + pa[1].astype(np.float64) * pb[0])
    assert (np.abs(three - exact) / np.abs(exact)).max() > 2.0 ** -19
