"""The arithmetic claim behind the split-operand GEMM (cnmf_amd/csrc/kernels_gemm3.hip.h), checked on
the CPU with a numpy emulation of the device's bf16 round-to-nearest-even split:

  x = h + m + l  exactly for float32 x   (3 x 8 significand bits >= 24; the residuals are exact), and
  a*b  ~=  ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)   drops only terms below 2^-25 |a*b|,
  i.e. the six-term sum is at least as accurate as the correctly rounded float32 product.
"""
import numpy as np


def bf16_rne(x):
    """float32 -> bfloat16 (kept as float32 with the low 16 bits cleared), round to nearest even --
    the bit trick of bf16_rne() in kernels_gemm3.hip.h."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    l = bf16_rne(r2)
    return h, m, l


def _samples(n, seed):
    rs = np.random.RandomState(seed)
    x = (rs.standard_normal(n) * np.exp(8 * rs.standard_normal(n))).astype(np.float32)   # ~ 1e-10 .. 1e10
    x[: n // 10] = np.abs(x[: n // 10])
    x[n // 10: n // 5] = rs.randint(0, 5000, size=n // 5 - n // 10).astype(np.float32)   # counts
    x[-1], x[-2], x[-3] = 0.0, 1.0, np.float32(1.0) + np.finfo(np.float32).eps
    return x


def test_three_planes_represent_float32_exactly():
    x = _samples(200000, 0)
    h, m, l = split3(x)
    s = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    assert np.array_equal(s, x.astype(np.float64))
    # each plane really is a bf16 number, and the planes shrink by ~2^-8 each
    for p in (h, m, l):
        assert not (p.view(np.uint32) & 0xFFFF).any()
    nz = x != 0
    assert (np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all()
    assert (np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()


def test_six_partial_products_are_float32_accurate():
    a, b = _samples(200000, 1), _samples(200000, 2)[::-1].copy()
    ah, am, al = (p.astype(np.float64) for p in split3(a))
    bh, bm, bl = (p.astype(np.float64) for p in split3(b))
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = ah * bh + (ah * bm + am * bh) + (ah * bl + am * bm + al * bh)
    dropped = am * bl + al * bm + al * bl
    nz = exact != 0
    rel = np.abs(six - exact)[nz] / np.abs(exact)[nz]
    assert np.allclose(six + dropped, exact, rtol=1e-15, atol=0)
    assert rel.max() <= 2.0 ** -24                       # below half an ulp of the float32 product
    f32_product = (a * b).astype(np.float64)             # what the exact-f32 pipe would feed its accumulator
    rel32 = np.abs(f32_product - exact)[nz] / np.abs(exact)[nz]
    assert rel.max() <= max(rel32.max(), 2.0 ** -24)
    # five terms would NOT do: without am*bm the error is ~2^-17
    five = ah * bh + (ah * bm + am * bh) + (ah * bl + al * bh)
    assert (np.abs(five - exact)[nz] / np.abs(exact)[nz]).max() > 2.0 ** -20


def test_count_planes_are_exact():
    """The count path (kernels_counts.hip.h): every integer n <= 65 535 is lo + hi256 with lo <= 256 and
    hi256 a multiple of 256 <= 65 280 -- both exactly representable in bfloat16 -- so (ah + am + al) * n is
    formed from exact partial products."""
    n = np.arange(0, 65536, dtype=np.float32)
    hi = np.where(n <= 256, 0.0, np.floor(n / 256.0) * 256.0).astype(np.float32)
    lo = (n - hi).astype(np.float32)
    assert np.array_equal(lo + hi, n)
    assert lo.max() <= 256 and hi.max() == 65280
    for p in (lo, hi):
        assert np.array_equal(bf16_rne(p), p)                    # exact in bf16
    # and the products with a three-plane operand are exact in float32 arithmetic of the planes
    a = _samples(4096, 5)
    ah, am, al = (q.astype(np.float64) for q in split3(a))
    for c in (1.0, 3.0, 255.0, 256.0):
        exact = a.astype(np.float64) * c
        assert np.array_equal(ah * c + am * c + al * c, exact)
        for q in (ah, am, al):                                   # each partial product fits a float32
            assert np.array_equal((q * c).astype(np.float32).astype(np.float64), q * c)
