"""The arithmetic of the split-fp32 mode (csrc/conv_split.hip), stated in numpy (oracle/lp.py: split_bf16x3) -- CPU only.

  * an fp32 value IS the sum of its three bf16 pieces, bit for bit, over the whole exponent range the nets use;
  * every piece is a bfloat16 value, piece i is bounded by 2^-8i of the value (to rounding);
  * the six products with i + j <= 2, each exact in float32, sum (in float64) to the exact product within 2^-22 of it: the
    three dropped ones are the size of float32's own rounding of the product.
"""
import numpy as np

from oracle import lp as LP


def _is_bf16(v):
    return np.array_equal(v.view(np.uint32) & 0xffff, np.zeros(v.shape, np.uint32))


def test_three_pieces_sum_to_the_value_exactly():
    rng = np.random.RandomState(0)
    a = (rng.randn(200000) * np.exp2(rng.randint(-80, 80, size=200000))).astype(np.float32)
    a[:6] = [0.0, -0.0, 1.0, 3.0e38, -1.1754944e-38 * 4096, np.float32(1) + np.float32(2.0 ** -23)]
    p0, p1, p2 = LP.split_bf16x3(a)
    assert all(_is_bf16(p) for p in (p0, p1, p2))
    total = (p0.astype(np.float64) + p1.astype(np.float64) + p2.astype(np.float64)).astype(np.float32)
    assert np.array_equal(total, a)
    nz = a != 0
    assert np.all(np.abs(p1[nz]) <= np.abs(a[nz]) * 2.0 ** -8 * 1.01)
    assert np.all(np.abs(p2[nz]) <= np.abs(a[nz]) * 2.0 ** -16 * 1.01)


def test_six_products_are_the_product_to_fp32_accuracy():
    rng = np.random.RandomState(1)
    a = (rng.randn(100000) * np.exp(rng.randn(100000) * 3)).astype(np.float32)
    b = (rng.randn(100000) * np.exp(rng.randn(100000) * 3)).astype(np.float32)
    pa, pb = LP.split_bf16x3(a), LP.split_bf16x3(b)
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = np.zeros_like(exact)
    for i, j in LP.split_product_terms():
        prod32 = pa[i] * pb[j]                                   # float32 product of two bf16 values ...
        assert np.array_equal(prod32.astype(np.float64), pa[i].astype(np.float64) * pb[j].astype(np.float64))      # ... is exact
        six += prod32.astype(np.float64)
    assert len(LP.split_product_terms()) == 6
    err = np.abs(six - exact) / np.abs(exact)
    assert err.max() < 2.0 ** -22, err.max()
    # float32's own rounding of the same product, for scale
    err32 = np.abs((a * b).astype(np.float64) - exact) / np.abs(exact)
    assert err.max() < 8 * err32.max()
    # and the bf16-rounded product the reduced-precision modes compute is 2^13 times further away
    lp_err = np.abs(LP.round_bf16(a).astype(np.float64) * LP.round_bf16(b).astype(np.float64) - exact) / np.abs(exact)
    assert np.median(lp_err) > 1000 * np.median(err[err > 0])


def test_two_piece_operands_and_three_products():
    """'bf16x2' (BASELINE config 4's arithmetic): two pieces carry >= 16 significant bits, the three kept products reproduce
    a dot product to ~2^-16 per term -- three orders of magnitude below plain bf16 operands (2^-9)"""
    rng = np.random.RandomState(3)
    a = (rng.randn(4096) * np.exp2(rng.randint(-30, 30, 4096))).astype(np.float32)
    p0, p1 = LP.split_bf16x2(a)
    assert np.array_equal(p0, LP.round_bf16(a))
    assert np.all(np.abs((a.astype(np.float64) - p0 - p1)) <= np.abs(a) * 2.0 ** -16)
    x, w = rng.randn(64, 512).astype(np.float32), rng.randn(512, 32).astype(np.float32)
    px, pw = LP.split_bf16x2(x), LP.split_bf16x2(w)
    exact = x.astype(np.float64) @ w.astype(np.float64)
    got = sum(px[i].astype(np.float64) @ pw[j].astype(np.float64) for i, j in LP.split2_product_terms())
    one = LP.round_bf16(x).astype(np.float64) @ LP.round_bf16(w).astype(np.float64)
    e2 = np.linalg.norm(got - exact) / np.linalg.norm(exact)
    e1 = np.linalg.norm(one - exact) / np.linalg.norm(exact)
    assert e2 < 1.5e-5 and e1 > 100 * e2, (e2, e1)
    assert len(LP.split2_product_terms()) == 3
