"""The reference's double-precision sums are SEQUENTIAL (RMSNorm sum of squares, ggml.c:11874-11879; softmax denominator, :2619-2671); the GPU
reduces them as a fixed tree.  Both round to f32 at once (mean, 1 / sum), so the order shows only when the sum sits next to a rounding
boundary of that f32.  Round 3 closed the hole: f32_rounding_safe (bamd_device.h) bounds the reordering error, and when it cannot rule a
difference out one lane redoes the sum in the reference's order.  Pinned here on a CONSTRUCTED worst case (tests/golden/f64_order_kat.npz,
tools/f64_order_search.py) on which the two orders do differ — the means by one f32 ulp, 13 bytes of the Q8_K activations with them — and on
the guard's logic restated in numpy (it must fire on every vector on which the orders differ)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "f64_order_kat.npz"))


def test_orders_differ_on_the_constructed_vector(kat, po):
    import f64_order_search as fs
    t = fs.terms(kat["x"])
    m_seq, m_tree = fs.mean32(fs.sum_seq(t)), fs.mean32(fs.sum_tree(t))
    assert m_seq.view(np.uint32) == kat["mean_seq"].view(np.uint32) and m_tree.view(np.uint32) == kat["mean_tree"].view(np.uint32)
    assert abs(int(m_seq.view(np.uint32)) - int(m_tree.view(np.uint32))) == 1
    # the oracle follows the reference: sequential
    assert np.array_equal(po.quantize_q8_K(po.rms_norm(kat["x"], float(kat["eps"]))), kat["q8k_seq"])
    assert not np.array_equal(kat["q8k_seq"], kat["q8k_tree"])


def test_guard_fires_on_the_constructed_vector_and_gives_the_sequential_mean(kat):
    import f64_order_search as fs
    m, fired = fs.guarded_mean32(fs.terms(kat["x"]))
    assert fired and m.view(np.uint32) == kat["mean_seq"].view(np.uint32)


def test_guard_never_misses(kat):
    """property: whenever the two orders give different f32 means the guard has fired — on random vectors and on a walk of the constructed
    vector's last element across the rounding boundary (where the orders disagree on a whole interval of steps)"""
    import f64_order_search as fs
    rng = np.random.default_rng(11)
    fired = 0
    for _ in range(1500):
        x = (rng.standard_normal(fs.K) * np.exp(rng.standard_normal(fs.K) * 2.0)).astype(np.float32)
        t = fs.terms(x)
        m, f = fs.guarded_mean32(t)
        fired += f
        assert m.view(np.uint32) == fs.mean32(fs.sum_seq(t)).view(np.uint32)
    assert fired < 15                                                   # ~1e-5 .. 1e-4 of natural reductions: the slow path stays rare
    x = kat["x"].copy()
    v0 = x[-1]
    differ = 0
    for k in range(-300, 300):
        x[-1] = np.float32(v0 * np.float32(1.0 + k * 2.0 ** -9))
        t = fs.terms(x)
        m, f = fs.guarded_mean32(t)
        a, b = fs.mean32(fs.sum_seq(t)), fs.mean32(fs.sum_tree(t))
        differ += int(a.view(np.uint32) != b.view(np.uint32))
        assert m.view(np.uint32) == a.view(np.uint32)
    assert differ >= 1


def test_random_vectors_agree():
    import f64_order_search as fs
    assert fs.random_trials(2000, np.random.default_rng(5)) == 0


@pytest.mark.gpu
def test_gpu_equals_the_reference_on_the_constructed_vector(kat, bamd):
    """the worst case through the HIP path: the guard sends it down the sequential order, so the bytes are the reference's"""
    got = bamd.op_quantize_q8_K(kat["x"], norm_w=np.ones(kat["x"].size, np.float32), eps=float(kat["eps"]))
    assert np.array_equal(got, kat["q8k_seq"])
    assert not np.array_equal(got, kat["q8k_tree"])


# ---- the softmax denominator (ggml.c:2619-2671): a constructed row of 64 scores (tests/golden/f64_softmax_kat.npz, tools/f64_softmax_search.py)
#      whose reciprocal sits on an f32 rounding boundary — the sequential order and a tree order of the same eight partial sums round to
#      DIFFERENT f32 (one ulp apart), so every probability of the row depends on the order ----
@pytest.fixture(scope="module")
def skat():
    return np.load(os.path.join(GOLDEN, "f64_softmax_kat.npz"))


def _softmax_row_orders(po, scores):
    import f64_softmax_search as ss
    e = ss.expf_table(po, scores * np.float32(0.125))
    return e, ss.denominators(e)


def test_softmax_orders_differ_on_the_constructed_row(skat, po):
    import f64_softmax_search as ss
    e, den = _softmax_row_orders(po, skat["scores"])
    inv_seq, inv_tree = np.float32(1.0 / den["seq"]), np.float32(1.0 / den["tree"])
    assert inv_seq.view(np.uint32) == skat["inv_seq"].view(np.uint32) and inv_tree.view(np.uint32) == skat["inv_tree"].view(np.uint32)
    assert abs(int(inv_seq.view(np.uint32)) - int(inv_tree.view(np.uint32))) == 1
    # the oracle follows the reference: sequential
    p = po.soft_max(skat["scores"], None, np.float32(0.125))
    assert np.array_equal(p.view(np.uint32), (e * inv_seq).astype(np.float32).view(np.uint32))
    assert not np.array_equal(p.view(np.uint32), (e * inv_tree).astype(np.float32).view(np.uint32))
    # whatever order a kernel adds the eight partial sums in, its reciprocal lies inside the band in which f32_rounding_safe (bamd_device.h)
    # refuses the fast path: |low 29 bits - 2^28| <= 2 n + 8, n = n_kv / 8
    for t in den.values():
        assert int(ss.dist(np.array([1.0 / t]))[0]) <= ss.GUARD_ULPS


@pytest.mark.gpu
@pytest.mark.parametrize("prefill", [False, True])
@pytest.mark.parametrize("long_path", [False, True])
def test_gpu_attention_equals_the_reference_on_the_constructed_row(skat, bamd, po, prefill, long_path):
    """the row through the HIP attention (single-launch kernel and the scores | softmax + P.V path): q = e_0, K rows = score x e_0 (f16-exact),
    identity RoPE, head dimension 64 (scale 1/8): the kernel's scores ARE the constructed row, its guard must take the sequential order.
    (A build with -DBAMD_NO_F64_GUARD fails this test on the single-launch kernel — 44 of 64 outputs one ulp off — and the RMSNorm test above;
    the softmax + P.V kernel's own tree happens to agree with the sequential order on this row, its fallback runs all the same.)"""
    from test_gpu_ops import oracle_attention, assert_bits
    H = Hkv = 1; hd = 64; n_ctx = 64; pos = 63
    s = skat["scores"]
    rng = np.random.default_rng(99)
    kc = np.zeros(n_ctx * hd, np.float16); kc[0::hd] = s.astype(np.float16); kc[pos * hd] = 0    # the token's own row comes from k
    kc = kc.view(np.uint16).copy()
    vc = rng.standard_normal(hd * n_ctx).astype(np.float16).view(np.uint16).copy()
    q = np.zeros(hd, np.float32); q[0] = 1.0
    k = np.zeros(hd, np.float32); k[0] = s[pos]
    v = rng.standard_normal(hd).astype(np.float32)
    rope = np.tile(np.array([1.0, 0.0], np.float32), hd // 2)
    kc2, vc2 = kc.copy(), vc.copy()
    want, wprobs = oracle_attention(po, q, k, v, kc2, vc2, rope, H, Hkv, hd, n_ctx, pos, prefill)
    e, den = _softmax_row_orders(po, s)
    assert np.array_equal(wprobs.view(np.uint32), (e * np.float32(1.0 / den["seq"])).astype(np.float32).view(np.uint32))    # the scores are the row
    if long_path:
        got, gprobs = bamd.op_attention(q, k, v, kc, vc, rope, H, Hkv, hd, n_ctx, pos, prefill_mode=prefill, want_probs=True)
        assert_bits(gprobs[:wprobs.size], wprobs, "softmax of the constructed row")
    else:
        got = bamd.op_attention(q, k, v, kc, vc, rope, H, Hkv, hd, n_ctx, pos, prefill_mode=prefill)
    assert_bits(got, want, "attention out on the constructed row")
