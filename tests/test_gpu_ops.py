"""GPU: every HIP kernel, called through the C-ABI (include/bamd.h bamd_op_*), against the CPU oracle on the same
seeded inputs.  Integer/byte results and f32 results alike must be BIT-IDENTICAL (the kernels restate the
reference's AVX2 operation order), so every comparison is on the raw bits."""
import numpy as np
import pytest

from booster_amd.gguf import random_kquant_tensor

pytestmark = pytest.mark.gpu
TYPES = [12, 13, 14]
BB = {12: 144, 13: 176, 14: 210}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bits(a, b, what=""):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    bad = np.flatnonzero(bits(a) != bits(b))
    assert bad.size == 0, "%s: %d/%d elements differ, first at %d: %r vs %r" % (what, bad.size, a.size, bad[0], a.flat[bad[0]], b.flat[bad[0]])


@pytest.mark.parametrize("K", [256, 768, 4096, 14336])
@pytest.mark.parametrize("scale", [1e-3, 1.0, 80.0])
def test_quantize_q8_K(bamd, po, K, scale):
    rng = np.random.default_rng(K)
    x = (rng.standard_normal(K) * scale).astype(np.float32)
    if K >= 768:
        x[256:512] = 0.0                                   # all-zero block
    x[3] = -np.abs(x).max() * 2                            # negative extremum
    x[700 % K] = x[(700 % K) - 1] = np.float32(1.25)       # ties
    got = bamd.op_quantize_q8_K(x)
    want = po.quantize_q8_K(x)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("K", [512, 4096])
def test_rmsnorm_quantize(bamd, po, K):
    rng = np.random.default_rng(K + 1)
    for trial in range(4):
        x = (rng.standard_normal(K) * 10 ** rng.uniform(-2, 2)).astype(np.float32)
        w = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
        eps = 1e-5
        got = bamd.op_quantize_q8_K(x, norm_w=w, eps=eps)
        y = po.rms_norm(x, eps) * w
        assert np.array_equal(got, po.quantize_q8_K(y.astype(np.float32)))


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("K,rows", [(256, 8), (768, 40), (768, 13), (2048, 4104), (4096, 512), (4096, 510), (8192, 24), (14336, 64), (14336, 2064),
                                    (28672, 8192), (28672, 2056), (28672, 8200), (28672, 24)])   # K = 28672: the 70B ffn_down (sixteen-wave split-K with compact term buffers; mode 0 = the launcher's own choice)
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_mul_mat_vec(bamd, po, t, K, rows, mode):
    """mode 1: one wave per row-group; mode 2: split-K over the 8 waves of a workgroup (where the shape allows;
    the launcher falls back to mode 1 otherwise).  Both must give the reference's bits."""
    rng = np.random.default_rng(1000 * t + K)
    W = random_kquant_tensor(t, K, rows, rng)
    x = (rng.standard_normal(K) * 3).astype(np.float32)
    res = rng.standard_normal(rows).astype(np.float32) if rows % 16 in (8, 13, 14) else None
    got = bamd.op_mul_mat_vec(t, W, rows, K, x, residual=res, mode=mode)
    want = po.mul_mat_q(t, W, rows, K, x, nthreads=8)[0]
    if res is not None:
        want = want + res
    assert_bits(got, want, "mul_mat_vec type %d K %d mode %d" % (t, K, mode))


@pytest.mark.parametrize("t", TYPES)
def test_mul_mat_vec_kats(bamd, kats, t):
    """reference-quantised weights + the reference's own dot products (committed fixture)"""
    name = {12: "q4_K", 13: "q5_K", 14: "q6_K"}[t]
    for K in (256, 768, 4096):
        key = "%s_K%d" % (name, K)
        blocks, x, dots = kats[key + "_blocks"], kats[key + "_x"], kats[key + "_dots"]
        got = bamd.op_mul_mat_vec(t, blocks, dots.size, K, x)
        assert_bits(got, dots, key)


@pytest.mark.parametrize("t", TYPES)
def test_mul_mat_vec_norm_residual(bamd, po, t):
    K, rows = 1024, 256
    rng = np.random.default_rng(77 + t)
    W = random_kquant_tensor(t, K, rows, rng)
    x = rng.standard_normal(K).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    res = rng.standard_normal(rows).astype(np.float32)
    got = bamd.op_mul_mat_vec(t, W, rows, K, x, norm_w=w, eps=1e-5, residual=res)
    a = (po.rms_norm(x, 1e-5) * w).astype(np.float32)
    want = po.mul_mat_q(t, W, rows, K, a)[0] + res
    assert_bits(got, want, "norm+residual")


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_wo_four_row_groups(bamd, po, t, mode):
    """plain prologue + residual add, K = 8192, 8192 rows = four row-groups per workgroup on 256 CUs (the 70B wo): mode 0 takes all four in one batch"""
    K, rows = 8192, 8192
    rng = np.random.default_rng(57 + t)
    W = random_kquant_tensor(t, K, rows, rng)
    x = (rng.standard_normal(K) * 2).astype(np.float32)
    res = rng.standard_normal(rows).astype(np.float32)
    got = bamd.op_mul_mat_vec(t, W, rows, K, x, residual=res, mode=mode)
    want = po.mul_mat_q(t, W, rows, K, x, nthreads=8)[0] + res
    assert_bits(got, want, "wo, four row-groups, mode %d" % mode)


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_fused_qkv_one_type(bamd, po, t, mode):
    """RMSNorm prologue + store, K = 4096, 6144 rows = three row-groups per workgroup on 256 CUs: the fused QKV launch of a layer whose wq | wk | wv share a type
    (mode 0: all three row-groups in one batch on the mixed-type kernel's body; modes 1 / 2: one wave per row-group / two batches)"""
    K, rows = 4096, 6144
    rng = np.random.default_rng(31 + t)
    W = random_kquant_tensor(t, K, rows, rng)
    x = (rng.standard_normal(K) * 2).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    got = bamd.op_mul_mat_vec(t, W, rows, K, x, norm_w=w, eps=1e-5, mode=mode)
    a = (po.rms_norm(x, 1e-5) * w).astype(np.float32)
    want = po.mul_mat_q(t, W, rows, K, a, nthreads=8)[0]
    assert_bits(got, want, "fused qkv, one type, mode %d" % mode)


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("K,rows", [(512, 768), (4096, 1024), (4096, 14336), (8192, 28672)])   # 14336 rows on 256 CUs: seven row-group pairs per workgroup (matvec_gateup7_kernel); 28672 rows at K = 8192: fourteen (matvec_gateup14_kernel: half pairs cross waves through LDS)
def test_ffn_gate_up(bamd, po, t, K, rows):
    rng = np.random.default_rng(5 * t + K)
    Wg = random_kquant_tensor(t, K, rows, rng, amp=4.0)
    Wu = random_kquant_tensor(t, K, rows, rng, amp=4.0)
    x = (rng.standard_normal(K) * 2).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    got = bamd.op_ffn_gate_up(t, Wg, Wu, rows, K, x, norm_w=w, eps=1e-5)
    a = (po.rms_norm(x, 1e-5) * w).astype(np.float32)
    g = po.mul_mat_q(t, Wg, rows, K, a)[0]
    u = po.mul_mat_q(t, Wu, rows, K, a)[0]
    L = po.lib()
    want = np.array([L.bo_v_silu(float(v)) for v in g], np.float32) * u
    assert_bits(got, want, "ffn gate/up")


@pytest.mark.parametrize("t", TYPES)
def test_get_row(bamd, po, t):
    K, rows = 768, 24
    rng = np.random.default_rng(t)
    W = random_kquant_tensor(t, K, rows, rng)
    rb = K // 256 * BB[t]
    for row in (0, 7, 23):
        got = bamd.op_get_row(t, W, rows, K, row)
        assert_bits(got, po.dequantize(t, W[row * rb:(row + 1) * rb], K), "get_row")


def test_rope_row(bamd, po):
    ff = (1 + np.arange(64) / 16.0).astype(np.float32)
    for pos in (0, 1, 127, 2047, 8191):
        for theta in (500000.0, 10000.0):
            assert_bits(bamd.op_rope_row(pos, 128, theta), po.rope_cache(pos, 128, theta))
            assert_bits(bamd.op_rope_row(pos, 128, theta, 1.0, ff), po.rope_cache(pos, 128, theta, 1.0, ff))


def oracle_attention(po, q, k, v, kc, vc, rope, H, Hkv, hd, n_ctx, pos, prefill):
    """llm_build_kv for one token with the oracle's primitives; updates kc/vc (numpy uint16) in place."""
    import ctypes as C
    L = po.lib()
    q = q.copy(); k = k.copy()
    for h in range(H):
        L.bo_rope_apply(q[h * hd:].ctypes.data_as(C.c_void_p), rope.ctypes.data_as(C.c_void_p), hd)
    for h in range(Hkv):
        L.bo_rope_apply(k[h * hd:].ctypes.data_as(C.c_void_p), rope.ctypes.data_as(C.c_void_p), hd)
    Ekv = Hkv * hd
    kc[pos * Ekv:(pos + 1) * Ekv] = k.astype(np.float16).view(np.uint16)
    vc.reshape(Ekv, n_ctx)[:, pos] = v.astype(np.float16).view(np.uint16)
    n_kv = min(n_ctx, max(32, (pos + 1 + 31) // 32 * 32))
    mask = np.where(np.arange(n_kv) <= pos, 0.0, -np.inf).astype(np.float32)
    out = np.zeros(H * hd, np.float32)
    probs0 = None
    gq = H // Hkv
    for h in range(H):
        hk = h // gq
        qh = np.ascontiguousarray(q[h * hd:(h + 1) * hd])
        s = np.zeros(n_kv, np.float32)
        q16 = qh.astype(np.float16).view(np.uint16)
        for i in range(n_kv):
            krow = np.ascontiguousarray(kc[i * Ekv + hk * hd: i * Ekv + (hk + 1) * hd])
            if prefill:
                s[i] = L.bo_vec_dot_f16(hd, krow.ctypes.data_as(C.c_void_p), q16.ctypes.data_as(C.c_void_p))
            else:
                s[i] = L.bo_dot_f16_f32_tinyblas(krow.ctypes.data_as(C.c_void_p), qh.ctypes.data_as(C.c_void_p), hd)
        p = po.soft_max(s, mask, np.float32(1.0) / np.sqrt(np.float32(hd)))
        if h == 0:
            probs0 = p
        for d in range(hd):
            vrow = np.ascontiguousarray(vc.reshape(Ekv, n_ctx)[hk * hd + d, :n_kv])
            out[h * hd + d] = L.bo_dot_f16_f32_tinyblas(vrow.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), n_kv)
    return out, probs0


@pytest.mark.parametrize("H,Hkv,hd", [(4, 1, 128), (4, 2, 64), (8, 8, 64), (8, 1, 256)])
@pytest.mark.parametrize("prefill", [False, True])
@pytest.mark.parametrize("long_path", [False, True])
def test_attention(bamd, po, H, Hkv, hd, prefill, long_path):
    n_ctx = 256
    rng = np.random.default_rng(H * 100 + hd + prefill)
    Ekv = Hkv * hd
    for pos in (0, 5, 31, 32, 100, 255):
        kc = (rng.standard_normal(n_ctx * Ekv) * 0.7).astype(np.float16).view(np.uint16).copy()
        vc = rng.standard_normal(Ekv * n_ctx).astype(np.float16).view(np.uint16).copy()
        q = (rng.standard_normal(H * hd) * 2).astype(np.float32)
        k = rng.standard_normal(Ekv).astype(np.float32)
        v = rng.standard_normal(Ekv).astype(np.float32)
        rope = po.rope_cache(pos, hd, 500000.0)
        kc2, vc2 = kc.copy(), vc.copy()
        want, wprobs = oracle_attention(po, q, k, v, kc2, vc2, rope, H, Hkv, hd, n_ctx, pos, prefill)
        if long_path:      # three-kernel path (contexts beyond the fused kernel's LDS budget); exposes the probabilities
            got, gprobs = bamd.op_attention(q, k, v, kc, vc, rope, H, Hkv, hd, n_ctx, pos, prefill_mode=prefill, want_probs=True)
            assert_bits(gprobs[:wprobs.size], wprobs, "softmax pos %d" % pos)
        else:              # fused single-launch path
            got = bamd.op_attention(q, k, v, kc, vc, rope, H, Hkv, hd, n_ctx, pos, prefill_mode=prefill)
        assert np.array_equal(kc, kc2) and np.array_equal(vc, vc2), "KV store differs at pos %d" % pos
        assert_bits(got, want, "attention out pos %d" % pos)


@pytest.mark.parametrize("t,K,rows,T,impl", [(12, 1024, 64, 5, 0), (14, 2048, 40, 19, 0), (13, 1024, 24, 9, 0),
                                               (12, 1024, 64, 16, 0), (12, 2048, 40, 21, 0), (12, 4096, 528, 37, 0), (12, 14336, 32, 16, 0),
                                               (14, 1024, 64, 16, 0), (14, 2048, 40, 21, 0), (14, 4096, 528, 37, 0), (14, 14336, 32, 33, 0),
                                               (12, 768, 40, 21, 0), (14, 2816, 24, 17, 0), (12, 11008, 32, 16, 0), (14, 256, 16, 3, 0),
                                               (13, 1024, 64, 16, 0), (13, 2048, 40, 21, 0), (13, 4096, 528, 37, 0), (13, 8192, 48, 33, 0), (13, 768, 24, 9, 0),
                                               (12, 28672, 24, 11, 0), (14, 28672, 16, 6, 0), (13, 28672, 8, 5, 0),      # K = 28672 (the 70B ffn_down): token tiles of four
                                               (12, 28672, 24, 11, 2), (14, 28672, 16, 6, 2),
                                               # impl 2: the matrix-core kernels (64-row x 64-token workgroups, fragments built once per workgroup, load-time side tables)
                                               (12, 1024, 64, 16, 2), (12, 2048, 40, 21, 2), (12, 4096, 528, 37, 2), (12, 14336, 32, 16, 2), (12, 768, 40, 70, 2),
                                               (12, 11008, 32, 16, 2), (12, 256, 8, 1, 2), (12, 4096, 200, 129, 2),
                                               (13, 1024, 64, 16, 2), (13, 2048, 40, 21, 2), (13, 4096, 528, 37, 2), (13, 8192, 48, 33, 2), (13, 768, 24, 65, 2),
                                               (14, 1024, 64, 16, 2), (14, 2048, 40, 21, 2), (14, 4096, 528, 37, 2), (14, 14336, 32, 33, 2), (14, 2816, 24, 17, 2),
                                               (14, 256, 16, 3, 2), (14, 4096, 200, 129, 2),
                                               ])
def test_mul_mat_batch(bamd, po, t, K, rows, T, impl):
    """batched prefill mat-mul (impl 0: integer-dot kernel, 2: matrix-core kernel) == the reference's mul_mat per activation row, bit for bit;
    ragged token tiles, rows % 16 != 0, residual epilogue, K with an odd number of super-blocks (Llama-2's 11008)"""
    rng = np.random.default_rng(77 * t + K + T)
    W = random_kquant_tensor(t, K, rows, rng)
    X = (rng.standard_normal((T, K)) * 3).astype(np.float32)
    res = rng.standard_normal((T, rows)).astype(np.float32) if T % 2 else None
    got = bamd.op_mul_mat_batch(t, W, rows, K, X, residual=res, impl=impl)
    for i in range(T):
        want = po.mul_mat_q(t, W, rows, K, X[i], nthreads=8)[0]
        if res is not None:
            want = want + res[i]
        assert_bits(got[i], want, "mul_mat_batch type %d K %d T %d impl %d token %d" % (t, K, T, impl, i))


def test_attention_long_context_eight_heads_per_kv_head(bamd, po):
    """the 70B head layout at a long context: gq = 8 -> four query heads per workgroup of the softmax + P.V launch (128 KB of probability rows in
    LDS at n_ctx 8192), the V^T ring over 125 blocks, a half block at the end (n_kv % 64 == 32)"""
    H, Hkv, hd, n_ctx = 8, 1, 128, 8192
    rng = np.random.default_rng(808)
    Ekv = Hkv * hd
    kc = (rng.standard_normal(n_ctx * Ekv) * 0.7).astype(np.float16).view(np.uint16).copy()
    vc = rng.standard_normal(Ekv * n_ctx).astype(np.float16).view(np.uint16).copy()
    for pos in (7999, 8170):
        q = (rng.standard_normal(H * hd) * 2).astype(np.float32)
        k = rng.standard_normal(Ekv).astype(np.float32)
        v = rng.standard_normal(Ekv).astype(np.float32)
        rope = po.rope_cache(pos, hd, 500000.0)
        kc2, vc2 = kc.copy(), vc.copy()
        want, wprobs = oracle_attention(po, q, k, v, kc2, vc2, rope, H, Hkv, hd, n_ctx, pos, False)
        got, gprobs = bamd.op_attention(q, k, v, kc, vc, rope, H, Hkv, hd, n_ctx, pos, prefill_mode=False, want_probs=True)
        assert_bits(gprobs[:wprobs.size], wprobs, "softmax pos %d" % pos)
        assert np.array_equal(kc, kc2) and np.array_equal(vc, vc2), "KV store differs at pos %d" % pos
        assert_bits(got, want, "attention out pos %d" % pos)


@pytest.mark.parametrize("prefill", [False, True])
def test_attention_long_context(bamd, po, prefill):
    """the three-kernel path at real long-context sizes: many 64-position tiles per workgroup, the softmax kernel both with its
    register-cached pass (n_kv <= 8192) and with the multi-pass fallback, the P.V kernel over hundreds of blocks"""
    H, Hkv, hd, n_ctx = 4, 1, 128, 16384
    rng = np.random.default_rng(4242 + prefill)
    Ekv = Hkv * hd
    kc = (rng.standard_normal(n_ctx * Ekv) * 0.7).astype(np.float16).view(np.uint16).copy()
    vc = rng.standard_normal(Ekv * n_ctx).astype(np.float16).view(np.uint16).copy()
    for pos in (4100, 9001, 16383):
        q = (rng.standard_normal(H * hd) * 2).astype(np.float32)
        k = rng.standard_normal(Ekv).astype(np.float32)
        v = rng.standard_normal(Ekv).astype(np.float32)
        rope = po.rope_cache(pos, hd, 500000.0)
        kc2, vc2 = kc.copy(), vc.copy()
        want, wprobs = oracle_attention(po, q, k, v, kc2, vc2, rope, H, Hkv, hd, n_ctx, pos, prefill)
        got, gprobs = bamd.op_attention(q, k, v, kc, vc, rope, H, Hkv, hd, n_ctx, pos, prefill_mode=prefill, want_probs=True)
        assert_bits(gprobs[:wprobs.size], wprobs, "softmax pos %d" % pos)
        assert np.array_equal(kc, kc2) and np.array_equal(vc, vc2), "KV store differs at pos %d" % pos
        assert_bits(got, want, "attention out pos %d" % pos)
