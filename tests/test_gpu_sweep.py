"""GPU: a seeded random sweep over mat-vec shapes — K any multiple of 256 up to 16384, row counts that are not multiples of 8 or 16, all three
K-quant types, both launch modes, with and without the RMSNorm prologue / residual epilogue.  Every case runs three ways and all must
agree bit for bit: the host-dispatched specialised kernels (default), the generic kernels with run-time dispatch (mode bit 4), and — where
the CPU finishes in well under a second — the oracle.  The fixed lists of tests/test_gpu_ops.py pin the shapes the models use; this
sweep is for the ones nobody thought of (ragged tails, odd K/256, one row-group, shapes where the launcher must fall back)."""
import numpy as np
import pytest

from booster_amd.gguf import random_kquant_tensor

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        t = int(rng.choice([12, 13, 14]))
        nb = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 24, 32, 40, 56, 64])) if i % 3 else int(rng.integers(1, 65))
        rows = int(rng.choice([1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 500, 512, 513, 1000, 2047, 4100])) if i % 2 else int(rng.integers(1, 3000))
        out.append((i, t, nb * 256, rows, int(rng.integers(0, 3)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("i,t,K,rows,mode,norm,resid", cases(72, 20260927))
def test_mul_mat_vec_sweep(bamd, po, i, t, K, rows, mode, norm, resid):
    rng = np.random.default_rng(7000 + i)
    W = random_kquant_tensor(t, K, rows, rng, amp=float(10 ** rng.uniform(-1, 1)))
    x = (rng.standard_normal(K) * 10 ** rng.uniform(-2, 2)).astype(np.float32)
    if i % 5 == 0:
        x[: min(256, K)] = 0.0                              # an all-zero activation block
    w = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32) if norm else None
    res = rng.standard_normal(rows).astype(np.float32) if resid else None
    fast = bamd.op_mul_mat_vec(t, W, rows, K, x, norm_w=w, eps=1e-5, residual=res, mode=mode)
    generic = bamd.op_mul_mat_vec(t, W, rows, K, x, norm_w=w, eps=1e-5, residual=res, mode=mode + 16)
    assert np.array_equal(bits(fast), bits(generic)), "case %d: specialised and generic kernels differ (type %d K %d rows %d mode %d)" % (i, t, K, rows, mode)
    if K * rows <= 4096 * 1100:                             # the oracle in well under a second
        a = x if w is None else (po.rms_norm(x, 1e-5) * w).astype(np.float32)
        want = po.mul_mat_q(t, W, rows, K, a, nthreads=8)[0]
        if res is not None:
            want = want + res
        assert np.array_equal(bits(fast), bits(want)), "case %d: differs from the oracle (type %d K %d rows %d mode %d norm %d)" % (i, t, K, rows, mode, norm)


@pytest.mark.parametrize("i,t,K,rows", [(j, t, K, rows) for j, (t, K, rows) in enumerate(
    [(12, 2048, 8), (12, 2048, 2056), (13, 4096, 520), (14, 4096, 24), (14, 8192, 1032), (12, 1280, 264), (13, 256, 40), (14, 14336, 136)])])
def test_ffn_gate_up_sweep(bamd, po, i, t, K, rows):
    """the fused gate/up launch (SiLU epilogue) at ragged shapes against the oracle"""
    rng = np.random.default_rng(9100 + i)
    Wg = random_kquant_tensor(t, K, rows, rng, amp=3.0)
    Wu = random_kquant_tensor(t, K, rows, rng, amp=3.0)
    x = (rng.standard_normal(K) * 2).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    got = bamd.op_ffn_gate_up(t, Wg, Wu, rows, K, x, norm_w=w, eps=1e-5)
    a = (po.rms_norm(x, 1e-5) * w).astype(np.float32)
    g = po.mul_mat_q(t, Wg, rows, K, a, nthreads=8)[0]
    u = po.mul_mat_q(t, Wu, rows, K, a, nthreads=8)[0]
    L = po.lib()
    want = np.array([L.bo_v_silu(float(v)) for v in g], np.float32) * u
    assert np.array_equal(bits(got), bits(want)), "gate/up case %d (type %d K %d rows %d)" % (i, t, K, rows)


def batch_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        t = int(rng.choice([12, 13, 14]))
        nb = int(rng.choice([1, 2, 3, 4, 5, 8, 11, 16, 43, 56]))
        rows = int(rng.choice([8, 16, 24, 40, 120, 128, 136, 264, 520]))
        T = int(rng.choice([2, 3, 15, 16, 17, 31, 32, 33, 63, 65, 100]))
        out.append((i, t, nb * 256, rows, T, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("i,t,K,rows,T,norm,resid", batch_cases(36, 424242))
def test_mul_mat_batch_sweep(bamd, po, i, t, K, rows, T, norm, resid):
    """batched prefill: the exact MFMA kernel == the integer-dot kernel == the oracle per activation row, at random (odd K/256, ragged token
    tiles, row counts off the 16-row tile, RMSNorm prologue, residual epilogue)"""
    rng = np.random.default_rng(8800 + i)
    W = random_kquant_tensor(t, K, rows, rng, amp=float(10 ** rng.uniform(-1, 1)))
    X = (rng.standard_normal((T, K)) * 10 ** rng.uniform(-1, 1)).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32) if norm else None
    res = rng.standard_normal((T, rows)).astype(np.float32) if resid else None
    mfma = bamd.op_mul_mat_batch(t, W, rows, K, X, norm_w=w, eps=1e-5, residual=res, impl=1)
    idot = bamd.op_mul_mat_batch(t, W, rows, K, X, norm_w=w, eps=1e-5, residual=res, impl=0)
    assert np.array_equal(bits(mfma), bits(idot)), "case %d: MFMA and integer-dot kernels differ (type %d K %d rows %d T %d)" % (i, t, K, rows, T)
    for tok in sorted(set([0, T // 2, T - 1])):
        a = X[tok] if w is None else (po.rms_norm(X[tok], 1e-5) * w).astype(np.float32)
        want = po.mul_mat_q(t, W, rows, K, a, nthreads=8)[0]
        if res is not None:
            want = want + res[tok]
        assert np.array_equal(bits(mfma[tok]), bits(want)), "case %d token %d: differs from the oracle (type %d K %d rows %d T %d)" % (i, tok, t, K, rows, T)


def attn_cases(n, seed):
    rng = np.random.default_rng(seed)
    shapes = [(4, 1, 128), (4, 2, 64), (8, 8, 64), (8, 1, 256), (8, 2, 128), (6, 3, 192), (16, 2, 64)]
    out = []
    for i in range(n):
        H, Hkv, hd = shapes[int(rng.integers(0, len(shapes)))]
        n_ctx = int(rng.choice([64, 128, 448, 512, 1024]))
        pos = int(rng.integers(0, n_ctx))
        out.append((i, H, Hkv, hd, n_ctx, pos, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("i,H,Hkv,hd,n_ctx,pos,prefill,long_path", attn_cases(28, 99))
def test_attention_sweep(bamd, po, i, H, Hkv, hd, n_ctx, pos, prefill, long_path):
    """one token's attention at random head layouts (GQA 1 / 2 / 4 / 8, head_dim 64 ... 256), context sizes and positions, through the
    single-launch kernel and through the three-launch path, T = 1 and T > 1 score arithmetic: output, KV-cache bytes (and probabilities on
    the three-launch path) against the oracle"""
    from test_gpu_ops import oracle_attention
    rng = np.random.default_rng(5100 + i)
    Ekv = Hkv * hd
    kc = (rng.standard_normal(n_ctx * Ekv) * 0.7).astype(np.float16).view(np.uint16).copy()
    vc = rng.standard_normal(Ekv * n_ctx).astype(np.float16).view(np.uint16).copy()
    q = (rng.standard_normal(H * hd) * 2).astype(np.float32)
    k = rng.standard_normal(Ekv).astype(np.float32)
    v = rng.standard_normal(Ekv).astype(np.float32)
    rope = po.rope_cache(pos, hd, 10000.0 if i % 2 else 500000.0)
    kc2, vc2 = kc.copy(), vc.copy()
    want, wprobs = oracle_attention(po, q, k, v, kc2, vc2, rope, H, Hkv, hd, n_ctx, pos, prefill)
    if long_path:
        got, gprobs = bamd.op_attention(q, k, v, kc, vc, rope, H, Hkv, hd, n_ctx, pos, prefill_mode=prefill, want_probs=True)
        assert np.array_equal(bits(gprobs[:wprobs.size]), bits(wprobs)), "case %d: softmax" % i
    else:
        got = bamd.op_attention(q, k, v, kc, vc, rope, H, Hkv, hd, n_ctx, pos, prefill_mode=prefill)
    assert np.array_equal(kc, kc2) and np.array_equal(vc, vc2), "case %d: KV store" % i
    assert np.array_equal(bits(got), bits(want)), "case %d: attention output (H %d Hkv %d hd %d n_ctx %d pos %d prefill %d long %d)" % (i, H, Hkv, hd, n_ctx, pos, prefill, long_path)
