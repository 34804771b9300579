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
    mfma = bamd.op_mul_mat_batch(t, W, rows, K, X, norm_w=w, eps=1e-5, residual=res, impl=2)
    idot = bamd.op_mul_mat_batch(t, W, rows, K, X, norm_w=w, eps=1e-5, residual=res, impl=0)
    assert np.array_equal(bits(mfma), bits(idot)), "case %d: matrix-core and integer-dot kernels differ (type %d K %d rows %d T %d)" % (i, t, K, rows, T)
    for tok in sorted(set([0, T // 2, T - 1])):
        a = X[tok] if w is None else (po.rms_norm(X[tok], 1e-5) * w).astype(np.float32)
        want = po.mul_mat_q(t, W, rows, K, a, nthreads=8)[0]
        if res is not None:
            want = want + res[tok]
        assert np.array_equal(bits(mfma[tok]), bits(want)), "case %d token %d: differs from the oracle (type %d K %d rows %d T %d)" % (i, tok, t, K, rows, T)


def attn_cases(n, seed):
    rng = np.random.default_rng(seed)
    shapes = [(4, 1, 128), (4, 2, 64), (8, 8, 64), (8, 1, 256), (8, 2, 128), (6, 3, 192), (16, 2, 64),
              (6, 2, 128), (12, 4, 64), (10, 2, 64), (7, 1, 128), (12, 2, 64)]      # heads per KV head 3 (Llama-3.2-3B), 5, 7, 6
    out = []
    for i in range(n):
        H, Hkv, hd = shapes[int(rng.integers(0, len(shapes)))]
        n_ctx = int(rng.choice([64, 128, 448, 512, 1024]))
        pos = int(rng.integers(0, n_ctx))
        out.append((i, H, Hkv, hd, n_ctx, pos, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("i,H,Hkv,hd,n_ctx,pos,prefill,long_path", attn_cases(48, 99))
def test_attention_sweep(bamd, po, i, H, Hkv, hd, n_ctx, pos, prefill, long_path):
    """one token's attention at random head layouts (GQA 1 / 2 / 4 / 8, head_dim 64 ... 256), context sizes and positions, through the
    single-launch kernel and through the three-launch path (heads per KV head 1 .. 8), T = 1 and T > 1 score arithmetic: output, KV-cache bytes (and probabilities on
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


ARCHS = [
    # E, H, Hkv, L, F, V, type mix, rope_freqs, theta, embd type
    dict(E=768, H=6, Hkv=2, L=2, F=1280, V=300, mix="q4km", rope_freqs=False, theta=10000.0),       # K/256 = 3 and 5: off the specialised tables
    dict(E=512, H=8, Hkv=8, L=3, F=1024, V=256, mix="q6k", rope_freqs=True, theta=500000.0),        # MHA, head_dim 64, every matrix Q6_K, llama-3.1 factors
    dict(E=1024, H=4, Hkv=1, L=2, F=2816, V=520, mix="q5v", rope_freqs=False, theta=500000.0),      # head_dim 256, GQA 4, Q5_K attn_v, F = 11 super-blocks
    dict(E=2048, H=16, Hkv=2, L=2, F=2048, V=1000, mix="q4km", rope_freqs=False, theta=1000000.0),  # GQA 8, K/256 = 8: the specialised tables
    dict(E=768, H=4, Hkv=2, L=2, F=2304, V=264, mix="q4k", rope_freqs=True, theta=10000.0),         # head_dim 192
]


def arch_type_fn(mix, L):
    from booster_amd import gguf
    if mix == "q6k":
        return lambda name, il: gguf.Q6_K
    if mix == "q4k":
        return lambda name, il: gguf.Q4_K
    if mix == "q5v":
        return lambda name, il: gguf.Q5_K if name == "attn_v" else gguf.q4_k_m_type(name, il, L)
    return None


@pytest.mark.parametrize("ai", range(len(ARCHS)))
def test_model_architecture_sweep(bamd, po, tmp_path, ai):
    """whole models of odd proportions against the oracle: a 19-token batched prompt (MFMA prefill kernels), the same prompt token by token,
    then 24 single-token steps and the device-side greedy loop — logits bit for bit"""
    from booster_amd import gguf
    a = dict(ARCHS[ai]); mix = a.pop("mix")
    p = str(tmp_path / ("arch%d.gguf" % ai))
    gguf.write_synthetic_llama(p, seed=31 + ai, type_fn=arch_type_fn(mix, a["L"]), embd_type=gguf.Q6_K if mix == "q6k" else gguf.Q4_K, **a)
    r = gguf.GGUFReader(p)
    om = po.OracleModel(r); oc = po.OracleContext(om, 64, nthreads=8)
    m = bamd.Model(p); ctx = bamd.Context(m, 64); ctx2 = bamd.Context(m, 64)
    V = a["V"]
    prompt = [(7919 * i + 13) % V for i in range(19)]
    lg_o = oc.decode(prompt, 0)
    lg_g = ctx.decode(prompt, 0)                            # one micro-batch
    assert np.array_equal(bits(lg_g), bits(lg_o)), "arch %d: batched prompt, max |d| = %g" % (ai, np.abs(lg_g - lg_o).max())
    for i, t in enumerate(prompt):                          # the same prompt one token per call: T = 1 arithmetic differs from T > 1 by design
        lg1 = ctx2.decode([t], i)
    lg1_o = None
    oc2 = po.OracleContext(om, 64, nthreads=8)
    for i, t in enumerate(prompt):
        lg1_o = oc2.decode([t], i)
    assert np.array_equal(bits(lg1), bits(lg1_o)), "arch %d: token-by-token prompt" % ai
    n_past = len(prompt)
    toks = []
    for s in range(24):
        t = int(np.argmax(lg_o)); toks.append(t)
        lg_o = oc.decode([t], n_past); lg_g = ctx.decode([t], n_past); n_past += 1
        assert np.array_equal(bits(lg_g), bits(lg_o)), "arch %d: step %d, max |d| = %g" % (ai, s, np.abs(lg_g - lg_o).max())
    # the device-side loop from the batched prompt's state
    ctx3 = bamd.Context(m, 64)
    ctx3.decode(prompt, 0)
    out, _ = ctx3.generate_greedy(len(prompt), 24)
    assert [int(x) for x in out[:24]] == toks, "arch %d: device-side greedy tokens" % ai
    assert np.array_equal(bits(ctx3.last_logits()), bits(lg_o)), "arch %d: logits after the device-side loop" % ai
    for c in (ctx, ctx2, ctx3):
        c.close()
    oc.close(); oc2.close(); m.close()


@pytest.mark.parametrize("t", [12, 13, 14])
@pytest.mark.parametrize("K,rows", [(28672, 264), (16384, 72), (10240, 1032)])
def test_mul_mat_vec_large_k(bamd, po, t, K, rows):
    """K beyond 8192 with one wave per row-group (Llama-3-70B's ffn_down: K = 28672; these shapes run on the generic kernel — a specialised
    instance measured no faster) against the oracle, with the residual epilogue"""
    rng = np.random.default_rng(31 * t + K)
    W = random_kquant_tensor(t, K, rows, rng)
    x = (rng.standard_normal(K) * 2).astype(np.float32)
    res = rng.standard_normal(rows).astype(np.float32)
    fast = bamd.op_mul_mat_vec(t, W, rows, K, x, residual=res, mode=1)
    generic = bamd.op_mul_mat_vec(t, W, rows, K, x, residual=res, mode=17)
    want = po.mul_mat_q(t, W, rows, K, x, nthreads=8)[0] + res
    assert np.array_equal(bits(fast), bits(generic)) and np.array_equal(bits(fast), bits(want)), "type %d K %d rows %d" % (t, K, rows)


@pytest.mark.parametrize("t", [12, 13, 14])
@pytest.mark.parametrize("K,rows,norm", [(11008, 520, False), (13824, 264, False), (5120, 640, True), (5120, 1032, False), (4352, 72, True), (12032, 40, False), (3072, 3072, False), (3072, 1032, True), (2304, 24, False), (3840, 520, True)])
def test_mul_mat_vec_uneven_split(bamd, po, t, K, rows, norm):
    """split-K when K / 256 is not a multiple of 8 (Llama-2's 11008 = 43, 13824 = 54, 5120 = 20 super-blocks, Llama-3.2-3B's 3072 = 12; 9, 15, 17
    and 47 at the edges of the ranges): uneven shares per wave == one wave per row-group == the oracle"""
    rng = np.random.default_rng(13 * t + K + rows)
    W = random_kquant_tensor(t, K, rows, rng)
    x = (rng.standard_normal(K) * 2).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32) if norm else None
    res = None if norm else rng.standard_normal(rows).astype(np.float32)
    split = bamd.op_mul_mat_vec(t, W, rows, K, x, norm_w=w, eps=1e-5, residual=res, mode=2)
    rowwise = bamd.op_mul_mat_vec(t, W, rows, K, x, norm_w=w, eps=1e-5, residual=res, mode=1)
    a = x if w is None else (po.rms_norm(x, 1e-5) * w).astype(np.float32)
    want = po.mul_mat_q(t, W, rows, K, a, nthreads=8)[0]
    if res is not None:
        want = want + res
    assert np.array_equal(bits(split), bits(want)), "split-K differs from the oracle (type %d K %d rows %d)" % (t, K, rows)
    assert np.array_equal(bits(rowwise), bits(want)), "row-wise differs from the oracle (type %d K %d rows %d)" % (t, K, rows)
