"""GPU: the weight-stream engine (csrc/bamd_wse.hip: one persistent launch per decode step, LDS-DMA loader / consumer / chainer waves) against
the launch kernels and against the genuine reference.

* single pieces through the engine kernel (bamd_op_wse_matvec) == the launch kernels' mat-vecs (which tests/test_gpu_ops.py pins to the oracle),
  bit for bit, for every K-quant type, both prologues, every epilogue, ragged row counts;
* BASELINE config 2 end to end ON THE ENGINE: the full-size Llama-3-8B Q4_K_M fixture recorded from gotzmann/booster's own llama_decode
  (tests/golden/fullsize_8b.bgld) — every greedy token and a digest of all logits of every step, bit for bit."""
import numpy as np
import pytest

from booster_amd.gguf import random_kquant_tensor

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


CASES = [(12, 14336, 4096, True, False), (12, 6144, 4096, True, False), (12, 4096, 4096, False, True), (14, 4096, 14336, False, True),
         (12, 4096, 14336, False, True), (13, 1024, 4096, True, False), (14, 32000, 4096, True, False), (12, 4100, 4096, False, False),
         (13, 4096, 8192, False, True), (14, 2048, 2048, True, True)]


@pytest.mark.parametrize("t,rows,K,norm,res", CASES)
@pytest.mark.parametrize("nc,nch", [(10, 1), (9, 2), (13, 2)])
def test_engine_piece_equals_launch_kernel(bamd, po, t, rows, K, norm, res, nc, nch):
    rng = np.random.default_rng(31 * t + rows + K)
    W = random_kquant_tensor(t, K, rows, rng, amp=4.0)
    x = (rng.standard_normal(K) * 2).astype(np.float32)
    nw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32) if norm else None
    r = rng.standard_normal(rows).astype(np.float32) if res else None
    want = bamd.op_mul_mat_vec(t, W, rows, K, x, norm_w=nw, eps=1e-5, residual=r)
    got, info = bamd.op_wse_matvec(t, W, rows, K, x, norm_w=nw, eps=1e-5, residual=r, nc=nc, nch=nch)
    assert np.array_equal(bits(got), bits(want)), "engine piece differs from the launch kernel (type %d, %d x %d)" % (t, rows, K)
    if rows * (K // 256) <= 64 * 1024:                        # and from the oracle directly where that takes seconds
        a = (po.rms_norm(x, 1e-5) * nw).astype(np.float32) if norm else x
        ref = po.mul_mat_q(t, W, rows, K, a, nthreads=8)[0]
        if res:
            ref = ref + r
        assert np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("t", [12, 13, 14])
@pytest.mark.parametrize("rows,K", [(14336, 4096), (1024, 2048)])
def test_engine_gate_up_equals_launch_kernel(bamd, t, rows, K):
    rng = np.random.default_rng(7 * t + rows)
    Wg = random_kquant_tensor(t, K, rows, rng, amp=4.0); Wu = random_kquant_tensor(t, K, rows, rng, amp=4.0)
    x = (rng.standard_normal(K) * 2).astype(np.float32)
    nw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    want = bamd.op_ffn_gate_up(t, Wg, Wu, rows, K, x, norm_w=nw, eps=1e-5)
    got, _ = bamd.op_wse_matvec(t, Wg, rows, K, x, norm_w=nw, eps=1e-5, w_up_raw=Wu, nc=10)
    assert np.array_equal(bits(got), bits(want))


def test_engine_f64_order_worst_case(bamd):
    """the constructed 4096-vector on which the tree and the sequential sum of squares round to different f32 means (tests/test_f64_order.py):
    the engine's consumers must take the reference's order there as the launch kernels do"""
    import os
    kat = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f64_order_kat.npz"))
    x = kat["x"].astype(np.float32); K = x.size
    rng = np.random.default_rng(3)
    W = random_kquant_tensor(12, K, 512, rng, amp=4.0)
    nw = np.ones(K, np.float32)
    eps = float(kat["eps"]) if "eps" in kat.files else 1e-5
    want = bamd.op_mul_mat_vec(12, W, 512, K, x, norm_w=nw, eps=eps)
    got, _ = bamd.op_wse_matvec(12, W, 512, K, x, norm_w=nw, eps=eps, nc=10)
    assert np.array_equal(bits(got), bits(want))


def test_config2_8b_decode_on_the_engine_matches_reference(bamd):
    import test_gpu_fullsize_ref as fr
    bamd.set_wse(1)
    try:
        fx = fr.load_fixture("8b")
        _, n_prompt, n_decode, n_ctx = fr.gen.CONFIGS["8b"]
        m = bamd.Model(fr.model_for("8b", fx)); ctx = bamd.Context(m, n_ctx)
        V = m.n_vocab
        prompt = [(7919 * i + 13) % V for i in range(n_prompt)]
        lg = ctx.decode(prompt, 0)
        fr.check_step(fx, 0, lg, "8b prompt")
        n_past = n_prompt
        for k in range(1, 7):
            lg = ctx.decode([int(fx["tokens"][k - 1])], n_past); n_past += 1
            fr.check_step(fx, k, lg, "8b decode on the engine")
        active, why = ctx.wse_active()
        assert active, "the engine was not used: " + why
        rest = n_decode - 6
        out, _ = ctx.generate_greedy(n_past, rest)
        assert list(out[:rest + 1]) == [int(t) for t in fx["tokens"][6:n_decode + 1]]
        fr.check_step(fx, n_decode, ctx.last_logits(), "8b last greedy step on the engine")
        ctx.close(); m.close()
    finally:
        bamd.set_wse(0)


def _run_on_engine(bamd, cfg, stepwise, expect_active):
    import test_gpu_fullsize_ref as fr
    bamd.set_wse(1)
    try:
        fx = fr.load_fixture(cfg)
        _, n_prompt, n_decode, n_ctx = fr.gen.CONFIGS[cfg]
        m = bamd.Model(fr.model_for(cfg, fx)); ctx = bamd.Context(m, n_ctx)
        prompt = [(7919 * i + 13) % m.n_vocab for i in range(n_prompt)]
        fr.check_step(fx, 0, ctx.decode(prompt, 0), cfg + " prompt")
        n_past = n_prompt
        for k in range(1, stepwise + 1):
            lg = ctx.decode([int(fx["tokens"][k - 1])], n_past); n_past += 1
            fr.check_step(fx, k, lg, cfg + " decode, engine requested")
        active, why = ctx.wse_active()
        assert active == expect_active, (active, why)
        rest = n_decode - stepwise
        out, _ = ctx.generate_greedy(n_past, rest)
        assert list(out[:rest + 1]) == [int(t) for t in fx["tokens"][stepwise:n_decode + 1]]
        fr.check_step(fx, n_decode, ctx.last_logits(), cfg + " last greedy step, engine requested")
        ctx.close(); m.close()
        return why
    finally:
        bamd.set_wse(0)


def test_config4_70b_stage_on_the_engine_matches_reference(bamd):
    """ten layers at the Llama-3-70B widths (K = 8192 / 28672, 64 query heads on 64 workgroups, Q5_K / Q6_K attn_v: CUs whose QKV run crosses a type
    boundary get two pieces; ffn_down: twelve blocks of the hidden vector per consumer wave, two sweeps) on the engine, against the genuine reference"""
    _run_on_engine(bamd, "70b_stage", 4, True)


def test_engine_declines_shapes_without_a_program_and_the_launch_sequence_answers(bamd):
    """Llama-2-7B: n_ff = 11008 = 43 super-blocks per ffn_down row — no engine program (8-record term chunks): with the engine requested the context
    says why, runs the launch sequence, and still reproduces the reference"""
    why = _run_on_engine(bamd, "l2_7b", 3, False)
    assert "multiple of 8" in why


def test_engine_soak_equals_launch_sequence(bamd):
    """several hundred consecutive engine steps per context, four prompts, against the launch sequence's greedy tokens and final logits on the same model: a stale granule,
    a tag that repeats, a ring slot refilled too early or a term chunk released twice would change a token somewhere (every run re-uses the granule vectors, the tags advance
    with the device step and the host serial).  Positions stay below the single-launch attention's limit, where the engine is active."""
    import test_gpu_fullsize_ref as fr
    fx = fr.load_fixture("8b")
    path = fr.model_for("8b", fx)
    V = 128256
    runs = []
    for on in (0, 1):
        bamd.set_wse(on)
        try:
            m = bamd.Model(path); ctx = bamd.Context(m, 512)
            res = []
            for seed in range(4):
                prompt = [(104729 * i + 7 * seed + 1) % V for i in range(24 + 8 * seed)]
                ctx.decode(prompt, 0)
                out, _ = ctx.generate_greedy(len(prompt), 380)
                res.append((out.copy(), ctx.last_logits().view(np.uint32).copy()))
            if on:
                assert ctx.wse_active()[0]
            ctx.close(); m.close()
            runs.append(res)
        finally:
            bamd.set_wse(0)
    for (ta, la), (tb, lb) in zip(*runs):
        assert np.array_equal(ta, tb) and np.array_equal(la, lb)
