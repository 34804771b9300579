"""GPU: whole-model parity through the C-ABI (bamd_model_load / bamd_decode / bamd_generate_greedy).
  * the committed fixtures of the genuine reference (tiny_a / tiny_b: prefill of 8 tokens + 40 greedy steps):
    logits bit-identical, tokens identical;
  * a larger synthetic model against the CPU oracle on the same GGUF;
  * invariants: hipGraph greedy loop == step-by-step decode; layer-split stages == single stage."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from goldenio import load_bgld
from booster_amd import gguf

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("variant", ["a", "b"])
def test_reference_fixture(bamd, variant):
    g = load_bgld(os.path.join(GOLDEN, "tiny_%s.bgld" % variant))
    m = bamd.Model(os.path.join(GOLDEN, "tiny_%s.gguf" % variant))
    ctx = bamd.Context(m, 128)
    prompt, logits, toks = g["meta/prompt"], g["greedy/logits"], g["greedy/tokens"]
    lg = ctx.decode(prompt, 0)                                  # one micro-batch of 8 tokens (T > 1 semantics)
    assert np.array_equal(bits(lg), bits(logits[0])), "prefill logits: max |d| = %g" % np.abs(lg - logits[0]).max()
    n_past = len(prompt)
    for s, t in enumerate(toks):
        assert int(np.argmax(lg)) == int(t)
        lg = ctx.decode([int(t)], n_past)
        n_past += 1
        assert np.array_equal(bits(lg), bits(logits[s + 1])), "decode step %d: max |d| = %g" % (s, np.abs(lg - logits[s + 1]).max())
    ctx.close(); m.close()


@pytest.mark.parametrize("variant", ["a", "b"])
def test_graph_greedy_matches_reference_tokens(bamd, variant):
    g = load_bgld(os.path.join(GOLDEN, "tiny_%s.bgld" % variant))
    m = bamd.Model(os.path.join(GOLDEN, "tiny_%s.gguf" % variant))
    ctx = bamd.Context(m, 128)
    prompt, logits, toks = g["meta/prompt"], g["greedy/logits"], g["greedy/tokens"]
    ctx.decode(prompt, 0)
    out, ms = ctx.generate_greedy(len(prompt), len(toks))       # device-side loop, one hipGraph per step
    assert np.array_equal(out[:len(toks)], toks)
    assert np.array_equal(bits(ctx.last_logits()), bits(logits[len(toks)]))
    # a second call continues from the KV state
    out2, _ = ctx.generate_greedy(len(prompt) + len(toks), 5)
    assert out2[0] == out[-1]
    ctx.close(); m.close()


def test_synthetic_vs_oracle(bamd, po, tmp_path):
    """Llama-3-8B proportions shrunk: GQA 4:1, hd 128, Q4_K_M type mixture over 4 layers, F = 7 super-blocks."""
    p = str(tmp_path / "syn.gguf")
    gguf.write_synthetic_llama(p, E=1024, H=8, Hkv=2, L=4, F=1792, V=1024, seed=11)
    r = gguf.GGUFReader(p)
    om = po.OracleModel(r); oc = po.OracleContext(om, 96, nthreads=8)
    m = bamd.Model(p); ctx = bamd.Context(m, 96)
    prompt = [(7919 * i + 13) % 1024 for i in range(5)]
    lg_o = oc.decode(prompt, 0); lg_g = ctx.decode(prompt, 0)
    assert np.array_equal(bits(lg_g), bits(lg_o)), "prefill: max |d| = %g" % np.abs(lg_g - lg_o).max()
    n_past = len(prompt)
    for s in range(40):                                          # crosses n_kv = 32 -> 64
        t = int(np.argmax(lg_o))
        lg_o = oc.decode([t], n_past); lg_g = ctx.decode([t], n_past)
        n_past += 1
        assert np.array_equal(bits(lg_g), bits(lg_o)), "step %d: max |d| = %g" % (s, np.abs(lg_g - lg_o).max())
    oc.close(); ctx.close(); m.close()


def test_layer_split_stages_equal_single_stage(bamd, tmp_path):
    """N virtual stages on one device give the logits of one stage (hidden state handed over
    as a device buffer exactly as the multi-process RCCL path does)."""
    import torch
    p = str(tmp_path / "syn2.gguf")
    gguf.write_synthetic_llama(p, E=512, H=4, Hkv=1, L=4, F=768, V=512, seed=3)
    full = bamd.Model(p); cf = bamd.Context(full, 64)
    stages = [bamd.Model(p, 0, 0, 1, True, False), bamd.Model(p, 0, 1, 3, False, False), bamd.Model(p, 0, 3, 4, False, True)]
    ctxs = [bamd.Context(s, 64) for s in stages]
    hid = [torch.zeros(512, dtype=torch.float32, device="cuda") for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    toks = [13, 7, 400, 3]
    lg = None
    for pos, t in enumerate(toks):
        lg = cf.decode([t], pos)
        ctxs[0].stage_step(t, pos, None, hid[0].data_ptr(), False, False, stream)
        ctxs[1].stage_step(t, pos, hid[0].data_ptr(), hid[1].data_ptr(), False, False, stream)
        ctxs[2].stage_step(t, pos, hid[1].data_ptr(), None, True, False, stream)
        tok = ctxs[2].stage_argmax(stream)
        assert tok == int(np.argmax(lg)), "pos %d" % pos
        tdev = torch.zeros(1, dtype=torch.int32, device="cuda")
        ctxs[2].stage_token_to(tdev.data_ptr(), stream)
        assert int(tdev.item()) == tok
    for c in ctxs + [cf]:
        c.close()
    for s in stages + [full]:
        s.close()


def test_layer_split_context_shift_equals_single_stage(bamd, tmp_path):
    """Booster's context shift (bamd_kv_seq_rm + bamd_kv_seq_add, applied to every stage's context) through three layer-split stages
    gives the logits of the single-stage run, which tests/test_gpu_fullsize_ref.py pins to the genuine reference: cells refilled in
    the same order and K rows re-rotated per stage, greedy tokens identical for 90 steps across two shifts (n_ctx 64)."""
    import torch
    p = str(tmp_path / "syn_shift.gguf")
    gguf.write_synthetic_llama(p, E=512, H=8, Hkv=2, L=4, F=768, V=512, seed=11)
    n_ctx, n_keep = 64, 4
    full = bamd.Model(p); cf = bamd.Context(full, n_ctx)
    stages = [bamd.Model(p, 0, 0, 1, True, False), bamd.Model(p, 0, 1, 3, False, False), bamd.Model(p, 0, 3, 4, False, True)]
    ctxs = [bamd.Context(s, n_ctx) for s in stages]
    hid = [torch.zeros(512, dtype=torch.float32, device="cuda") for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    prompt = [(7919 * i + 13) % 512 for i in range(12)]
    lg = None
    for pos, t in enumerate(prompt):
        lg = cf.decode([t], pos)
        ctxs[0].stage_step(t, pos, None, hid[0].data_ptr(), False, False, stream)
        ctxs[1].stage_step(t, pos, hid[0].data_ptr(), hid[1].data_ptr(), False, False, stream)
        ctxs[2].stage_step(t, pos, hid[1].data_ptr(), None, True, False, stream)
    n_past, shifts = len(prompt), 0
    for step in range(90):
        tok = ctxs[2].stage_argmax(stream)
        assert tok == int(np.argmax(lg)), "step %d after %d shifts" % (step, shifts)
        if n_past + 1 > n_ctx:
            n_new = cf.context_shift(n_keep, n_past)
            for c in ctxs:
                assert c.context_shift(n_keep, n_past) == n_new
            n_past = n_new; shifts += 1
        lg = cf.decode([tok], n_past)
        ctxs[0].stage_step(tok, n_past, None, hid[0].data_ptr(), False, False, stream)
        ctxs[1].stage_step(tok, n_past, hid[0].data_ptr(), hid[1].data_ptr(), False, False, stream)
        ctxs[2].stage_step(tok, n_past, hid[1].data_ptr(), None, True, False, stream)
        n_past += 1
    assert shifts == 2
    assert np.array_equal(ctxs[2].stage_logits(stream).view(np.uint32), lg.view(np.uint32)), "all logits of the last step"
    for c in ctxs + [cf]:
        c.close()
    for s in stages + [full]:
        s.close()


def test_pipeline_schedule_single_gpu(bamd, tmp_path):
    """booster_amd.pipeline.run_pipeline with world = 1 on the GPU == bamd_decode greedy loop (token feedback on the device)."""
    import torch
    from booster_amd import pipeline
    p = str(tmp_path / "syn3.gguf")
    gguf.write_synthetic_llama(p, E=512, H=4, Hkv=1, L=3, F=768, V=512, seed=5)
    prompt = [5, 9, 300, 17]
    st = pipeline.HipStage(bamd, torch, p, 0, (0, 3), True, True, 64, 2)
    fed = pipeline.run_pipeline(st, None, 0, 1, prompt, 6, 2)
    st.close()
    m = bamd.Model(p); ctx = bamd.Context(m, 64)
    lg = ctx.decode(prompt, 0)
    want = []
    n_past = len(prompt)
    for i in range(6):
        t = int(np.argmax(lg)); want.append(t)
        lg = ctx.decode([t], n_past); n_past += 1
    assert fed[0] == want and fed[1] == want
    ctx.close(); m.close()


def test_batched_prefill_equals_token_by_token(bamd, tmp_path):
    """SURVEY §8 a8 (ne11 = T): a micro-batch through the batched kernels == the same tokens one by one (T>1 attention semantics
    in both), bit for bit — ragged tile (T % 8 != 0), a second micro-batch on top of the first, mixed Q4_K/Q6_K fused QKV."""
    p = str(tmp_path / "synb.gguf")
    gguf.write_synthetic_llama(p, E=1024, H=8, Hkv=2, L=3, F=1792, V=1024, seed=21)
    m = bamd.Model(p)
    toks = [(7919 * i + 13) % 1024 for i in range(37 + 22)]
    out = {}
    for mode in (1, 2, 0):                                   # batched with MFMA for Q4_K | batched, integer-dot only | token by token
        bamd.set_prefill_batch(mode)
        ctx = bamd.Context(m, 128)
        l1 = ctx.decode(toks[:37], 0).copy()
        l2 = ctx.decode(toks[37:], 37).copy()
        l3 = ctx.decode([5], 59).copy()                      # a decode step on top of the batched KV cache
        out[mode] = (l1, l2, l3)
        if mode == 1:                                        # one call with more than 512 tokens = llama_decode's n_ubatch split
            long_toks = [(11 * i + 5) % 1024 for i in range(700)]
            c2 = bamd.Context(m, 1024); la = c2.decode(long_toks, 0).copy(); c2.close()
            c3 = bamd.Context(m, 1024); c3.decode(long_toks[:512], 0); lb = c3.decode(long_toks[512:], 512).copy(); c3.close()
            assert np.array_equal(bits(la), bits(lb))
        ctx.close()
    bamd.set_prefill_batch(1)
    for mode in (1, 2):
        for a, b in zip(out[mode], out[0]):
            assert np.array_equal(bits(a), bits(b)), "mode %d: max |d| = %g" % (mode, np.abs(a - b).max())
    m.close()


def test_split_gguf_model_equals_single_file(bamd, tmp_path):
    """a gguf-split model (three shards, opened by its first shard) runs exactly like the un-split file"""
    single = str(tmp_path / "one.gguf")
    kw = dict(E=512, H=4, Hkv=2, L=3, F=768, V=512, seed=41)
    gguf.write_synthetic_llama(single, **kw)
    shards = gguf.write_synthetic_llama(str(tmp_path / "model"), n_split=3, **kw)
    toks = [(31 * i + 7) % 512 for i in range(19)]
    out = []
    for path in (single, shards[0]):
        m = bamd.Model(path); ctx = bamd.Context(m, 64)
        l1 = ctx.decode(toks, 0).copy(); l2 = ctx.decode([3], len(toks)).copy()
        out.append((l1, l2)); ctx.close(); m.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(bits(a), bits(b))


def test_large_context_uses_the_same_kernels(bamd, tmp_path):
    """n_ctx = 32768 (a Mistral / Llama-3.1 GGUF opened with its training context): the LDS score rows of the batched and the
    single-launch attention kernels are sized by the sequence, not by n_ctx, so prompts are still evaluated in micro-batches and short
    sequences still decode through the single launch — bit-identical to token-by-token evaluation, also for a micro-batch that
    starts beyond position 8192 (on a zero-initialised cache, the same in both modes)."""
    p = str(tmp_path / "bigctx.gguf")
    gguf.write_synthetic_llama(p, E=1024, H=8, Hkv=2, L=2, F=1792, V=1024, seed=23)
    m = bamd.Model(p)
    toks = [(7919 * i + 13) % 1024 for i in range(70)]
    out = {}
    for mode in (1, 0):
        bamd.set_prefill_batch(mode)
        ctx = bamd.Context(m, 32768)
        l1 = ctx.decode(toks[:41], 0).copy()
        l2 = ctx.decode([7], 41).copy()                      # single-launch attention at n_ctx 32768
        l3 = ctx.decode(toks[41:], 9000).copy()              # micro-batch at positions 9000..9028
        l4 = ctx.decode([9], 9029).copy()
        out[mode] = (l1, l2, l3, l4)
        ctx.close()
    bamd.set_prefill_batch(1)
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(bits(a), bits(b)), "max |d| = %g" % np.abs(a - b).max()
    m.close()


@pytest.mark.parametrize("cfg", ["mistral-q6k-8k", "70b-proportions-q5k"])
def test_baseline_config_shapes_vs_oracle(bamd, po, tmp_path, cfg):
    """BASELINE.json configs 4 and 5 as parity cases at reduced width: (5) every tensor Q6_K, theta 10000, n_ctx 8192 (the
    long-context three-kernel attention in decode, batched prefill with the large score buffer); (4) GQA 8:1 with attn_v in Q5_K
    (three differently typed segments in the fused QKV mat-vec) and an FFN width that is not a multiple of 2048 (ring depth 4)."""
    p = str(tmp_path / "cfg.gguf")
    if cfg == "mistral-q6k-8k":
        gguf.write_synthetic_llama(p, E=512, H=8, Hkv=2, L=2, F=1024, V=512, theta=10000.0, seed=31, type_fn=lambda n, il: gguf.Q6_K, embd_type=gguf.Q6_K)
        n_ctx, prompt = 8192, [(31 * i + 7) % 512 for i in range(45)]
    else:
        def tf(name, il):
            return gguf.Q5_K if name == "attn_v" else gguf.Q6_K if name in ("output", "ffn_down") else gguf.Q4_K
        gguf.write_synthetic_llama(p, E=1024, H=8, Hkv=1, L=2, F=3584, V=512, seed=33, type_fn=tf)
        n_ctx, prompt = 256, [(17 * i + 3) % 512 for i in range(21)]
    r = gguf.GGUFReader(p)
    om = po.OracleModel(r); oc = po.OracleContext(om, n_ctx, nthreads=8)
    m = bamd.Model(p); ctx = bamd.Context(m, n_ctx)
    lg_o = oc.decode(prompt, 0); lg_g = ctx.decode(prompt, 0)
    assert np.array_equal(bits(lg_g), bits(lg_o)), "prefill: max |d| = %g" % np.abs(lg_g - lg_o).max()
    n_past = len(prompt)
    for s in range(24):
        t = int(np.argmax(lg_o))
        lg_o = oc.decode([t], n_past); lg_g = ctx.decode([t], n_past)
        n_past += 1
        assert np.array_equal(bits(lg_g), bits(lg_o)), "step %d: max |d| = %g" % (s, np.abs(lg_g - lg_o).max())
    oc.close(); ctx.close(); m.close()


def test_stage_prefill_equals_single_stage(bamd, tmp_path):
    """batched prompt micro-batches through three virtual stages (hidden states [T][E] handed over as device buffers) == the
    single-stage batched prefill == token by token; then decode steps on top of the stage KV caches."""
    import os
    import torch
    if os.environ.get("BAMD_ATTN_FUSED") == "0" or os.environ.get("BAMD_PREFILL_BATCH") == "0":
        pytest.skip("the batched prefill kernels are switched off by the environment: bamd_stage_prefill reports 'no batched kernels' by design")
    p = str(tmp_path / "syn4.gguf")
    gguf.write_synthetic_llama(p, E=1024, H=8, Hkv=2, L=4, F=2048, V=512, seed=9)
    full = bamd.Model(p); cf = bamd.Context(full, 128)
    stages = [bamd.Model(p, 0, 0, 1, True, False), bamd.Model(p, 0, 1, 3, False, False), bamd.Model(p, 0, 3, 4, False, True)]
    ctxs = [bamd.Context(s, 128) for s in stages]
    side = torch.cuda.Stream()
    toks = [(37 * i + 11) % 512 for i in range(45)]
    with torch.cuda.stream(side):
        stream = torch.cuda.current_stream().cuda_stream
        hid = [torch.zeros(45 * 1024, dtype=torch.float32, device="cuda") for _ in range(2)]
        for (a, b) in ((0, 30), (30, 45)):                      # two micro-batches
            n = b - a
            assert ctxs[0].stage_prefill(toks[a:b], n, a, None, hid[0].data_ptr(), False, stream)
            assert ctxs[1].stage_prefill(None, n, a, hid[0].data_ptr(), hid[1].data_ptr(), False, stream)
            assert ctxs[2].stage_prefill(None, n, a, hid[1].data_ptr(), None, True, stream)
        tok = ctxs[2].stage_argmax(stream)
    lg = cf.decode(toks[:30], 0); lg = cf.decode(toks[30:], 30)
    assert tok == int(np.argmax(lg))
    # one decode step through the stages on top of the batched KV caches
    with torch.cuda.stream(side):
        stream = torch.cuda.current_stream().cuda_stream
        h1 = [torch.zeros(1024, dtype=torch.float32, device="cuda") for _ in range(2)]
        ctxs[0].stage_step(tok, 45, None, h1[0].data_ptr(), False, False, stream)
        ctxs[1].stage_step(tok, 45, h1[0].data_ptr(), h1[1].data_ptr(), False, False, stream)
        ctxs[2].stage_step(tok, 45, h1[1].data_ptr(), None, True, False, stream)
        tok2 = ctxs[2].stage_argmax(stream)
    lg2 = cf.decode([tok], 45)
    assert tok2 == int(np.argmax(lg2))
    for c in ctxs + [cf]:
        c.close()
    for s in stages + [full]:
        s.close()


class _NoPrefill:
    """a HipStage without the batched prompt phase: run_pipeline's hasattr fallback (one token per round)"""
    def __init__(self, st):
        self._st = st

    def __getattr__(self, name):
        if name == "prefill":
            raise AttributeError(name)
        return getattr(self._st, name)


def test_pipeline_prompt_phase_on_real_kernels(bamd, tmp_path):
    """booster_amd.pipeline's batched prompt phase on the REAL stage (HipStage.prefill -> bamd_stage_prefill) against the one-token-per-round path and
    the plain level-1 evaluation: a 513-token prompt = a 512-position micro-batch + a ONE-token tail (bamd_stage_prefill declines T = 1: the per-token fallback
    inside HipStage.prefill), a 1100-token prompt = three micro-batches, and the same with the batched kernels switched off (every micro-batch takes the fallback)"""
    import os
    import torch
    from booster_amd import pipeline
    if os.environ.get("BAMD_ATTN_FUSED") == "0" or os.environ.get("BAMD_PREFILL_BATCH") == "0":
        pytest.skip("the batched prefill kernels are switched off by the environment")
    p = str(tmp_path / "syn_pp.gguf")
    gguf.write_synthetic_llama(p, E=1024, H=8, Hkv=2, L=3, F=2048, V=512, seed=13)
    n_ctx = 1280
    for n_prompt in (513, 1100):
        prompt = [(37 * i + 11) % 512 for i in range(n_prompt)]
        m = bamd.Model(p); ctx = bamd.Context(m, n_ctx)
        for i in range(0, n_prompt, 512):
            lg = ctx.decode(prompt[i:i + 512], i)
        want, n_past = [], n_prompt
        for _ in range(5):
            t = int(np.argmax(lg)); want.append(t)
            lg = ctx.decode([t], n_past); n_past += 1
        ctx.close(); m.close()
        for mode in ("batched", "no prefill method", "batched kernels off"):
            st = pipeline.HipStage(bamd, torch, p, 0, (0, 3), True, True, n_ctx, 1)
            if mode == "batched kernels off":
                bamd.set_prefill_batch(False)
            try:
                fed = pipeline.run_pipeline(_NoPrefill(st) if mode == "no prefill method" else st, None, 0, 1, prompt, 5, 1)
            finally:
                bamd.set_prefill_batch(True)
                st.close()
            assert fed[0] == want, "%d-token prompt, %s" % (n_prompt, mode)
