"""CPU: the oracle (oracle/booster_oracle.c) against the fixtures produced by the genuine reference build
(tests/golden/*.bgld from oracle/harness/gen_golden.cpp, block_kats.npz from tests/golden/gen_block_kats.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from goldenio import load_bgld
from booster_amd.gguf import GGUFReader

TYPES = {"q4_K": 12, "q5_K": 13, "q6_K": 14}
BB = {12: 144, 13: 176, 14: 210}


@pytest.mark.parametrize("tname", sorted(TYPES))
@pytest.mark.parametrize("K", [256, 768, 4096])
def test_block_kats(po, kats, tname, K):
    t = TYPES[tname]
    key = "%s_K%d" % (tname, K)
    blocks, deq, x, q8, dots = (kats[key + s] for s in ("_blocks", "_deq", "_x", "_q8", "_dots"))
    rows = deq.shape[0]
    # activation quantisation: bytes identical (d, qs, bsums) — except bsums of all-zero blocks, which the
    # reference leaves uninitialised (ggml-quants.c:3608-3612)
    mine = po.quantize_q8_K(x).reshape(-1, 292)
    ref = q8.reshape(-1, 292)
    for b in range(mine.shape[0]):
        if ref[b, :4].view(np.float32)[0] == 0.0:
            assert np.array_equal(mine[b, :260], ref[b, :260])
        else:
            assert np.array_equal(mine[b], ref[b])
    # dequantisation bit-exact
    for r in range(rows):
        rb = K // 256 * BB[t]
        assert np.array_equal(po.dequantize(t, blocks[r * rb:(r + 1) * rb], K).view(np.uint32), deq[r].view(np.uint32))
    # dot products bit-exact with the reference's AVX2 kernels
    y = po.mul_mat_q(t, blocks, rows, K, x)[0]
    assert np.array_equal(y.view(np.uint32), dots.view(np.uint32))


@pytest.mark.parametrize("variant", ["a", "b"])
def test_model_golden(po, variant):
    g = load_bgld(os.path.join(GOLDEN, "tiny_%s.bgld" % variant))
    r = GGUFReader(os.path.join(GOLDEN, "tiny_%s.gguf" % variant))
    m = po.OracleModel(r)
    ctx = po.OracleContext(m, 128, nthreads=4)
    ctx.enable_taps()
    prompt = g["meta/prompt"]
    lg = ctx.decode(prompt, 0)
    for k in ["attn_norm-0", "Vcur-0", "kqv_merged_cont-0", "kqv_out-0", "ffn_inp-0", "ffn_norm-0", "ffn_gate-0", "ffn_up-0",
              "ffn_gate_par-0", "ffn_out-0", "l_out-0", "l_out-1", "result_norm", "result_output"]:
        ref = g["prefill/" + k].reshape(-1)
        assert np.array_equal(ref.view(np.uint32), ctx.taps[k].view(np.uint32)), k
    logits, toks = g["greedy/logits"], g["greedy/tokens"]
    assert np.array_equal(lg.view(np.uint32), logits[0].view(np.uint32))
    n_past = len(prompt)
    for s, t in enumerate(toks):
        assert int(np.argmax(lg)) == int(t)
        lg = ctx.decode([t], n_past)
        n_past += 1
        assert np.array_equal(lg.view(np.uint32), logits[s + 1].view(np.uint32)), "decode step %d" % s
    # decode-step taps (T = 1 path: tinyBLAS attention)
    ctx.close()


def test_fp16_roundtrip(po):
    L = po.lib()
    h = np.arange(65536, dtype=np.uint16)
    f = h.view(np.float16).astype(np.float32)
    mine = np.array([L.bo_fp16_to_fp32(int(v)) for v in h[::7]], np.float32)
    ref = f[::7]
    ok = (mine.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(mine) & np.isnan(ref))
    assert ok.all()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 1e3, 7e4)])
    x = np.concatenate([x, np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-7, 5.96e-8, 2.98e-8, 2.9802325e-8, np.inf, -np.inf], np.float32)])
    mine = np.array([L.bo_fp32_to_fp16(float(v)) for v in x], np.uint16)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(mine, ref)
