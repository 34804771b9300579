"""GPU: the own AQL queue (csrc/bamd_aql.h) is the path that RUNS by default — not a silent fall-back to the hipGraph — and gives the hipGraph path's bits:
  * the device-side greedy loop on the genuine reference's tiny fixture (single-launch attention) and on a longer sequence (scores | softmax + P.V);
  * single-token bamd_decode steps (the bridge's token loop: state from the pinned host inbox);
  * more packets than the queue's ring holds (the wrap), many short replays.
BAMD_AQL=0 in the environment (tools/switch_matrix.sh) turns the assertions about which path ran around."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from goldenio import load_bgld
from booster_amd import gguf

pytestmark = pytest.mark.gpu
OWN_QUEUE = os.environ.get("BAMD_AQL", "1") != "0"


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def greedy(bamd, path, n_ctx, prompt, n_steps, aql):
    bamd.set_aql(aql)
    try:
        m = bamd.Model(path); ctx = bamd.Context(m, n_ctx)
        for i in range(0, len(prompt), 512):
            ctx.decode(prompt[i:i + 512], i)
        out, _ = ctx.generate_greedy(len(prompt), n_steps)
        lg = ctx.last_logits()
        runs = ctx.aql_runs()
        ctx.close(); m.close()
        return out, lg, runs
    finally:
        bamd.set_aql(True)


def test_greedy_loop_runs_on_the_own_queue_and_matches_the_hipgraph(bamd):
    g = load_bgld(os.path.join(GOLDEN, "tiny_a.bgld"))
    path = os.path.join(GOLDEN, "tiny_a.gguf")
    prompt, toks = [int(t) for t in g["meta/prompt"]], g["greedy/tokens"]
    out1, lg1, runs1 = greedy(bamd, path, 128, prompt, len(toks), True)
    out0, lg0, runs0 = greedy(bamd, path, 128, prompt, len(toks), False)
    assert runs0 == 0
    assert (runs1 >= 1) == OWN_QUEUE, "the greedy loop did not run where it should (BAMD_AQL_VERBOSE=1 says why the own queue is not used)"
    assert np.array_equal(out1, out0) and np.array_equal(out1[:len(toks)], toks)
    assert np.array_equal(bits(lg1), bits(lg0)) and np.array_equal(bits(lg1), bits(g["greedy/logits"][len(toks)]))


def test_long_sequence_path_on_the_own_queue(bamd, tmp_path):
    """beyond 448 positions the attention is scores | softmax + P.V: the step still replays from the own queue (bamd_attention_split_is_ik_clean)"""
    p = str(tmp_path / "aql_long.gguf")
    gguf.write_synthetic_llama(p, E=1024, H=8, Hkv=2, L=3, F=1792, V=1024, seed=23)
    prompt = [(7919 * i + 13) % 1024 for i in range(700)]
    out1, lg1, runs1 = greedy(bamd, p, 1024, prompt, 24, True)
    out0, lg0, runs0 = greedy(bamd, p, 1024, prompt, 24, False)
    assert runs0 == 0 and (runs1 >= 1) == OWN_QUEUE
    assert np.array_equal(out1, out0) and np.array_equal(bits(lg1), bits(lg0))


def test_single_token_decode_steps_on_the_own_queue(bamd):
    """bamd_decode with one token (the bridge's loop): the step's state travels through the pinned host inbox; every step's logits == the hipGraph path's"""
    path = os.path.join(GOLDEN, "tiny_b.gguf")
    prompt = [5, 9, 2, 77, 31, 8, 1, 40]
    res = {}
    for aql in (True, False):
        bamd.set_aql(aql)
        try:
            m = bamd.Model(path); ctx = bamd.Context(m, 128)
            lg = ctx.decode(prompt, 0)
            rows, n_past = [], len(prompt)
            for _ in range(30):
                t = int(np.argmax(lg))
                lg = ctx.decode([t], n_past); n_past += 1
                rows.append(lg)
            res[aql] = (np.array(rows), ctx.aql_runs())
            ctx.close(); m.close()
        finally:
            bamd.set_aql(True)
    assert res[False][1] == 0 and (res[True][1] >= 30) == OWN_QUEUE
    assert np.array_equal(bits(res[True][0]), bits(res[False][0]))


def test_many_replays_across_the_ring_wrap(bamd):
    """the queue's ring holds 16384 packets: 400 greedy calls of 20 steps on the tiny model write ~ 5 x that, one doorbell never spanning the wrap (bamd_aql.cpp)"""
    path = os.path.join(GOLDEN, "tiny_a.gguf")
    m = bamd.Model(path); ctx = bamd.Context(m, 128)
    first = None
    for i in range(400):
        ctx.decode([1, 2, 3, 4, 5, 6, 7, 8], 0)
        out, _ = ctx.generate_greedy(8, 20)
        if first is None:
            first = out.copy()
        assert np.array_equal(out, first), "replay %d differs" % i
    assert (ctx.aql_runs() >= 400) == OWN_QUEUE
    ctx.close(); m.close()
