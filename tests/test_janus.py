"""CPU: the host Janus sampler (booster_amd/csrc/bamd_bridge.cpp: init_janus, sample_janus) against fixtures recorded from the
genuine reference (tests/golden/gen_janus_kats.py: initJanus + sample_janus_token + llama_sample_token of cpp/janus.cpp, compiled in
place) — per-token type / scale tables, logits after the penalties (bit for bit), the token drawn with the same mt19937 seed."""
import ctypes as C
import os

import numpy as np
import pytest

import booster_amd
from booster_amd import gguf
from janus_cases import N_LAST, case_logits, digest

KATS = os.path.join(os.path.dirname(__file__), "golden", "janus_kats.npz")


@pytest.fixture(scope="module")
def L():
    lib = booster_amd.lib()
    lib.bamd_vocab_load.restype = C.c_void_p; lib.bamd_vocab_load.argtypes = [C.c_char_p]
    lib.bamd_vocab_free.argtypes = [C.c_void_p]
    lib.bamd_janus_test_new.restype = C.c_void_p
    lib.bamd_janus_test_new.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int]
    lib.bamd_janus_test_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bamd_janus_test_sample.restype = C.c_int
    lib.bamd_janus_test_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
    lib.bamd_janus_test_free.argtypes = [C.c_void_p]
    return lib


@pytest.mark.parametrize("name", ["l2", "l2b", "l3"])
def test_host_sampler_equals_reference(L, name, tmp_path):
    k = np.load(KATS)
    g = lambda key: k[name + "_" + key]
    V = int(g("V")[0]); scale, hi, lo, depth = g("params")
    vocab = gguf.synthetic_janus_vocab(V)
    path = str(tmp_path / "janus.gguf")
    gguf.write_synthetic_llama(path, E=256, H=2, Hkv=1, L=1, F=256, V=V, seed=3, vocab=vocab)
    vh = L.bamd_vocab_load(path.encode())
    assert vh
    j = L.bamd_janus_test_new(vh, float(scale), float(hi), float(lo), int(depth))
    try:
        types = np.zeros(V, np.float32); scales = np.zeros(V, np.float32)
        L.bamd_janus_test_tables(j, types.ctypes.data_as(C.c_void_p), scales.ctypes.data_as(C.c_void_p))
        assert np.array_equal(types.astype(np.uint8), g("types")), "token classes (tokType, janus.cpp:723-800)"
        bad = np.nonzero(scales.view(np.uint32) != g("scales").view(np.uint32))[0]
        assert len(bad) == 0, "scale table (initJanus) differs at ids %s: %s vs %s" % (bad[:8], scales[bad[:8]], g("scales")[bad[:8]])
        n = len(g("token"))
        for c in range(n):
            logits = case_logits(g("seed")[c], V, g("negative")[c], g("ov_ids")[c], g("ov_vals")[c])
            before = logits.copy()
            last = np.ascontiguousarray(g("last")[c], np.int32)
            tok = L.bamd_janus_test_sample(j, logits.ctypes.data_as(C.c_void_p), last.ctypes.data_as(C.c_void_p), N_LAST, int(g("prompt_len")[c]),
                                           int(g("pos")[c]), int(g("max")[c]), int(g("rng_seed")[c]))
            if digest(logits) != str(g("digest")[c]):
                ids = g("changed_ids")[c]; ids = ids[ids >= 0]
                mine = np.nonzero(logits.view(np.uint32) != before.view(np.uint32))[0]
                raise AssertionError("case %d: logits after the penalties differ; reference changed %s -> %s, here %s -> %s"
                                     % (c, ids[:8], g("changed_vals")[c][:8], mine[:8], logits[mine[:8]]))
            assert tok == int(g("token")[c]), "case %d: sampled token" % c
    finally:
        L.bamd_janus_test_free(j); L.bamd_vocab_free(vh)
