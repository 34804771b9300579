"""GPU: the nine cgo symbols (include/booster_bridge.h) driven the way pkg/server/server.go drives them:
initContext -> init -> doInference on one thread while status/getPromptTokenCount are polled, then the counters."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from booster_amd import gguf

pytestmark = pytest.mark.gpu


def bind(bamd):
    L = bamd.lib()
    f, i, u = C.c_float, C.c_int, C.c_uint32
    L.init.argtypes = [C.c_char_p, C.c_char_p]; L.init.restype = None
    L.initContext.restype = C.c_void_p
    L.initContext.argtypes = [i, C.c_char_p, i, i, i, i, i, i, i, i, C.c_int32, f, f, f, i, f, f, f, i, C.c_int32, C.c_int32, f, f, f, u, C.c_char_p]
    L.doInference.restype = C.c_int64; L.doInference.argtypes = [i, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
    L.stopInference.argtypes = [i]; L.stopInference.restype = None
    L.status.restype = C.c_char_p; L.status.argtypes = [C.c_char_p]
    for n in ("promptEval", "getPromptTokenCount", "timing"):
        getattr(L, n).restype = C.c_int64; getattr(L, n).argtypes = [C.c_char_p]
    L.getSeed.restype = C.c_uint32; L.getSeed.argtypes = [C.c_char_p]
    L.bamd_bridge_tokenize.argtypes = [C.c_void_p, C.c_char_p, i, i, C.c_void_p, i]
    return L


@pytest.fixture(scope="module")
def lib(bamd):
    return bind(bamd)


def make_model(tmp_path, name, L=3):
    vocab = gguf.synthetic_spm_vocab()
    p = str(tmp_path / name)
    gguf.write_synthetic_llama(p, E=512, H=4, Hkv=1, L=L, F=768, V=len(vocab["tokens"]), seed=21, vocab=vocab)
    return p, vocab


EIGHT_GPUS_L16 = "3,2,2,2,2,2,2,2"          # sixteen layers + the output layer over eight devices: 3 | 2 | 2 | 2 | 2 | 2 | 2 | 1 + output


def stage_layout(lib, ctx):
    out = np.zeros(3 * 8, np.int32)
    lib.bamd_bridge_stage_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    n = lib.bamd_bridge_stage_layout(ctx, out.ctypes.data_as(C.c_void_p), 8)
    return [tuple(int(v) for v in out[3 * s:3 * s + 3]) for s in range(n)]


def ctx_args(idx, path, gpus, n_ctx, predict, hi=1.0, lo=1.0):
    return (idx, path.encode(), 4, 512, gpus[0], gpus[1], gpus[2], gpus[3], n_ctx, predict, 0, 0.0, 0.0, 0.8, 40, 0.9, 1.0, 1.1, 64,
            1, 200, 0.97, hi, lo, 42, b"")


def test_scripted_session(lib, bamd, tmp_path):
    path, vocab = make_model(tmp_path, "bridge.gguf")
    ctx = lib.initContext(*ctx_args(0, path, (100, 0, 0, 0), 128, 24))
    assert ctx, "initContext failed"
    lib.init(b"", b"")                                   # config mode calls init AFTER initContext (server.go:532-553)
    prompt = b"the cat sat on the hat"
    ids = np.zeros(256, np.int32)
    n_prompt = lib.bamd_bridge_tokenize(ctx, prompt, 0, 1, ids.ctypes.data_as(C.c_void_p), 256)
    seen = []
    done = threading.Event()

    def poll():
        while not done.is_set():
            seen.append(lib.status(b"job-1"))
            lib.getPromptTokenCount(b"job-1")
            time.sleep(0.001)
    t = threading.Thread(target=poll); t.start()
    n = lib.doInference(0, ctx, b"job-1", b"sess", prompt)
    done.set(); t.join()
    assert lib.getPromptTokenCount(b"job-1") == n_prompt
    text = lib.status(b"job-1")
    assert text.startswith(b" the cat sat on the hat") or text.startswith(prompt)    # SPM pieces re-insert the leading space
    # n_p_eval + n_eval: the prompt batch, then one eval per generated token except the last one (never fed back)
    assert n_prompt + 1 <= n <= n_prompt + 24
    assert lib.promptEval(b"job-1") >= 0 and lib.timing(b"job-1") >= 0
    assert lib.getSeed(b"job-1") > 1_600_000_000
    assert all(text.startswith(s) for s in seen if s)    # every polled snapshot is a prefix of the final text
    # same prompt again: Janus with hi = lo = 1.0 collapses to the arg-max -> deterministic text
    n2 = lib.doInference(0, ctx, b"job-2", b"sess", prompt)
    assert n2 == n and lib.status(b"job-2") == text
    # prompt longer than n_ctx - 4 -> 0
    assert lib.doInference(0, ctx, b"job-3", b"", b"a " * 400) == 0


def test_layer_split_on_one_device_and_stop(lib, bamd, tmp_path):
    """settings that would need a CPU path fail cleanly; a split named for more GPUs than present lands on the devices that exist;
    stopInference ends a long generation."""
    path, vocab = make_model(tmp_path, "bridge2.gguf")
    assert not lib.initContext(*ctx_args(1, path, (0, 0, 0, 0), 64, 8))      # no CPU path
    assert not lib.initContext(*ctx_args(1, path, (2, 0, 0, 0), 64, 8))      # sum <= n_layer: partial offload unsupported
    if bamd.device_count() < 2:                                            # named for two GPUs, one present: all on device 0, like the reference
        ctx2 = lib.initContext(*ctx_args(6, path, (2, 2, 0, 0), 64, 8))
        assert ctx2 and lib.doInference(6, ctx2, b"job-22", b"", b"hello") > 0
    ctx = lib.initContext(*ctx_args(1, path, (10, 0, 0, 0), 2048, 1500))
    assert ctx
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("n", lib.doInference(1, ctx, b"job-long", b"", b"hello there")))
    th.start(); time.sleep(0.15); lib.stopInference(1); th.join(timeout=60)
    assert not th.is_alive() and 0 < res["n"] < 1500


@pytest.mark.parametrize("name,idx", [("l2", 2), ("l2b", 3), ("l3", 4)])
def test_device_sampler_equals_reference(lib, bamd, tmp_path, name, idx):
    """the Janus penalties + shortlist on the device (bamd_logits_shortlist) against the fixtures recorded from the genuine
    reference sampler (tests/golden/gen_janus_kats.py): logits after the penalties bit for bit, the same token from the same mt19937
    seed — through the device shortlist and through the host sampler, including the cases that must fall back to the full sort."""
    import os
    if os.environ.get("BAMD_JANUS_GPU") == "0":
        pytest.skip("the device sampler is switched off by the environment (the host sampler has its own fixtures: tests/test_janus.py)")
    from janus_cases import N_LAST, case_logits, digest
    k = np.load(os.path.join(os.path.dirname(__file__), "golden", "janus_kats.npz"))
    g = lambda key: k[name + "_" + key]
    V = int(g("V")[0]); scale, hi, lo, depth = g("params")
    vocab = gguf.synthetic_janus_vocab(V)
    path = str(tmp_path / "janus.gguf")
    gguf.write_synthetic_llama(path, E=256, H=2, Hkv=1, L=1, F=256, V=V, seed=3, vocab=vocab)
    ctx = lib.initContext(idx, path.encode(), 4, 512, 100, 0, 0, 0, 128, 16, 0, 0.0, 0.0, 0.8, 40, 0.9, 1.0, 1.1, 64,
                          1, int(depth), float(scale), float(hi), float(lo), 42, b"")
    assert ctx
    lib.bamd_bridge_sample_test.restype = C.c_int
    lib.bamd_bridge_sample_test.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    counts = np.zeros(2, np.int64)
    n = len(g("token"))
    for c in range(n):
        logits = case_logits(g("seed")[c], V, g("negative")[c], g("ov_ids")[c], g("ov_vals")[c])
        last = np.ascontiguousarray(g("last")[c], np.int32)
        for device in (1, 0):
            after = np.zeros(V, np.float32)
            tok = lib.bamd_bridge_sample_test(ctx, logits.ctypes.data_as(C.c_void_p), last.ctypes.data_as(C.c_void_p), N_LAST, int(g("prompt_len")[c]),
                                              int(g("pos")[c]), int(g("max")[c]), int(g("rng_seed")[c]), device, after.ctypes.data_as(C.c_void_p),
                                              counts.ctypes.data_as(C.c_void_p))
            assert digest(after) == str(g("digest")[c]), "case %d device=%d: logits after the penalties" % (c, device)
            assert tok == int(g("token")[c]), "case %d device=%d: sampled token" % (c, device)
    assert counts[0] >= n // 2 and counts[1] >= 1 and counts[0] + counts[1] == n     # most draws from the device shortlist, the tie / negative cases through the host path


def test_device_sampler_fallbacks(lib, bamd, tmp_path):
    """device shortlist vs host sampler where the device path must hand over: more candidates inside the cut-off than its buffer holds
    (1024), and no penalties at all (first generated token); same token, same logits after the penalties."""
    import os
    if os.environ.get("BAMD_JANUS_GPU") == "0":
        pytest.skip("the device sampler is switched off by the environment")
    V = 30100
    vocab = gguf.synthetic_janus_vocab(V)
    path = str(tmp_path / "janus_fb.gguf")
    gguf.write_synthetic_llama(path, E=256, H=2, Hkv=1, L=1, F=256, V=V, seed=3, vocab=vocab)
    ctx = lib.initContext(5, path.encode(), 4, 512, 100, 0, 0, 0, 128, 16, 0, 0.0, 0.0, 0.8, 40, 0.9, 1.0, 1.1, 64, 1, 200, 0.96, 0.99, 0.90, 42, b"")
    assert ctx
    lib.bamd_bridge_sample_test.restype = C.c_int
    lib.bamd_bridge_sample_test.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(99)
    counts = np.zeros(2, np.int64)
    for case in range(6):
        logits = rng.normal(2.0, 1.0, V).astype(np.float32)
        if case < 3:                                     # 3000 candidates within 0.5 % of a unique top: inside either cut-off
            ids = rng.choice(V - 300, 3000, replace=False) + 300
            logits[ids] = (30.0 * (1.0 - rng.uniform(0.0, 0.004, 3000))).astype(np.float32)
            logits[ids[0]] = np.float32(30.01)
        else:
            logits[int(rng.integers(300, V))] = np.float32(25.0)
        last = rng.integers(300, V, 64).astype(np.int32)
        prompt_len, pos = 10, (10 if case >= 3 else 40)   # pos == prompt_len: depth 0, only the EOS factor (x 1.0)
        res = []
        for device in (1, 0):
            after = np.zeros(V, np.float32)
            tok = lib.bamd_bridge_sample_test(ctx, logits.ctypes.data_as(C.c_void_p), last.ctypes.data_as(C.c_void_p), 64, prompt_len, pos, 100, 1234 + case,
                                              device, after.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p))
            res.append((tok, after))
        assert res[0][0] == res[1][0] >= 0, "case %d" % case
        assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32)), "case %d: logits after the penalties" % case
    assert counts[1] >= 3 and counts[0] >= 3            # the crowded cases went through the host path, the others stayed on the device


def test_long_prompt_pipelines_through_three_stages(lib, bamd, tmp_path, monkeypatch):
    """a prompt of five micro-batches (> 2048 tokens, n_batch 512) through three layer-split stages: the micro-batches are enqueued
    back to back — double-buffered hand-off, no host wait until the logits are needed (GGML_SCHED_MAX_COPIES, ggml-backend.c:1030) — and
    the text must be the one the single-stage pod produces (Janus hi = lo = 1: arg-max, deterministic)"""
    path, vocab = make_model(tmp_path, "bridge_long.gguf")
    prompt = (b"the cat sat on the hat and the hen ate the ham " * 120)[:2900]
    texts = []
    for pod, gpus in ((0, None), (1, "1,1,2")):            # (pods 0 and 1 of the earlier tests are replaced)
        if gpus:
            monkeypatch.setenv("BOOSTER_GPUS", gpus)
            if bamd.device_count() < 3:
                monkeypatch.setenv("BAMD_VIRTUAL_DEVICES", "3")
        ctx = lib.initContext(*ctx_args(pod, path, (100, 0, 0, 0), 4096, 12))
        assert ctx
        job = ("long-%d" % pod).encode()
        n = lib.doInference(pod, ctx, job, b"", prompt)
        assert n > 2048 + 1, "the prompt must span at least five micro-batches (%d evaluations)" % n
        texts.append((n, lib.status(job)))
    assert texts[0] == texts[1]


def test_long_prompt_pipelines_through_eight_stages(lib, bamd, tmp_path, monkeypatch):
    """the EIGHT-stage schedule of BASELINE config 4 inside one process — BOOSTER_GPUS with eight weights over eight devices (virtual ones on a
    one-GPU box: eight stage streams, seven event-ordered hand-off copies per evaluation, double-buffered prompt micro-batches) — on a sixteen-layer model: a prompt
    of six micro-batches and the generated text equal the single-stage pod's"""
    path, vocab = make_model(tmp_path, "bridge_l16.gguf", L=16)
    prompt = (b"the cat sat on the hat and the hen ate the ham " * 120)[:2900]
    texts = []
    for pod, gpus in ((0, None), (1, EIGHT_GPUS_L16)):
        if gpus:
            monkeypatch.setenv("BOOSTER_GPUS", gpus)
            if bamd.device_count() < 8:
                monkeypatch.setenv("BAMD_VIRTUAL_DEVICES", "8")
        ctx = lib.initContext(*ctx_args(pod, path, (100, 0, 0, 0), 4096, 12))
        assert ctx
        lay = stage_layout(lib, ctx)
        if gpus:
            assert lay == [(0, 0, 3)] + [(d, 1 + 2 * d, 3 + 2 * d) for d in range(1, 7)] + [(7, 15, 16)], lay
        else:
            assert lay == [(0, 0, 16)]
        job = ("long8-%d" % pod).encode()
        n = lib.doInference(pod, ctx, job, b"", prompt)
        assert n > 2048 + 1, "the prompt must span at least five micro-batches (%d evaluations)" % n
        texts.append((n, lib.status(job)))
    assert texts[0] == texts[1]


@pytest.mark.parametrize("split", ["one stage", "eight stages"])
def test_reference_abi_transcript_sixteen_layers(lib, bamd, tmp_path, split, monkeypatch):
    """the same scripted session of the nine symbols recorded from the GENUINE reference bridge on a SIXTEEN-layer model (tests/golden/abi_transcript_l16.json),
    replayed on one stage and through Booster's split over eight devices (BOOSTER_GPUS=3,2,2,2,2,2,2,2: seven hand-offs per evaluation): every return value,
    status() text and token count the reference's, unchanged by the split"""
    if split == "eight stages":
        monkeypatch.setenv("BOOSTER_GPUS", EIGHT_GPUS_L16)
        if bamd.device_count() < 8:
            monkeypatch.setenv("BAMD_VIRTUAL_DEVICES", "8")
    ctxs = replay_transcript(lib, bamd, tmp_path, "l16" + split[:3], "abi_transcript_l16.json")
    for c in ctxs.values():
        assert len(stage_layout(lib, c)) == (8 if split == "eight stages" else 1)


@pytest.mark.parametrize("split", ["one stage", "three stages"])
def test_reference_abi_transcript(lib, bamd, tmp_path, split, monkeypatch):
    """the nine cgo symbols replayed against the transcript recorded from the GENUINE reference bridge
    (tests/golden/abi_transcript.json <- tests/golden/gen_abi_transcript.py: cpp/bridge.cpp + cpp/janus.cpp compiled in place, CPU
    path): every doInference return value (n_p_eval + n_eval, 0 for a prompt beyond n_ctx - 4), every status() text byte for byte
    (prompt echo + generated pieces; a reused job id; a job that never ran), every getPromptTokenCount, on two pods.  The model is
    the same deterministic synthetic GGUF; here it runs on the GPU (gpu1 = 100).
    "three stages": the same session through Booster's layer split — BOOSTER_GPUS=2,1,1 over three devices (virtual ones on a box
    with fewer GPUs: BAMD_VIRTUAL_DEVICES), i.e. the event-ordered multi-stage path with its hand-off copies and stage graphs —
    must give the reference's results unchanged."""
    if split == "three stages":
        monkeypatch.setenv("BOOSTER_GPUS", "2,1,1")
        if bamd.device_count() < 3:
            monkeypatch.setenv("BAMD_VIRTUAL_DEVICES", "3")
    replay_transcript(lib, bamd, tmp_path, split[:3])


def replay_transcript(lib, bamd, tmp_path, tag, fixture="abi_transcript.json"):
    import json
    import os
    t = json.load(open(os.path.join(os.path.dirname(__file__), "golden", fixture)))
    vocab = gguf.synthetic_janus_vocab(t["n_vocab"])
    path = str(tmp_path / "abi.gguf")
    gguf.write_synthetic_llama(path, V=len(vocab["tokens"]), vocab=vocab, **t["model"])
    ctxs = {}
    pod_of = {0: 5, 1: 7}                               # the other tests of this module use pods 0..4 and 6
    for line, want in zip(t["script"], t["results"]):
        p = line.split()
        op = p[0]
        if op == "ctx":
            idx, n_ctx, n_predict, hi, lo = int(p[1]), int(p[2]), int(p[3]), float(p[4]), float(p[5])
            ctxs[idx] = lib.initContext(*ctx_args(pod_of[idx], path, (100, 0, 0, 0), n_ctx, n_predict, hi, lo))
            assert bool(ctxs[idx]) == bool(want["ok"]), line
        elif op == "init":
            lib.init(b"", b"")
        elif op == "infer":
            idx = int(p[1]); pod = pod_of[idx]
            got = lib.doInference(pod, ctxs[idx], ("abi-%s-" % tag + p[2]).encode(), b"sess", bytes.fromhex(p[3]))
            assert got == want["ret"], "%s: doInference returned %d, the reference %d" % (line[:40], got, want["ret"])
        elif op == "status":
            got = lib.status(("abi-%s-" % tag + p[1]).encode())
            assert got == bytes.fromhex(want["hex"]), "%s: status() differs\n ours %r\n ref  %r" % (line, got, bytes.fromhex(want["hex"]))
        elif op == "count":
            assert lib.getPromptTokenCount(("abi-%s-" % tag + p[1]).encode()) == want["ret"], line
        elif op == "stop":
            lib.stopInference(pod_of[int(p[1])])
        elif op == "seed":
            assert (lib.getSeed(("abi-%s-" % tag + p[1]).encode()) != 0) == bool(want["nonzero"])
        elif op == "evalms":
            assert (lib.promptEval(("abi-%s-" % tag + p[1]).encode()) >= 0) == bool(want["nonneg"])
        elif op == "genms":
            assert (lib.timing(("abi-%s-" % tag + p[1]).encode()) >= 0) == bool(want["nonneg"])
    return ctxs
