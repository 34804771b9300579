#!/usr/bin/env python3
"""Full-size fixtures from the GENUINE reference (build container only; needs /root/reference and oracle/_ref/ref_run).

For every BASELINE.json configuration the deterministic synthetic GGUF of that shape (booster_amd.gguf.write_synthetic_llama,
numpy Generator seed 7 — the GPU box regenerates the same bytes; the fixture carries the file's size, a sha256 of its first 64 MiB and
an xxh3-128 of the WHOLE file) is evaluated by the reference CPU path (oracle/_ref/ref_run -> llama_decode, cpp/src/llama.cpp:14537) on the synthetic
prompt tok[i] = (7919 i + 13) mod V, greedy.  What is committed is DATA ONLY: arg-max tokens, 32 probe logits per step, the top
logit and a 64-bit digest of all logits per step (tests/golden/fullsize_<cfg>.bgld, a few KB each).

    python tests/golden/gen_fullsize_fixtures.py [cfg ...]      cfg in: 8b 8b_prefill2048 70b_stage 70b_full m7q6k_8k shift selfextend yarn l2_7b l32_3b
"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from booster_amd import gguf  # noqa: E402

Q6 = gguf.Q6_K
CONFIGS = {
    # name: (model kwargs, n_prompt, n_decode, n_ctx, threads)
    "8b": (dict(E=4096, H=32, Hkv=8, L=32, F=14336, V=128256, theta=500000.0), 128, 128, 512),
    "8b_prefill2048": (dict(E=4096, H=32, Hkv=8, L=32, F=14336, V=128256, theta=500000.0), 2048, 8, 4096),
    # one pipeline stage of Llama-3-70B Q4_K_M (the last: 10 layers + output layer), 70B widths, Q5_K attn_v outside the "more bits" layers
    "70b_stage": (dict(E=8192, H=64, Hkv=8, L=10, F=28672, V=128256, theta=500000.0, type_fn="70b"), 8, 16, 256),
    # config 4 at FULL depth: the whole Llama-3-70B Q4_K_M (80 layers, 41.9 GB of weights, attn_v Q5_K / Q6_K) — the N = 1 point of the layer split; the file is
    # the one `bench.py --model 70b` writes
    "70b_full": (dict(E=8192, H=64, Hkv=8, L=80, F=28672, V=128256, theta=500000.0, type_fn="70b_full"), 16, 8, 64),
    "m7q6k_8k": (dict(E=4096, H=32, Hkv=8, L=32, F=14336, V=32000, theta=10000.0, type_fn="q6k", embd_type=Q6), 8064, 16, 8192),
    # generation past n_ctx with Booster's context shift (cpp/bridge.cpp:487-503; ref_run's n_keep argument).  A small GQA model,
    # n_ctx 96, 150 generated tokens: three shifts, holes refilled in cell order, K rows re-rotated in place three times over
    "shift": (dict(E=512, H=8, Hkv=2, L=3, F=768, V=512, theta=500000.0), 40, 150, 96),
    # YaRN rope scaling (llama.rope.scaling.type = "yarn", factor 4, original context 64, attn_factor 1.25): positions on both sides of the
    # original context, a batched prompt and single-token steps
    "yarn": (dict(E=512, H=8, Hkv=2, L=3, F=768, V=512, theta=10000.0, n_ctx_train=256, rope_scaling=dict(type="yarn", factor=4.0, orig_ctx=64, attn_factor=1.25)), 90, 40, 256),
    # Self-Extend (cpp/bridge.cpp:507-523 with ga_n = 2, ga_w = 16; ref_run's n_keep = -(100 ga_n + ga_w)): positions compressed window by
    # window, every cell its own rotation delta
    "selfextend": (dict(E=512, H=8, Hkv=2, L=3, F=768, V=512, theta=500000.0), 40, 60, 128),
    # two more architectures at FULL size (not BASELINE configurations: the kernels' other shape classes).  Llama-2-7B Q4_K_M: no GQA, n_ff 11008
    # (43 super-blocks: split-K with uneven shares per wave), SPM-sized vocabulary.  Llama-3.2-3B Q4_K_M: n_embd 3072 (12 super-blocks), three
    # query heads per KV head, rope_freqs, and NO output.weight — lm_head runs on the Q6_K token_embd (tied embeddings, llama.cpp:6070-6076)
    "l2_7b": (dict(E=4096, H=32, Hkv=32, L=32, F=11008, V=32000, theta=10000.0, n_ctx_train=4096), 64, 64, 256),
    "l32_3b": (dict(E=3072, H=24, Hkv=8, L=28, F=8192, V=128256, theta=500000.0, rope_freqs=True, tied=True, embd_type=Q6), 64, 64, 256),
}
N_KEEP = {"shift": 8, "selfextend": -216}


def type_fn_of(tag, L):
    if tag == "q6k":
        return lambda name, il: gguf.Q6_K
    if tag == "70b_full":
        return lambda name, il: gguf.q4_k_m_type_70b(name, il, L)
    if tag == "70b":
        # Q4_K_M recipe of an 80-layer model seen from its LAST stage (layers 70..79): attn_v Q6_K in the "more bits" layers, else Q5_K
        def f(name, il):
            t = gguf.q4_k_m_type(name, 70 + il, 80)
            if name == "attn_v" and t == gguf.Q4_K:
                return gguf.Q5_K
            return t
        return f
    return None


def model_path(cfg, d="/dev/shm"):
    if cfg.startswith("8b"):
        return os.path.join(d, "bamd_llama3_8b_q4_k_m_synth.gguf")      # the file bench.py and tests/test_gpu_fullsize.py use
    if cfg == "70b_full":
        return os.path.join(d, "bamd_llama3_70b_q4_k_m_synth.gguf")     # the file `bench.py --model 70b` uses
    return os.path.join(d, "bamd_fx_%s.gguf" % cfg)


def ensure_model(cfg):
    kw = dict(CONFIGS[cfg][0])
    p = model_path(cfg)
    if not os.path.exists(p + ".done"):
        tf = kw.pop("type_fn", None)
        gguf.write_synthetic_llama(p, seed=7, reuse_layers=True, type_fn=type_fn_of(tf, kw["L"]), **kw)
        open(p + ".done", "w").write("ok")
    return p


def file_digest(p):
    """(sha256 of the first 64 MiB, size) — the round-2 check, kept — see file_digest_full for the whole file"""
    h = hashlib.sha256()
    with open(p, "rb") as f:
        h.update(f.read(64 << 20))
    return h.hexdigest(), os.path.getsize(p)


def file_digest_full(p):
    """xxh3-128 of the WHOLE file (several GB/s: a 5 GB model in about a second): a generator drift in any tensor says "wrong file", not
    "logits differ" """
    import xxhash
    h = xxhash.xxh3_128()
    with open(p, "rb") as f:
        while True:
            b = f.read(64 << 20)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def refresh_digests(cfgs):
    """rewrite the digest line of the committed side files from the GGUFs the reference evaluated (they are still in /dev/shm), without
    re-running the reference:  python tests/golden/gen_fullsize_fixtures.py --digests [cfg ...]"""
    for cfg in cfgs:
        p = ensure_model(cfg)
        side = os.path.join(ROOT, "tests", "golden", "fullsize_%s.bgld.txt" % cfg)
        lines = open(side).read().splitlines()
        dg, sz = file_digest(p)
        old = dict(t.split("=", 1) for t in lines[-1].split() if "=" in t)
        assert int(old["gguf_bytes"]) == sz and old["gguf_sha256_first64MiB"] == dg, "%s: the file in /dev/shm is not the one the fixture was recorded on" % cfg
        lines[-1] = "gguf_bytes=%d gguf_sha256_first64MiB=%s gguf_xxh3_128=%s" % (sz, dg, file_digest_full(p))
        open(side, "w").write("\n".join(lines) + "\n")
        print(cfg, lines[-1], flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--digests":
        return refresh_digests(sys.argv[2:] or list(CONFIGS))
    cfgs = sys.argv[1:] or list(CONFIGS)
    threads = int(os.environ.get("REF_THREADS", str(os.cpu_count() or 8)))
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_run")
    for cfg in cfgs:
        _, n_prompt, n_decode, n_ctx = CONFIGS[cfg]
        p = ensure_model(cfg)
        out = os.path.join(ROOT, "tests", "golden", "fullsize_%s.bgld" % cfg)
        t0 = time.time()
        extra = [str(N_KEEP[cfg])] if cfg in N_KEEP else []
        r = subprocess.run([exe, p, str(threads), str(n_prompt), str(n_decode), str(n_ctx), out] + extra, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
        dg, sz = file_digest(p)
        line = r.stdout.decode().strip().splitlines()[-1]
        with open(out + ".txt", "w") as f:
            f.write("%s\ngguf_bytes=%d gguf_sha256_first64MiB=%s gguf_xxh3_128=%s\n" % (line, sz, dg, file_digest_full(p)))
        print(cfg, line, "(%.0f s)" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
