"""Generates tests/golden/janus_kats.npz with the GENUINE reference Janus sampler (oracle/_ref/janus_ref, built by
`make -C oracle janusref`: cpp/janus.cpp, cpp/common/common.cpp and llama_sample_token of the reference, compiled in place): on
synthetic GGUFs whose vocabularies mix scripts (booster_amd.gguf.synthetic_janus_vocab; 30 100 tokens = the reference's Llama-2 id
table, 128 300 tokens = its Llama-3 rules), scripted logits / histories go through initJanus + sample_janus_token (the calls of
cpp/bridge.cpp:196 and :589).  Recorded: the per-token type and scale tables, a digest of the logits after the penalties (plus the
changed entries when they are few), the sampled token.  Build container only."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, ".."))
from booster_amd import gguf  # noqa: E402
from janus_cases import N_LAST, case_logits, digest  # noqa: E402

JANUS_REF = os.path.join(ROOT, "oracle", "_ref", "janus_ref")
SETS = [("l2", 30100, (0.96, 0.99, 0.96, 200), 40), ("l2b", 30100, (0.90, 0.97, 0.80, 24), 30), ("l3", 128300, (0.97, 0.99, 0.96, 200), 30)]
N_OV = 16


def token_classes(vocab):
    """rough script classes of the pieces, only to steer the scripted cases (the expected values come from the reference)"""
    ru, en, other = [], [], []
    for i, t in enumerate(vocab["tokens"]):
        if i < 259:
            continue
        body = t.replace("▁", "")
        if body and all("Ѐ" <= c <= "ӿ" for c in body):
            ru.append(i)
        elif body and all(c.isascii() and c.isalpha() for c in body):
            en.append(i)
        elif any(ord(c) > 0x7f for c in body):
            other.append(i)
    return np.array(ru), np.array(en), np.array(other)


def make_cases(vocab, rng, n):
    V = len(vocab["tokens"])
    ru, en, other = token_classes(vocab)
    cases = []
    for c in range(n):
        kind = c % 10
        seed = int(rng.integers(1, 2**31))
        top = 16.0 + float(rng.uniform(0, 8))
        k = int(rng.integers(2, 12))
        pool = ru if kind in (1, 2, 3) else (en if kind == 4 else np.arange(3, V))
        if kind == 8:
            pool = np.concatenate([[2], other[:20], ru[:8]])
        picks = rng.choice(pool, size=min(k, len(pool)), replace=False)
        ov_ids = np.full(N_OV, -1, np.int32); ov_vals = np.zeros(N_OV, np.float32)
        # a cluster near the top: ratios between ~0.9 and 1 so that both cut-offs split it somewhere
        ov_ids[:len(picks)] = picks
        ov_vals[:len(picks)] = (top * (1.0 - rng.uniform(0.0, 0.09, len(picks)))).astype(np.float32)
        negative = kind == 5                                    # every logit negative: the ratio test never cuts (whole vocabulary)
        if negative:
            ov_ids[:] = -1
        if kind == 6:
            ov_vals[:2] = np.float32(top)                       # two equal top logits
        if kind == 7:
            ov_ids[len(picks)] = 2; ov_vals[len(picks)] = np.float32(top * 0.995)    # EOS just below the top: the EOS boost decides
        gen = int(rng.integers(0, 60))                          # generated tokens so far
        prompt_len = int(rng.integers(1, 30))
        pos = prompt_len + gen
        mx = int(rng.integers(max(gen, 1), 200))
        hist_pool = np.concatenate([picks, rng.choice(pool, size=6), rng.integers(3, V, 6)])
        if kind in (7, 8):
            hist_pool = np.concatenate([hist_pool, [2, 2]])
        last = rng.choice(hist_pool, size=N_LAST).astype(np.int32)
        if kind in (1, 2):
            last[-1] = ru[int(rng.integers(0, len(ru)))]        # last token Cyrillic: the continuation rule and the x0.5 pass
        elif kind == 3:
            last[-1] = en[int(rng.integers(0, len(en)))]
        cases.append(dict(seed=seed, negative=int(negative), ov_ids=ov_ids, ov_vals=ov_vals, last=last, prompt_len=prompt_len, pos=pos, max=mx,
                          rng_seed=int(rng.integers(1, 2**31))))
    return cases


def main():
    rng = np.random.default_rng(17)
    out = {}
    for name, V, (scale, hi, lo, depth), n_cases in SETS:
        vocab = gguf.synthetic_janus_vocab(V)
        assert len(vocab["tokens"]) == V
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "janus.gguf")
            gguf.write_synthetic_llama(path, E=256, H=2, Hkv=1, L=1, F=256, V=V, seed=3, vocab=vocab)
            cases = make_cases(vocab, rng, n_cases)
            cin, cout = os.path.join(td, "cases.bin"), os.path.join(td, "out.bin")
            with open(cin, "wb") as f:
                f.write(struct.pack("<ifffi", len(cases), scale, hi, lo, depth))
                for c in cases:
                    f.write(struct.pack("<iiiiI", N_LAST, c["prompt_len"], c["pos"], c["max"], c["rng_seed"]))
                    f.write(c["last"].tobytes()); f.write(case_logits(c["seed"], V, c["negative"], c["ov_ids"], c["ov_vals"]).tobytes())
            subprocess.run([JANUS_REF, path, cin, cout], check=True)
            raw = open(cout, "rb").read()
        assert struct.unpack_from("<i", raw, 0)[0] == V
        off = 4
        types = np.frombuffer(raw, np.float32, V, off); off += 4 * V
        scales = np.frombuffer(raw, np.float32, V, off); off += 4 * V
        toks, digests, ch_ids, ch_vals = [], [], [], []
        for c in cases:
            toks.append(struct.unpack_from("<i", raw, off)[0]); off += 4
            after = np.frombuffer(raw, np.float32, V, off); off += 4 * V
            before = case_logits(c["seed"], V, c["negative"], c["ov_ids"], c["ov_vals"])
            changed = np.nonzero(after.view(np.uint32) != before.view(np.uint32))[0]
            digests.append(digest(after))
            ids = np.full(64, -1, np.int32); vals = np.zeros(64, np.float32)
            if len(changed) <= 64:
                ids[:len(changed)] = changed; vals[:len(changed)] = after[changed]
            ch_ids.append(ids); ch_vals.append(vals)
        assert off == len(raw)
        p = name + "_"
        out[p + "V"] = np.array([V]); out[p + "params"] = np.array([scale, hi, lo, depth], np.float64)
        out[p + "types"] = types.astype(np.uint8); out[p + "scales"] = scales
        for key in ("seed", "negative", "prompt_len", "pos", "max", "rng_seed"):
            out[p + key] = np.array([c[key] for c in cases], np.int64)
        out[p + "ov_ids"] = np.stack([c["ov_ids"] for c in cases]); out[p + "ov_vals"] = np.stack([c["ov_vals"] for c in cases])
        out[p + "last"] = np.stack([c["last"] for c in cases])
        out[p + "token"] = np.array(toks, np.int32); out[p + "digest"] = np.array(digests)
        out[p + "changed_ids"] = np.stack(ch_ids); out[p + "changed_vals"] = np.stack(ch_vals)
        print(name, "tokens", toks[:14], "types", sorted(set(types.tolist())), "scales", len(set(scales.tolist())))
    np.savez_compressed(os.path.join(HERE, "janus_kats.npz"), **out)
    print("wrote janus_kats.npz", os.path.getsize(os.path.join(HERE, "janus_kats.npz")))


if __name__ == "__main__":
    main()
