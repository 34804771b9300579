#!/usr/bin/env python3
"""tests/golden/abi_transcript.json: what the GENUINE reference bridge returns for a scripted session of the nine cgo symbols
(cpp/bridge.h:132-165, implementation cpp/bridge.cpp:697-835) — recorded with oracle/_ref/bridge_ref (oracle/harness/bridge_ref.cpp
linked against the reference's bridge.cpp / janus.cpp / common compiled in place) on the CPU path, on the deterministic synthetic
30 100-token SentencePiece-vocabulary model (booster_amd.gguf.synthetic_janus_vocab + write_synthetic_llama seed 21).  Janus runs with
hi = lo = 1.0, which leaves only the top candidate: the session is deterministic.  tests/test_gpu_bridge.py replays the same script
against libbooster_amd.so and compares every return value and every status() text.  Build container only."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
from booster_amd import gguf  # noqa: E402

MODEL = dict(E=512, H=4, Hkv=1, L=3, F=768, seed=21)
# round 6: the same session on a SIXTEEN-layer model (abi_transcript_l16.json), deep enough for Booster's split over eight devices
# (BOOSTER_GPUS with eight weights: seven hand-offs per evaluation) — tests/test_gpu_bridge.py::test_reference_abi_transcript_eight_stages
MODEL_L16 = dict(E=512, H=4, Hkv=1, L=16, F=768, seed=23)
N_VOCAB = 30100      # the reference's initJanus indexes its Llama-2 id table without bounds: the vocabulary must cover it (tests/golden/gen_janus_kats.py)


def script():
    hx = lambda s: s.encode().hex()
    return [
        "ctx 0 128 24 1.0 1.0",             # config mode: initContext first, init afterwards (server.go:532-553)
        "init",
        "status nojob", "count nojob",      # a job nobody started
        "infer 0 job-1 " + hx("the cat sat on the hat"),
        "status job-1", "count job-1", "seed job-1", "evalms job-1", "genms job-1",
        "infer 0 job-2 " + hx("the cat sat on the hat"),      # same prompt, fresh KV cache: the same text
        "status job-2", "count job-2",
        "infer 0 job-3 " + hx("a " * 400),                    # longer than n_ctx - 4: returns 0
        "status job-3", "count job-3",
        "stop 0",                                             # a stop BEFORE the job starts is cleared by it (bridge.cpp:186)
        "infer 0 job-4 " + hx("hello there"),
        "status job-4", "count job-4",
        "init",                                               # idempotent
        "infer 0 job-5 " + hx("<s>hi</s> there<unk>x"),       # special tokens are parsed (parse_special = true)
        "status job-5", "count job-5",
        "ctx 1 64 200 1.0 1.0",                               # a second pod: n_predict beyond the context -> stops at n_ctx - 4
        "infer 1 job-6 " + hx("the cat"),
        "status job-6", "count job-6",
        "infer 0 job-1 " + hx("hat"),                         # a job id used twice: the text starts over
        "status job-1", "count job-1",
    ]


def main(MODEL=MODEL, out_name="abi_transcript.json"):
    vocab = gguf.synthetic_janus_vocab(N_VOCAB)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "bridge.gguf")
        gguf.write_synthetic_llama(path, V=len(vocab["tokens"]), vocab=vocab, **MODEL)
        sp = os.path.join(td, "script.txt")
        lines = script()
        open(sp, "w").write("\n".join(lines) + "\n")
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "bridge_ref"), path, sp], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    out = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(out) == len(lines), (len(out), len(lines))
    json.dump(dict(model=MODEL, n_vocab=N_VOCAB, script=lines, results=out), open(os.path.join(HERE, out_name), "w"), indent=0)
    for l, o in zip(lines, out):
        print(l[:40].ljust(42), {k: (bytes.fromhex(v)[:60] if k == "hex" else v) for k, v in o.items() if k not in ("op", "job")})


if __name__ == "__main__":
    main()
    main(MODEL_L16, "abi_transcript_l16.json")
