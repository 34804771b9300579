"""Server-level rate through the nine-symbol bridge (what Booster's Go code would see): initContext + doInference on an 8B-shaped
synthetic GGUF with a 128256-token byte-level BPE vocabulary; Janus sampling on the host (GPU box only).
usage: python tools/bridge_bench.py [n_predict=128]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from booster_amd import gguf, build
n_predict = int(sys.argv[1]) if len(sys.argv) > 1 else 128
path = "/dev/shm/bamd_bridge_8b.gguf"
if not os.path.exists(path):
    v = gguf.synthetic_bpe_vocab(n_merges=2000)
    ctrl = v["tokens"][-4:]; toks = v["tokens"][:-4]; types = v["types"][:-4]
    while len(toks) < 128256 - 4:
        toks.append("Ġfill%d" % len(toks)); types.append(1)            # filler tokens (never produced by a merge)
    v["tokens"] = toks + ctrl; v["types"] = types + [3] * 4
    n = len(v["tokens"]); v["bos_token_id"] = n - 4; v["eos_token_id"] = n - 3
    gguf.write_synthetic_llama(path, E=4096, H=32, Hkv=8, L=32, F=14336, V=n, seed=7, reuse_layers=True, vocab=v)
import torch  # noqa: F401  (one HIP runtime per process)
L = C.CDLL(build.build())
i, f, u = C.c_int, C.c_float, C.c_uint32
L.initContext.restype = C.c_void_p
L.initContext.argtypes = [i, C.c_char_p, i, i, i, i, i, i, i, i, C.c_int32, f, f, f, i, f, f, f, i, C.c_int32, C.c_int32, f, f, f, u, C.c_char_p]
L.doInference.restype = C.c_int64; L.doInference.argtypes = [i, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
for nme in ("promptEval", "getPromptTokenCount", "timing"):
    getattr(L, nme).restype = C.c_int64; getattr(L, nme).argtypes = [C.c_char_p]
L.status.restype = C.c_char_p; L.status.argtypes = [C.c_char_p]
ctx = L.initContext(0, path.encode(), 4, 512, 100, 0, 0, 0, 2048, n_predict, 0, 0.0, 0.0, 0.8, 40, 0.9, 1.0, 1.1, 64, 1, 200, 0.97, 0.99, 0.96, 42, b"")
assert ctx, "initContext failed"
L.init(b"", b"")
prompt = ("the quick brown fox jumps over the lazy dog and then " * 24).encode()
for job in (b"warm", b"run"):
    t0 = time.perf_counter()
    n = L.doInference(0, ctx, job, b"s", prompt)
    dt = time.perf_counter() - t0
    np_ = L.getPromptTokenCount(job)
    print("%s: %d tokens processed (%d prompt) in %.1f ms  ->  %.1f generated tok/s end to end; bridge timing(): %d ms/token, promptEval(): %d ms/token"
          % (job.decode(), n, np_, dt * 1e3, (n - np_) / dt, L.timing(job), L.promptEval(job)))
