"""Prompt-evaluation rate of the batched MFMA path only (no cross-checks): for A/B runs of kernel variants (BAMD_LIB / BAMD_PREFILL_V).
usage: python tools/prefill_quick.py [n_prompt=512] [reps=4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import booster_amd as b
from booster_amd import gguf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
path = "/dev/shm/bamd_prefill_8b.gguf"
if not os.path.exists(path):
    gguf.write_synthetic_llama(path, E=4096, H=32, Hkv=8, L=32, F=14336, V=128256, seed=7, reuse_layers=True)
m = b.Model(path)
toks = [(7919 * i + 13) % 128256 for i in range(n)]
b.set_prefill_batch(1)
ctx = b.Context(m, 2048 if n <= 2044 else 4096)
ctx.decode(toks[:16], 0)
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter()
    for i in range(0, n, 512):
        ctx.decode(toks[i:i + 512], i)
    best = min(best, time.perf_counter() - t0)
print("%s V=%s %4d tokens: %8.2f ms  %9.1f tok/s" % (os.environ.get("BAMD_LIB", "default"), os.environ.get("BAMD_PREFILL_V", "2"), n, best * 1e3, n / best))
