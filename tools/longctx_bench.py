"""Decode rate at long context (BASELINE config 5 shape of the problem): 8B synthetic GGUF, n_ctx 8192, prompt of n tokens through the
batched prefill, then greedy decode steps with n_kv ~ n (three-kernel attention path).  usage: longctx_bench.py [n_prompt=7936] [n_ctx=8192] (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import booster_amd as b
from booster_amd import gguf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7936
path = "/dev/shm/bamd_prefill_8b.gguf"
if not os.path.exists(path):
    gguf.write_synthetic_llama(path, E=4096, H=32, Hkv=8, L=32, F=14336, V=128256, seed=7, reuse_layers=True)
n_ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
m = b.Model(path); ctx = b.Context(m, n_ctx)
toks = [(7919 * i + 13) % 128256 for i in range(n)]
t0 = time.perf_counter()
for i in range(0, n, 512):
    ctx.decode(toks[i:i + 512], i)
tp = time.perf_counter() - t0
print("prefill %d tokens: %.1f ms (%.0f tok/s)" % (n, tp * 1e3, n / tp))
ctx.generate_greedy(n, 8)
out, ms = ctx.generate_greedy(n + 8, 64)
kvb = 131072 * (n + 40)
print("decode at n_kv ~ %d: %.3f ms/token = %.1f tok/s; bytes/token %.2f GB -> %.0f GB/s (%.1f %% of 8 TB/s)" % (n + 40, ms / 64, 64e3 / ms, (4.6174e9 + kvb) / 1e9, (4.6174e9 + kvb) / (ms / 64 * 1e-3) / 1e9, (4.6174e9 + kvb) / (ms / 64 * 1e-3) / 8e12 * 100))
