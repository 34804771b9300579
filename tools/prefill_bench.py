"""Prompt-evaluation rate of the synthetic Llama-3-8B Q4_K_M GGUF: batched prefill kernels vs token by token (GPU box only).
usage: python tools/prefill_bench.py [n_prompt=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import booster_amd as b
from booster_amd import gguf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
path = "/dev/shm/bamd_prefill_8b.gguf"
if not os.path.exists(path):
    gguf.write_synthetic_llama(path, E=4096, H=32, Hkv=8, L=32, F=14336, V=128256, seed=7, reuse_layers=True)
m = b.Model(path)
toks = [(7919 * i + 13) % 128256 for i in range(n)]
res = {}
for mode, reps in ((1, 3), (2, 2), (0, 1)):
    b.set_prefill_batch(mode)
    ctx = b.Context(m, 2048 if n <= 2044 else 4096)
    ctx.decode(toks[:16], 0)                               # warm-up (buffer allocation)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for i in range(0, n, 512):
            lg = ctx.decode(toks[i:i + 512], i)
        best = min(best, time.perf_counter() - t0)
    res[mode] = (best, lg.copy())
    ctx.close()
    print("%-22s %4d tokens: %8.1f ms  %9.1f tok/s" % ({1: "batched (MFMA)", 2: "batched (integer dot)", 0: "token-by-token"}[mode], n, best * 1e3, n / best))
print("logits bit-identical:", bool(np.array_equal(res[1][1].view(np.uint32), res[0][1].view(np.uint32))) and bool(np.array_equal(res[2][1].view(np.uint32), res[0][1].view(np.uint32))))
flops = 2 * 6.979e9 * n
print("batched: %.1f TFLOP/s of mat-mul work (13.96 GFLOP/token, SURVEY 8d)" % (flops / res[1][0] / 1e12))
