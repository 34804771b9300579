"""BASELINE.json configs 4 and 5 on ONE MI355X (GPU box).
  m7q6k : Mistral-7B shapes, every matrix Q6_K (config 5), n_ctx 8192 — greedy decode at a short and at a long sequence
  70b   : the last stage of Llama-3-70B Q4_K_M split over 8 GPUs (config 4): 10 of the 80 layers + output layer — prompt micro-batch and
          greedy decode of that stage alone (what one of the eight GPUs does per token)
  l2-7b : Llama-2-7B Q4_K_M shapes (MHA: 32 KV heads, n_ff 11008 = 43 super-blocks, vocabulary 32000) — the model family the reference's Janus id
          table was written for; not a BASELINE configuration
usage: config_bench.py m7q6k|70b|l2-7b"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import booster_amd as b
from booster_amd import gguf

which = sys.argv[1]


def run(path, V, n_ctx, n_long, kv_bytes_per_pos, W):
    m = b.Model(path); ctx = b.Context(m, n_ctx)
    toks = [(7919 * i + 13) % V for i in range(max(n_long, 512))]
    t0 = time.perf_counter(); ctx.decode(toks[:512], 0); ctx.decode(toks[:512], 0); tp = (time.perf_counter() - t0) / 2
    print("prompt micro-batch of 512: %.1f ms (%.0f tok/s incl. the first-call buffer allocation in one of the two runs)" % (tp * 1e3, 512 / tp))
    t0 = time.perf_counter(); ctx.decode(toks[:512], 0); tp = time.perf_counter() - t0
    print("prompt micro-batch of 512: %.1f ms = %.0f tok/s" % (tp * 1e3, 512 / tp))
    for n in (128, n_long):
        for i in range(0, n, 512):
            ctx.decode(toks[i:min(i + 512, n)], i)
        ctx.generate_greedy(n, 8)
        out, ms = ctx.generate_greedy(n + 8, 64)
        nb = W + kv_bytes_per_pos * (n + 40)
        print("decode at n_kv ~ %d: %.3f ms/token = %.1f tok/s; %.2f GB/token -> %.0f GB/s = %.1f %% of 8 TB/s"
              % (n + 40, ms / 64, 64e3 / ms, nb / 1e9, nb / (ms / 64 * 1e-3) / 1e9, nb / (ms / 64 * 1e-3) / 8e12 * 100))


if which == "m7q6k":
    path = "/dev/shm/bamd_m7_q6k.gguf"
    if not os.path.exists(path):
        t0 = time.time()
        gguf.write_synthetic_llama(path, E=4096, H=32, Hkv=8, L=32, F=14336, V=32000, theta=1e6, seed=7, reuse_layers=True,
                                   type_fn=lambda name, il: gguf.Q6_K)
        print("gguf %.1f s" % (time.time() - t0))
    E, F, L, V = 4096, 14336, 32, 32000
    W = (L * (E * (E + 2 * 1024) + E * E + 3 * E * F) + V * E) // 256 * 210
    run(path, V, 8192, 7936, 2 * L * 1024 * 2, W)
elif which == "l2-7b":
    path = "/dev/shm/bamd_l2_7b.gguf"
    if not os.path.exists(path):
        t0 = time.time()
        gguf.write_synthetic_llama(path, E=4096, H=32, Hkv=32, L=32, F=11008, V=32000, theta=10000.0, n_ctx_train=4096, seed=7, reuse_layers=True)
        print("gguf %.1f s" % (time.time() - t0))
    m_ = b.Model(path); W = m_.weight_bytes; m_.close()
    run(path, 32000, 4096, 3584, 2 * 32 * 4096 * 2, W)
else:
    path = "/dev/shm/bamd_70b_stage.gguf"

    def tf(name, il):                      # Q4_K_M rule of the LAST 10 of 80 layers (il + 70 >= 7*80/8): attn_v and ffn_down Q6_K, output Q6_K
        return gguf.Q6_K if name in ("output", "attn_v", "ffn_down") else gguf.Q4_K
    if not os.path.exists(path):
        t0 = time.time()
        gguf.write_synthetic_llama(path, E=8192, H=64, Hkv=8, L=10, F=28672, V=128256, seed=7, reuse_layers=True, type_fn=tf)
        print("gguf %.1f s" % (time.time() - t0))
    E, F, L, V = 8192, 28672, 10, 128256
    q4 = lambda n: n // 256 * 144
    q6 = lambda n: n // 256 * 210
    W = L * (q4(E * E) + q4(E * 1024) + q6(E * 1024) + q4(E * E) + 2 * q4(E * F) + q6(E * F)) + q6(V * E)
    run(path, V, 2048, 1536, 2 * L * 1024 * 2, W)
