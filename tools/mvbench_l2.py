"""Mat-vec launch shapes of Llama-2-7B Q4_K_M (E 4096, F 11008 = 43 super-blocks: not a multiple of 8) — GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import booster_amd as b
BB = {12: 144, 14: 210}
for name, t, rows, k, pro, epi in [("qkv", 12, 12288, 4096, 1, 0), ("wo", 12, 4096, 4096, 0, 1), ("gate/up", 12, 11008, 4096, 1, 2), ("down q4k", 12, 4096, 11008, 0, 1),
                                   ("down q6k", 14, 4096, 11008, 0, 1), ("lm_head", 14, 32000, 4096, 1, 3)]:
    mb = rows * (k // 256) * BB[t] * (2 if epi == 2 else 1) / 1e6
    us = b.bench_matvec(t, rows, k, pro, epi, 0, 300)
    print("%-10s %8.2f us  %7.1f MB  %7.1f GB/s" % (name, us, mb, mb / us * 1e3))
