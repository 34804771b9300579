"""Mat-vec launch shapes of Llama-3.2-3B (E 3072 = 12 super-blocks, F 8192) and Llama-3.2-1B (E 2048, F 8192) Q4_K_M — GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import booster_amd as b
BB = {12: 144, 14: 210}
for name, t, rows, k, pro, epi in [("3B qkv", 12, 5120, 3072, 1, 0), ("3B wo", 12, 3072, 3072, 0, 1), ("3B gate/up", 12, 8192, 3072, 1, 2), ("3B down q4k", 12, 3072, 8192, 0, 1),
                                   ("3B down q6k", 14, 3072, 8192, 0, 1), ("lm_head 3B", 14, 128256, 3072, 1, 3),
                                   ("1B qkv", 12, 3072, 2048, 1, 0), ("1B wo", 12, 2048, 2048, 0, 1), ("1B gate/up", 12, 8192, 2048, 1, 2), ("1B down q6k", 14, 2048, 8192, 0, 1)]:
    mb = rows * (k // 256) * BB[t] * (2 if epi == 2 else 1) / 1e6
    us = b.bench_matvec(t, rows, k, pro, epi, 0, 300)
    print("%-12s %8.2f us  %7.1f MB  %7.1f GB/s" % (name, us, mb, mb / us * 1e3))
