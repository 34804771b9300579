"""Mode-A mat-vec shapes only (gate/up, ffn_down forced to mode A, lm_head in Q6_K and Q4_K) — a quick A/B target for kernel edits (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import booster_amd as b
BB = {12: 144, 14: 210}
shapes = [("gate/up q4k", 12, 14336, 4096, 1, 2, 1), ("down q4k A", 12, 4096, 14336, 0, 1, 1), ("lm_head q6k", 14, 128256, 4096, 1, 3, 1), ("lm q4k", 12, 128256, 4096, 1, 3, 1)]
for name, t, rows, k, pro, epi, mode in shapes:
    mb = rows * (k // 256) * BB[t] * (2 if epi == 2 else 1) / 1e6
    us = b.bench_matvec(t, rows, k, pro, epi, mode, 200)
    print("%-16s %8.2f us %7.1f MB %7.1f GB/s" % (name, us, mb, mb / us * 1e3))
