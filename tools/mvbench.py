"""Micro-benchmark of the mat-vec launch shapes of Llama-3-8B Q4_K_M (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import booster_amd as b
BB = {12: 144, 14: 210}
shapes = [("prologue-only K4096", 12, 8, 4096, 0, 0), ("prologue-only norm K4096", 12, 8, 4096, 1, 0), ("prologue-only K14336", 12, 8, 14336, 0, 0),
          ("qkv q4k", 12, 6144, 4096, 1, 0), ("wo q4k", 12, 4096, 4096, 0, 1), ("gate/up q4k", 12, 14336, 4096, 1, 2),
          ("down q4k", 12, 4096, 14336, 0, 1), ("down q6k", 14, 4096, 14336, 0, 1), ("lm_head q6k", 14, 128256, 4096, 1, 3)]
for name, t, rows, k, pro, epi in shapes:
    mb = rows * (k // 256) * BB[t] * (2 if epi == 2 else 1) / 1e6
    for mode in ((1, 2) if epi in (0, 1) else (1,)):
        us = b.bench_matvec(t, rows, k, pro, epi, mode, 300)
        print("%-26s mode %d: %8.2f us  %7.1f MB  %7.1f GB/s" % (name, mode, us, mb, mb / us * 1e3 / 1e3 * 1e3 / 1e3 if False else mb / us * 1e3))
