"""Streaming rate of the mode-A mat-vec kernel by matrix size: Infinity-Cache-resident (75 MB) up to HBM-bound (0.9 GB) — the ceiling curve (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import booster_amd as b
BB = {12: 144, 14: 210}
for name, t, rows, k in [("q4k 32768x4096 (75MB, MALL-resident)", 12, 32768, 4096), ("q4k 65536x4096 (151MB)", 12, 65536, 4096), ("q4k 262144x4096 (604MB)", 12, 262144, 4096),
                         ("q6k 32768x4096 (110MB)", 14, 32768, 4096), ("q6k 262144x4096 (881MB)", 14, 262144, 4096)]:
    mb = rows * (k // 256) * BB[t] / 1e6
    us = b.bench_matvec(t, rows, k, 0, 0, 1, 100)
    print("%-40s %8.2f us %7.1f MB %7.1f GB/s" % (name, us, mb, mb / us * 1e3))
