"""One mat-vec shape, a few launches — target of rocprofv3 --pmc passes (GPU box only).  usage: pmc_one.py gu|head|down|wo [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import booster_amd as b
shapes = {"gu": (12, 14336, 4096, 1, 2, 1), "head": (14, 128256, 4096, 1, 3, 1), "down": (12, 4096, 14336, 0, 1, 2), "wo": (12, 4096, 4096, 0, 1, 2),
          "qkv": (12, 6144, 4096, 1, 0, 2)}
t, rows, k, pro, epi, mode = shapes[sys.argv[1]]
print(sys.argv[1], b.bench_matvec(t, rows, k, pro, epi, mode, int(sys.argv[2]) if len(sys.argv) > 2 else 20), "us")
