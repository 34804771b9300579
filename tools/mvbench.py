"""Micro-benchmark of mat-vec launch shapes, back to back on one matrix (GPU box only).

    python tools/mvbench.py [8b | 70b | modea | ceiling | l2-7b | l32]

8b (default): the launch shapes of Llama-3-8B Q4_K_M, both launch modes where both exist, and prologue-only launches; modea: the mode-A
kernels only (a quick A/B target for kernel edits); ceiling: streaming rate by matrix size, Infinity-Cache-resident (75 MB) up to HBM-bound
(0.9 GB); l2-7b: Llama-2-7B (n_ff 11008 = 43 super-blocks: uneven split-K); l32: Llama-3.2-3B / 1B (n_embd 3072 / 2048)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import booster_amd as b

BB = {12: 144, 13: 176, 14: 210}
# (name, type, rows, K, prologue (1 = RMSNorm), epilogue (0 store, 1 +residual, 2 silu(gate)*up, 3 arg-max), modes)
SETS = {
    "8b": [("prologue-only K4096", 12, 8, 4096, 0, 0, (1, 2)), ("prologue-only norm K4096", 12, 8, 4096, 1, 0, (1, 2)), ("prologue-only K14336", 12, 8, 14336, 0, 0, (1, 2)),
           ("qkv q4k", 12, 6144, 4096, 1, 0, (1, 2)), ("wo q4k", 12, 4096, 4096, 0, 1, (1, 2)), ("gate/up q4k", 12, 14336, 4096, 1, 2, (1,)),
           ("down q4k", 12, 4096, 14336, 0, 1, (1, 2)), ("down q6k", 14, 4096, 14336, 0, 1, (1, 2)), ("lm_head q6k", 14, 128256, 4096, 1, 3, (1,))],
    "modea": [("gate/up q4k", 12, 14336, 4096, 1, 2, (1,)), ("down q4k A", 12, 4096, 14336, 0, 1, (1,)), ("lm_head q6k", 14, 128256, 4096, 1, 3, (1,)),
              ("lm q4k", 12, 128256, 4096, 1, 3, (1,))],
    "ceiling": [("q4k 32768x4096 (75MB, MALL-resident)", 12, 32768, 4096, 0, 0, (1,)), ("q4k 65536x4096 (151MB)", 12, 65536, 4096, 0, 0, (1,)),
                ("q4k 262144x4096 (604MB)", 12, 262144, 4096, 0, 0, (1,)), ("q6k 32768x4096 (110MB)", 14, 32768, 4096, 0, 0, (1,)),
                ("q6k 262144x4096 (881MB)", 14, 262144, 4096, 0, 0, (1,))],
    "70b": [("qkv q4k (one type)", 12, 10240, 8192, 1, 0, (1, 2)), ("wo q4k", 12, 8192, 8192, 0, 1, (1, 2)), ("gate/up q4k", 12, 28672, 8192, 1, 2, (1,)),
            ("down q4k", 12, 8192, 28672, 0, 1, (1, 2)), ("down q6k", 14, 8192, 28672, 0, 1, (1, 2)), ("prologue-only K28672", 12, 8, 28672, 0, 0, (1,)),
            ("prologue-only norm K8192", 12, 8, 8192, 1, 0, (1,))],
    "l2-7b": [("qkv", 12, 12288, 4096, 1, 0, (0,)), ("wo", 12, 4096, 4096, 0, 1, (0,)), ("gate/up", 12, 11008, 4096, 1, 2, (0,)), ("down q4k", 12, 4096, 11008, 0, 1, (0,)),
              ("down q6k", 14, 4096, 11008, 0, 1, (0,)), ("lm_head", 14, 32000, 4096, 1, 3, (0,))],
    "l32": [("3B qkv", 12, 5120, 3072, 1, 0, (0,)), ("3B wo", 12, 3072, 3072, 0, 1, (0,)), ("3B gate/up", 12, 8192, 3072, 1, 2, (0,)), ("3B down q4k", 12, 3072, 8192, 0, 1, (0,)),
            ("3B down q6k", 14, 3072, 8192, 0, 1, (0,)), ("lm_head 3B", 14, 128256, 3072, 1, 3, (0,)), ("1B qkv", 12, 3072, 2048, 1, 0, (0,)), ("1B wo", 12, 2048, 2048, 0, 1, (0,)),
            ("1B gate/up", 12, 8192, 2048, 1, 2, (0,)), ("1B down q6k", 14, 2048, 8192, 0, 1, (0,))],
}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "8b"
    for name, t, rows, k, pro, epi, modes in SETS[which]:
        mb = rows * (k // 256) * BB[t] * (2 if epi == 2 else 1) / 1e6
        for mode in modes:
            us = b.bench_matvec(t, rows, k, pro, epi, mode, 300 if which != "ceiling" else 100)
            print("%-38s mode %d: %8.2f us  %7.1f MB  %7.1f GB/s" % (name, mode, us, mb, mb / us * 1e3))


if __name__ == "__main__":
    main()
