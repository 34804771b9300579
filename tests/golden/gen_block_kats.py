"""Generates tests/golden/block_kats.npz from the GENUINE reference (oracle/_ref/libggml_ref.so, built by
`make -C oracle ref` from /root/reference).  Run in the build container only; the .npz is the committed fixture.

Known-answer vectors per K-quant type (SURVEY.md §4 test plan items 1-2):
  * reference-quantised weight blocks (ggml_quantize_chunk) for a [rows][K] matrix,
  * reference dequantisation of them (dequantize_row_q*_K),
  * activations, their reference Q8_K quantisation (quantize_row_q8_K),
  * reference dot products ggml_vec_dot_q*_K_q8_K (AVX2 build) for every row,
  * element ops through the reference's own graph executor: rms_norm, silu, soft_max_ext.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "libggml_ref.so")
L = C.CDLL(REF)


class _InitParams(C.Structure):
    _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]


# ggml_init fills the f16->f32 lookup table that GGML_FP16_TO_FP32 reads on x86 (ggml.c ggml_init; ggml-impl.h)
L.ggml_init.restype = C.c_void_p
L.ggml_init.argtypes = [_InitParams]
L.ggml_init(_InitParams(1 << 20, None, False))
Q4_K, Q5_K, Q6_K, Q8_K = 12, 13, 14, 15
BB = {Q4_K: 144, Q5_K: 176, Q6_K: 210}
L.ggml_quantize_chunk.restype = C.c_size_t
L.ggml_quantize_chunk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
L.quantize_row_q8_K.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
for n in ("dequantize_row_q4_K", "dequantize_row_q5_K", "dequantize_row_q6_K"):
    getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
for n in ("ggml_vec_dot_q4_K_q8_K", "ggml_vec_dot_q5_K_q8_K", "ggml_vec_dot_q6_K_q8_K"):
    getattr(L, n).argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
DEQ = {Q4_K: L.dequantize_row_q4_K, Q5_K: L.dequantize_row_q5_K, Q6_K: L.dequantize_row_q6_K}
DOT = {Q4_K: L.ggml_vec_dot_q4_K_q8_K, Q5_K: L.ggml_vec_dot_q5_K_q8_K, Q6_K: L.ggml_vec_dot_q6_K_q8_K}


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def main():
    rng = np.random.default_rng(20240918)
    out = {}
    for t, name in ((Q4_K, "q4_K"), (Q5_K, "q5_K"), (Q6_K, "q6_K")):
        for K, rows in ((256, 16), (768, 24), (4096, 16)):
            w = (rng.standard_normal((rows, K)) * (1.0 / np.sqrt(K))).astype(np.float32)
            w[0, :7] = 0.0
            blocks = np.zeros(rows * (K // 256) * BB[t], np.uint8)
            L.ggml_quantize_chunk(t, p(w), p(blocks), 0, rows, K, None)
            deq = np.zeros((rows, K), np.float32)
            DEQ[t](p(blocks), p(deq), rows * K)
            x = (rng.standard_normal(K) * rng.choice([0.01, 1.0, 30.0])).astype(np.float32)
            if K >= 768:
                x[256:512] = 0.0                     # an all-zero activation block (amax == 0 branch)
            x[5] = -np.abs(x).max() * 1.5            # negative extremum: iscale = -127/max with max < 0
            q8 = np.zeros((K // 256) * 292, np.uint8)
            L.quantize_row_q8_K(p(x), p(q8), K)
            dots = np.zeros(rows, np.float32)
            rb = (K // 256) * BB[t]
            for r in range(rows):
                s = C.c_float(0)
                DOT[t](K, C.byref(s), 0, C.c_void_p(blocks.ctypes.data + r * rb), 0, p(q8), 0, 1)
                dots[r] = s.value
            key = "%s_K%d" % (name, K)
            out[key + "_blocks"] = blocks; out[key + "_deq"] = deq; out[key + "_x"] = x; out[key + "_q8"] = q8; out[key + "_dots"] = dots
    np.savez_compressed(os.path.join(HERE, "block_kats.npz"), **out)
    print("wrote block_kats.npz with", len(out), "arrays")


if __name__ == "__main__":
    sys.exit(main())
