"""CPU, build container only: the oracle against the genuine reference compiled in place (oracle/_ref/libggml_ref.so).
Skipped where that library does not exist or cannot be loaded."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "libggml_ref.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libggml_ref.so not built (needs /root/reference: make -C oracle ref)")
    try:
        L = C.CDLL(REF)
    except OSError as e:
        pytest.skip("cannot load reference build: %s" % e)
    class _InitParams(C.Structure):
        _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]
    # ggml_init fills the f16->f32 lookup table that GGML_FP16_TO_FP32 reads on x86
    L.ggml_init.restype = C.c_void_p
    L.ggml_init.argtypes = [_InitParams]
    L.ggml_init(_InitParams(1 << 20, None, False))
    L.ggml_quantize_chunk.restype = C.c_size_t
    L.ggml_quantize_chunk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    L.quantize_row_q8_K.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    for n in ("ggml_vec_dot_q4_K_q8_K", "ggml_vec_dot_q5_K_q8_K", "ggml_vec_dot_q6_K_q8_K"):
        getattr(L, n).argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    return L


def p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("t,bb,fn", [(12, 144, "ggml_vec_dot_q4_K_q8_K"), (13, 176, "ggml_vec_dot_q5_K_q8_K"), (14, 210, "ggml_vec_dot_q6_K_q8_K")])
@pytest.mark.parametrize("K", [256, 1024, 14336])
def test_dot_random_bytes(po, ref, t, bb, fn, K):
    """Random raw blocks (every bit pattern of scales/quants), random activations, several magnitudes."""
    from booster_amd.gguf import random_kquant_tensor
    rng = np.random.default_rng(K + t)
    rows = 32
    blocks = random_kquant_tensor(t, K, rows, rng)
    for scale in (1e-3, 1.0, 50.0):
        x = (rng.standard_normal(K) * scale).astype(np.float32)
        q8r = np.zeros(K // 256 * 292, np.uint8)
        ref.quantize_row_q8_K(p(x), p(q8r), K)
        assert np.array_equal(po.quantize_q8_K(x), q8r)
        want = np.zeros(rows, np.float32)
        rb = K // 256 * bb
        for r in range(rows):
            s = C.c_float(0)
            getattr(ref, fn)(K, C.byref(s), 0, C.c_void_p(blocks.ctypes.data + r * rb), 0, p(q8r), 0, 1)
            want[r] = s.value
        got = po.mul_mat_q(t, blocks, rows, K, x)[0]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
