"""ctypes glue for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F32, F16, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 12, 13, 14, 15
BLOCK_BYTES = {Q4_K: 144, Q5_K: 176, Q6_K: 210, Q8_K: 292}


class BoLayer(C.Structure):
    _fields_ = [("attn_norm", C.c_void_p), ("wq", C.c_void_p), ("wk", C.c_void_p), ("wv", C.c_void_p), ("wo", C.c_void_p),
                ("tq", C.c_int), ("tk", C.c_int), ("tv", C.c_int), ("to", C.c_int),
                ("ffn_norm", C.c_void_p), ("wg", C.c_void_p), ("wu", C.c_void_p), ("wd", C.c_void_p),
                ("tg", C.c_int), ("tu", C.c_int), ("td", C.c_int)]


class BoModel(C.Structure):
    _fields_ = [("E", C.c_int), ("H", C.c_int), ("Hkv", C.c_int), ("hd", C.c_int), ("L", C.c_int), ("F", C.c_int), ("V", C.c_int),
                ("eps", C.c_float), ("rope_theta", C.c_float), ("rope_freq_scale", C.c_float),
                ("n_ctx_orig", C.c_int),
                ("rope_freqs", C.c_void_p),
                ("tok_embd", C.c_void_p), ("t_embd", C.c_int),
                ("out_norm", C.c_void_p),
                ("output", C.c_void_p), ("t_out", C.c_int),
                ("layers", C.POINTER(BoLayer))]


TAP_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_int64)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        src = os.path.join(_HERE, "booster_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        L = C.CDLL(path)
        L.bo_fp16_to_fp32.restype = C.c_float; L.bo_fp16_to_fp32.argtypes = [C.c_uint16]
        L.bo_fp32_to_fp16.restype = C.c_uint16; L.bo_fp32_to_fp16.argtypes = [C.c_float]
        L.bo_row_size.restype = C.c_size_t; L.bo_row_size.argtypes = [C.c_int, C.c_int64]
        L.bo_quantize_row_q8_K.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.bo_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        for n in ("bo_vec_dot_q4_K_q8_K", "bo_vec_dot_q5_K_q8_K", "bo_vec_dot_q6_K_q8_K"):
            getattr(L, n).restype = C.c_float; getattr(L, n).argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.bo_mul_mat_q.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.bo_rms_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float]
        L.bo_v_expf.restype = C.c_float; L.bo_v_expf.argtypes = [C.c_float]
        L.bo_v_silu.restype = C.c_float; L.bo_v_silu.argtypes = [C.c_float]
        L.bo_soft_max.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int]
        L.bo_rope_cache.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_float,
                                    C.c_int, C.c_float, C.c_float]
        L.bo_rope_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.bo_dot_f16_f32_tinyblas.restype = C.c_float; L.bo_dot_f16_f32_tinyblas.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.bo_vec_dot_f16.restype = C.c_float; L.bo_vec_dot_f16.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.bo_ctx_new.restype = C.c_void_p; L.bo_ctx_new.argtypes = [C.POINTER(BoModel), C.c_int, C.c_int]
        L.bo_ctx_free.argtypes = [C.c_void_p]
        L.bo_ctx_set_tap.argtypes = [C.c_void_p, TAP_FN, C.c_void_p]
        L.bo_kv_clear.argtypes = [C.c_void_p]
        L.bo_decode.restype = C.c_int; L.bo_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.bo_get_logits.restype = C.POINTER(C.c_float); L.bo_get_logits.argtypes = [C.c_void_p]
        L.bo_kv_k.restype = C.POINTER(C.c_uint16); L.bo_kv_k.argtypes = [C.c_void_p, C.c_int]
        L.bo_kv_v.restype = C.POINTER(C.c_uint16); L.bo_kv_v.argtypes = [C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- array-level helpers -----------------------------------------------------------------------------------
def quantize_q8_K(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(x.size // 256 * 292, np.uint8)
    lib().bo_quantize_row_q8_K(_p(x), _p(out), x.size)
    return out


def dequantize(ttype, raw, k):
    raw = np.ascontiguousarray(raw, np.uint8)
    y = np.zeros(k, np.float32)
    lib().bo_dequantize_row(ttype, _p(raw), _p(y), k)
    return y


def mul_mat_q(ttype, W, nrows, K, x, nthreads=1):
    W = np.ascontiguousarray(W, np.uint8)
    x = np.ascontiguousarray(x, np.float32).reshape(-1, K)
    T = x.shape[0]
    y = np.zeros((T, nrows), np.float32)
    lib().bo_mul_mat_q(ttype, _p(W), nrows, K, _p(x), T, _p(y), nthreads)
    return y


def rms_norm(x, eps):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().bo_rms_norm(_p(x), _p(y), x.size, eps)
    return y


def soft_max(s, mask, scale):
    s = np.ascontiguousarray(s, np.float32)
    p = np.empty_like(s)
    m = None if mask is None else np.ascontiguousarray(mask, np.float32)
    lib().bo_soft_max(_p(s), None if m is None else _p(m), scale, _p(p), s.size)
    return p


def rope_cache(pos, n_dims, freq_base, freq_scale=1.0, freq_factors=None, ext_factor=0.0, attn_factor=1.0, n_ctx_orig=8192,
               beta_fast=32.0, beta_slow=1.0):
    c = np.zeros(n_dims, np.float32)
    ff = None if freq_factors is None else np.ascontiguousarray(freq_factors, np.float32)
    lib().bo_rope_cache(_p(c), pos, n_dims, freq_base, freq_scale, None if ff is None else _p(ff), ext_factor, attn_factor,
                        n_ctx_orig, beta_fast, beta_slow)
    return c


# ---- whole model -----------------------------------------------------------------------------------------
class OracleModel:
    """Builds a bo_model from a booster_amd.gguf.GGUFReader (tensor bytes stay mapped)."""

    def __init__(self, reader, n_layers=None):
        kv, T = reader.kv, reader.tensors
        self.reader = reader
        self._keep = []
        E = int(kv["llama.embedding_length"]); H = int(kv["llama.attention.head_count"])
        Hkv = int(kv.get("llama.attention.head_count_kv", H)); L = int(kv["llama.block_count"])
        if n_layers is not None:
            L = min(L, n_layers)
        Fd = int(kv["llama.feed_forward_length"])
        V = int(T["token_embd.weight"]["shape"][1])
        self.E, self.H, self.Hkv, self.L, self.F, self.V = E, H, Hkv, L, Fd, V
        self.hd = E // H

        def ptr(name):
            a = np.ascontiguousarray(T[name]["data"])
            self._keep.append(a)
            return a.ctypes.data

        layers = (BoLayer * L)()
        for il in range(L):
            p = "blk.%d." % il
            ly = layers[il]
            ly.attn_norm = ptr(p + "attn_norm.weight"); ly.ffn_norm = ptr(p + "ffn_norm.weight")
            for f, t, nm in (("wq", "tq", "attn_q"), ("wk", "tk", "attn_k"), ("wv", "tv", "attn_v"), ("wo", "to", "attn_output"),
                             ("wg", "tg", "ffn_gate"), ("wu", "tu", "ffn_up"), ("wd", "td", "ffn_down")):
                setattr(ly, f, ptr(p + nm + ".weight")); setattr(ly, t, T[p + nm + ".weight"]["type"])
        m = BoModel()
        m.E, m.H, m.Hkv, m.hd, m.L, m.F, m.V = E, H, Hkv, E // H, L, Fd, V
        m.eps = float(kv["llama.attention.layer_norm_rms_epsilon"])
        m.rope_theta = float(kv.get("llama.rope.freq_base", 10000.0))
        m.rope_freq_scale = 1.0
        m.n_ctx_orig = int(kv.get("llama.context_length", 8192))
        m.rope_freqs = ptr("rope_freqs.weight") if "rope_freqs.weight" in T else None
        m.tok_embd = ptr("token_embd.weight"); m.t_embd = T["token_embd.weight"]["type"]
        m.out_norm = ptr("output_norm.weight")
        oname = "output.weight" if "output.weight" in T else "token_embd.weight"     # tied embeddings, llama.cpp:6070-6076
        m.output = ptr(oname); m.t_out = T[oname]["type"]
        m.layers = layers
        self._layers = layers
        self.m = m


class OracleContext:
    def __init__(self, model, n_ctx, nthreads=1):
        self.model = model
        self.n_ctx = n_ctx
        self.c = lib().bo_ctx_new(C.byref(model.m), n_ctx, nthreads)
        self._tap = None
        self.taps = {}

    def close(self):
        if self.c:
            lib().bo_ctx_free(self.c)
            self.c = None

    def enable_taps(self):
        def fn(ud, name, il, data, n):
            key = name.decode() + ("-%d" % il if il >= 0 else "")
            self.taps[key] = np.ctypeslib.as_array(data, shape=(n,)).copy()
        self._tap = TAP_FN(fn)
        lib().bo_ctx_set_tap(self.c, self._tap, None)

    def decode(self, tokens, n_past):
        t = np.ascontiguousarray(tokens, np.int32)
        self.taps = {}
        rc = lib().bo_decode(self.c, _p(t), t.size, n_past)
        assert rc == 0, "bo_decode failed"
        return np.ctypeslib.as_array(lib().bo_get_logits(self.c), shape=(self.model.V,)).copy()

    def kv_k(self, il):
        n = self.n_ctx * self.model.Hkv * self.model.hd
        return np.ctypeslib.as_array(lib().bo_kv_k(self.c, il), shape=(n,)).copy()

    def kv_v(self, il):
        n = self.n_ctx * self.model.Hkv * self.model.hd
        return np.ctypeslib.as_array(lib().bo_kv_v(self.c, il), shape=(n,)).copy()
