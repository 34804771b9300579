"""booster_amd — Python host side above the C-ABI of libbooster_amd.so (include/bamd.h, include/booster_bridge.h).

The reference's host language is Go (pkg/server, cgo); Go is not in this image, so this thin ctypes layer plays
the host role for tests, bench and multi-process layer split.  It contains NO compute: every number comes out
of the HIP library, and loading fails loudly if that library is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BAMD_LIB") or os.path.join(_HERE, "lib", "libbooster_amd.so")   # BAMD_LIB: experiment builds (tools/)
_lib = None

F32, F16, Q4_K, Q5_K, Q6_K = 0, 1, 12, 13, 14


class BamdError(RuntimeError):
    pass


def lib():
    """The HIP library.  No fallback: a missing .so is an error (run `python -m booster_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BamdError("libbooster_amd.so is not built (python -m booster_amd.build); there is no CPU fallback")
        # PyTorch bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both own the GPU, and the
        # first one loaded wins the SONAME.  Import torch first so that both sides share torch's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, ci, cf, i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
        L.bamd_last_error.restype = C.c_char_p
        L.bamd_model_load.restype = vp; L.bamd_model_load.argtypes = [C.c_char_p, ci, ci, ci, ci, ci]
        L.bamd_model_free.argtypes = [vp]
        for n in ("bamd_model_n_vocab", "bamd_model_n_embd", "bamd_model_n_layer", "bamd_model_n_ctx_train"):
            getattr(L, n).argtypes = [vp]
        L.bamd_model_weight_bytes.restype = i64; L.bamd_model_weight_bytes.argtypes = [vp]
        L.bamd_model_tensor_raw.restype = i64; L.bamd_model_tensor_raw.argtypes = [vp, C.c_char_p, vp, i64]
        L.bamd_context_new.restype = vp; L.bamd_context_new.argtypes = [vp, ci]
        L.bamd_context_free.argtypes = [vp]
        L.bamd_n_ctx.argtypes = [vp]
        L.bamd_kv_cache_clear.argtypes = [vp]
        L.bamd_decode.argtypes = [vp, vp, ci, ci]
        L.bamd_get_logits.restype = C.POINTER(C.c_float); L.bamd_get_logits.argtypes = [vp]
        L.bamd_generate_greedy.argtypes = [vp, ci, ci, vp, C.POINTER(C.c_float)]
        L.bamd_kv_seq_rm.argtypes = [vp, ci, ci]; L.bamd_kv_seq_add.argtypes = [vp, ci, ci, ci]; L.bamd_kv_seq_div.argtypes = [vp, ci, ci, ci]
        L.bamd_stage_step.argtypes = [vp, C.c_int32, vp, ci, vp, vp, ci, ci, vp]
        L.bamd_stage_token_to.argtypes = [vp, vp, vp]
        L.bamd_stage_prefill.argtypes = [vp, vp, ci, ci, vp, vp, ci, vp]
        L.bamd_stage_argmax.argtypes = [vp, vp, C.POINTER(C.c_int32)]
        L.bamd_profile_step.argtypes = [vp, ci, vp, vp, vp]; L.bamd_profile_step_kinds.argtypes = [vp, ci, vp, vp, vp]
        L.bamd_timeline_step.argtypes = [vp, ci, ci, vp, ci, C.POINTER(ci)]
        L.bamd_set_prefill_batch.argtypes = [ci]; L.bamd_set_prefill_batch.restype = None
        L.bamd_bench_matvec.argtypes = [ci, ci, ci, ci, ci, ci, ci, C.POINTER(C.c_float)]
        L.bamd_op_quantize_q8_K.argtypes = [vp, i64, vp, cf, vp]
        L.bamd_op_mul_mat_vec.argtypes = [ci, vp, ci, ci, vp, vp, cf, vp, vp, ci]
        L.bamd_op_ffn_gate_up.argtypes = [ci, vp, vp, ci, ci, vp, vp, cf, vp]
        L.bamd_op_mul_mat_batch.argtypes = [ci, vp, ci, ci, vp, ci, vp, cf, vp, vp, ci]
        L.bamd_op_get_row.argtypes = [ci, vp, ci, ci, ci, vp]
        L.bamd_op_attention.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp]
        L.bamd_op_rope_row.argtypes = [ci, ci, cf, cf, vp, vp]
        L.bamd_set_aql.argtypes = [ci]; L.bamd_set_aql.restype = None
        L.bamd_aql_runs.argtypes = [vp]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise BamdError(lib().bamd_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_prefill_batch(on):
    """True (default): prompts of 2..512 tokens go through the batched prefill kernels; False: token by token (same bits)."""
    lib().bamd_set_prefill_batch(int(on))      # 2: batched without the MFMA kernel


def set_aql(on):
    """True (default): Context.generate_greedy replays the step as AQL packets with fence scope NONE on the library's own HSA queue (csrc/bamd_aql.h) where it
    can; False: one hipGraph per step on the context's HIP stream.  Same kernels, same bits.  Takes effect at the next generate_greedy call."""
    lib().bamd_set_aql(int(bool(on)))


def device_count():
    return lib().bamd_device_count()


class Model:
    """llama_model analogue: one layer-split stage of a GGUF Llama model resident on one GPU."""

    def __init__(self, path, device=0, layer_first=0, layer_last=-1, with_embd=True, with_output=True):
        self.h = lib().bamd_model_load(os.fsencode(path), device, layer_first, layer_last, int(with_embd), int(with_output))
        if not self.h:
            raise BamdError(lib().bamd_last_error().decode())
        self.n_vocab = lib().bamd_model_n_vocab(self.h)
        self.n_embd = lib().bamd_model_n_embd(self.h)
        self.n_layer = lib().bamd_model_n_layer(self.h)
        self.weight_bytes = lib().bamd_model_weight_bytes(self.h)

    def tensor_raw(self, name):
        """the GGUF bytes of a tensor as the loader sees them (bamd_model_tensor_raw): uint8 array"""
        n = lib().bamd_model_tensor_raw(self.h, name.encode(), None, 0)
        if n < 0:
            raise BamdError(lib().bamd_last_error().decode())
        out = np.zeros(n, np.uint8)
        if lib().bamd_model_tensor_raw(self.h, name.encode(), _p(out), n) != n:
            raise BamdError("bamd_model_tensor_raw: size changed")
        return out

    def close(self):
        if self.h:
            lib().bamd_model_free(self.h)
            self.h = None


class Context:
    """llama_context analogue (KV cache + scratch + device-side step state)."""

    def __init__(self, model, n_ctx):
        self.model = model
        self.h = lib().bamd_context_new(model.h, n_ctx)
        if not self.h:
            raise BamdError(lib().bamd_last_error().decode())
        self.n_ctx = n_ctx

    def close(self):
        if self.h:
            lib().bamd_context_free(self.h)
            self.h = None

    def decode(self, tokens, n_past):
        """llama_decode(llama_batch_get_one(tokens, n, n_past, 0)); returns the last token's logits."""
        t = np.ascontiguousarray(tokens, np.int32)
        if lib().bamd_decode(self.h, _p(t), t.size, n_past) != 0:
            raise BamdError(lib().bamd_last_error().decode())
        return np.ctypeslib.as_array(lib().bamd_get_logits(self.h), shape=(self.model.n_vocab,)).copy()

    def generate_greedy(self, n_past, n_steps):
        out = np.zeros(n_steps + 1, np.int32)
        ms = C.c_float(0)
        _chk(lib().bamd_generate_greedy(self.h, n_past, n_steps, _p(out), C.byref(ms)))
        return out, float(ms.value)

    def aql_runs(self):
        """generate_greedy calls of this context that ran on the own AQL queue so far"""
        return int(lib().bamd_aql_runs(self.h))

    def last_logits(self):
        return np.ctypeslib.as_array(lib().bamd_get_logits(self.h), shape=(self.model.n_vocab,)).copy()

    def kv_seq_rm(self, p0, p1):
        """llama_kv_cache_seq_rm(ctx, 0, p0, p1)"""
        _chk(lib().bamd_kv_seq_rm(self.h, int(p0), int(p1)))

    def kv_seq_add(self, p0, p1, delta):
        """llama_kv_cache_seq_add(ctx, 0, p0, p1, delta)"""
        _chk(lib().bamd_kv_seq_add(self.h, int(p0), int(p1), int(delta)))

    def kv_seq_div(self, p0, p1, d):
        """llama_kv_cache_seq_div(ctx, 0, p0, p1, d)"""
        _chk(lib().bamd_kv_seq_div(self.h, int(p0), int(p1), int(d)))

    def context_shift(self, n_keep, n_past):
        """Booster's context shift (cpp/bridge.cpp:487-503); returns the new n_past"""
        n_discard = (n_past - n_keep) // 2
        self.kv_seq_rm(n_keep, n_keep + n_discard)
        self.kv_seq_add(n_keep + n_discard, n_past, -n_discard)
        return n_past - n_discard

    def profile_step_kinds(self, pos):
        """per launch kind: [qkv, attention, other, wo, gate/up, ffn_down, lm_head, empty event pair] -> (launches, ms, bytes)"""
        launches = np.zeros(8, np.int32); ms = np.zeros(8, np.float64); nbytes = np.zeros(8, np.float64)
        _chk(lib().bamd_profile_step_kinds(self.h, pos, _p(launches), _p(ms), _p(nbytes)))
        return launches, ms, nbytes

    def profile_step(self, pos):
        launches = np.zeros(4, np.int32); ms = np.zeros(4, np.float64); nbytes = np.zeros(4, np.float64)
        _chk(lib().bamd_profile_step(self.h, pos, _p(launches), _p(ms), _p(nbytes)))
        return launches, ms, nbytes

    def timeline_step(self, pos, replays=3):
        """phase stamps of one decode step (BAMD_LIB=.../libbooster_amd_timing.so): u64 [launches][512 workgroups][24], 100 MHz;
        per workgroup: 8 phases of wave 0, 8 phases of wave 7, the exit stamp of each of the 8 waves"""
        cap = 5 * self.model.n_layer + 1
        out = np.zeros((cap, 512, 24), np.uint64)
        n = C.c_int(0)
        _chk(lib().bamd_timeline_step(self.h, pos, replays, _p(out), cap, C.byref(n)))
        return out[:n.value]

    def stage_step(self, token, pos, hidden_in_ptr, hidden_out_ptr, want_logits, prefill_mode, stream_ptr, token_dev_ptr=None):
        _chk(lib().bamd_stage_step(self.h, int(token), token_dev_ptr, int(pos), hidden_in_ptr, hidden_out_ptr, int(want_logits),
                                   int(prefill_mode), stream_ptr))

    def stage_prefill(self, tokens, n_tokens, n_past, hidden_in_ptr, hidden_out_ptr, want_logits, stream_ptr):
        """batched prompt micro-batch through this stage (tokens: host ids on the first stage, else None);
        returns False when the shape has no batched kernels (fall back to stage_step per token)"""
        t = None if tokens is None else np.ascontiguousarray(tokens, np.int32)
        rc = lib().bamd_stage_prefill(self.h, _p(t), int(n_tokens), int(n_past), hidden_in_ptr, hidden_out_ptr, 1 if want_logits else 0, stream_ptr)
        if rc == 2:
            return False
        _chk(rc)
        return True

    def stage_token_to(self, token_dev_ptr, stream_ptr):
        _chk(lib().bamd_stage_token_to(self.h, token_dev_ptr, stream_ptr))

    def stage_logits(self, stream_ptr):
        """host logits of the last stage_step(want_logits=True) on the last stage (synchronises the stream)"""
        lib().bamd_stage_get_logits.restype = C.POINTER(C.c_float); lib().bamd_stage_get_logits.argtypes = [C.c_void_p, C.c_void_p]
        ptr = lib().bamd_stage_get_logits(self.h, stream_ptr)
        if not ptr:
            raise BamdError("bamd_stage_get_logits failed")
        return np.ctypeslib.as_array(ptr, shape=(self.model.n_vocab,)).copy()

    def stage_argmax(self, stream_ptr):
        t = C.c_int32(0)
        _chk(lib().bamd_stage_argmax(self.h, stream_ptr, C.byref(t)))
        return int(t.value)


def bench_matvec(ttype, nrows, k, pro=0, epi=0, mode=0, iters=200):
    us = C.c_float(0)
    _chk(lib().bamd_bench_matvec(ttype, nrows, k, pro, epi, mode, iters, C.byref(us)))
    return float(us.value)


# ---- op-level wrappers (parity tests) ------------------------------------------------------------------------
def op_quantize_q8_K(x, norm_w=None, eps=0.0):
    x = np.ascontiguousarray(x, np.float32)
    w = None if norm_w is None else np.ascontiguousarray(norm_w, np.float32)
    out = np.zeros(x.size // 256 * 292, np.uint8)
    _chk(lib().bamd_op_quantize_q8_K(_p(x), x.size, _p(w), eps, _p(out)))
    return out


def op_mul_mat_vec(ttype, w_raw, nrows, k, x, norm_w=None, eps=0.0, residual=None, mode=0):
    w_raw = np.ascontiguousarray(w_raw, np.uint8)
    x = np.ascontiguousarray(x, np.float32)
    nw = None if norm_w is None else np.ascontiguousarray(norm_w, np.float32)
    res = None if residual is None else np.ascontiguousarray(residual, np.float32)
    y = np.zeros(nrows, np.float32)
    _chk(lib().bamd_op_mul_mat_vec(ttype, _p(w_raw), nrows, k, _p(x), _p(nw), eps, _p(res), _p(y), mode))
    return y


def op_mul_mat_batch(ttype, w_raw, nrows, k, x, norm_w=None, eps=0.0, residual=None, impl=0):
    """Y[t] = W . Q8_K(x[t]) for T rows at once through the prefill kernels: impl 0 = integer-dot kernel, 1 = round-2 MFMA kernel, 2 = round-5 MFMA kernel (3: its eight-wave Q4_K / Q5_K layout)."""
    w_raw = np.ascontiguousarray(w_raw, np.uint8); x = np.ascontiguousarray(x, np.float32)
    T = x.shape[0]
    nw = None if norm_w is None else np.ascontiguousarray(norm_w, np.float32)
    res = None if residual is None else np.ascontiguousarray(residual, np.float32)
    y = np.zeros((T, nrows), np.float32)
    _chk(lib().bamd_op_mul_mat_batch(ttype, _p(w_raw), nrows, k, _p(x), T, _p(nw), eps, _p(res), _p(y), impl))
    return y


def op_ffn_gate_up(ttype, wg_raw, wu_raw, nrows, k, x, norm_w=None, eps=0.0):
    wg_raw = np.ascontiguousarray(wg_raw, np.uint8); wu_raw = np.ascontiguousarray(wu_raw, np.uint8)
    x = np.ascontiguousarray(x, np.float32)
    nw = None if norm_w is None else np.ascontiguousarray(norm_w, np.float32)
    y = np.zeros(nrows, np.float32)
    _chk(lib().bamd_op_ffn_gate_up(ttype, _p(wg_raw), _p(wu_raw), nrows, k, _p(x), _p(nw), eps, _p(y)))
    return y


def op_get_row(ttype, w_raw, nrows, k, row):
    w_raw = np.ascontiguousarray(w_raw, np.uint8)
    y = np.zeros(k, np.float32)
    _chk(lib().bamd_op_get_row(ttype, _p(w_raw), nrows, k, row, _p(y)))
    return y


def op_rope_row(pos, n_dims, freq_base, freq_scale=1.0, freq_factors=None):
    row = np.zeros(n_dims, np.float32)
    ff = None if freq_factors is None else np.ascontiguousarray(freq_factors, np.float32)
    _chk(lib().bamd_op_rope_row(pos, n_dims, freq_base, freq_scale, _p(ff), _p(row)))
    return row


def op_attention(q, k, v, k_cache, v_cache_t, rope_row, H, Hkv, hd, n_ctx, pos, prefill_mode=False, want_probs=False, long_path=False):
    q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float32); v = np.ascontiguousarray(v, np.float32)
    rope_row = np.ascontiguousarray(rope_row, np.float32)
    assert k_cache.dtype == np.uint16 and v_cache_t.dtype == np.uint16 and k_cache.flags.c_contiguous and v_cache_t.flags.c_contiguous
    out = np.zeros(H * hd, np.float32)
    probs = np.zeros(n_ctx, np.float32) if want_probs else None
    _chk(lib().bamd_op_attention(_p(q), _p(k), _p(v), _p(k_cache), _p(v_cache_t), _p(rope_row), H, Hkv, hd, n_ctx, pos, int(prefill_mode) | (2 if (long_path or want_probs) else 0),
                                 _p(out), _p(probs)))
    return (out, probs) if want_probs else out
