// bamd_engine.cpp — model runtime of libbooster_amd.so: GGUF -> HBM (wave-stream repack), KV cache, the fixed
// Llama decode pipeline as a static launch sequence / hipGraph, and the level-1 C-ABI of include/bamd.h.
//
// Replaces, for general.architecture = "llama" only, what the reference does in cpp/src/llama.cpp
// (llama_load_model_from_file :16539, llm_load_tensors :5899, llama_new_context_with_model :16592,
// llama_kv_cache_init :2926, build_llama :8781, llama_decode_internal :14537) and cpp/ggml/src/ggml-backend.c
// (graph split / scheduling) with one static pipeline: 6 kernels per layer, activations never leave HBM/LDS,
// the step's control state (position, KV length, token) lives on the device.
#include "../../include/bamd.h"
#include "bamd_formats.h"
#include "bamd_gguf.h"
#include "bamd_kernels.h"
#include "bamd_aql.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

static thread_local std::string g_err;
// BAMD_PREFILL_BATCH=0: evaluate prompts token by token through the decode kernels instead of the batched kernels (same bits)
static int g_prefill_batch = [] { const char * e = getenv("BAMD_PREFILL_BATCH"); return (e && e[0] == '0') ? 0 : 1; }();
// BAMD_PREFILL_MFMA=0: Q4_K mat-muls of the batched prefill on the integer-dot kernel instead of the MFMA kernel (same bits)
static int g_prefill_mfma = [] { const char * e = getenv("BAMD_PREFILL_MFMA"); return (e && e[0] == '0') ? 0 : 1; }();
// The matrix-core kernels need a side table per matrix (bamd_prefill2.hip): built at MODEL LOAD, all matrices or none, against an explicit memory budget — after the
// weights are resident the tables (+ 78 % of the Q4_K / Q5_K bytes, + 63 % of the Q6_K bytes) must leave BAMD_PREFILL_AUX_RESERVE_GB (default 8) GiB of the device free,
// else the model runs its prompts on the integer-dot kernel (token by token where that has no instance).  BAMD_PREFILL_AUX=0 skips them (decode-only deployments).
static const int g_prefill_aux = [] { const char * e = getenv("BAMD_PREFILL_AUX"); return (e && e[0] == '0') ? 0 : 1; }();
// BAMD_STAGE_GRAPH=0: bamd_stage_step enqueues its kernels one by one instead of replaying a captured hipGraph
static const int g_stage_graph = [] { const char * e = getenv("BAMD_STAGE_GRAPH"); return (e && e[0] == '0') ? 0 : 1; }();
static const bool g_attn_fused = [] { const char * e = getenv("BAMD_ATTN_FUSED"); return !(e && e[0] == '0'); }();   // default: fused single-launch attention (BAMD_ATTN_FUSED=0: three-kernel path)
static int fail(const std::string & m) { g_err = m; return 1; }
// BAMD_AQL=0: the device-side greedy loop replays one hipGraph per step on the context's HIP stream (rounds 1-5) instead of AQL packets with fence scope
// NONE on the library's own queue (bamd_aql.h).  Same kernels, same bits; BAMD_AQL_VERBOSE=1 says why when the own queue is not used
static int g_aql = [] { const char * e = getenv("BAMD_AQL"); return (e && e[0] == '0') ? 0 : 1; }();
extern "C" __attribute__((visibility("default"))) void bamd_set_aql(int on) { g_aql = on ? 1 : 0; }
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return 1; } } while (0)
#define HIPP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return nullptr; } } while (0)

extern "C" __attribute__((visibility("default"))) const char * bamd_last_error(void) { return g_err.c_str(); }

// owners for the short-lived HIP objects of the measurement entry points (an early error return must not leak them)
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
    hipError_t create() { hipError_t e = hipEventCreate(&a); return e != hipSuccess ? e : hipEventCreate(&b); }
};
struct OwnedStream { hipStream_t s = nullptr; ~OwnedStream() { if (s) hipStreamDestroy(s); } };
struct OwnedDevMem { void * p = nullptr; ~OwnedDevMem() { if (p) hipFree(p); } };
struct OwnedGraphExec { hipGraphExec_t g = nullptr; ~OwnedGraphExec() { if (g) hipGraphExecDestroy(g); } };

// -------------------------------------------------------------------------------------------------------
// RoPE table on the host — ggml_rope_cache_init / rope_yarn / ggml_rope_yarn_corr_dims (ggml.c:13994-14041).
// cos/sin come from the host libm exactly like the reference's CPU path; the device only looks them up.
// -------------------------------------------------------------------------------------------------------
static float rope_yarn_ramp(const float low, const float high, const int i0) {
    const float y = (i0 / 2 - low) / fmaxf(0.001f, high - low);
    return 1 - fminf(1, fmaxf(0, y));
}
static float rope_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}
static void rope_row(float * cache, int32_t pos, int n_dims, float freq_base, float freq_scale, const float * freq_factors,
                     float ext_factor, float attn_factor, int n_ctx_orig, float beta_fast, float beta_slow) {
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    float corr_dims[2];
    {
        const float start = floorf(rope_corr_dim(n_dims, n_ctx_orig, beta_fast, freq_base));
        const float end = ceilf(rope_corr_dim(n_dims, n_ctx_orig, beta_slow, freq_base));
        corr_dims[0] = start > 0 ? start : 0;
        corr_dims[1] = end < n_dims - 1 ? end : n_dims - 1;
    }
    float theta = (float) pos;
    for (int i0 = 0; i0 < n_dims; i0 += 2) {
        const float ff = freq_factors ? freq_factors[i0 / 2] : 1.0f;
        const float theta_extrap = theta / ff;
        const float theta_interp = freq_scale * theta_extrap;
        float th = theta_interp, mscale = attn_factor;
        if (ext_factor != 0.0f) {
            const float ramp_mix = rope_yarn_ramp(corr_dims[0], corr_dims[1], i0) * ext_factor;
            th = theta_interp * (1 - ramp_mix) + theta_extrap * ramp_mix;
            mscale *= 1.0f + 0.1f * logf(1.0f / freq_scale);
        }
        cache[i0 + 0] = cosf(th) * mscale;
        cache[i0 + 1] = sinf(th) * mscale;
        theta *= theta_scale;
    }
}

// -------------------------------------------------------------------------------------------------------
struct DevMat {                      // one quantised matrix resident in HBM
    void * stream = nullptr;         // wave-stream records (matmul operand)
    void * raw = nullptr;            // GGUF layout (kept only for token_embd: row gather)
    int type = 0, nrows = 0, K = 0;  // nrows = rows in the GGUF
    int nrows_pad = 0;               // rows of the stream (next multiple of 8; missing rows are zero records)
    size_t bytes = 0;
};
struct DevLayer {
    float * attn_norm = nullptr, * ffn_norm = nullptr;
    DevMat wq, wk, wv, wo, wg, wu, wd;
    // prefill side tables (bamd_prefill2.hip), built at the first batched evaluation: one per QKV segment as enqueue_prefill_batch merges them, wo, gate, up, down
    void * aux_qkv[3] = { nullptr, nullptr, nullptr }, * aux_o = nullptr, * aux_g = nullptr, * aux_u = nullptr, * aux_d = nullptr;
};

struct bamd_model {
    int device = 0;
    int E = 0, H = 0, Hkv = 0, hd = 0, L = 0, F = 0, V = 0, n_ctx_train = 0, n_rot = 0;
    float eps = 1e-5f, rope_theta = 10000.f, rope_freq_scale = 1.f, rope_ext_factor = 0.f, rope_attn_factor = 1.f;
    int rope_n_ctx_orig = 0;
    int layer_first = 0, layer_last = 0;
    bool with_embd = true, with_output = true;
    std::vector<float> rope_freqs;   // host copy (optional)
    std::vector<DevLayer> layers;    // [layer_last - layer_first]
    DevMat tok_embd, output;
    float * out_norm = nullptr;
    int64_t weight_bytes = 0;
    int n_cu = 256;
    std::unique_ptr<GgufFile> file;  // stays mapped (bamd_model_tensor_raw)
    std::vector<void *> allocs;
    bool aux_ok = false; int64_t aux_bytes = 0; std::string aux_why;      // prefill side tables: all matrices or none (build_prefill_aux)
};

struct bamd_context {
    bamd_model * m = nullptr;
    int n_ctx = 0;
    std::vector<unsigned short *> kc, vc;
    float * rope = nullptr, * rope_cur = nullptr;   // rope_cur [hd]: the cos / sin row of the current step's position (step_begin_kernel -> attn_qk_kernel)
    float * x = nullptr, * x2 = nullptr, * q = nullptr, * k = nullptr, * v = nullptr, * att = nullptr, * h = nullptr, * scores = nullptr, * probs = nullptr;
    int n_ctx_pad = 0;              // KV row stride: n_ctx rounded up to 64 (V^T rows are stored in 64-position blocks)
    float * logits = nullptr;        // device
    float * logits_host = nullptr;   // pinned
    bool logits_readback = true, logits_host_valid = false;
    bool status_pending = false;     // a bamd_decode without read-back returned before its stream was waited for: the next synchronising call checks the give-up words
    // sampler prefilter (bamd_logits_shortlist): per-vocabulary tables, penalty list and result (device + pinned mirrors)
    uint8_t * samp_cls = nullptr; float * samp_cut = nullptr;
    bamd_logit_penalty * samp_pen = nullptr, * samp_pen_host = nullptr;
    unsigned char * samp_out = nullptr, * samp_out_host = nullptr;
    bamd_step_state * st = nullptr;
    int32_t * forced = nullptr; int forced_cap = 0;
    int32_t * out_tokens = nullptr; int out_cap = 0;
    hipStream_t stream = nullptr;
    hipGraphExec_t graph = nullptr; int graph_fused = -1;
    bamd_aql_graph * aql = nullptr; int aql_key = -1; bool aql_failed = false; int aql_runs = 0;    // the same step as AQL packets for the own queue (bamd_aql.h)
    bamd_aql_graph * aql_step = nullptr; int aql_step_key = -1; int aql_steps = 0;                 // bamd_stage_step's single-token step on the own queue
    bamd_step_state * inbox = nullptr;           // pinned host memory: the state of that step, read by step_begin_kernel itself (no copy in front of the step)
    unsigned long long * co_gran = nullptr;   // co-launch granules [H * hd] {value, tag} + give-up counter behind them (bamd_colaunch.hip), zero-initialised
    uint32_t * co_err = nullptr;
    int32_t * slots = nullptr; int slots_cap = 0;   // device-side greedy loop after a context shift: {cell, padded KV length} of every step (bamd_generate_greedy)
    int host_serial = 0;             // host calls that set the device state so far (bamd_step_state.serial)
    // KV cell metadata (llama_kv_cache cells: pos / delta / head / used, llama.cpp:2700-2760) — inactive (cell i holds position i, nothing to
    // track) until the first bamd_kv_seq_rm / bamd_kv_seq_add
    struct Cells { bool active = false, has_shift = false; std::vector<int32_t> pos, delta; int head = 0, used = 0; } cells;
    int n_cached = 0;               // positions evaluated so far (highest n_past + n_tokens seen): what the cells hold when tracking starts
    int32_t * cellpos = nullptr;    // device copy of cells.pos (attention mask of the shifted path)
    int32_t * shift_idx = nullptr; float * shift_tab = nullptr; int shift_tab_cap = 0;
    // stage-step graphs (bamd_stage_step): one per (want_logits, prefill_mode), valid for the pointers it was captured with
    struct StageGraph { hipGraphExec_t exec = nullptr; const void * token_src = nullptr, * hin = nullptr; void * hout = nullptr; int fused = -1; };
    StageGraph sgraph[2][2];
    // batched prefill buffers, [bcap] tokens each (allocated at the first multi-token decode)
    // phase-stamp blocks (bamd_timeline_step, BAMD_TIMING builds): one block of BAMD_TL_SLOT_WORDS u64 per launch
    unsigned long long * tl_base = nullptr; int tl_slot = 0, tl_cap = 0;
    int bcap = 0;
    float * attn_bscr = nullptr; size_t attn_bscr_bytes = 0, attn_bscr_failed = 0;   // score rows of the matrix-core prefill attention beyond BAMD_AM_MAXPOS = 512 positions (grow-only; absent = the VALU kernel runs)
    float * bx = nullptr, * bx2 = nullptr, * bqkv = nullptr, * batt = nullptr, * bh = nullptr; unsigned char * bblob = nullptr, * bblob16 = nullptr;
    std::vector<void *> allocs;
};

static int dev_alloc(std::vector<void *> & keep, void ** p, size_t bytes) {
    HIPC(hipMalloc(p, bytes + 4096));           // + 4 KiB: the sixteen-wave prefill kernel's nibble loads of the two super-blocks behind the end of K run past
                                                // the last record group (bamd_prefill2.hip)
    keep.push_back(*p);
    return 0;
}

extern "C" __attribute__((visibility("default"))) int bamd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { g_err = "hipGetDeviceCount failed (no HIP device?)"; return -1; }
    return n;
}
extern "C" __attribute__((visibility("default"))) int bamd_backend_init(void) { return bamd_device_count(); }

// upload one matrix: raw bytes -> (optional raw copy) + wave-stream repack on the device
static int upload_mat(bamd_model * m, const GgufTensor * t, DevMat & d, bool keep_raw, bool want_stream, void * staging, hipStream_t s,
                      void * stream_dst = nullptr) {
    if (t->ne.size() != 2) return fail("tensor " + t->name + ": expected 2 dims");
    d.type = t->type; d.K = (int) t->ne[0]; d.nrows = (int) t->ne[1]; d.bytes = t->nbytes;
    if (d.type != BAMD_F32 && d.type != BAMD_F16 && !bamd_is_kquant(d.type)) return fail("tensor " + t->name + ": type " + std::to_string(d.type) + " not supported (F32, F16, Q4_K, Q5_K, Q6_K)");
    if (bamd_is_kquant(d.type) && d.K % 256) return fail("tensor " + t->name + ": row length not a multiple of 256");
    if (want_stream) {
        if (!bamd_is_kquant(d.type)) return fail("tensor " + t->name + ": only Q4_K/Q5_K/Q6_K matrices are supported on the matmul path");
        if (d.K % 256) return fail("tensor " + t->name + ": row length not a multiple of 256");
    }
    d.nrows_pad = (d.nrows + 7) / 8 * 8;
    if (d.nrows_pad != d.nrows && stream_dst) return fail("tensor " + t->name + ": fused QKV needs row counts that are multiples of 8");
    void * rawdev = staging;
    if (keep_raw) { if (dev_alloc(m->allocs, &d.raw, t->nbytes)) return 1; rawdev = d.raw; }
    HIPC(hipMemcpyAsync(rawdev, t->data, t->nbytes, hipMemcpyHostToDevice, s));
    if (want_stream) {
        const size_t stream_bytes = bamd_stream_bytes(d.type, d.K, d.nrows_pad);
        if (stream_bytes >= (size_t) 0x7fffffff) return fail("tensor " + t->name + ": wave-stream copy exceeds 2 GiB (32-bit record offsets)");
        if (stream_dst) d.stream = stream_dst;
        else { if (dev_alloc(m->allocs, &d.stream, stream_bytes)) return 1; if (d.nrows_pad != d.nrows) HIPC(hipMemsetAsync(d.stream, 0, stream_bytes, s)); }
        bamd_launch_repack(rawdev, d.stream, d.type, d.nrows, d.K, s);
        HIPC(hipGetLastError());
        m->weight_bytes += (int64_t) t->nbytes;
    }
    HIPC(hipStreamSynchronize(s));     // staging buffer is reused by the next tensor
    return 0;
}
static int upload_f32(bamd_model * m, const GgufTensor * t, float ** p, int n, hipStream_t s) {
    if (!t) return fail("missing norm tensor");
    if (t->type != BAMD_F32 || (int) t->ne[0] != n) return fail("tensor " + t->name + ": expected f32[" + std::to_string(n) + "]");
    if (dev_alloc(m->allocs, (void **) p, (size_t) n * 4)) return 1;
    HIPC(hipMemcpyAsync(*p, t->data, (size_t) n * 4, hipMemcpyHostToDevice, s));
    return 0;
}

static void build_prefill_aux(bamd_model * m, hipStream_t s);
static int model_load_impl(bamd_model * m, const char * path, int device, int lf, int ll, int with_embd, int with_output) {
    // the file is read and its header validated before any device is touched: a bad file is reported as such on any machine
    m->file.reset(new GgufFile());
    std::string err;
    if (!m->file->open(path, err)) return fail(err);
    GgufFile & g = *m->file;
    std::string arch;
    if (!g.get_str("general.architecture", arch) || arch != "llama") return fail("general.architecture must be \"llama\" (got \"" + arch + "\")");
    uint32_t u;
    if (!g.get_u32("llama.embedding_length", u)) return fail("missing llama.embedding_length"); m->E = (int) u;
    if (!g.get_u32("llama.block_count", u)) return fail("missing llama.block_count"); m->L = (int) u;
    if (!g.get_u32("llama.feed_forward_length", u)) return fail("missing llama.feed_forward_length"); m->F = (int) u;
    if (!g.get_u32("llama.attention.head_count", u)) return fail("missing llama.attention.head_count"); m->H = (int) u;
    m->Hkv = m->H; if (g.get_u32("llama.attention.head_count_kv", u)) m->Hkv = (int) u;
    if (!g.get_f32("llama.attention.layer_norm_rms_epsilon", m->eps)) return fail("missing llama.attention.layer_norm_rms_epsilon");
    m->n_ctx_train = 2048; if (g.get_u32("llama.context_length", u)) m->n_ctx_train = (int) u;
    if (m->E <= 0 || m->L <= 0 || m->F <= 0 || m->H <= 0 || m->Hkv <= 0) return fail("embedding_length, block_count, feed_forward_length and head counts must be positive");
    if (m->E % m->H) return fail("llama.embedding_length is not a multiple of llama.attention.head_count");
    m->hd = m->E / m->H;
    m->n_rot = m->hd; if (g.get_u32("llama.rope.dimension_count", u)) m->n_rot = (int) u;
    if (m->n_rot != m->hd) return fail("llama.rope.dimension_count != n_embd/n_head is not supported");
    g.get_f32("llama.rope.freq_base", m->rope_theta);
    {   // RoPE scaling as llm_load_hparams reads it (llama.cpp:4630-4650): the type defaults to "linear" when the key is absent, the
        // factor comes from rope.scaling.factor or the legacy rope.scale_linear, freq_scale = 1 / factor; a "none" type never scales
        // (llama_new_context_with_model, llama.cpp:16682-16684).  YaRN: ext_factor = 1 (:16686-16688), the magnitude factor is
        // rope.scaling.attn_factor for every type (:16690), the correction range comes from rope.scaling.original_context_length
        // (default: the training context; :4630-4631, :16670-16672) with beta_fast 32 / beta_slow 1 (:16441-16442).
        std::string st = "linear"; float factor = 0.f;
        g.get_str("llama.rope.scaling.type", st);
        if (st != "none" && st != "linear" && st != "yarn") return fail("rope scaling type \"" + st + "\" not supported");
        if (!g.get_f32("llama.rope.scaling.factor", factor)) g.get_f32("llama.rope.scale_linear", factor);
        m->rope_freq_scale = (factor == 0.f || st == "none") ? 1.0f : 1.0f / factor;
        m->rope_ext_factor = st == "yarn" ? 1.0f : 0.0f;
        m->rope_attn_factor = 1.0f; g.get_f32("llama.rope.scaling.attn_factor", m->rope_attn_factor);
        m->rope_n_ctx_orig = m->n_ctx_train;
        if (g.get_u32("llama.rope.scaling.original_context_length", u) && u != 0) m->rope_n_ctx_orig = (int) u;
    }
    if (m->H % m->Hkv) return fail("n_head % n_head_kv != 0");
    const int gq = m->H / m->Hkv;
    if (gq < 1 || gq > 8) return fail("GQA ratio (heads per KV head) must be 1 .. 8");
    if (m->hd % 64 || m->hd > 256) return fail("head dim must be 64, 128, 192 or 256");
    if (m->E % 256 || m->F % 256) return fail("n_embd and n_ff must be multiples of 256");
    if (m->E > 32768 || m->F > 131072) return fail("n_embd / n_ff beyond what the kernels' LDS budgets were sized for");
    if (ll < 0 || ll > m->L) ll = m->L;
    if (lf < 0 || lf > ll) return fail("bad layer range");
    m->layer_first = lf; m->layer_last = ll; m->with_embd = with_embd != 0; m->with_output = with_output != 0;

    const GgufTensor * te = g.tensor("token_embd.weight");
    if (!te) return fail("missing token_embd.weight");
    if (te->ne.size() != 2 || (int) te->ne[0] != m->E) return fail("token_embd.weight: row length != llama.embedding_length");
    m->V = (int) te->ne[1];
    if (m->V <= 0) return fail("token_embd.weight: empty vocabulary");
    if (const GgufTensor * rf = g.tensor("rope_freqs.weight")) {
        if (rf->type != BAMD_F32 || (int) rf->ne[0] < m->hd / 2) return fail("bad rope_freqs.weight");
        m->rope_freqs.assign((const float *) rf->data, (const float *) rf->data + m->hd / 2);
    }
    const int ndev = bamd_device_count();
    if (ndev <= 0) return fail("no HIP device available: libbooster_amd has no CPU fallback");
    if (device < 0 || device >= ndev) return fail("bad device index");
    HIPC(hipSetDevice(device));
    m->device = device;
    { hipDeviceProp_t p; HIPC(hipGetDeviceProperties(&p, device)); m->n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }
    hipStream_t s; HIPC(hipStreamCreate(&s));
    size_t stage_bytes = 0;
    for (auto & t : g.tensors) stage_bytes = std::max(stage_bytes, t.nbytes);
    void * staging = nullptr;
    { const hipError_t e = hipMalloc(&staging, stage_bytes ? stage_bytes : 16); if (e != hipSuccess) { hipStreamDestroy(s); return fail(std::string("staging buffer: ") + hipGetErrorString(e)); } }
    int rc = 0;
    do {
        if (m->with_embd) { if ((rc = upload_mat(m, te, m->tok_embd, true, false, staging, s))) break; }
        if (m->with_output) {
            if ((rc = upload_f32(m, g.tensor("output_norm.weight"), &m->out_norm, m->E, s))) break;
            const GgufTensor * to = g.tensor("output.weight");
            if (!to) to = te;                                              // tied embeddings, llama.cpp:6070-6076
            if ((rc = upload_mat(m, to, m->output, false, true, staging, s))) break;
            if (m->output.K != m->E) { rc = fail("output.weight: row length != llama.embedding_length"); break; }
        }
        m->layers.resize((size_t) (ll - lf));
        for (int il = lf; il < ll && !rc; ++il) {
            DevLayer & ly = m->layers[(size_t) (il - lf)];
            const std::string p = "blk." + std::to_string(il) + ".";
            if ((rc = upload_f32(m, g.tensor(p + "attn_norm.weight"), &ly.attn_norm, m->E, s))) break;
            if ((rc = upload_f32(m, g.tensor(p + "ffn_norm.weight"), &ly.ffn_norm, m->E, s))) break;
            struct { const char * n; DevMat * d; } mats[] = { { "attn_q", &ly.wq }, { "attn_k", &ly.wk }, { "attn_v", &ly.wv }, { "attn_output", &ly.wo },
                                                              { "ffn_gate", &ly.wg }, { "ffn_up", &ly.wu }, { "ffn_down", &ly.wd } };
            // wq | wk | wv share one allocation so that equal-typed neighbours form ONE weight stream (fused QKV mat-vec)
            char * qkv_base = nullptr; size_t qkv_off = 0;
            {
                size_t tot = 0;
                for (int j = 0; j < 3; ++j) {
                    const GgufTensor * t = g.tensor(p + mats[j].n + ".weight");
                    if (t && t->ne.size() == 2 && bamd_is_kquant(t->type)) tot += bamd_stream_bytes(t->type, t->ne[0], ((int64_t) t->ne[1] + 7) / 8 * 8);
                }
                if (dev_alloc(m->allocs, (void **) &qkv_base, tot)) { rc = 1; break; }
            }
            int mi = 0;
            for (auto & mm : mats) {
                const GgufTensor * t = g.tensor(p + mm.n + ".weight");
                if (!t) { rc = fail("missing tensor " + p + mm.n + ".weight"); break; }
                void * dst = nullptr;
                if (mi < 3) { dst = qkv_base + qkv_off; if (t->ne.size() == 2 && bamd_is_kquant(t->type)) qkv_off += bamd_stream_bytes(t->type, t->ne[0], ((int64_t) t->ne[1] + 7) / 8 * 8); }
                if ((rc = upload_mat(m, t, *mm.d, false, true, staging, s, dst))) break;
                ++mi;
            }
            if (rc) break;
            if (ly.wq.nrows != m->E || ly.wq.K != m->E || ly.wk.nrows != m->Hkv * m->hd || ly.wk.K != m->E || ly.wv.nrows != m->Hkv * m->hd || ly.wv.K != m->E ||
                ly.wo.nrows != m->E || ly.wo.K != m->E || ly.wg.nrows != m->F || ly.wg.K != m->E || ly.wu.nrows != m->F || ly.wu.K != m->E ||
                ly.wd.nrows != m->E || ly.wd.K != m->F) { rc = fail("layer " + std::to_string(il) + ": unexpected tensor shapes"); break; }
            if (ly.wg.type != ly.wu.type) { rc = fail("ffn_gate and ffn_up must share one quantisation type"); break; }
        }
    } while (0);
    hipStreamSynchronize(s);
    hipFree(staging);
    if (!rc) {
        build_prefill_aux(m, s);
        if (!m->aux_ok && getenv("BAMD_PREFILL_VERBOSE")) fprintf(stderr, "bamd: prompts run without the matrix-core kernels: %s\n", m->aux_why.c_str());
    }
    hipStreamDestroy(s);
    return rc;
}

extern "C" __attribute__((visibility("default"))) bamd_model * bamd_model_load(const char * path, int device, int layer_first, int layer_last, int with_embd, int with_output) {
    bamd_model * m = new bamd_model();
    int rc = 1;
    try { rc = model_load_impl(m, path, device, layer_first, layer_last, with_embd, with_output); }
    catch (const std::exception & e) { g_err = std::string("model load: ") + e.what(); rc = 1; }     // nothing may unwind through the C boundary
    if (rc) { bamd_model_free(m); return nullptr; }
    return m;
}
extern "C" __attribute__((visibility("default"))) void bamd_model_free(bamd_model * m) {
    if (!m) return;
    hipSetDevice(m->device);
    for (void * p : m->allocs) hipFree(p);
    delete m;
}
extern "C" __attribute__((visibility("default"))) int bamd_model_n_vocab(const bamd_model * m) { return m->V; }
extern "C" __attribute__((visibility("default"))) int bamd_model_n_embd(const bamd_model * m) { return m->E; }
extern "C" __attribute__((visibility("default"))) int bamd_model_n_layer(const bamd_model * m) { return m->L; }
extern "C" __attribute__((visibility("default"))) int bamd_model_n_ctx_train(const bamd_model * m) { return m->n_ctx_train; }
extern "C" __attribute__((visibility("default"))) int64_t bamd_model_weight_bytes(const bamd_model * m) { return m->weight_bytes; }
extern "C" __attribute__((visibility("default"))) int64_t bamd_model_tensor_raw(const bamd_model * m, const char * name, void * dst, int64_t cap) {
    const GgufTensor * t = m->file->tensor(name);
    if (!t) { g_err = std::string("no tensor ") + name; return -1; }
    if ((int64_t) t->nbytes > cap) return (int64_t) t->nbytes;
    memcpy(dst, t->data, t->nbytes);
    return (int64_t) t->nbytes;
}

// -------------------------------------------------------------------------------------------------------
static int context_init(bamd_context * c, bamd_model * m, int n_ctx) {
    HIPC(hipSetDevice(m->device));
    c->m = m; c->n_ctx = n_ctx;
    if (n_ctx < 32 || n_ctx % 32) return fail("n_ctx must be a positive multiple of 32");
    HIPC(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const int nl = (int) m->layers.size(), Ekv = m->Hkv * m->hd;
    c->kc.resize((size_t) nl); c->vc.resize((size_t) nl);
    c->n_ctx_pad = (n_ctx + 63) / 64 * 64;
    const size_t kvb = (size_t) c->n_ctx_pad * Ekv * 2;
    for (int i = 0; i < nl; ++i) {                                   // llama_kv_cache_init :2926-3026 — zero-initialised
        if (dev_alloc(c->allocs, (void **) &c->kc[(size_t) i], kvb) || dev_alloc(c->allocs, (void **) &c->vc[(size_t) i], kvb)) return 1;
        HIPC(hipMemsetAsync(c->kc[(size_t) i], 0, kvb, c->stream)); HIPC(hipMemsetAsync(c->vc[(size_t) i], 0, kvb, c->stream));
    }
    {   // RoPE table for every position (host libm, like the reference), uploaded once
        std::vector<float> tab((size_t) n_ctx * m->hd);
        for (int p = 0; p < n_ctx; ++p)
            rope_row(tab.data() + (size_t) p * m->hd, p, m->hd, m->rope_theta, m->rope_freq_scale, m->rope_freqs.empty() ? nullptr : m->rope_freqs.data(),
                     m->rope_ext_factor, m->rope_attn_factor, m->rope_n_ctx_orig, 32.0f, 1.0f);
        if (dev_alloc(c->allocs, (void **) &c->rope, tab.size() * 4)) return 1;
        HIPC(hipMemcpy(c->rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        if (dev_alloc(c->allocs, (void **) &c->rope_cur, (size_t) m->hd * 4)) return 1;
        HIPC(hipMemcpy(c->rope_cur, tab.data(), (size_t) m->hd * 4, hipMemcpyHostToDevice));
    }
    if (dev_alloc(c->allocs, (void **) &c->x, (size_t) m->E * 4) || dev_alloc(c->allocs, (void **) &c->x2, (size_t) m->E * 4) ||
        dev_alloc(c->allocs, (void **) &c->q, (size_t) (m->E + 2 * Ekv) * 4) || dev_alloc(c->allocs, (void **) &c->att, (size_t) m->E * 4) ||
        dev_alloc(c->allocs, (void **) &c->h, (size_t) m->F * 4) || dev_alloc(c->allocs, (void **) &c->scores, (size_t) m->H * c->n_ctx_pad * 4) || dev_alloc(c->allocs, (void **) &c->probs, (size_t) m->H * c->n_ctx_pad * 4) ||
        dev_alloc(c->allocs, (void **) &c->logits, (size_t) m->V * 4) || dev_alloc(c->allocs, (void **) &c->st, sizeof(bamd_step_state))) return 1;
    c->k = c->q + m->E; c->v = c->k + Ekv;                           // q | k | v contiguous: rows of the fused QKV mat-vec
    HIPC(hipMemsetAsync(c->st, 0, sizeof(bamd_step_state), c->stream));
    if (dev_alloc(c->allocs, (void **) &c->co_gran, (size_t) m->H * m->hd * 8 + 64)) return 1;
    HIPC(hipMemsetAsync(c->co_gran, 0, (size_t) m->H * m->hd * 8 + 64, c->stream));
    c->co_err = (uint32_t *) (c->co_gran + (size_t) m->H * m->hd);
    HIPC(hipMemsetAsync(c->logits, 0, (size_t) m->V * 4, c->stream));
    HIPC(hipHostMalloc((void **) &c->logits_host, (size_t) m->V * 4 + 64));     // + the give-up words of the co-launch / engine kernels, read back with the logits
    memset(c->logits_host + m->V, 0, 64);
    c->forced_cap = 4096; c->out_cap = n_ctx + 8;
    if (dev_alloc(c->allocs, (void **) &c->forced, (size_t) c->forced_cap * 4) || dev_alloc(c->allocs, (void **) &c->out_tokens, (size_t) c->out_cap * 4)) return 1;
    HIPC(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" __attribute__((visibility("default"))) bamd_context * bamd_context_new(bamd_model * m, int n_ctx) {
    bamd_context * c = new bamd_context();
    if (context_init(c, m, n_ctx)) { bamd_context_free(c); return nullptr; }
    return c;
}
extern "C" __attribute__((visibility("default"))) void bamd_context_free(bamd_context * c) {
    if (!c) return;
    if (c->m) hipSetDevice(c->m->device);
    if (c->graph) hipGraphExecDestroy(c->graph);
    bamd_aql_free(c->aql); bamd_aql_free(c->aql_step);
    if (c->inbox) hipHostFree(c->inbox);
    for (auto & row : c->sgraph) for (auto & g : row) if (g.exec) hipGraphExecDestroy(g.exec);
    for (void * p : c->allocs) hipFree(p);
    if (c->logits_host) hipHostFree(c->logits_host);
    if (c->attn_bscr) hipFree(c->attn_bscr);
    if (c->samp_pen_host) hipHostFree(c->samp_pen_host);
    if (c->samp_out_host) hipHostFree(c->samp_out_host);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}
extern "C" __attribute__((visibility("default"))) int bamd_n_ctx(const bamd_context * c) { return c->n_ctx; }
extern "C" __attribute__((visibility("default"))) void bamd_kv_cache_clear(bamd_context * c) {
    // llama_kv_cache_clear resets the cell metadata; positions are passed per call here, so only the shifted-cell tracking has anything to forget
    c->cells.active = false; c->cells.has_shift = false; c->n_cached = 0;
}

// -------------------------------------------------------------------------------------------------------
// the static per-token pipeline
// -------------------------------------------------------------------------------------------------------
struct StepTimer {                 // optional per-launch HIP-event timing (bamd_profile_step)
    bool on = false;
    std::vector<hipEvent_t> ev; std::vector<int> cls, kind; std::vector<double> bytes;
    // c: 0 mat-vec, 1 attention, 2 other; k: which launch of the layer (0 qkv, 1 attention, 2 other, 3 wo, 4 gate/up, 5 ffn_down, 6 lm_head)
    void begin(hipStream_t s, int c, double b, int k = -1) { if (!on) return; hipEvent_t a; hipEventCreate(&a); hipEventRecord(a, s); ev.push_back(a); cls.push_back(c); kind.push_back(k < 0 ? c : k); bytes.push_back(b); }
    void end(hipStream_t s) { if (!on) return; hipEvent_t b; hipEventCreate(&b); hipEventRecord(b, s); ev.push_back(b); }
    void cancel() { if (!on) return; hipEventDestroy(ev.back()); ev.pop_back(); cls.pop_back(); kind.pop_back(); bytes.pop_back(); }   // the launch behind begin() did not happen
};

static unsigned long long * tl_next(bamd_context * c) {
    if (!c->tl_base || c->tl_slot >= c->tl_cap) return nullptr;
    return c->tl_base + (size_t) (c->tl_slot++) * BAMD_TL_SLOT_WORDS;
}
static void seg_of(bamd_mv_seg & sg, const DevMat & d, float * out) { sg.w = d.stream; sg.out = out; sg.type = d.type; sg.nrows = d.nrows_pad; sg.nvalid = d.nrows; }

// enqueue the layers of this stage for the token whose hidden state is in c->x; leaves the result in c->x
// pos_hi: the highest position this enqueue (or every replay of the graph being captured) will see.  The single-launch attention
// kernel (one workgroup per query head, serial over the sequence) wins below ~450 positions (measured crossover, 8B shape: 1.73 vs
// 1.85 ms/token at 240, equal at 440, 2.06 vs 1.88 at 740) at any n_ctx (its LDS score rows are sized by the sequence bound, not by n_ctx); longer sequences take the
// three-kernel path, whose cost is nearly flat up to a few thousand positions.
static bool attn_fused_for(const bamd_context * c, int pos_hi) { return g_attn_fused && pos_hi < 448 && !c->cells.active; }   // shifted cells: three-launch path (attn_qk_kernel<.., SH>)
// LDS row length of the single-launch / batched attention kernels for sequences up to position pos_hi: a multiple of 64, independent of n_ctx
static int attn_lds_ld(const bamd_context * c, int pos_hi) { return std::min((pos_hi + 1 + 63) / 64 * 64, c->n_ctx_pad); }
static int enqueue_layers(bamd_context * c, int prefill_mode, hipStream_t s, StepTimer * tm, int pos_hi) {
    bamd_model * m = c->m;
    const int gq = m->H / m->Hkv;
    static const int qk_tiles = [] { const char * e = getenv("BAMD_QK_TILES"); return e ? atoi(e) : 64; }();   // score-kernel workgroups per KV head: 64 = two per CU at Hkv = 8 (16 waves per CU: 2.066 -> 2.038 ms/token at 8000 positions; 128: 2.13)
    const int tiles = std::min(std::max(c->n_ctx / 64, 1), qk_tiles);
    for (size_t il = 0; il < m->layers.size(); ++il) {
        const DevLayer & ly = m->layers[il];
        bamd_mv_args a; memset(&a, 0, sizeof a);
        // 1. q,k,v = W{q,k,v} . Q8_K(rms_norm(x) * attn_norm)          (llama.cpp:8810-8835)
        seg_of(a.seg[0], ly.wq, c->q); a.nseg = 1;
        if (ly.wk.type == ly.wq.type) { a.seg[0].nrows += ly.wk.nrows; a.seg[0].nvalid += ly.wk.nrows; }   // streams and outputs are contiguous: extend
        else { seg_of(a.seg[a.nseg], ly.wk, c->k); a.nseg++; }
        if (ly.wv.type == ly.wk.type) { a.seg[a.nseg - 1].nrows += ly.wv.nrows; a.seg[a.nseg - 1].nvalid += ly.wv.nrows; }
        else { seg_of(a.seg[a.nseg], ly.wv, c->v); a.nseg++; }
        a.x = c->x; a.normw = ly.attn_norm; a.eps = m->eps; a.K = m->E; a.tl = tl_next(c);
        if (tm) tm->begin(s, 0, (double) (ly.wq.bytes + ly.wk.bytes + ly.wv.bytes));
        bamd_launch_matvec(a, BAMD_PRO_NORM, BAMD_EPI_STORE, m->n_cu, s);
        if (tm) tm->end(s);
        // 2. RoPE, KV store, softmax(QK^T) V                               (llama.cpp:8837-8849, :8318-8353)
        bamd_attn_args t; memset(&t, 0, sizeof t);
        t.st = c->st; t.q = c->q; t.k = c->k; t.v = c->v; t.kc = c->kc[il]; t.vc = c->vc[il]; t.rope = c->rope; t.rope_cur = c->rope_cur; t.scores = c->scores; t.probs = c->probs; t.out = c->att;
        t.n_ctx = c->n_ctx_pad; t.hd = m->hd; t.Hkv = m->Hkv;
        // the probability rows in global memory are the data path of the softmax | P.V PAIR only (score rows beyond the LDS); the one-launch softmax + P.V keeps them in
        // LDS and wrote them out for the tests' sake — 2 x 31 KB of stores by sixteen of its workgroups, the ones the launch then ended with
        if (bamd_attention_split_is_ik_clean(t, m->H / m->Hkv)) t.probs = nullptr;
        t.hd = m->hd; t.Hkv = m->Hkv; t.n_ctx = c->n_ctx_pad; t.kq_scale = 1.0f / sqrtf((float) m->hd); t.prefill_mode = prefill_mode;
        // single-launch kernel below 448 positions, three kernels (scores | softmax | P.V) above (attn_fused_for)
        t.lds_ld = std::min(512, c->n_ctx_pad);               // single-launch kernel only (sequences < 448 positions): constant, so captured graphs stay valid as pos advances
        t.cellpos = c->cells.active ? c->cellpos : nullptr;
        // 3. x2 = x + Wo . Q8_K(att)                                        (llama.cpp:8294-8303, :8864)
        memset(&a, 0, sizeof a);
        seg_of(a.seg[0], ly.wo, c->x2); a.nseg = 1; a.x = c->att; a.K = m->E; a.res = c->x;
        // 2 + 3 in ONE launch when the attention takes its single-launch kernel and the wo shape has a co-launch instance: the wo workgroups
        // fetch their weights on the CUs the attention leaves idle (bamd_colaunch.hip)
        bool co = false;
        if (attn_fused_for(c, pos_hi) && !prefill_mode) {
            unsigned long long * tl = tl_next(c);
            t.tl = tl; a.tl = tl;
            if (tm) tm->begin(s, 1, (double) ly.wo.bytes);
            co = bamd_launch_attn_wo(t, gq, a, m->n_cu, c->co_gran, (int) (il & 255), c->co_err, s) == 0;
            if (tm) { if (co) tm->end(s); else tm->cancel(); }
            if (!co && tl) c->tl_slot--;
        }
        if (!co) {
            if (tm) tm->begin(s, 1, 0.0);
            t.tl = tl_next(c);
            if (bamd_launch_attention(t, gq, attn_fused_for(c, pos_hi) ? tiles : -tiles, s)) return fail("attention launch: unsupported head configuration");
            if (tm) tm->end(s);
            a.tl = tl_next(c);
            if (tm) tm->begin(s, 0, (double) ly.wo.bytes, 3);
            bamd_launch_matvec(a, BAMD_PRO_PLAIN, BAMD_EPI_ADD, m->n_cu, s);
            if (tm) tm->end(s);
        }
        // 4. h = silu(Wg . a) * (Wu . a),  a = Q8_K(rms_norm(x2) * ffn_norm)  (llama.cpp:8869-8885)
        memset(&a, 0, sizeof a);
        seg_of(a.seg[0], ly.wg, c->h); seg_of(a.seg[1], ly.wu, c->h); a.nseg = 2; a.x = c->x2; a.normw = ly.ffn_norm; a.eps = m->eps; a.K = m->E; a.tl = tl_next(c);
        if (tm) tm->begin(s, 0, (double) (ly.wg.bytes + ly.wu.bytes), 4);
        bamd_launch_matvec(a, BAMD_PRO_NORM, BAMD_EPI_SILU_MUL, m->n_cu, s);
        if (tm) tm->end(s);
        // 5. x = x2 + Wd . Q8_K(h)                                          (llama.cpp:8885, :8902)
        memset(&a, 0, sizeof a);
        seg_of(a.seg[0], ly.wd, c->x); a.nseg = 1; a.x = c->h; a.K = m->F; a.res = c->x2; a.tl = tl_next(c);
        if (tm) tm->begin(s, 0, (double) ly.wd.bytes, 5);
        bamd_launch_matvec(a, BAMD_PRO_PLAIN, BAMD_EPI_ADD, m->n_cu, s);
        if (tm) tm->end(s);
    }
    return 0;
}
static void enqueue_lm_head(bamd_context * c, hipStream_t s, StepTimer * tm) {
    bamd_model * m = c->m;
    bamd_mv_args a; memset(&a, 0, sizeof a);
    seg_of(a.seg[0], m->output, c->logits); a.nseg = 1; a.x = c->x; a.normw = m->out_norm; a.eps = m->eps; a.K = m->E; a.best_key = &c->st->best_key; a.tl = tl_next(c);
    if (tm) tm->begin(s, 0, (double) m->output.bytes, 6);
    bamd_launch_matvec(a, BAMD_PRO_NORM, BAMD_EPI_ARGMAX, m->n_cu, s);
    if (tm) tm->end(s);
}
static void enqueue_begin(bamd_context * c, int n_forced, int do_embed, hipStream_t s, bool with_slots = false) {
    bamd_model * m = c->m;
    bamd_launch_step_begin(c->st, c->forced, n_forced, c->out_tokens, m->tok_embd.raw, m->tok_embd.type, m->E, m->V, c->x, do_embed, s,
                           with_slots ? c->slots : nullptr, with_slots ? c->cellpos : nullptr, c->rope, c->rope_cur, m->hd);
}

// The tag of a granule hand-over (bamd_colaunch.hip) is (host serial : 12, device step : 12, layer : 8); a word must never already hold the tag a
// consumer is about to wait for.  The serial runs 1 .. 0xffe (0 is what zero-initialised granules carry, 0xfff is reserved) and every time it wraps ALL granule
// vectors of the context are overwritten with 0xff bytes — tag 0xffffffff, which no launch produces — in stream order ahead of the launch that reuses serial 1
// (with a bare 12-bit counter a stale granule of 4096 host calls ago carried the awaited tag and a gather passed without waiting).
static int next_serial(bamd_context * c, hipStream_t s) {
    if (++c->host_serial > 0xffe) {
        c->host_serial = 1;
        const bamd_model * m = c->m;
        if (c->co_gran) HIPC(hipMemsetAsync(c->co_gran, 0xff, (size_t) m->H * m->hd * 8, s));
    }
    return 0;
}
static int set_state(bamd_context * c, int pos_base, hipStream_t s, bool keep_key) {
    bamd_step_state h; memset(&h, 0, sizeof h);
    if (next_serial(c, s)) return 1;
    h.pos_base = pos_base; h.n_ctx = c->n_ctx; h.serial = c->host_serial;
    if (keep_key) {
        // keep best_key (the arg-max of the previous lm_head): rewrite only the leading fields
        HIPC(hipMemcpyAsync(c->st, &h, offsetof(bamd_step_state, best_key), hipMemcpyHostToDevice, s));
    } else HIPC(hipMemcpyAsync(c->st, &h, sizeof h, hipMemcpyHostToDevice, s));
    return 0;
}

// ---- KV cell bookkeeping after position edits (llama_kv_cache_seq_rm / _seq_add / find_slot / update) ------------------------------------
// The reference keeps, per cache cell, the position it holds and a pending rotation delta (llama.cpp:2700-2760).  As long as nobody edits
// positions, cell i holds position i and nothing is tracked here.  Booster's context shift (cpp/bridge.cpp:487-503) is
//     llama_kv_cache_seq_rm (ctx, 0, n_keep, n_keep + n_discard);  llama_kv_cache_seq_add(ctx, 0, n_keep + n_discard, n_past, -n_discard);
// after which cells and positions differ: freed cells are refilled in cell order (find_slot), the attention runs over CELLS (mask by the
// position each holds), and the K rows of the moved cells are re-rotated by their delta before the next evaluation (K-shift).
// the cells are back to "cell i holds position i" with nothing pending (e.g. after a plain truncation, seq_rm(n, -1)): stop tracking, so that
// micro-batches, the device-side greedy loop and the single-launch attention are available again (llama_decode has no such modes to lose)
static void kv_try_deactivate(bamd_context * c) {
    bamd_context::Cells & k = c->cells;
    if (!k.active || k.has_shift) return;
    for (int i = 0; i < c->n_ctx; ++i) if (k.pos[(size_t) i] != (i < k.used ? i : -1)) return;
    if (k.head != (k.used >= c->n_ctx ? 0 : k.used)) return;
    k.active = false;
    c->n_cached = k.used;
    if (c->graph) { hipGraphExecDestroy(c->graph); c->graph = nullptr; }
    for (auto & row : c->sgraph) for (auto & g : row) if (g.exec) { hipGraphExecDestroy(g.exec); g.exec = nullptr; }
}
static int kv_activate(bamd_context * c) {
    // cell metadata is edited on the host and copied with blocking copies: nothing of this context may be in flight (the caller's stage
    // streams are hipStreamNonBlocking, so the null-stream copies below would not wait for them)
    HIPC(hipDeviceSynchronize());
    if (c->cells.active) return 0;
    bamd_context::Cells & k = c->cells;
    k.pos.assign((size_t) c->n_ctx, -1); k.delta.assign((size_t) c->n_ctx, 0);
    const int n = std::min(c->n_cached, c->n_ctx);
    for (int i = 0; i < n; ++i) k.pos[(size_t) i] = i;
    k.used = n; k.head = n >= c->n_ctx ? 0 : n;          // llama_decode_internal: head += n_tokens, wraps to 0 at size (llama.cpp:14743-14748)
    if (!c->cellpos && dev_alloc(c->allocs, (void **) &c->cellpos, (size_t) c->n_ctx_pad * 4)) return 1;
    std::vector<int32_t> h((size_t) c->n_ctx_pad, -1);
    memcpy(h.data(), k.pos.data(), (size_t) c->n_ctx * 4);
    HIPC(hipMemcpy(c->cellpos, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    k.active = true;
    if (c->graph) { hipGraphExecDestroy(c->graph); c->graph = nullptr; }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int bamd_kv_seq_rm(bamd_context * c, int p0, int p1) {     // llama_kv_cache_seq_rm(ctx, 0, p0, p1): llama.cpp:3150-3217
    HIPC(hipSetDevice(c->m->device));
    if (kv_activate(c)) return 1;
    bamd_context::Cells & k = c->cells;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = 0x7fffffff;
    int new_head = c->n_ctx;
    for (int i = 0; i < c->n_ctx; ++i) {
        if (k.pos[(size_t) i] >= p0 && k.pos[(size_t) i] < p1) {           // one sequence: a cell is empty once it leaves it
            if (k.pos[(size_t) i] >= 0) k.used--;
            k.pos[(size_t) i] = -1;
            if (new_head == c->n_ctx) new_head = i;
        }
    }
    if (new_head != c->n_ctx && new_head < k.head) k.head = new_head;
    HIPC(hipMemcpy(c->cellpos, k.pos.data(), (size_t) c->n_ctx * 4, hipMemcpyHostToDevice));
    kv_try_deactivate(c);
    return 0;
}
extern "C" __attribute__((visibility("default"))) int bamd_kv_seq_add(bamd_context * c, int p0, int p1, int delta) {   // llama_kv_cache_seq_add: llama.cpp:3268-3313
    HIPC(hipSetDevice(c->m->device));
    if (kv_activate(c)) return 1;
    bamd_context::Cells & k = c->cells;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = 0x7fffffff;
    if (p0 == p1) return 0;
    int new_head = c->n_ctx;
    for (int i = 0; i < c->n_ctx; ++i) {
        if (k.pos[(size_t) i] >= 0 && k.pos[(size_t) i] >= p0 && k.pos[(size_t) i] < p1) {
            k.has_shift = true;
            k.pos[(size_t) i] += delta; k.delta[(size_t) i] += delta;
            if (k.pos[(size_t) i] < 0) { k.used--; k.pos[(size_t) i] = -1; if (new_head == c->n_ctx) new_head = i; }
        }
    }
    k.head = new_head != c->n_ctx ? new_head : 0;
    HIPC(hipMemcpy(c->cellpos, k.pos.data(), (size_t) c->n_ctx * 4, hipMemcpyHostToDevice));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int bamd_kv_seq_div(bamd_context * c, int p0, int p1, int d) {       // llama_kv_cache_seq_div: llama.cpp:3315-3350 (Self-Extend)
    HIPC(hipSetDevice(c->m->device));
    if (d < 1) return fail("bamd_kv_seq_div: divisor < 1");
    if (kv_activate(c)) return 1;
    bamd_context::Cells & k = c->cells;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = 0x7fffffff;
    if (p0 == p1) return 0;
    for (int i = 0; i < c->n_ctx; ++i) {
        if (k.pos[(size_t) i] >= 0 && k.pos[(size_t) i] >= p0 && k.pos[(size_t) i] < p1) {
            k.has_shift = true;
            const int32_t p_old = k.pos[(size_t) i];
            k.pos[(size_t) i] /= d; k.delta[(size_t) i] += k.pos[(size_t) i] - p_old;
        }
    }
    HIPC(hipMemcpy(c->cellpos, k.pos.data(), (size_t) c->n_ctx * 4, hipMemcpyHostToDevice));
    return 0;
}
// llama_kv_cache_update_internal (llama.cpp:15245-15277): apply the pending K-shift to every layer's K cache, clear the deltas
static int kv_update(bamd_context * c, hipStream_t s) {
    bamd_context::Cells & k = c->cells;
    if (!k.has_shift) return 0;
    bamd_model * m = c->m;
    std::vector<int32_t> vals(1, 0), idx((size_t) c->n_ctx_pad, 0);     // row 0: delta 0 (the reference rotates EVERY cell, most by zero)
    {
        std::vector<int32_t> sorted(k.delta.begin(), k.delta.end());
        std::sort(sorted.begin(), sorted.end());
        sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
        for (int32_t d : sorted) if (d != 0) vals.push_back(d);           // a context shift: one value; Self-Extend: up to one per cell
        for (int i = 0; i < c->n_ctx; ++i) {
            const int32_t d = k.delta[(size_t) i];
            idx[(size_t) i] = d == 0 ? 0 : (int32_t) (std::lower_bound(vals.begin() + 1, vals.end(), d) - vals.begin());
        }
    }
    std::vector<float> tab(vals.size() * (size_t) m->hd);
    for (size_t j = 0; j < vals.size(); ++j)
        rope_row(tab.data() + j * m->hd, vals[j], m->hd, m->rope_theta, m->rope_freq_scale, m->rope_freqs.empty() ? nullptr : m->rope_freqs.data(),
                 m->rope_ext_factor, m->rope_attn_factor, m->rope_n_ctx_orig, 32.0f, 1.0f);   // the parameters of the context's own table (bamd_context_new)
    if (!c->shift_idx && dev_alloc(c->allocs, (void **) &c->shift_idx, (size_t) c->n_ctx_pad * 4)) return 1;
    if (!c->shift_tab) { if (dev_alloc(c->allocs, (void **) &c->shift_tab, (size_t) (c->n_ctx + 1) * m->hd * 4)) return 1; c->shift_tab_cap = c->n_ctx + 1; }
    HIPC(hipMemcpyAsync(c->shift_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, s));
    HIPC(hipMemcpyAsync(c->shift_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    for (size_t il = 0; il < m->layers.size(); ++il) bamd_launch_k_shift(c->kc[il], c->n_ctx, m->Hkv, m->hd, c->shift_idx, c->shift_tab, s);
    HIPC(hipStreamSynchronize(s));                           // (idx / tab are host temporaries)
    k.has_shift = false;
    std::fill(k.delta.begin(), k.delta.end(), 0);
    return 0;
}
// llama_kv_cache_find_slot for one token (llama.cpp:3028-3127) with llama_decode_internal's head handling (:14686-14688, :14743-14748);
// *n_kv = the padded length the attention runs over (:14693-14701)
static int kv_find_slot(bamd_context * c, int pos, int * cell, int * n_kv) {
    bamd_context::Cells & k = c->cells;
    const int size = c->n_ctx;
    if (k.head > k.used + 2) k.head = 0;
    int n_tested = 0;
    for (;;) {
        if (k.head + 1 > size) { n_tested += size - k.head; k.head = 0; continue; }
        if (k.pos[(size_t) k.head] >= 0) { k.head += 1; n_tested += 1; if (n_tested >= size) return fail("KV cache full: no free cell (llama_decode would return 1)"); continue; }
        break;
    }
    *cell = k.head;
    k.pos[(size_t) k.head] = pos; k.used += 1;
    int cell_max = 0;
    for (int i = size; i > 0; --i) if (k.pos[(size_t) (i - 1)] >= 0) { cell_max = i; break; }
    *n_kv = std::min(size, std::max(32, (cell_max + 31) / 32 * 32));
    k.head += 1; if (k.head >= size) k.head = 0;
    return 0;
}

extern "C" int bamd_stage_step(bamd_context * c, int32_t token, const void * token_dev, int pos, const void * hidden_in_dev, void * hidden_out_dev, int want_logits,
                               int prefill_mode, void * hip_stream);
// ---- batched prefill: a micro-batch of T > 1 tokens through the layers at once (llama_decode with n_tokens > 1) -----------
#define BAMD_PREFILL_CAP 512            /* the reference's default n_batch / n_ubatch */
// pos_hi: last position of the micro-batch — its score rows (2 x padded length floats per head) must fit the LDS: 18 432 positions
static bool prefill_batch_supported(const bamd_context * c, int pos_hi) {
    const bamd_model * m = c->m;
    const int gq = m->H / m->Hkv;
    if (!(g_prefill_batch && g_attn_fused && (size_t) attn_lds_ld(c, pos_hi) * 8 <= 144 * 1024 && m->hd <= 256 && (m->hd & 63) == 0 && gq >= 1 && gq <= 8)) return false;
    // every mat-mul needs a kernel: the matrix-core kernels take every K-quant at any K when the model has its side tables; the integer-dot kernel
    // takes any K-quant while 4 tokens of Q8_K activations fit the LDS (K <= 35840; tiles of 8 tokens up to K = 17920)
    auto ok = [&](int type, int K) { return (g_prefill_mfma && m->aux_ok && (type == BAMD_Q4_K || type == BAMD_Q5_K || type == BAMD_Q6_K)) || 4 * bamd_blob_bytes(K) <= 160 * 1024; };
    for (const DevLayer & ly : m->layers)
        if (!ok(ly.wq.type, m->E) || !ok(ly.wk.type, m->E) || !ok(ly.wv.type, m->E) || !ok(ly.wo.type, m->E) || !ok(ly.wg.type, m->E) || !ok(ly.wu.type, m->E) || !ok(ly.wd.type, m->F)) return false;
    return true;
}
// prefill side tables of every layer matrix (bamd_prefill2.hip), built at model load (round 6; round 5 built them lazily inside the first batched evaluation — a
// multi-gigabyte allocation after the contexts existed, and a partial failure mixed kernel generations: ADVICE r5).  The QKV segments are the ones
// enqueue_prefill_batch forms (equal-typed neighbours of the fused wq | wk | wv stream merge into one matrix).  All or nothing.
static void build_prefill_aux(bamd_model * m, hipStream_t s) {
    if (!g_prefill_aux) { m->aux_why = "switched off (BAMD_PREFILL_AUX=0)"; return; }
    if (!bamd_prefill_mfma_supported()) { m->aux_why = "the device refuses the matrix-core kernels' LDS size"; return; }
    struct Item { const void * w; int type, nrows, K; void ** slot; };
    std::vector<Item> items;
    for (DevLayer & ly : m->layers) {
        struct Seg { const void * w; int type, nrows; } seg[3]; int n = 1;
        seg[0] = { ly.wq.stream, ly.wq.type, ly.wq.nrows_pad };
        if (ly.wk.type == ly.wq.type) seg[0].nrows += ly.wk.nrows_pad; else seg[n++] = { ly.wk.stream, ly.wk.type, ly.wk.nrows_pad };
        if (ly.wv.type == ly.wk.type) seg[n - 1].nrows += ly.wv.nrows_pad; else seg[n++] = { ly.wv.stream, ly.wv.type, ly.wv.nrows_pad };
        for (int i = 0; i < n; ++i) items.push_back({ seg[i].w, seg[i].type, seg[i].nrows, m->E, &ly.aux_qkv[i] });
        items.push_back({ ly.wo.stream, ly.wo.type, ly.wo.nrows_pad, m->E, &ly.aux_o });
        items.push_back({ ly.wg.stream, ly.wg.type, ly.wg.nrows_pad, m->E, &ly.aux_g });
        items.push_back({ ly.wu.stream, ly.wu.type, ly.wu.nrows_pad, m->E, &ly.aux_u });
        items.push_back({ ly.wd.stream, ly.wd.type, ly.wd.nrows_pad, m->F, &ly.aux_d });
    }
    size_t need = 0;
    for (const Item & it : items) { const size_t b = bamd_prefill_aux_bytes(it.type, it.nrows, it.K); if (!b) { m->aux_why = "a matrix type / shape without a matrix-core kernel"; return; } need += b + 4096; }
    if (items.empty()) return;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); m->aux_why = "hipMemGetInfo failed"; return; }
    double reserve_gb = 8.0; if (const char * e = getenv("BAMD_PREFILL_AUX_RESERVE_GB")) reserve_gb = atof(e);
    const size_t reserve = (size_t) (reserve_gb * 1073741824.0);
    if (need + reserve > free_b) {
        char b[256]; snprintf(b, sizeof b, "side tables need %.2f GiB, %.2f GiB free, %.1f GiB must stay free (BAMD_PREFILL_AUX_RESERVE_GB)", need / 1073741824.0, free_b / 1073741824.0, reserve_gb);
        m->aux_why = b; return;
    }
    std::vector<void *> got;
    for (const Item & it : items) {
        void * p = nullptr;
        if (hipMalloc(&p, bamd_prefill_aux_bytes(it.type, it.nrows, it.K)) != hipSuccess) {
            (void) hipGetLastError();
            for (void * q : got) hipFree(q);
            for (const Item & jt : items) *jt.slot = nullptr;
            m->aux_why = "allocation failed"; m->aux_bytes = 0; return;
        }
        got.push_back(p); *it.slot = p; m->aux_bytes += (int64_t) bamd_prefill_aux_bytes(it.type, it.nrows, it.K);
        bamd_launch_prefill_aux(it.w, it.type, it.nrows, it.K, p, s);
    }
    for (void * q : got) m->allocs.push_back(q);
    hipStreamSynchronize(s);
    m->aux_ok = true;
}
static int ensure_batch_buffers(bamd_context * c) {
    if (c->bcap) return 0;
    bamd_model * m = c->m;
    const size_t T = BAMD_PREFILL_CAP, Ekv = (size_t) m->Hkv * m->hd;
    if (dev_alloc(c->allocs, (void **) &c->bx, T * m->E * 4) || dev_alloc(c->allocs, (void **) &c->bx2, T * m->E * 4) ||
        dev_alloc(c->allocs, (void **) &c->bqkv, T * (m->E + 2 * Ekv) * 4) || dev_alloc(c->allocs, (void **) &c->batt, T * m->E * 4) ||
        dev_alloc(c->allocs, (void **) &c->bh, T * m->F * 4) ||
        dev_alloc(c->allocs, (void **) &c->bblob, T * bamd_blob_bytes(std::max(m->E, m->F))) ||
        dev_alloc(c->allocs, (void **) &c->bblob16, T * bamd_blob16_bytes(std::max(m->E, m->F)))) return 1;
    c->bcap = (int) T;
    return 0;
}
// one batched mat-mul: K-quant segments with a side table on the matrix-core kernels, the rest on the integer-dot kernel (identical bits either way).
// aux[i]: side table of segment i (null: none)
static int batch_mm(bamd_context * c, bamd_mm_args a, int epi, int T, hipStream_t s, const void * const * aux) {
    bamd_model * m = c->m;
    auto on_mfma = [&](int i) { return g_prefill_mfma && aux[i] && (a.seg[i].type == BAMD_Q4_K || a.seg[i].type == BAMD_Q5_K || a.seg[i].type == BAMD_Q6_K); };
    auto mm_mfma = [&](const bamd_mv_seg & sg, const void * ax, int nv, float * out, const float * res, int e) {
        return bamd_launch_matmul_mfma2(sg.w, ax, sg.type, nv, sg.nrows, a.K, c->bblob16, T, out, res, e, a.ldo, s);
    };
    if (epi == BAMD_EPI_SILU_MUL) {
        if (on_mfma(0) && on_mfma(1) && a.seg[1].type == a.seg[0].type) {
            const int nv = a.seg[0].nvalid > 0 ? a.seg[0].nvalid : a.seg[0].nrows;
            if (mm_mfma(a.seg[0], aux[0], nv, a.seg[0].out, nullptr, BAMD_EPI_STORE)) return 1;   // gate -> h
            // up, with h = silu(gate) * up as its epilogue (every element is read and rewritten by the one lane that owns it)
            if (mm_mfma(a.seg[1], aux[1], nv, a.seg[0].out, a.seg[0].out, BAMD_EPI_SILU_MUL)) return 1;
            return 0;
        }
        return bamd_launch_matmul_batch(a, epi, m->n_cu, s);
    }
    bamd_mm_args rest = a; rest.nseg = 0;
    for (int i = 0; i < a.nseg; ++i) {
        if (on_mfma(i)) {
            const int nv = a.seg[i].nvalid > 0 ? a.seg[i].nvalid : a.seg[i].nrows;
            const float * res = epi == BAMD_EPI_ADD ? a.res + (a.seg[i].out - a.seg[0].out) : nullptr;
            if (mm_mfma(a.seg[i], aux[i], nv, a.seg[i].out, res, res ? BAMD_EPI_ADD : BAMD_EPI_STORE)) return 1;
        } else rest.seg[rest.nseg++] = a.seg[i];
    }
    if (rest.nseg) {
        if (epi == BAMD_EPI_ADD) rest.res = a.res + (rest.seg[0].out - a.seg[0].out);
        return bamd_launch_matmul_batch(rest, epi, m->n_cu, s);
    }
    return 0;
}
// first stage: tokens already in c->forced; later stages: hidden_in [T][E] f32 on this device.  Last stage: leaves the hidden state of
// the LAST token in c->x (for lm_head); other stages: writes hidden_out [T][E] (possibly on the next device: peer copy).
static int enqueue_prefill_batch(bamd_context * c, int T, int n_past, hipStream_t s, const void * hidden_in = nullptr, void * hidden_out = nullptr) {
    bamd_model * m = c->m;
    const int E = m->E, F = m->F, Ekv = m->Hkv * m->hd, ldq = E + 2 * Ekv, gq = m->H / m->Hkv;
    bamd_step_state h; memset(&h, 0, sizeof h);
    h.pos_base = n_past; h.pos = n_past; h.n_ctx = c->n_ctx; h.step = T;
    h.n_kv = std::min(c->n_ctx, (n_past + T + 31) / 32 * 32);
    HIPC(hipMemcpyAsync(c->st, &h, sizeof h, hipMemcpyHostToDevice, s));
    if (m->with_embd) bamd_launch_embed_batch(c->forced, T, m->tok_embd.raw, m->tok_embd.type, E, m->V, c->bx, s);
    else HIPC(hipMemcpyAsync(c->bx, hidden_in, (size_t) T * E * 4, hipMemcpyDeviceToDevice, s));
    for (size_t il = 0; il < m->layers.size(); ++il) {
        const DevLayer & ly = m->layers[il];
        bamd_mm_args a; memset(&a, 0, sizeof a);
        // q,k,v                                                            (llama.cpp:8810-8835)
        bamd_launch_quantize_batch(c->bx, ly.attn_norm, m->eps, E, T, c->bblob, c->bblob16, s);
        seg_of(a.seg[0], ly.wq, c->bqkv); a.nseg = 1;
        if (ly.wk.type == ly.wq.type) { a.seg[0].nrows += ly.wk.nrows; a.seg[0].nvalid += ly.wk.nrows; }
        else { seg_of(a.seg[a.nseg], ly.wk, c->bqkv + E); a.nseg++; }
        if (ly.wv.type == ly.wk.type) { a.seg[a.nseg - 1].nrows += ly.wv.nrows; a.seg[a.nseg - 1].nvalid += ly.wv.nrows; }
        else { seg_of(a.seg[a.nseg], ly.wv, c->bqkv + E + Ekv); a.nseg++; }
        a.blob = c->bblob; a.K = E; a.T = T; a.ldo = ldq;
        if (batch_mm(c, a, BAMD_EPI_STORE, T, s, ly.aux_qkv)) return fail("batched mat-mul: unsupported shape");
        // RoPE, KV store, attention with the T>1 semantics                  (llama.cpp:8837-8849, :8318-8353)
        bamd_attn_args t; memset(&t, 0, sizeof t);
        t.st = c->st; t.q = c->bqkv; t.k = c->bqkv + E; t.v = c->bqkv + E + Ekv; t.kc = c->kc[il]; t.vc = c->vc[il]; t.rope = c->rope; t.out = c->batt;
        t.hd = m->hd; t.Hkv = m->Hkv; t.n_ctx = c->n_ctx_pad; t.kq_scale = 1.0f / sqrtf((float) m->hd); t.prefill_mode = 1;
        t.batch = 1; t.ld_qkv = ldq; t.ld_out = E; t.lds_ld = attn_lds_ld(c, n_past + T - 1); t.batch_pos0p1 = n_past + 1;
        {
            const size_t need = m->hd == 128 ? bamd_attention_batch_mfma_scratch(m->Hkv, gq, T, t.lds_ld) : 0;
            if (need > c->attn_bscr_bytes && !(c->attn_bscr_failed && need >= c->attn_bscr_failed)) {   // (freed with the context; replaced only while nothing of this context is in flight: stream order)
                HIPC(hipStreamSynchronize(s));
                if (c->attn_bscr) hipFree(c->attn_bscr);
                c->attn_bscr = nullptr; c->attn_bscr_bytes = 0;
                // ~ H * T * ld * 4 bytes (0.5 GB at 8 K positions on the 8B shape, 2.4 GB at 18 K on a 70B stage).  If the device cannot spare it the
                // prompt is not lost: without a scratch block the matrix-core launcher declines and attn_batch_kernel (VALU, no scratch) runs
                if (hipMalloc((void **) &c->attn_bscr, need) == hipSuccess) c->attn_bscr_bytes = need;
                else { (void) hipGetLastError(); c->attn_bscr = nullptr; c->attn_bscr_failed = need; }   // not retried for this size or larger (every layer of every micro-batch drained the stream and failed again)
            }
            t.batch_scratch = need && c->attn_bscr_bytes >= need ? c->attn_bscr : nullptr;
        }
        if (bamd_launch_attention_batch(t, gq, T, s)) return fail("batched attention: unsupported head configuration");
        // x2 = x + Wo . att
        bamd_launch_quantize_batch(c->batt, nullptr, 0.f, E, T, c->bblob, c->bblob16, s);
        memset(&a, 0, sizeof a);
        seg_of(a.seg[0], ly.wo, c->bx2); a.nseg = 1; a.blob = c->bblob; a.K = E; a.T = T; a.ldo = E; a.res = c->bx;
        { const void * ax[3] = { ly.aux_o, nullptr, nullptr }; if (batch_mm(c, a, BAMD_EPI_ADD, T, s, ax)) return fail("batched mat-mul: unsupported shape"); }
        // h = silu(Wg . a) * (Wu . a)
        bamd_launch_quantize_batch(c->bx2, ly.ffn_norm, m->eps, E, T, c->bblob, c->bblob16, s);
        memset(&a, 0, sizeof a);
        seg_of(a.seg[0], ly.wg, c->bh); seg_of(a.seg[1], ly.wu, c->bh); a.nseg = 2; a.blob = c->bblob; a.K = E; a.T = T; a.ldo = F;
        { const void * ax[3] = { ly.aux_g, ly.aux_u, nullptr }; if (batch_mm(c, a, BAMD_EPI_SILU_MUL, T, s, ax)) return fail("batched mat-mul: unsupported shape"); }
        // x = x2 + Wd . h
        bamd_launch_quantize_batch(c->bh, nullptr, 0.f, F, T, c->bblob, c->bblob16, s);
        memset(&a, 0, sizeof a);
        seg_of(a.seg[0], ly.wd, c->bx); a.nseg = 1; a.blob = c->bblob; a.K = F; a.T = T; a.ldo = E; a.res = c->bx2;
        { const void * ax[3] = { ly.aux_d, nullptr, nullptr }; if (batch_mm(c, a, BAMD_EPI_ADD, T, s, ax)) return fail("batched mat-mul: unsupported shape"); }
    }
    if (m->with_output) HIPC(hipMemcpyAsync(c->x, c->bx + (size_t) (T - 1) * E, (size_t) E * 4, hipMemcpyDeviceToDevice, s));
    else HIPC(hipMemcpyAsync(hidden_out, c->bx, (size_t) T * E * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

// a wo workgroup of a co-launch (bamd_colaunch.hip) that never saw the attention role's flags gave up instead of hanging: the results are void
// the give-up words of the kernels that wait inside a launch (attention || wo co-launch: word 0), copied behind
// the host logits by status_readback and checked once the stream has been synchronised.  A give-up invalidates that step only: the words are
// cleared after reporting, so the context stays usable.
static hipError_t status_readback(bamd_context * c, hipStream_t s) { return hipMemcpyAsync(c->logits_host + c->m->V, c->co_err, 32, hipMemcpyDeviceToHost, s); }
static int co_gave_up(bamd_context * c) {
    uint32_t w[8]; memcpy(w, c->logits_host + c->m->V, 32);
    if (!w[0]) return 0;
    hipMemsetAsync(c->co_err, 0, 32, c->stream);
    hipStreamSynchronize(c->stream);
    memset(c->logits_host + c->m->V, 0, 32);
    return fail("co-launched attention + wo: a workgroup gave up waiting for the attention role (results invalid)");
}
extern "C" __attribute__((visibility("default"))) int bamd_decode(bamd_context * c, const int32_t * tokens, int n_tokens, int n_past) {
    bamd_model * m = c->m;
    if (!m->with_embd || !m->with_output) { fail("bamd_decode needs a stage that owns embedding and output"); return 1; }
    if (!tokens || n_tokens < 1) { fail("n_tokens out of range"); return 1; }
    if (n_past < 0 || n_past > c->n_ctx - n_tokens) { fail("context overflow"); return 1; }
    if (n_tokens > BAMD_PREFILL_CAP && g_prefill_batch) {
        // llama_decode's n_ubatch split (llama.cpp:14615): micro-batches of 512, the logits are those of the last token
        for (int i = 0; i < n_tokens; i += BAMD_PREFILL_CAP)
            if (bamd_decode(c, tokens + i, std::min(BAMD_PREFILL_CAP, n_tokens - i), n_past + i)) return 1;
        return 0;
    }
    if (n_tokens > c->forced_cap) { fail("n_tokens out of range (token by token evaluation takes at most 4096 per call)"); return 1; }
    if (hipSetDevice(m->device) != hipSuccess) { fail("hipSetDevice"); return 1; }
    hipStream_t s = c->stream;
    if (c->cells.active && n_tokens > 1) { fail("after a context shift (bamd_kv_seq_add) tokens are evaluated one per call"); return 1; }
    if (!c->cells.active) c->n_cached = std::max(c->n_cached, n_past + n_tokens);
    if (hipMemcpyAsync(c->forced, tokens, (size_t) n_tokens * 4, hipMemcpyHostToDevice, s) != hipSuccess) { fail("H2D tokens"); return 1; }
    if (n_tokens > 1 && n_tokens <= BAMD_PREFILL_CAP && prefill_batch_supported(c, n_past + n_tokens - 1)) {
        // one micro-batch: every layer once for all tokens (each weight record unpacked once per 8 tokens), lm_head for the last
        if (ensure_batch_buffers(c)) return 1;
        if (enqueue_prefill_batch(c, n_tokens, n_past, s)) return 1;
        enqueue_lm_head(c, s, nullptr);                                  // n_outputs = 1: last token only (llama.cpp:14580-14593)
    } else if (n_tokens == 1) {
        // single token: the captured stage-step graph (state and token are two small host copies, then one graph launch)
        if (bamd_stage_step(c, tokens[0], nullptr, n_past, nullptr, nullptr, 1, 0, s)) return 1;
    } else {
        if (set_state(c, n_past, s, false)) return 1;
        const int prefill_mode = n_tokens > 1;
        for (int t = 0; t < n_tokens; ++t) {
            enqueue_begin(c, n_tokens, 1, s);
            if (enqueue_layers(c, prefill_mode, s, nullptr, n_past + n_tokens)) return 1;
            if (t == n_tokens - 1) enqueue_lm_head(c, s, nullptr);
        }
    }
    {   // a launch that the runtime rejected (LDS / grid beyond the device's limits, wrong architecture) leaves no other trace
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) { fail(std::string("kernel launch failed: ") + hipGetErrorString(le)); return 1; }
    }
    c->logits_host_valid = c->logits_readback;
    if (c->logits_readback && hipMemcpyAsync(c->logits_host, c->logits, (size_t) m->V * 4, hipMemcpyDeviceToHost, s) != hipSuccess) { fail("D2H logits"); return 1; }
    if (status_readback(c, s) != hipSuccess) { fail("D2H co-launch status"); return 1; }
    if (!c->logits_readback) {
        // nothing of this evaluation is read on the host yet (bamd_set_logits_readback(c, 0): the sampler prefilter runs where the lm_head left the logits): do not wait
        // here — the caller's next synchronising call (bamd_logits_shortlist, bamd_get_logits) enqueues behind this evaluation, waits ONCE and checks the status words
        // then.  One host round trip per token instead of two (round 5: 2.31 -> 2.27 ms per token through the bridge at 7850 cached positions)
        c->status_pending = true;
        return 0;
    }
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) { fail(std::string("decode failed: ") + hipGetErrorString(e)); return 1; }
    if (co_gave_up(c)) return 1;
    return 0;
}
// a micro-batch of 2..512 prompt tokens through ONE layer-split stage (bamd_stage_step's batched counterpart): tokens (host) on the
// stage that owns the embedding, hidden_in_dev [n_tokens][n_embd] f32 elsewhere; hidden_out_dev [n_tokens][n_embd] on every stage but
// the last, which computes the logits of the last token when want_logits.  Returns 2 when the shape has no batched kernels (caller
// falls back to bamd_stage_step per token).
extern "C" __attribute__((visibility("default"))) int bamd_stage_prefill(bamd_context * c, const int32_t * tokens, int n_tokens, int n_past, const void * hidden_in_dev,
                                                                           void * hidden_out_dev, int want_logits, void * hip_stream) {
    bamd_model * m = c->m;
    HIPC(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t) hip_stream;
    if (n_tokens < 2 || n_tokens > BAMD_PREFILL_CAP || !prefill_batch_supported(c, n_past + n_tokens - 1)) return 2;
    if (n_past < 0 || n_past + n_tokens > c->n_ctx) return fail("context overflow");
    if (c->cells.active) return fail("after a context shift (bamd_kv_seq_add) tokens are evaluated one per call");
    c->n_cached = std::max(c->n_cached, n_past + n_tokens);
    if (m->with_embd) { if (!tokens) return fail("bamd_stage_prefill: tokens required on the first stage"); HIPC(hipMemcpyAsync(c->forced, tokens, (size_t) n_tokens * 4, hipMemcpyHostToDevice, s)); }
    else if (!hidden_in_dev) return fail("bamd_stage_prefill: hidden_in required");
    if (!m->with_output && !hidden_out_dev) return fail("bamd_stage_prefill: hidden_out required");
    if (ensure_batch_buffers(c)) return 1;
    if (enqueue_prefill_batch(c, n_tokens, n_past, s, hidden_in_dev, hidden_out_dev)) return 1;
    if (m->with_output && want_logits) enqueue_lm_head(c, s, nullptr);
    {   // a launch the runtime rejected (LDS size, grid) leaves no other trace (bamd_decode checks the same way)
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(le));
    }
    return 0;
}
extern "C" __attribute__((visibility("default"))) void bamd_set_prefill_batch(int on) { g_prefill_batch = on ? 1 : 0; g_prefill_mfma = on == 2 ? 0 : 1; }
extern "C" __attribute__((visibility("default"))) const float * bamd_get_logits(bamd_context * c) {
    if (!c->logits_host_valid) {                                          // bamd_set_logits_readback(c, 0): copy on demand
        if (hipSetDevice(c->m->device) != hipSuccess) return nullptr;
        if (hipMemcpyAsync(c->logits_host, c->logits, (size_t) c->m->V * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return nullptr;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return nullptr;
        if (c->status_pending) { c->status_pending = false; if (co_gave_up(c)) return nullptr; }
        c->logits_host_valid = true;
    }
    return c->logits_host;
}
// waits for everything enqueued on the context's stream and reports what a bamd_decode without read-back left pending (the co-launch give-up words): the
// synchronisation point of a caller that evaluated without sampling afterwards (a request that stops during its prompt, or at n_ctx - 4)
extern "C" __attribute__((visibility("default"))) int bamd_synchronize(bamd_context * c) {
    HIPC(hipSetDevice(c->m->device));
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(std::string("bamd_synchronize: ") + hipGetErrorString(e));
    if (c->status_pending) { c->status_pending = false; if (co_gave_up(c)) return 1; }
    return 0;
}
extern "C" __attribute__((visibility("default"))) void bamd_set_logits_readback(bamd_context * c, int on) { c->logits_readback = on != 0; }
extern "C" __attribute__((visibility("default"))) void * bamd_context_stream(bamd_context * c) { return (void *) c->stream; }
extern "C" __attribute__((visibility("default"))) int bamd_set_logits_test(bamd_context * c, const float * logits) {
    if (hipSetDevice(c->m->device) != hipSuccess) return 1;
    HIPC(hipMemcpy(c->logits, logits, (size_t) c->m->V * 4, hipMemcpyHostToDevice));
    c->logits_host_valid = false;
    return 0;
}

// ---- sampler prefilter (kernels in bamd_sampler.hip) --------------------------------------------------------
extern "C" __attribute__((visibility("default"))) int bamd_sampler_tables(bamd_context * c, const uint8_t * halve_class, const float * cutoff_of, int n_vocab) {
    bamd_model * m = c->m;
    if (n_vocab != m->V || !m->with_output) { fail("bamd_sampler_tables: needs the stage that owns the output layer, n_vocab of the model"); return 1; }
    if (hipSetDevice(m->device) != hipSuccess) { fail("hipSetDevice"); return 1; }
    if (!c->samp_cls) {
        const size_t out_bytes = sizeof(bamd_shortlist_head) + (size_t) BAMD_SHORTLIST_CAP * 8;
        if (dev_alloc(c->allocs, (void **) &c->samp_cls, (size_t) m->V) || dev_alloc(c->allocs, (void **) &c->samp_cut, (size_t) m->V * 4) ||
            dev_alloc(c->allocs, (void **) &c->samp_pen, sizeof(bamd_logit_penalty) * BAMD_PENALTY_CAP) || dev_alloc(c->allocs, (void **) &c->samp_out, out_bytes)) return 1;
        HIPC(hipHostMalloc((void **) &c->samp_pen_host, sizeof(bamd_logit_penalty) * BAMD_PENALTY_CAP));
        HIPC(hipHostMalloc((void **) &c->samp_out_host, out_bytes));
    }
    HIPC(hipMemcpy(c->samp_cls, halve_class, (size_t) m->V, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(c->samp_cut, cutoff_of, (size_t) m->V * 4, hipMemcpyHostToDevice));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int bamd_logits_shortlist(bamd_context * c, const bamd_logit_penalty * pen, int n_pen, int halve,
                                                                            bamd_shortlist_head * head, int32_t * ids, float * vals, void * hip_stream) {
    bamd_model * m = c->m;
    if (!c->samp_cls) { fail("bamd_logits_shortlist before bamd_sampler_tables"); return 1; }
    if (n_pen < 0 || n_pen > BAMD_PENALTY_CAP) { fail("bamd_logits_shortlist: too many penalty entries"); return 1; }
    if (hipSetDevice(m->device) != hipSuccess) { fail("hipSetDevice"); return 1; }
    hipStream_t s = (hipStream_t) hip_stream;
    const size_t out_bytes = sizeof(bamd_shortlist_head) + (size_t) BAMD_SHORTLIST_CAP * 8;
    if (n_pen) {
        memcpy(c->samp_pen_host, pen, sizeof(bamd_logit_penalty) * (size_t) n_pen);
        HIPC(hipMemcpyAsync(c->samp_pen, c->samp_pen_host, sizeof(bamd_logit_penalty) * (size_t) n_pen, hipMemcpyHostToDevice, s));
    }
    bamd_shortlist_head * dh = (bamd_shortlist_head *) c->samp_out;
    int32_t * dids = (int32_t *) (c->samp_out + sizeof(bamd_shortlist_head));
    float * dvals = (float *) (c->samp_out + sizeof(bamd_shortlist_head) + (size_t) BAMD_SHORTLIST_CAP * 4);
    bamd_launch_sampler_shortlist(c->logits, c->samp_pen, n_pen, c->samp_cls, halve, c->samp_cut, m->V, dh, dids, dvals, BAMD_SHORTLIST_CAP, s);
    c->logits_host_valid = false;
    HIPC(hipMemcpyAsync(c->samp_out_host, c->samp_out, out_bytes, hipMemcpyDeviceToHost, s));
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) { fail(std::string("sampler prefilter failed: ") + hipGetErrorString(e)); return 1; }
    if (c->status_pending) { c->status_pending = false; if (co_gave_up(c)) return 1; }      // the evaluation in front of this call (bamd_decode without read-back) was not waited for
    memcpy(head, c->samp_out_host, sizeof *head);
    const int n = std::min(std::max(head->count, 0), BAMD_SHORTLIST_CAP);
    memcpy(ids, c->samp_out_host + sizeof(bamd_shortlist_head), (size_t) n * 4);
    memcpy(vals, c->samp_out_host + sizeof(bamd_shortlist_head) + (size_t) BAMD_SHORTLIST_CAP * 4, (size_t) n * 4);
    return 0;
}

static int build_graph(bamd_context * c, int pos_hi) {
    hipStream_t s = c->stream;
    hipGraph_t g = nullptr;
    HIPC(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    enqueue_begin(c, 0, 1, s, c->cells.active);
    int rc = enqueue_layers(c, 0, s, nullptr, pos_hi);
    enqueue_lm_head(c, s, nullptr);
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(&c->graph, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) return fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    return 0;
}

extern "C" __attribute__((visibility("default"))) int bamd_generate_greedy(bamd_context * c, int n_past, int n_steps, int32_t * out_tokens, float * elapsed_ms) {
    bamd_model * m = c->m;
    if (!m->with_embd || !m->with_output) return fail("bamd_generate_greedy needs a stage that owns embedding and output");
    if (n_steps < 1 || n_past < 1 || n_past + n_steps > c->n_ctx || n_steps + 1 > c->out_cap) return fail("bad n_past / n_steps");
    HIPC(hipSetDevice(m->device));
    hipStream_t s = c->stream;
    int attn_hi = n_past + n_steps;
    if (c->cells.active) {
        // cells no longer follow positions (context shift / Self-Extend): the reference keeps generating at full speed (cpp/bridge.cpp:487-503).
        // llama_kv_cache_find_slot depends on the cell metadata only, not on the tokens: run it for all n_steps here and hand the device loop
        // the cell and the padded KV length of every step (step_begin_kernel); pending rotations first (llama_kv_cache_update)
        if (kv_update(c, s)) return 1;
        if (c->slots_cap < n_steps) {
            if (dev_alloc(c->allocs, (void **) &c->slots, (size_t) c->out_cap * 8)) return 1;
            c->slots_cap = c->out_cap;
            if (c->graph) { hipGraphExecDestroy(c->graph); c->graph = nullptr; }            // (captured with the old pointer)
        }
        std::vector<int32_t> hs((size_t) n_steps * 2);
        for (int t = 0; t < n_steps; ++t) {
            int cell = 0, n_kv = 0;
            if (kv_find_slot(c, n_past + t, &cell, &n_kv)) return 1;
            hs[(size_t) 2 * t] = cell; hs[(size_t) 2 * t + 1] = n_kv; attn_hi = std::max(attn_hi, n_kv - 1);
        }
        HIPC(hipMemcpyAsync(c->slots, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, s));
        HIPC(hipStreamSynchronize(s));                                    // (hs is a host temporary)
    } else c->n_cached = std::max(c->n_cached, n_past + n_steps);
    const int fused = (attn_fused_for(c, attn_hi) ? 1 : (c->cells.active ? 2 : 0));     // 2: the shifted-cell kernels and the slot table (other arguments: recapture)
    float ms_steps = 0.f;
    EventPair ev; bool timed_by_events = false;
    // the step as AQL packets on the library's own queue (bamd_aql.h): the launch sequences whose kernels keep the inter-kernel rules of bamd_device.h — the
    // single-launch attention path and the scores | softmax + P.V path of long sequences; anything else (shifted cells, score rows beyond the LDS) replays the hipGraph
    bool aql_path_ok = fused == 1;
    if (fused == 0) {                                                 // long sequences: scores | softmax + P.V, when its kernels are the inter-kernel-clean ones
        bamd_attn_args t; memset(&t, 0, sizeof t); t.hd = m->hd; t.Hkv = m->Hkv; t.n_ctx = c->n_ctx_pad;
        aql_path_ok = bamd_attention_split_is_ik_clean(t, m->H / m->Hkv) != 0;
    }
    const bool want_aql = g_aql && aql_path_ok && !c->aql_failed;
    if (c->aql && (c->aql_key != fused || !want_aql)) { bamd_aql_free(c->aql); c->aql = nullptr; }
    if (want_aql && !c->aql) {
        bamd_aql_recording rec;
        bamd_aql_rec = &rec;
        enqueue_begin(c, 0, 1, s, false);
        const int rc = enqueue_layers(c, 0, s, nullptr, attn_hi);
        enqueue_lm_head(c, s, nullptr);
        bamd_aql_rec = nullptr;
        const char * why = "recording failed";
        if (!rc) c->aql = bamd_aql_build(m->device, rec, &why);
        if (!c->aql) { c->aql_failed = true; if (getenv("BAMD_AQL_VERBOSE")) fprintf(stderr, "bamd: own AQL queue not used: %s\n", why ? why : "?"); }
        c->aql_key = fused;
    }
    if (c->aql) {
        if (set_state(c, n_past, s, true)) return 1;
        HIPC(hipStreamSynchronize(s));                                // the state and everything before it is in memory: the first packet acquires at system scope
        double sec = 0.0; const char * why = nullptr;
        if (bamd_aql_run(c->aql, n_steps, &sec, &why)) return fail(std::string("own AQL queue: ") + (why ? why : "?"));
        ms_steps = (float) (sec * 1e3); c->aql_runs += 1;
    } else {
        if (c->graph && c->graph_fused != fused) { hipGraphExecDestroy(c->graph); c->graph = nullptr; }
        if (!c->graph && build_graph(c, attn_hi)) return 1;
        c->graph_fused = fused;
        if (set_state(c, n_past, s, true)) return 1;
        HIPC(ev.create()); timed_by_events = true;
        HIPC(hipEventRecord(ev.a, s));
        for (int t = 0; t < n_steps; ++t) HIPC(hipGraphLaunch(c->graph, s));
        HIPC(hipEventRecord(ev.b, s));
    }
    enqueue_begin(c, 0, 0, s);                                        // flush the last arg-max into out_tokens
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(out_tokens, c->out_tokens, (size_t) (n_steps + 1) * 4, hipMemcpyDeviceToHost, s));
    HIPC(hipMemcpyAsync(c->logits_host, c->logits, (size_t) m->V * 4, hipMemcpyDeviceToHost, s));
    HIPC(status_readback(c, s));
    HIPC(hipStreamSynchronize(s));
    c->logits_host_valid = true;
    if (co_gave_up(c)) return 1;
    if (timed_by_events) HIPC(hipEventElapsedTime(&ms_steps, ev.a, ev.b));
    if (elapsed_ms) *elapsed_ms = ms_steps;
    return 0;
}
// how many bamd_generate_greedy calls of this context ran on the own AQL queue so far (tests, bench: which path produced the number)
extern "C" __attribute__((visibility("default"))) int bamd_aql_runs(const bamd_context * c) { return c->aql_runs + c->aql_steps; }

// ---- layer-split stage ---------------------------------------------------------------------------------
extern "C" __attribute__((visibility("default"))) int bamd_stage_step(bamd_context * c, int32_t token, const void * token_dev, int pos, const void * hidden_in_dev,
                               void * hidden_out_dev, int want_logits, int prefill_mode, void * hip_stream) {
    bamd_model * m = c->m;
    HIPC(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t) hip_stream;            // NULL = the HIP default (null) stream, as for any HIP API
    if (pos < 0 || pos >= c->n_ctx) return fail("position out of range");
    // ---- own AQL queue (bamd_aql.h): the whole single-stage step — state from a pinned host inbox, layers, lm_head — as packets with fence scope NONE, one host
    //      wait at its end.  For the caller that owns embedding and output, feeds host tokens on the context's own stream and wants the logits (bamd_decode from
    //      the bridge's token loop); layer-split stages, device-side tokens, shifted cells and prefill-mode steps keep the stage graphs below ----
    if (g_aql && !c->aql_failed && m->with_embd && m->with_output && !token_dev && !hidden_in_dev && !hidden_out_dev && want_logits && !prefill_mode &&
        !c->cells.active && s == c->stream && s != nullptr) {
        const int fused = attn_fused_for(c, pos) ? 1 : 0;
        bool ok = fused == 1;
        if (!ok) { bamd_attn_args t; memset(&t, 0, sizeof t); t.hd = m->hd; t.Hkv = m->Hkv; t.n_ctx = c->n_ctx_pad; ok = bamd_attention_split_is_ik_clean(t, m->H / m->Hkv) != 0; }
        if (ok && !c->inbox && hipHostMalloc((void **) &c->inbox, sizeof(bamd_step_state)) != hipSuccess) { (void) hipGetLastError(); c->inbox = nullptr; ok = false; }
        if (ok) {
            if (c->aql_step && c->aql_step_key != fused) { bamd_aql_free(c->aql_step); c->aql_step = nullptr; }
            if (!c->aql_step) {
                bamd_aql_recording rec;
                bamd_aql_rec = &rec;
                bamd_launch_step_begin(c->st, c->forced, 1, c->out_tokens, m->tok_embd.raw, m->tok_embd.type, m->E, m->V, c->x, 1, s, nullptr, nullptr, c->rope, c->rope_cur, m->hd, c->inbox);
                const int rc = enqueue_layers(c, 0, s, nullptr, pos);
                enqueue_lm_head(c, s, nullptr);
                bamd_aql_rec = nullptr;
                const char * why = "recording failed";
                if (!rc) c->aql_step = bamd_aql_build(m->device, rec, &why);
                if (!c->aql_step) { c->aql_failed = true; if (getenv("BAMD_AQL_VERBOSE")) fprintf(stderr, "bamd: own AQL queue not used: %s\n", why ? why : "?"); }
                c->aql_step_key = fused;
            }
        }
        if (ok && c->aql_step) {
            if (next_serial(c, s)) return 1;
            c->n_cached = std::max(c->n_cached, pos + 1);
            HIPC(hipStreamSynchronize(s));                               // whatever this context enqueued before (a prompt micro-batch nobody waited for) is done
            bamd_step_state * in = c->inbox;
            memset(in, 0, sizeof *in);
            in->pos_base = pos; in->n_ctx = c->n_ctx; in->serial = c->host_serial; in->token = token;
            const char * why = nullptr;
            if (bamd_aql_run(c->aql_step, 1, nullptr, &why)) return fail(std::string("own AQL queue: ") + (why ? why : "?"));
            c->aql_steps += 1;
            return 0;
        }
    }
    // state for exactly this token: pos_base = pos, step = 0, one forced token (from the host, or from a device int32)
    if (next_serial(c, s)) return 1;
    bamd_step_state h; memset(&h, 0, sizeof h); h.pos_base = pos; h.n_ctx = c->n_ctx; h.serial = c->host_serial;
    int attn_hi = pos;                                   // what decides single-launch vs three-launch attention
    if (c->cells.active) {
        if (prefill_mode) return fail("after a context shift (bamd_kv_seq_add) tokens are evaluated one per call");
        int cell = 0, n_kv = 0;
        if (kv_update(c, s) || kv_find_slot(c, pos, &cell, &n_kv)) return 1;
        h.cell_plus1 = cell + 1; h.n_kv_fixed = n_kv; attn_hi = n_kv - 1;
        HIPC(hipMemcpyAsync(c->cellpos + cell, &c->cells.pos[(size_t) cell], 4, hipMemcpyHostToDevice, s));
    } else c->n_cached = std::max(c->n_cached, pos + 1);
    HIPC(hipMemcpyAsync(c->st, &h, sizeof h, hipMemcpyHostToDevice, s));
    const int32_t * forced = c->forced;
    if (token_dev) forced = (const int32_t *) token_dev;
    else HIPC(hipMemcpyAsync(c->forced, &token, 4, hipMemcpyHostToDevice, s));
    // everything after the two small host copies is a fixed launch sequence for given pointers: replay it as one hipGraph
    // (a stage of 4 layers is ~22 launches at ~8 us of host time each; the layer-split pipeline is host-bound without this)
    auto enqueue = [&](hipStream_t q) -> int {
        if (m->with_embd) bamd_launch_step_begin(c->st, forced, 1, c->out_tokens, m->tok_embd.raw, m->tok_embd.type, m->E, m->V, c->x, 1, q, nullptr, nullptr, c->rope, c->rope_cur, m->hd);
        else {
            // no embedding on this stage: still advance the device state (pos, n_kv), then take the hidden state
            bamd_launch_step_begin(c->st, c->forced, 0, c->out_tokens, nullptr, BAMD_F32, 0, m->V, c->x, 1, q, nullptr, nullptr, c->rope, c->rope_cur, m->hd);
            HIPC(hipMemcpyAsync(c->x, hidden_in_dev, (size_t) m->E * 4, hipMemcpyDeviceToDevice, q));
        }
        if (enqueue_layers(c, prefill_mode, q, nullptr, attn_hi)) return 1;
        if (m->with_output) { if (want_logits) enqueue_lm_head(c, q, nullptr); }
        else HIPC(hipMemcpyAsync(hidden_out_dev, c->x, (size_t) m->E * 4, hipMemcpyDeviceToDevice, q));
        return 0;
    };
    if (!g_stage_graph || s == nullptr) {                            // the legacy default stream cannot be captured
        if (enqueue(s)) return 1;
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(le));
        return 0;
    }
    bamd_context::StageGraph & sg = c->sgraph[want_logits ? 1 : 0][prefill_mode ? 1 : 0];
    const int fused = (attn_fused_for(c, attn_hi) ? 1 : (c->cells.active ? 2 : 0));     // 2: the shifted-cell kernels (other arguments: recapture)
    if (sg.exec && (sg.token_src != (const void *) forced || sg.hin != hidden_in_dev || sg.hout != hidden_out_dev || sg.fused != fused)) { hipGraphExecDestroy(sg.exec); sg.exec = nullptr; }
    if (!sg.exec) {
        hipGraph_t g = nullptr;
        HIPC(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue(s);
        const hipError_t e = hipStreamEndCapture(s, &g);
        if (rc) { if (g) hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return fail(std::string("stage graph capture: ") + hipGetErrorString(e));
        const hipError_t e2 = hipGraphInstantiate(&sg.exec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e2 != hipSuccess) { sg.exec = nullptr; return fail(std::string("stage graph instantiate: ") + hipGetErrorString(e2)); }
        sg.token_src = forced; sg.hin = hidden_in_dev; sg.hout = hidden_out_dev; sg.fused = fused;
    }
    HIPC(hipGraphLaunch(sg.exec, s));
    return 0;
}
// write the arg-max token of the last lm_head of this (last) stage into a device int32 — no host round trip
extern "C" __attribute__((visibility("default"))) int bamd_stage_token_to(bamd_context * c, void * token_dev, void * hip_stream) {
    hipStream_t s = (hipStream_t) hip_stream;
    bamd_model * m = c->m;
    HIPC(hipSetDevice(m->device));
    // flush-only step_begin: decodes best_key into out_tokens[n_out]; n_out was reset to 0 by bamd_stage_step
    bamd_launch_step_begin(c->st, c->forced, 0, c->out_tokens, nullptr, BAMD_F32, 0, m->V, c->x, 0, s);
    HIPC(hipMemcpyAsync(token_dev, c->out_tokens, 4, hipMemcpyDeviceToDevice, s));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int bamd_stage_argmax(bamd_context * c, void * hip_stream, int32_t * token) {
    hipStream_t s = (hipStream_t) hip_stream;
    HIPC(hipSetDevice(c->m->device));
    bamd_step_state h;
    HIPC(hipMemcpyAsync(&h, c->st, sizeof h, hipMemcpyDeviceToHost, s));
    HIPC(status_readback(c, s));
    HIPC(hipStreamSynchronize(s));
    if (co_gave_up(c)) return 1;                                      // the stage graphs take the co-launch / engine paths as well
    *token = (int32_t) (0xffffffffu - (uint32_t) (h.best_key & 0xffffffffull));
    return 0;
}

// host logits of the last bamd_stage_step(want_logits=1) on the last stage; synchronises `hip_stream`
extern "C" __attribute__((visibility("default"))) const float * bamd_stage_get_logits(bamd_context * c, void * hip_stream) {
    hipStream_t s = (hipStream_t) hip_stream;
    bamd_model * m = c->m;
    if (hipSetDevice(m->device) != hipSuccess) return nullptr;
    if (hipMemcpyAsync(c->logits_host, c->logits, (size_t) m->V * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return nullptr;
    if (status_readback(c, s) != hipSuccess) return nullptr;
    if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
    if (co_gave_up(c)) return nullptr;
    return c->logits_host;
}
// raw GGUF access for the tokenizer side of the bridge
const GgufFile * bamd_model_gguf(const bamd_model * m) { return m->file.get(); }
extern "C" __attribute__((visibility("default"))) int bamd_model_device(const bamd_model * m) { return m->device; }

// ---- measurement -----------------------------------------------------------------------------------------
// one eager step at position pos with a HIP event pair around every launch; per launch kind (StepTimer::begin) the launch count, the summed
// event time and the summed algorithmic bytes; entry 7 = what an EMPTY event pair reads on this stream (median of 9)
static int profile_step_kinds(bamd_context * c, int pos, int * launches, double * ms, double * bytes) {
    bamd_model * m = c->m;
    if (!m->with_embd || !m->with_output) return fail("profile needs a full single-stage model");
    HIPC(hipSetDevice(m->device));
    hipStream_t s = c->stream;
    int32_t tok = 1;
    HIPC(hipMemcpyAsync(c->forced, &tok, 4, hipMemcpyHostToDevice, s));
    if (set_state(c, pos, s, false)) return 1;
    StepTimer tm; tm.on = true;
    tm.begin(s, 2, 0.0); enqueue_begin(c, 1, 1, s); tm.end(s);
    if (enqueue_layers(c, 0, s, &tm, pos)) return 1;
    enqueue_lm_head(c, s, &tm);
    HIPC(hipStreamSynchronize(s));
    for (int i = 0; i < 8; ++i) { launches[i] = 0; ms[i] = 0; bytes[i] = 0; }
    const int n_kv = std::min(c->n_ctx, (pos + 1 + 31) / 32 * 32);
    for (size_t i = 0; i < tm.cls.size(); ++i) {
        float t = 0.f; hipEventElapsedTime(&t, tm.ev[2 * i], tm.ev[2 * i + 1]);
        const int k = tm.kind[i];
        launches[k] += 1; ms[k] += t;
        bytes[k] += tm.cls[i] == 1 ? (double) n_kv * m->Hkv * m->hd * 2 * 2 : tm.bytes[i];
    }
    for (auto e : tm.ev) hipEventDestroy(e);
    {
        std::vector<float> ov;
        for (int i = 0; i < 9; ++i) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a, s); hipEventRecord(b, s); hipStreamSynchronize(s);
            float t = 0.f; hipEventElapsedTime(&t, a, b); ov.push_back(t);
            hipEventDestroy(a); hipEventDestroy(b);
        }
        std::sort(ov.begin(), ov.end());
        launches[7] = 9; ms[7] = ov[4]; bytes[7] = 0;
    }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int bamd_profile_step_kinds(bamd_context * c, int pos, int * launches, double * ms, double * bytes) {
    return profile_step_kinds(c, pos, launches, ms, bytes);
}
// the same folded into three classes: [0] all mat-vec launches, [1] attention, [2] other, [3] the empty event pair
extern "C" __attribute__((visibility("default"))) int bamd_profile_step(bamd_context * c, int pos, int * launches, double * ms, double * bytes) {
    int l[8]; double t[8], b[8];
    if (profile_step_kinds(c, pos, l, t, b)) return 1;
    for (int i = 0; i < 4; ++i) { launches[i] = 0; ms[i] = 0; bytes[i] = 0; }
    const int cls_of[7] = { 0, 1, 2, 0, 0, 0, 0 };
    for (int k = 0; k < 7; ++k) { launches[cls_of[k]] += l[k]; ms[cls_of[k]] += t[k]; bytes[cls_of[k]] += b[k]; }
    launches[3] = l[7]; ms[3] = t[7]; bytes[3] = 0;
    return 0;
}

// Phase stamps of one decode step at position `pos` (BAMD_TIMING builds only): the step is captured into a hipGraph whose launches
// carry their stamp blocks, replayed `replays` times back to back (each replay overwrites the stamps: the last one is read), and the
// blocks are copied to `out` ([n_launches][BAMD_TL_SLOT_WORDS] u64, launch order: per layer qkv, attention, wo, gate/up, down; then
// lm_head).  100 MHz device wall clock.  *n_launches receives the number of stamped launches.
extern "C" __attribute__((visibility("default"))) int bamd_timeline_step(bamd_context * c, int pos, int replays, unsigned long long * out, int cap_launches, int * n_launches) {
    bamd_model * m = c->m;
    if (!bamd_timing_enabled()) return fail("bamd_timeline_step: library built without -DBAMD_TIMING (use booster_amd/lib/libbooster_amd_timing.so)");
    if (!m->with_embd || !m->with_output) return fail("timeline needs a full single-stage model");
    HIPC(hipSetDevice(m->device));
    hipStream_t s = c->stream;
    const int cap = (int) m->layers.size() * 5 + 1;
    if (cap > cap_launches) return fail("bamd_timeline_step: output buffer too small");
    const size_t bytes = (size_t) cap * BAMD_TL_SLOT_WORDS * 8;
    OwnedDevMem mem;
    HIPC(hipMalloc(&mem.p, bytes));
    unsigned long long * buf = (unsigned long long *) mem.p;
    HIPC(hipMemsetAsync(buf, 0, bytes, s));
    int32_t tok = 1;
    HIPC(hipMemcpyAsync(c->forced, &tok, 4, hipMemcpyHostToDevice, s));
    c->tl_base = buf; c->tl_slot = 0; c->tl_cap = cap;
    hipGraph_t g = nullptr; OwnedGraphExec ge;
    {
        const hipError_t eb = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        if (eb != hipSuccess) { c->tl_base = nullptr; c->tl_slot = 0; c->tl_cap = 0; return fail(std::string("timeline capture: ") + hipGetErrorString(eb)); }
    }
    enqueue_begin(c, 1, 1, s);
    int rc = enqueue_layers(c, 0, s, nullptr, pos);
    enqueue_lm_head(c, s, nullptr);
    hipError_t e = hipStreamEndCapture(s, &g);
    *n_launches = c->tl_slot;
    c->tl_base = nullptr; c->tl_slot = 0; c->tl_cap = 0;
    if (rc || e != hipSuccess) { if (g) hipGraphDestroy(g); return rc ? rc : fail("timeline capture failed"); }
    e = hipGraphInstantiate(&ge.g, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) { ge.g = nullptr; return fail("timeline graph instantiate failed"); }
    for (int r = 0; r < replays; ++r) {
        if (set_state(c, pos, s, false)) return 1;
        HIPC(hipGraphLaunch(ge.g, s));
    }
    HIPC(hipStreamSynchronize(s));
    HIPC(hipMemcpy(out, buf, (size_t) *n_launches * BAMD_TL_SLOT_WORDS * 8, hipMemcpyDeviceToHost));
    return 0;
}

// -------------------------------------------------------------------------------------------------------
// op-level entry points (host in / host out): thin wrappers that run the SAME kernels on device 0
// -------------------------------------------------------------------------------------------------------
struct Tmp {
    std::vector<void *> p;
    ~Tmp() { for (void * x : p) hipFree(x); }
    void * up(const void * h, size_t n) { void * d = nullptr; if (hipMalloc(&d, n + 4096) != hipSuccess) return nullptr; p.push_back(d); if (h && hipMemcpy(d, h, n, hipMemcpyHostToDevice) != hipSuccess) return nullptr; return d; }
};
static int need_device() {
    if (bamd_device_count() <= 0) return fail("no HIP device available: libbooster_amd has no CPU fallback");
    HIPC(hipSetDevice(0));
    return 0;
}
static int n_cu0() { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 256; return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }

extern "C" __attribute__((visibility("default"))) int bamd_op_quantize_q8_K(const float * x, int64_t k, const float * norm_w, float eps, void * out_blocks) {
    if (need_device()) return 1;
    if (k <= 0 || k % 256) return fail("k must be a positive multiple of 256");
    Tmp t; const size_t ob = (size_t) (k / 256) * 292;
    float * dx = (float *) t.up(x, (size_t) k * 4); float * dw = norm_w ? (float *) t.up(norm_w, (size_t) k * 4) : nullptr; void * dout = t.up(nullptr, ob);
    if (!dx || !dout || (norm_w && !dw)) return fail("device alloc/copy failed");
    HIPC(hipMemset(dout, 0, ob));
    bamd_launch_quantize_q8k_test(dx, dw, eps, (int) k, norm_w != nullptr, dout, nullptr);
    HIPC(hipGetLastError());
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(out_blocks, dout, ob, hipMemcpyDeviceToHost));
    return 0;
}

static int op_matvec(int type, const void * wA, const void * wB, int nrows, int k, const float * x, const float * norm_w, float eps,
                     const float * residual, float * y, int epi, int mode) {
    if (need_device()) return 1;
    if (!bamd_is_kquant(type) || k <= 0 || k % 256 || nrows <= 0) return fail("bad type/shape");
    const int nrows_pad = (nrows + 7) / 8 * 8;
    Tmp t; const size_t wb = bamd_row_bytes(type, k) * (size_t) nrows, wbp = bamd_stream_bytes(type, k, nrows_pad);
    void * rawA = t.up(wA, wb), * strA = t.up(nullptr, wbp), * rawB = nullptr, * strB = nullptr;
    if (wB) { rawB = t.up(wB, wb); strB = t.up(nullptr, wbp); }
    if (strA) HIPC(hipMemset(strA, 0, wbp));
    if (strB) HIPC(hipMemset(strB, 0, wbp));
    float * dx = (float *) t.up(x, (size_t) k * 4); float * dw = norm_w ? (float *) t.up(norm_w, (size_t) k * 4) : nullptr;
    float * dres = residual ? (float *) t.up(residual, (size_t) nrows * 4) : nullptr; float * dy = (float *) t.up(nullptr, (size_t) nrows * 4);
    unsigned long long * key = (unsigned long long *) t.up(nullptr, 8);
    if (!rawA || !strA || !dx || !dy || !key || (wB && (!rawB || !strB)) || (norm_w && !dw) || (residual && !dres)) return fail("device alloc/copy failed");
    HIPC(hipMemset(key, 0, 8));
    bamd_launch_repack(rawA, strA, type, nrows, k, nullptr);
    if (wB) bamd_launch_repack(rawB, strB, type, nrows, k, nullptr);
    bamd_mv_args a; memset(&a, 0, sizeof a);
    a.seg[0].w = strA; a.seg[0].out = dy; a.seg[0].type = type; a.seg[0].nrows = nrows_pad; a.seg[0].nvalid = nrows; a.nseg = 1;
    if (wB) { a.seg[1] = a.seg[0]; a.seg[1].w = strB; a.nseg = 2; }
    a.x = dx; a.normw = dw; a.eps = eps; a.K = k; a.res = dres; a.best_key = key; a.mode = mode;
    bamd_launch_matvec(a, norm_w ? BAMD_PRO_NORM : BAMD_PRO_PLAIN, epi, n_cu0(), nullptr);
    HIPC(hipGetLastError());
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(y, dy, (size_t) nrows * 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int bamd_op_mul_mat_vec(int type, const void * w_raw, int nrows, int k, const float * x, const float * norm_w, float eps,
                                   const float * residual, float * y, int mode) {
    return op_matvec(type, w_raw, nullptr, nrows, k, x, norm_w, eps, residual, y, residual ? BAMD_EPI_ADD : BAMD_EPI_STORE, mode);
}
// batched mat-mul of T activation rows against one matrix through the prefill kernels: impl 0 = integer-dot kernel, 2 = matrix-core kernel (1 and 3 were the generations removed in round 6)
extern "C" __attribute__((visibility("default"))) int bamd_op_mul_mat_batch(int type, const void * w_raw, int nrows, int k, const float * x, int T, const float * norm_w,
                                                                              float eps, const float * residual, float * y, int impl) {
    if (need_device()) return 1;
    if (!bamd_is_kquant(type) || k <= 0 || k % 256 || nrows <= 0 || T <= 0) return fail("bad type/shape");
    const int nrows_pad = (nrows + 7) / 8 * 8;
    Tmp t; const size_t wb = bamd_row_bytes(type, k) * (size_t) nrows, wbp = bamd_stream_bytes(type, k, nrows_pad);
    void * raw = t.up(w_raw, wb), * str = t.up(nullptr, wbp);
    float * dx = (float *) t.up(x, (size_t) T * k * 4); float * dw = norm_w ? (float *) t.up(norm_w, (size_t) k * 4) : nullptr;
    float * dres = residual ? (float *) t.up(residual, (size_t) T * nrows * 4) : nullptr; float * dy = (float *) t.up(nullptr, (size_t) T * nrows * 4);
    void * blob = t.up(nullptr, (size_t) T * bamd_blob_bytes(k)), * blob16 = t.up(nullptr, (size_t) T * bamd_blob16_bytes(k));
    if (!raw || !str || !dx || !dy || !blob || !blob16 || (norm_w && !dw) || (residual && !dres)) return fail("device alloc/copy failed");
    HIPC(hipMemset(str, 0, wbp));
    bamd_launch_repack(raw, str, type, nrows, k, nullptr);
    bamd_launch_quantize_batch(dx, dw, eps, k, T, blob, blob16, nullptr);
    if (impl == 2) {                                                // the matrix-core kernel: side table built here, as the engine builds it at model load
        void * aux = t.up(nullptr, bamd_prefill_aux_bytes(type, nrows_pad, k));
        if (!aux) return fail("device alloc failed");
        bamd_launch_prefill_aux(str, type, nrows_pad, k, aux, nullptr);
        if (bamd_launch_matmul_mfma2(str, aux, type, nrows, nrows_pad, k, blob16, T, dy, dres, dres ? BAMD_EPI_ADD : BAMD_EPI_STORE, nrows, nullptr)) return fail("MFMA path: unsupported type/shape");
    } else {
        bamd_mm_args a; memset(&a, 0, sizeof a);
        a.seg[0].w = str; a.seg[0].out = dy; a.seg[0].type = type; a.seg[0].nrows = nrows_pad; a.seg[0].nvalid = nrows; a.nseg = 1;
        a.blob = (const uint8_t *) blob; a.K = k; a.T = T; a.ldo = nrows; a.res = dres;
        if (bamd_launch_matmul_batch(a, residual ? BAMD_EPI_ADD : BAMD_EPI_STORE, n_cu0(), nullptr)) return fail("batched mat-mul: unsupported shape");
    }
    HIPC(hipGetLastError());
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(y, dy, (size_t) T * nrows * 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int bamd_op_ffn_gate_up(int type, const void * wg_raw, const void * wu_raw, int nrows, int k, const float * x, const float * norm_w,
                                   float eps, float * y) {
    return op_matvec(type, wg_raw, wu_raw, nrows, k, x, norm_w, eps, nullptr, y, BAMD_EPI_SILU_MUL, 0);
}
extern "C" __attribute__((visibility("default"))) int bamd_op_get_row(int type, const void * w_raw, int nrows, int k, int row, float * y) {
    if (need_device()) return 1;
    if (type != BAMD_F32 && type != BAMD_F16 && !bamd_is_kquant(type)) return fail("bad type");
    if (k <= 0 || (bamd_is_kquant(type) && k % 256)) return fail("bad row length");
    if (row < 0 || row >= nrows) return fail("row out of range");
    Tmp t; const size_t wb = bamd_row_bytes(type, k) * (size_t) nrows;
    void * raw = t.up(w_raw, wb); float * dy = (float *) t.up(nullptr, (size_t) k * 4);
    bamd_step_state h; memset(&h, 0, sizeof h); h.n_ctx = 32;
    bamd_step_state * st = (bamd_step_state *) t.up(&h, sizeof h);
    int32_t * forced = (int32_t *) t.up(&row, 4); int32_t * outt = (int32_t *) t.up(nullptr, 64);
    if (!raw || !dy || !st || !forced || !outt) return fail("device alloc/copy failed");
    bamd_launch_step_begin(st, forced, 1, outt, raw, type, k, nrows, dy, 1, nullptr);
    HIPC(hipGetLastError());
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(y, dy, (size_t) k * 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int bamd_op_rope_row(int pos, int n_dims, float freq_base, float freq_scale, const float * freq_factors, float * row) {
    rope_row(row, pos, n_dims, freq_base, freq_scale, freq_factors, 0.0f, 1.0f, 8192, 32.0f, 1.0f);
    return 0;
}
// reference layout <-> chain-major device layout of the KV cache (bamd_device.h, "Attention": kperm / vperm)
static inline int kperm_host(int n) { const int l = n >> 3; return ((l >> 3) << 6) + ((n & 7) << 3) + (l & 7); }
static void kv_to_device_order(const uint16_t * k_ref, const uint16_t * v_ref, int n_ctx, int n_ctx_pad, int Hkv, int hd, std::vector<uint16_t> & kd, std::vector<uint16_t> & vd) {
    const int Ekv = Hkv * hd;
    kd.assign((size_t) n_ctx_pad * Ekv, 0); vd.assign((size_t) Ekv * n_ctx_pad, 0);
    for (int i = 0; i < n_ctx; ++i) for (int h = 0; h < Hkv; ++h) for (int n = 0; n < hd; ++n)
        kd[(size_t) i * Ekv + h * hd + kperm_host(n)] = k_ref[(size_t) i * Ekv + h * hd + n];
    for (int r = 0; r < Ekv; ++r) for (int p = 0; p < n_ctx; ++p)
        vd[(size_t) r * n_ctx_pad + (p & ~63) + ((p & 7) << 3) + ((p & 63) >> 3)] = v_ref[(size_t) r * n_ctx + p];
}
static void kv_from_device_order(uint16_t * k_ref, uint16_t * v_ref, int n_ctx, int n_ctx_pad, int Hkv, int hd, const std::vector<uint16_t> & kd, const std::vector<uint16_t> & vd) {
    const int Ekv = Hkv * hd;
    for (int i = 0; i < n_ctx; ++i) for (int h = 0; h < Hkv; ++h) for (int n = 0; n < hd; ++n)
        k_ref[(size_t) i * Ekv + h * hd + n] = kd[(size_t) i * Ekv + h * hd + kperm_host(n)];
    for (int r = 0; r < Ekv; ++r) for (int p = 0; p < n_ctx; ++p)
        v_ref[(size_t) r * n_ctx + p] = vd[(size_t) r * n_ctx_pad + (p & ~63) + ((p & 7) << 3) + ((p & 63) >> 3)];
}

extern "C" __attribute__((visibility("default"))) int bamd_op_attention(const float * q, const float * k, const float * v, uint16_t * k_cache, uint16_t * v_cache_t,
                                 const float * rope_row_h, int H, int Hkv, int hd, int n_ctx, int pos, int prefill_mode, float * out,
                                 float * probs_h0) {
    const bool split_path = (prefill_mode & 2) != 0;   // bit 1: force the three-kernel (long-context) path
    prefill_mode &= 1;
    if (need_device()) return 1;
    if (H <= 0 || Hkv <= 0 || hd <= 0 || hd % 64 || hd > 256 || n_ctx <= 0 || pos < 0 || pos >= n_ctx || n_ctx % 32 || H % Hkv) return fail("bad attention shape");
    Tmp t; const int Ekv = Hkv * hd; const int n_ctx_pad = (n_ctx + 63) / 64 * 64; const size_t kvb = (size_t) n_ctx_pad * Ekv * 2;
    std::vector<float> rope((size_t) n_ctx * hd, 0.f);
    memcpy(rope.data() + (size_t) pos * hd, rope_row_h, (size_t) hd * 4);
    std::vector<uint16_t> kd, vd;
    kv_to_device_order(k_cache, v_cache_t, n_ctx, n_ctx_pad, Hkv, hd, kd, vd);
    bamd_step_state h; memset(&h, 0, sizeof h); h.pos = pos; h.n_ctx = n_ctx; h.n_kv = std::min(n_ctx, std::max(32, (pos + 1 + 31) / 32 * 32));
    bamd_attn_args a; memset(&a, 0, sizeof a);
    a.st = (bamd_step_state *) t.up(&h, sizeof h);
    a.q = (float *) t.up(q, (size_t) H * hd * 4); a.k = (float *) t.up(k, (size_t) Ekv * 4); a.v = (float *) t.up(v, (size_t) Ekv * 4);
    a.kc = (unsigned short *) t.up(kd.data(), kvb); a.vc = (unsigned short *) t.up(vd.data(), kvb);
    a.rope = (float *) t.up(rope.data(), rope.size() * 4); if (a.rope) a.rope_cur = a.rope + (size_t) pos * hd; a.scores = (float *) t.up(nullptr, (size_t) H * n_ctx_pad * 4);
    a.probs = (float *) t.up(nullptr, (size_t) H * n_ctx_pad * 4); a.out = (float *) t.up(nullptr, (size_t) H * hd * 4);
    if (!a.st || !a.q || !a.k || !a.v || !a.kc || !a.vc || !a.rope || !a.scores || !a.probs || !a.out) return fail("device alloc/copy failed");
    a.hd = hd; a.Hkv = Hkv; a.n_ctx = n_ctx_pad; a.kq_scale = 1.0f / sqrtf((float) hd); a.prefill_mode = prefill_mode;
    { const int tiles = std::min(std::max(n_ctx / 64, 1), 64);
      if (bamd_launch_attention(a, H / Hkv, split_path ? -tiles : tiles, nullptr)) return fail("unsupported head configuration"); }
    HIPC(hipGetLastError());
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(out, a.out, (size_t) H * hd * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(kd.data(), a.kc, kvb, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(vd.data(), a.vc, kvb, hipMemcpyDeviceToHost));
    kv_from_device_order(k_cache, v_cache_t, n_ctx, n_ctx_pad, Hkv, hd, kd, vd);
    if (probs_h0 && split_path) {
        std::vector<float> pp((size_t) n_ctx_pad);
        HIPC(hipMemcpy(pp.data(), a.probs, (size_t) n_ctx_pad * 4, hipMemcpyDeviceToHost));
        for (int p = 0; p < h.n_kv; ++p) probs_h0[p] = pp[(size_t) ((p & ~63) + ((p & 7) << 3) + ((p & 63) >> 3))];
    }
    return 0;
}

// ---- micro-benchmark of one mat-vec launch shape (random resident weights; HIP-event timing of `iters` launches) ----
extern "C" __attribute__((visibility("default"))) int bamd_bench_matvec(int type, int nrows, int k, int pro, int epi, int mode, int iters,
                                                                        float * us_per_launch) {
    if (need_device()) return 1;
    if (!bamd_is_kquant(type) || k % 256 || nrows % 8) return fail("bad type/shape");
    Tmp t; const size_t wb = bamd_row_bytes(type, k) * (size_t) nrows;
    std::vector<uint8_t> hw(wb);
    uint32_t sd = 12345u; for (size_t i = 0; i < wb; ++i) { sd = sd * 1664525u + 1013904223u; hw[i] = (uint8_t) (sd >> 24); }
    const int bb = bamd_block_bytes(type);
    for (size_t b = 0; b < wb / bb; ++b) {                    // sane f16 scales (0x1c00 ~ 0.0039)
        uint8_t * p = hw.data() + b * bb;
        if (type == BAMD_Q6_K) { p[208] = 0x00; p[209] = 0x1c; } else { p[0] = 0; p[1] = 0x1c; p[2] = 0; p[3] = 0x1c; }
    }
    std::vector<float> hx((size_t) k); for (int i = 0; i < k; ++i) { sd = sd * 1664525u + 1013904223u; hx[i] = (float) (int) (sd >> 8) / 8388608.0f - 1.0f; }
    std::vector<float> hn((size_t) k, 1.0f);
    const size_t wbs = bamd_stream_bytes(type, k, nrows);
    void * raw = t.up(hw.data(), wb), * strA = t.up(nullptr, wbs), * strB = epi == BAMD_EPI_SILU_MUL ? t.up(nullptr, wbs) : nullptr;
    float * dx = (float *) t.up(hx.data(), (size_t) k * 4), * dw = (float *) t.up(hn.data(), (size_t) k * 4);
    float * dres = (float *) t.up(nullptr, (size_t) nrows * 4), * dy = (float *) t.up(nullptr, (size_t) nrows * 4);
    unsigned long long * key = (unsigned long long *) t.up(nullptr, 8);
    if (!raw || !strA || !dx || !dw || !dres || !dy || !key) return fail("device alloc/copy failed");
    HIPC(hipMemset(dres, 0, (size_t) nrows * 4)); HIPC(hipMemset(key, 0, 8));
    bamd_launch_repack(raw, strA, type, nrows, k, nullptr);
    if (strB) bamd_launch_repack(raw, strB, type, nrows, k, nullptr);
    bamd_mv_args a; memset(&a, 0, sizeof a);
    a.seg[0].w = strA; a.seg[0].out = dy; a.seg[0].type = type; a.seg[0].nrows = nrows; a.nseg = 1;
    if (strB) { a.seg[1] = a.seg[0]; a.seg[1].w = strB; a.nseg = 2; }
    a.x = dx; a.normw = dw; a.eps = 1e-5f; a.K = k; a.res = dres; a.best_key = key; a.mode = mode;
    const int ncu = n_cu0();
    if (iters < 1) return fail("iters < 1");
    OwnedStream os; HIPC(hipStreamCreate(&os.s));
    hipStream_t s = os.s;
    for (int i = 0; i < 3; ++i) bamd_launch_matvec(a, pro, epi, ncu, s);
    HIPC(hipGetLastError());
    EventPair ev; HIPC(ev.create());
    HIPC(hipEventRecord(ev.a, s));
    for (int i = 0; i < iters; ++i) bamd_launch_matvec(a, pro, epi, ncu, s);
    HIPC(hipEventRecord(ev.b, s));
    HIPC(hipStreamSynchronize(s));
    float ms = 0.f; HIPC(hipEventElapsedTime(&ms, ev.a, ev.b));
    *us_per_launch = ms * 1000.0f / (float) iters;
    return 0;
}
