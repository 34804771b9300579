// bamd_kernels.h — host-visible launch interface of the kernel files (bamd_matvec.hip, bamd_attention.hip, bamd_prefill.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bamd.h"   // bamd_logit_penalty, bamd_shortlist_head

enum { BAMD_PRO_PLAIN = 0, BAMD_PRO_NORM = 1 };
enum { BAMD_EPI_STORE = 0, BAMD_EPI_ADD = 1, BAMD_EPI_SILU_MUL = 2, BAMD_EPI_ARGMAX = 3 };

// device-resident decode state: lets a whole decode step (and a captured hipGraph of it) run without host input
struct bamd_step_state {
    int32_t pos_base;              // position of the token of step 0
    int32_t step;                  // steps begun so far
    int32_t pos, n_kv, token;      // this step: position, padded KV length (llama.cpp:14693-14701), token id
    int32_t n_ctx;
    int32_t n_out;                 // arg-max tokens appended to out_tokens so far
    int32_t cell;                  // KV cell this step's token is stored in: = pos until positions were shifted (bamd_kv_seq_add), then the slot
                                   // llama_kv_cache_find_slot picks (llama.cpp:3028-3127)
    int32_t cell_plus1;            // host: cell of step 0 + 1 (0 = cells follow positions)
    int32_t n_kv_fixed;            // host: padded KV length of this step when cells no longer follow positions (0 = from pos)
    int32_t serial;                // host: 12-bit counter of the host calls that set this state (tag of the co-launch flag words, bamd_colaunch.hip)
    unsigned long long best_key;   // arg-max key of the last lm_head (0 = none)
};

struct bamd_mv_seg { const void * w; float * out; int type; int nrows; int nvalid; };   // nrows: multiple of 8 (stream rows, zero-padded); nvalid: real rows (0 = nrows)
struct bamd_mv_args {
    bamd_mv_seg seg[3]; int nseg;
    const float * x;               // f32 activation [K]
    const float * normw;           // RMSNorm weight (BAMD_PRO_NORM)
    float eps; int K;
    const float * res;             // residual (BAMD_EPI_ADD), indexed like seg[0].out
    unsigned long long * best_key; // BAMD_EPI_ARGMAX
    int mode;                      // 0 auto, 1 force one-wave-per-row-group, 2 force split-K (tests); + 16: force the generic kernels
    int cnt_q, cnt_r;              // fast kernels: row-groups per wave slot (mode A) / per workgroup (mode B), quotient and remainder (set by the launcher)
    unsigned long long * tl;       // phase-stamp block of this launch (BAMD_TIMING builds; null = off)
};

struct bamd_attn_args {
    const bamd_step_state * st;
    const float * q, * k, * v;     // f32 [H*hd], [Hkv*hd], [Hkv*hd] of this token (pre-RoPE)
    unsigned short * kc, * vc;     // f16 caches of this layer, chain-major order (bamd_device.h): K [n_ctx][Hkv*hd], V^T [Hkv*hd][n_ctx]; n_ctx % 64 == 0
    const float * rope;            // [n_ctx][hd] (cos,sin) pairs
    const float * rope_cur;        // [hd]: the row of the CURRENT position, left at this fixed address by step_begin_kernel (the long-sequence score kernel requests it
                                   // without a dependent load of the position in front); null = take the row from `rope` (op-level tests, batched prefill)
    float * scores;                // [H][n_ctx] scratch: scores (long-context path)
    float * probs;                 // [H][n_ctx] scratch: probabilities in V^T position order (long-context path)
    float * out;                   // [H*hd]
    int hd, Hkv, n_ctx;
    float kq_scale;
    int prefill_mode;              // 1: KQ with the T>1 semantics of the reference (q -> f16, ggml_vec_dot_f16)
    int batch, ld_qkv, ld_out;     // batched prefill: q/k/v and out are [T][ld_*] f32, token = blockIdx.y, position st->pos + token
    float * batch_scratch;         // batched prefill, more than 512 positions (BAMD_AM_MAXPOS): score rows of the matrix-core kernel (bamd_attention_batch_mfma_scratch bytes); null: the VALU kernel
    int batch_pos0p1;              // batched prefill: the position of token 0, plus one (= st->pos + 1, known to the host: the matrix-core kernel takes it from here
                                   // and starts its requests without a dependent load of the device state); 0 = read st->pos
    int lds_ld;                    // single-launch / batched kernels: floats per score / probability row in LDS — a multiple of 64 that bounds the padded
                                   // sequence length of this launch (or of every replay of the graph it is captured in); 0 = n_ctx
    unsigned long long * tl;       // phase-stamp block of this launch (BAMD_TIMING builds; null = off)
    const int32_t * cellpos;       // [n_ctx] position held by every KV cell, -1 = free (after a context shift: cells no longer follow positions; the mask
                                   // is the reference's "cell.pos > pos or empty -> -inf", llama.cpp:14152-14200); null = cell i holds position i
};

// batched prefill mat-mul: Y[t][row] = W[row,:] . Q8_K(a_t), T tokens
struct bamd_mm_args {
    bamd_mv_seg seg[3]; int nseg;  // as bamd_mv_args; seg[].out = base of the [T][ldo] output of that segment's rows
    const uint8_t * blob;          // [T][bamd_blob_bytes(K)] quantised activations (bamd_launch_quantize_batch)
    int K, T, ldo;                 // ldo: floats between consecutive tokens in every output / residual matrix
    const float * res;             // BAMD_EPI_ADD: residual, indexed like seg[0].out
};

#define BAMD_TL_WG 512                 /* workgroups recorded per launch (phase stamps, BAMD_TIMING builds) */
#define BAMD_TL_WG_WORDS 24            /* per workgroup: [wave 0: 8 phases][wave 7: 8 phases][exit stamp of each of 8 waves] */
#define BAMD_TL_SLOT_WORDS (BAMD_TL_WG * BAMD_TL_WG_WORDS)
int  bamd_timing_enabled(void);   // 1 when the kernels were compiled with -DBAMD_TIMING
void bamd_launch_repack(const void * raw, void * dst, int type, int nrows, int K, hipStream_t s);
void bamd_launch_quantize_q8k_test(const float * x, const float * nw, float eps, int K, int norm, void * out, hipStream_t s);
void bamd_launch_matvec(const bamd_mv_args & a, int pro, int epi, int n_cu, hipStream_t s);
void bamd_launch_step_begin(bamd_step_state * st, const int32_t * forced, int n_forced, int32_t * out_tokens, const void * embd,
                            int embd_type, int E, int V, float * x, int do_embed, hipStream_t s, const int32_t * slots = nullptr, int32_t * cellpos = nullptr,
                            const float * rope = nullptr, float * rope_cur = nullptr, int hd = 0, const bamd_step_state * inbox = nullptr);
int  bamd_launch_attention(const bamd_attn_args & a, int gq, int max_tiles, hipStream_t s);
int  bamd_attention_split_is_ik_clean(const bamd_attn_args & a, int gq);      // the long-sequence path of this shape keeps the inter-kernel rules (bamd_device.h): own-queue replay allowed
// single-launch attention and the wo projection (+ residual) behind it in ONE launch (bamd_colaunch.hip); gran: [H * hd] zero-initialised 8-byte
// granules of the context (the attention output travels through them), il: layer index (part of the tag), err: give-up counter.
// 1 = this shape has no co-launch kernel: issue the two launches
int  bamd_launch_attn_wo(const bamd_attn_args & t, int gq, const bamd_mv_args & wo, int n_cu, unsigned long long * gran, int il, uint32_t * err, hipStream_t s);
// K-shift (build_k_shift, llama.cpp:8482-8512 -> ggml_compute_forward_rope_f16, ggml.c:14169-14290): every cell's K row re-rotated in place by
// the cos / sin row tab[tab_of_cell[cell]] (row 0 = delta 0); kc chain-major f16 [n_cells][Hkv*hd]
void bamd_launch_k_shift(unsigned short * kc, int n_cells, int Hkv, int hd, const int32_t * tab_of_cell, const float * tab, hipStream_t s);
size_t bamd_blob_bytes(int K);
size_t bamd_blob16_bytes(int K);
// blob: int8 activations for matmul_batch_kernel (may be null); blob16: f16 copy for the MFMA kernel (may be null)
void bamd_launch_quantize_batch(const float * x, const float * nw, float eps, int K, int T, void * blob, void * blob16, hipStream_t s);
// K-quant matrices on the matrix cores, exact (bamd_prefill2.hip): y[t][row] = W[row,:] . Q8_K(a_t); epi BAMD_EPI_STORE: out = y; BAMD_EPI_ADD: out = y + res;
// BAMD_EPI_SILU_MUL: out = silu(res) * y (res = the gate projection, may alias out).  The A fragments are built once per 64-row x 64-token workgroup from the
// wave-stream copy and the matrix's load-time side table (aux: bamd_prefill_aux_bytes bytes, filled by bamd_launch_prefill_aux).  1 = type / shape not supported or no table
int  bamd_prefill_mfma_supported(void);      // the current device accepts the kernels' LDS size (asked at model load)
size_t bamd_prefill_aux_bytes(int type, int nrows_pad, int K);
void bamd_launch_prefill_aux(const void * w_stream, int type, int nrows_pad, int K, void * aux, hipStream_t s);
int  bamd_launch_matmul_mfma2(const void * w_stream, const void * aux, int type, int nrows, int nrows_pad, int K, const void * blob16, int T, float * out, const float * res,
                              int epi, int ldo, hipStream_t s);
int  bamd_launch_matmul_batch(const bamd_mm_args & a, int epi, int n_cu, hipStream_t s);      // 1 = shape not supported
void bamd_launch_embed_batch(const int32_t * tokens, int T, const void * embd, int embd_type, int E, int V, float * x, hipStream_t s);
int  bamd_launch_attention_batch(const bamd_attn_args & a, int gq, int T, hipStream_t s);     // 1 = shape not supported
int  bamd_launch_attention_batch_mfma(const bamd_attn_args & a, int gq, int T, hipStream_t s);   // the same on the matrix cores (after the KV store); 1 = shape not covered
size_t bamd_attention_batch_mfma_scratch(int Hkv, int gq, int T, int ld);                         // bytes of a.batch_scratch it needs for sequences of up to ld positions (0: none)
void bamd_launch_sampler_shortlist(float * logits, const bamd_logit_penalty * pen, int n_pen, const uint8_t * halve_class, int halve,
                                   const float * cutoff_of, int V, bamd_shortlist_head * head, int32_t * ids, float * vals, int cap, hipStream_t s);
