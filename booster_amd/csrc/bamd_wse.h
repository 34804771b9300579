// bamd_wse.h — the weight-stream engine: a decode step as ONE persistent launch (bamd_wse.hip), driven by per-CU programs the host
// plans once per context (bamd_wse_plan.cpp).  Shared by host and device code.
//
// What it replaces: the per-token walk of the reference's scheduler over the Llama graph — ggml_backend_sched_compute_splits
// (cpp/ggml/src/ggml-backend.c:1751-1844) over build_llama (cpp/src/llama.cpp:8781-8925) — which rounds 1-3 ran as 130 dependent launches.
//
// Structure (MI355X_MICROARCH.md rows ldsdma-fill, prefetch-credit, gather-pass, engine-vs-launches): one workgroup per CU, resident for the
// whole step, with three wave roles:
//   wave 0      LOADER   walks the CU's weight records of ALL ops in program order and copies them HBM -> LDS ring with global_load_lds
//                        (16 KiB slots, 16 x 1 KiB per fill, nt).  Weights do not depend on activations, so the ring runs AHEAD across the
//                        dependency edges: while the CU waits for the next activation vector its next 7 slots are already landing.
//   wave 1      CHAINER  replays the reference's sequential f32 chains (one step per super-block, per lane (row, SIMD lane e)) from the
//                        terms the consumers park, in record order, runs the epilogues (store / + residual / silu(gate) * up / arg-max) and
//                        PUBLISHES the output rows as 8-byte {value, tag} granules (one sc1 store each: the data is its own flag).
//   waves 2..   CONSUMERS gather the op's activation vector (granules of the producing CUs, re-read until every tag matches; or a plain f32
//                        vector written by an earlier launch), RMSNorm + Q8_K it into LDS exactly as the launch kernels' prologue does, then
//                        take the records of the op round-robin from the ring, compute the per-record TERMS {d, fs, dmin, pm} (bamd_device.h:
//                        block_terms — the same code, the same bits) and park them in a small LDS term ring for the chainer.
// The attention of a layer is one more op: on the first H workgroups eight consumer waves run attn_fused_body (bamd_attn_fused.h).
// Every wait is bounded and reports through `err`; nothing is ever reset between steps (tags = host serial, device step, layer).
#pragma once
#include <stdint.h>
#include "bamd_kernels.h"

#define BAMD_WSE_SLOT 16384            /* bytes of one ring slot = one fill = 16 global_load_lds_dwordx4 of one wave */
#define BAMD_WSE_TERM_BYTES 576        /* one parked record: {fs, pm} per lane (512 B) + {d, dmin} per row (64 B) */
#define BAMD_WSE_MISC_BYTES 2048
#define BAMD_WSE_MAX_SLOTS 16
#define BAMD_WSE_MAX_TERMS 128
#define BAMD_WSE_STASH 128             /* gate values a CU keeps between its gate and up pieces (row-groups x 8) */
#define BAMD_WSE_NVEC 8

enum { BAMD_WSE_END = 0, BAMD_WSE_MATVEC = 1, BAMD_WSE_ATTN = 2 };
enum { BAMD_WSE_ACT_REUSE = 0, BAMD_WSE_ACT_GATHER = 1, BAMD_WSE_ACT_NORM = 2 };            /* act: bit 0 gather a new vector, bit 1 RMSNorm * weight first */
enum { BAMD_WSE_EPI_STORE = 0, BAMD_WSE_EPI_ADD = 1, BAMD_WSE_EPI_GATE = 2, BAMD_WSE_EPI_UP = 3, BAMD_WSE_EPI_ARGMAX = 4 };
/* vector ids (bamd_wse_args.vec): XIN = the step's input hidden state (plain f32, written by step_begin / the previous stage),
   X = hidden state between layers, QKV, ATT, X2 = hidden state after attention, HID = silu(gate) * up, XOUT / LOGITS = plain f32 outputs */
enum { BAMD_WSE_V_XIN = 0, BAMD_WSE_V_X = 1, BAMD_WSE_V_QKV = 2, BAMD_WSE_V_ATT = 3, BAMD_WSE_V_X2 = 4, BAMD_WSE_V_HID = 5, BAMD_WSE_V_XOUT = 6, BAMD_WSE_V_LOGITS = 7 };

// one op of one CU's program (64 bytes; read through the scalar cache)
struct bamd_wse_op {
    uint64_t src;          // MATVEC: first byte of this CU's records of the piece (wave-stream layout: row-group major, super-blocks consecutive)
    uint64_t normw;        // RMSNorm weight [K] f32 (act & NORM)
    uint32_t kind;         // BAMD_WSE_*
    uint32_t type;         // BAMD_Q4_K / Q5_K / Q6_K
    uint32_t nb;           // K / 256 (a multiple of 8)
    uint32_t ntask;        // row-groups of 8 rows this CU owns in the piece (consecutive)
    uint32_t row0;         // output row of task 0, lane row 0 (index into the out / residual vectors)
    uint32_t nvalid;       // rows >= nvalid are padding of the stream: never stored
    uint32_t gs0;          // global slot number of the piece's first slot (slots count over the CU's whole program)
    uint32_t rps;          // records per slot (the last slot of the piece may hold fewer); ATTN: the MATVEC pieces before this op in the CU's program
    uint32_t grec0;        // global record number of the piece's first record (a multiple of 8)
    uint8_t act;           // BAMD_WSE_ACT_* bits
    uint8_t actbuf;        // which of the two LDS activation buffers
    uint8_t in_vec, in_tag;    // activation vector and the tag byte its granules carry (the layer that consumes it)
    uint8_t epi;           // BAMD_WSE_EPI_*
    uint8_t out_vec, out_tag;
    uint8_t res_vec, res_tag;  // EPI_ADD: residual vector
    uint8_t layer;         // ATTN: layer (KV cache) index; MATVEC: for the timeline only
    uint8_t tlslot;        // timeline row of this op (0..tl_ops-1), 255 = none
    uint8_t pad;
};

struct bamd_wse_vec { void * p; uint32_t n; uint32_t gran; };     // gran: 1 = 8-byte {value, tag} granules, 0 = plain f32

struct bamd_wse_args {
    const bamd_wse_op * ops;           // [n_cu][ops_per_cu]
    int ops_per_cu;
    int ns, tr, nc, nch;               // ring slots, term-ring records (a power of two), consumer waves, chainer waves (1 or 2)
    uint32_t off_act[2], off_terms, off_misc, off_attn;   // LDS byte offsets (16-byte multiples): activation buffers, term ring, control words, attention scratch
    bamd_wse_vec vec[BAMD_WSE_NVEC];
    const bamd_step_state * st;
    uint32_t * err;                    // [0] give-ups, [1] first code, [2] first workgroup
    unsigned long long * tl;           // optional: [n_cu][tl_ops][8] wall-clock stamps
    int tl_ops;
    unsigned long long * best_key;     // EPI_ARGMAX
    float eps;
    // attention ops
    bamd_attn_args at;                 // everything but q / k / v / out / kc / vc
    unsigned short * const * kc;       // [layers] K caches
    unsigned short * const * vc;       // [layers] V^T caches
    int gq, H;
    int thin;                          // loader: keep one fill outstanding while the consumers of its CU gather (gather-pass)
};

// timeline events (per CU and op): stamps of the s_memrealtime clock (100 MHz)
enum { BAMD_WSE_TL_GATHER0 = 0,        // consumer 0 starts to gather the activation vector
       BAMD_WSE_TL_VALID = 1,          // ... all its granules carry the tag
       BAMD_WSE_TL_ACTREADY = 2,       // activations quantised, barrier passed
       BAMD_WSE_TL_FIRSTREC = 3,       // consumer 0 parked its first record
       BAMD_WSE_TL_LASTREC = 4,        // consumer 0 parked its last record
       BAMD_WSE_TL_CHAIN0 = 5,         // chainer: first chunk of the piece chained
       BAMD_WSE_TL_PUBLISHED = 6,      // chainer: last row-group of the piece published
       BAMD_WSE_TL_LOADED = 7 };       // loader: last slot of the piece issued

// ---- host side: plan the per-CU programs of a model (bamd_wse_plan.cpp; pure host code, unit-tested without a GPU) ----------------------------
struct bamd_wse_mat { uint64_t stream; int type, nrows_pad, nrows, K; };                   // a repacked matrix (DevMat)
struct bamd_wse_layer { bamd_wse_mat wq, wk, wv, wo, wg, wu, wd; uint64_t attn_norm, ffn_norm; };
struct bamd_wse_plan {
    int n_cu, ops_per_cu, ns, tr, nc, tl_ops;
    uint32_t off_act[2], off_terms, off_misc, off_attn;
    size_t lds_bytes;
    bamd_wse_op * ops;                 // malloc'ed [n_cu * ops_per_cu]
    char why[160];                     // when planning fails: the reason
};
// layers [l0, l1) of `L`; first: the input is XIN (plain); with_head: output.weight + arg-max behind the last layer (else XOUT plain);
// attn_lds: bytes of attention scratch (2 score rows + stage).  0 = ok, 1 = this shape has no engine program (plan->why says why)
int bamd_wse_plan_build(bamd_wse_plan * plan, const bamd_wse_layer * L, int l0, int l1, int n_cu, int E, int H, int Hkv, int hd, int F,
                        const bamd_wse_mat * head, uint64_t head_norm, int V, size_t attn_lds, int nc, int lds_limit);
void bamd_wse_plan_free(bamd_wse_plan * plan);
// single-piece test program (bamd_op_wse_matvec): one matrix (or a gate / up pair when wB), plain f32 in and out
int bamd_wse_plan_single(bamd_wse_plan * plan, const bamd_wse_mat * wA, const bamd_wse_mat * wB, uint64_t normw, int epi, int n_cu, int nc, int lds_limit);
// 0 = launched, 1 = refused
int bamd_launch_wse(const bamd_wse_args & a, int n_cu, size_t lds_bytes, hipStream_t s);
int bamd_wse_setup(int head_dim, size_t * static_lds = nullptr);      // once per head size, outside any stream capture (raises the kernels' dynamic-LDS limit; static_lds: the instance's static LDS bytes)
int bamd_wse_selftest_launch(const uint8_t * src, uint32_t * out, hipStream_t s);
