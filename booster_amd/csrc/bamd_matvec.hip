// bamd_matvec.hip — single-token mat-vec kernels (mode A: one wave per row-group; mode B: split-K), load-time repack, step begin.
// Numerics contract and reference citations: bamd_device.h.
#include "bamd_matvec_core.h"

// ===========================================================================================================
// Load-time repack: GGUF row-major blocks -> wave-stream records (bamd_formats.h).  One thread per (row, block).
// ===========================================================================================================
__global__ void repack_kernel(const uint8_t * __restrict__ raw, uint8_t * __restrict__ dst, int type, int nrows, int nb) {
    const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t) nrows * nb) return;
    const int row = (int) (idx / nb), i = (int) (idx % nb);
    const int rg = row >> 3, r = row & 7;
    const int bb = type == BAMD_Q4_K ? 144 : type == BAMD_Q5_K ? 176 : 210;
    const uint8_t * src = raw + ((int64_t) row * nb + i) * bb;
    uint8_t * rec = dst + ((int64_t) rg * nb + i) * bamd_record_bytes(type);
    if (type == BAMD_Q4_K || type == BAMD_Q5_K) {
        const uint8_t * qs = src + (type == BAMD_Q4_K ? 16 : 48);
        for (int e = 0; e < 8; ++e)
            for (int j = 0; j < 4; ++j)
                for (int t = 0; t < 4; ++t) rec[(r * 8 + e) * 16 + 4 * j + t] = qs[32 * j + 4 * e + t];
        int hdr_off = 1024;
        if (type == BAMD_Q5_K) {
            for (int e = 0; e < 8; ++e)
                for (int t = 0; t < 4; ++t) rec[1024 + (r * 8 + e) * 4 + t] = src[16 + 4 * e + t];
            hdr_off = 1280;
        }
        // header: d, dmin as in the file, then the eight scales and the eight mins of get_scale_min_k4 (ggml-quants.c:1891-1899), a byte each
#if BAMD_XSCALES
        uint8_t * h = rec + hdr_off + r * 16, * h2 = rec + hdr_off + 128 + r * 4;
        for (int t = 0; t < 4; ++t) h[t] = src[t];
        for (int j = 0; j < 8; ++j) {
            int sc, mn; get_scale_min_k4(j, src + 4, sc, mn);
            h[4 + j] = (uint8_t) sc;
            if (j < 4) h[12 + j] = (uint8_t) mn; else h2[j - 4] = (uint8_t) mn;
        }
#else
        for (int t = 0; t < 16; ++t) rec[hdr_off + r * 16 + t] = src[t];
#endif
    } else {
        const uint8_t * ql = src, * qh = src + 128, * sc = src + 192;
        for (int e = 0; e < 8; ++e) {
            for (int j = 0; j < 4; ++j)
                for (int t = 0; t < 4; ++t) rec[(r * 8 + e) * 16 + 4 * j + t] = ql[32 * j + 4 * e + t];
            for (int m = 0; m < 2; ++m)
                for (int t = 0; t < 4; ++t) rec[1024 + (r * 8 + e) * 8 + 4 * m + t] = qh[32 * m + 4 * e + t];
        }
        for (int hi = 0; hi < 2; ++hi)
            for (int c = 0; c < 8; ++c) rec[1536 + r * 16 + hi * 8 + c] = sc[2 * c + hi];
        rec[1664 + r * 2] = src[208]; rec[1664 + r * 2 + 1] = src[209];
    }
}

// test entry: standard block_q8_K bytes out of the prologue (for parity tests against quantize_row_q8_K)
__global__ void __launch_bounds__(512) quantize_q8k_test_kernel(const float * x, const float * nw, float eps, int K, int norm, uint8_t * out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nb = K >> 8;
    uint32_t * q8 = (uint32_t *) smem; int * S = (int *) (q8 + nb * 64); float * yd = (float *) (S + nb * 8);
    double * red = (double *) (smem + BAMD_ACT_RED_OFF(nb));
    if (norm) { ActPro<true> ap; ap.issue(x, nw, K, wave_id()); ap.finish(x, nw, eps, K, q8, S, yd, red); }
    else { ActPro<false> ap; ap.issue(x, nw, K, wave_id()); ap.finish(x, nw, eps, K, q8, S, yd, red); }
    for (int i = threadIdx.x; i < nb * 64; i += blockDim.x) {
        const int blk = i >> 6, e = (i >> 3) & 7, c = i & 7;
        const uint32_t w = q8[i];
        uint8_t * o = out + (size_t) blk * 292;
        for (int t = 0; t < 4; ++t) o[4 + 32 * c + 4 * e + t] = (uint8_t) (w >> (8 * t));
    }
    for (int i = threadIdx.x; i < nb; i += blockDim.x) *(float *) (out + (size_t) i * 292) = yd[i];
    __syncthreads();
    for (int i = threadIdx.x; i < nb * 16; i += blockDim.x) {          // bsums from the stored int8
        const int blk = i >> 4, j = i & 15;
        const int8_t * q = (const int8_t *) (out + (size_t) blk * 292 + 4);
        int s = 0; for (int t = 0; t < 16; ++t) s += q[j * 16 + t];
        *(int16_t *) (out + (size_t) blk * 292 + 260 + 2 * j) = (int16_t) (yd[blk] == 0.f ? 0 : s);
    }
}

template <int TYPE, typename REC, int EPI, int PRO>
__device__ __forceinline__ void stream_dispatch_depth(const uint8_t * wA, const uint8_t * wB, int nb, int first, int count, int stride,
                                                      float * out, const float * res, const ProArgs & pa, bool do_pro,
                                                      unsigned long long & best, int nvalid) {
    ActPro<PRO == BAMD_PRO_NORM> ap;
    if ((nb & 7) == 0)      stream_segment<TYPE, REC, 8, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, ap, do_pro, do_pro, best, nvalid);
    else if ((nb & 3) == 0) stream_segment<TYPE, REC, 4, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, ap, do_pro, do_pro, best, nvalid);
    else if ((nb & 1) == 0) stream_segment<TYPE, REC, 2, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, ap, do_pro, do_pro, best, nvalid);
    else                    stream_segment<TYPE, REC, 1, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, ap, do_pro, do_pro, best, nvalid);
}

template <int PRO, int EPI>
__global__ void __launch_bounds__(512) matvec_kernel(bamd_mv_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    const int wave = wave_id(), nwaves = blockDim.x >> 6;
    const int slot = blockIdx.x + gridDim.x * wave;          // consecutive row-groups land on different CUs
    const int stride = gridDim.x * nwaves;
    unsigned long long best = 0ull;
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
    bool pro_done = false;
    int off = 0;
    const int nseg = PAIR ? 1 : a.nseg;
    for (int s = 0; s < nseg; ++s) {
        const int nrg = a.seg[s].nrows >> 3;
        // my row-groups inside the concatenated index space [off, off+nrg): g = slot + k*stride
        const int k0 = off <= slot ? 0 : (off - slot + stride - 1) / stride;
        const int g0 = slot + k0 * stride;
        const int count = g0 < off + nrg ? (off + nrg - 1 - g0) / stride + 1 : 0;
        if (count > 0) {
            const int t = a.seg[s].type;
            const uint8_t * wA = (const uint8_t *) a.seg[s].w;
            const uint8_t * wB = PAIR ? (const uint8_t *) a.seg[1].w : wA;
            float * out = a.seg[s].out;
            const float * res = a.res;
            const int nv = a.seg[s].nvalid > 0 ? a.seg[s].nvalid : a.seg[s].nrows;
            if (t == BAMD_Q4_K)      stream_dispatch_depth<BAMD_Q4_K, RecQ4K, EPI, PRO>(wA, wB, nb, g0 - off, count, stride, out, res, pa, !pro_done, best, nv);
            else if (t == BAMD_Q5_K) stream_dispatch_depth<BAMD_Q5_K, RecQ5K, EPI, PRO>(wA, wB, nb, g0 - off, count, stride, out, res, pa, !pro_done, best, nv);
            else                     stream_dispatch_depth<BAMD_Q6_K, RecQ6K, EPI, PRO>(wA, wB, nb, g0 - off, count, stride, out, res, pa, !pro_done, best, nv);
            pro_done = true;
        }
        off += nrg;
    }
    if (!pro_done) { ActPro<PRO == BAMD_PRO_NORM> ap; BAMD_PRO_ISSUE(ap, pa); BAMD_PRO_FINISH(ap, pa); }   // idle waves still owe the block its barriers
    if (EPI == BAMD_EPI_ARGMAX) {
        // wave max -> block max -> one atomic per workgroup
        for (int o = 32; o; o >>= 1) { const unsigned long long ob = __shfl_xor(best, o); best = ob > best ? ob : best; }
        __syncthreads();
        unsigned long long * wb = (unsigned long long *) smem;
        if ((threadIdx.x & 63) == 0) wb[wave] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long b = 0ull;
            for (int w = 0; w < nwaves; ++w) b = wb[w] > b ? wb[w] : b;
            if (b) atomicMax(a.best_key, b);
        }
    }
    TL_STAMP(a.tl, 7);
}

template <int TYPE, typename REC, int EPI, int PRO>
__device__ __forceinline__ void split_dispatch(const uint8_t * w, int nb, int first, int count, int stride, float * out, const float * res,
                                               const ProArgs & pa, bool do_pro, float * part0, int & rgctr, int nvalid) {
    const int nbw = nb >> 3;
    ActPro<PRO == BAMD_PRO_NORM> ap, ap2;
    if (nb & 7) {                                        // uneven K-split (split_supported): NBW = the larger share
        if (nbw == 5)      split_stream<TYPE, REC, 6, 1, 2, EPI, PRO, false, false, true>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
        else if (nbw == 6) split_stream<TYPE, REC, 7, 1, 2, EPI, PRO, false, false, true>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
        else if (nbw == 2) split_stream<TYPE, REC, 3, 2, 2, EPI, PRO, false, false, true>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
        else __builtin_trap();
        return;
    }
    // (records per wave per row-group, row-groups per batch, term buffers): the batch is the prefetch depth.  K = 14336 with M = 2
    // (all of ffn_down's work per workgroup in flight from the first instruction, single-buffered) measured no better for Q4_K and
    // 14 % worse for Q6_K than M = 1: the kernel is instruction-issue bound, not latency bound.
    if (nbw == 2)       split_stream<TYPE, REC, 2, 4, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
    else if (nbw == 7)  split_stream<TYPE, REC, 7, 1, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
    else if (nbw == 4)  split_stream<TYPE, REC, 4, 2, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
    else if (nbw == 1)  split_stream<TYPE, REC, 1, 8, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
    else __builtin_trap();                               // the launcher only picks this kernel for the shapes above
}

// host must check bamd_split_supported(nb) before choosing this kernel
template <int PRO, int EPI>
__global__ void __launch_bounds__(512) matvec_split_kernel(bamd_mv_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    float * part0 = (float *) (smem + BAMD_ACT_RED_OFF(nb) + 16 * sizeof(double));
    int rgctr = 0;
    bool pro_done = false;
    int off = 0;
    const int slot = blockIdx.x, stride = gridDim.x;         // row-groups are dealt to WORKGROUPS here
    for (int s = 0; s < a.nseg; ++s) {
        const int nrg = a.seg[s].nrows >> 3;
        const int k0 = off <= slot ? 0 : (off - slot + stride - 1) / stride;
        const int g0 = slot + k0 * stride;
        const int count = g0 < off + nrg ? (off + nrg - 1 - g0) / stride + 1 : 0;
        if (count > 0) {
            const int t = a.seg[s].type;
            const uint8_t * w = (const uint8_t *) a.seg[s].w;
            const int nv = a.seg[s].nvalid > 0 ? a.seg[s].nvalid : a.seg[s].nrows;
            if (t == BAMD_Q4_K)      split_dispatch<BAMD_Q4_K, RecQ4K, EPI, PRO>(w, nb, g0 - off, count, stride, a.seg[s].out, a.res, pa, !pro_done, part0, rgctr, nv);
            else if (t == BAMD_Q5_K) split_dispatch<BAMD_Q5_K, RecQ5K, EPI, PRO>(w, nb, g0 - off, count, stride, a.seg[s].out, a.res, pa, !pro_done, part0, rgctr, nv);
            else                     split_dispatch<BAMD_Q6_K, RecQ6K, EPI, PRO>(w, nb, g0 - off, count, stride, a.seg[s].out, a.res, pa, !pro_done, part0, rgctr, nv);
            pro_done = true;
        }
        off += nrg;
    }
    if (!pro_done) { ActPro<PRO == BAMD_PRO_NORM> ap; BAMD_PRO_ISSUE(ap, pa); BAMD_PRO_FINISH(ap, pa); }
    TL_STAMP(a.tl, 7);
}


// ===========================================================================================================
// Step begin: pick the token of this step (forced prompt token, or the arg-max of the previous step's logits),
// advance the position, and dequantise its embedding row into the residual stream.
// ===========================================================================================================
// slots (null = cells follow positions, or ONE step whose cell / padded length the host put into the state): after a context shift the device-side
// greedy loop takes the cell and the padded KV length of step k from slots[2 k], slots[2 k + 1] — llama_kv_cache_find_slot does not depend on
// the tokens, so the host runs it for all steps ahead (bamd_generate_greedy) — and records the position in the cell's mask entry (cellpos)
__global__ void __launch_bounds__(1024) step_begin_kernel(bamd_step_state * st, const int32_t * forced, int n_forced,
                                                         int32_t * out_tokens, const uint8_t * embd, int embd_type, int E, int V,
                                                         float * x, int do_embed, const int32_t * slots, int32_t * cellpos,
                                                         const float * rope, float * rope_cur, int hd, const bamd_step_state * inbox) {
    __shared__ int tok_s, pos_s;
    if (threadIdx.x == 0) {
        // the state is inter-kernel data (bamd_device.h): sc1 loads, issued together (ONE round trip for the fields this step needs), sc1 stores
        bamd_step_state h;
        int step, ftok;
        if (inbox) {
            // the host left this step's state in pinned host memory (bamd_stage_step on the own AQL queue: no copy engine, no HIP stream in front of the step):
            // system-scope loads, then the device state is initialised as the host's copy of the whole struct would have left it
            h.pos_base = __hip_atomic_load(&inbox->pos_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            h.n_ctx = __hip_atomic_load(&inbox->n_ctx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            h.cell_plus1 = __hip_atomic_load(&inbox->cell_plus1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            h.n_kv_fixed = __hip_atomic_load(&inbox->n_kv_fixed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            h.serial = __hip_atomic_load(&inbox->serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ftok = __hip_atomic_load(&inbox->token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            h.step = 0; h.n_out = 0; h.best_key = 0ull; step = 0;
            ik_st(&st->pos_base, h.pos_base); ik_st(&st->n_ctx, h.n_ctx); ik_st(&st->cell_plus1, h.cell_plus1); ik_st(&st->n_kv_fixed, h.n_kv_fixed);
            ik_st(&st->serial, h.serial); ik_st(&st->n_out, 0);
        } else {
            h.pos_base = ik_ld(&st->pos_base); h.step = ik_ld(&st->step); h.n_ctx = ik_ld(&st->n_ctx); h.n_out = ik_ld(&st->n_out);
            h.cell_plus1 = ik_ld(&st->cell_plus1); h.n_kv_fixed = ik_ld(&st->n_kv_fixed); h.best_key = ik_ld(&st->best_key);
            step = h.step;
            ftok = step < n_forced ? ik_ld(forced + step) : 0;
        }
        int tok = 0;
        if (h.best_key != 0ull) {                             // arg-max of the previous lm_head, 0 = none ran
            tok = (int) (0xffffffffu - (uint32_t) (h.best_key & 0xffffffffull));
            ik_st(out_tokens + h.n_out, tok); h.n_out += 1;
            ik_st(&st->n_out, h.n_out);
        }
        if (step < n_forced) tok = ftok;
        if (tok < 0 || tok >= V) tok = 0;
        tok_s = tok;
        ik_st(&st->token, tok);
        if (do_embed) {
            h.pos = h.pos_base + step;
            pos_s = h.pos;
            h.cell = h.cell_plus1 ? h.cell_plus1 - 1 + step : h.pos;
            int n_kv = (h.pos + 1 + 31) / 32 * 32;
            if (n_kv > h.n_ctx) n_kv = h.n_ctx;
            h.n_kv = h.n_kv_fixed ? h.n_kv_fixed : n_kv;
            if (slots) { h.cell = ik_ld(slots + 2 * step); h.n_kv = ik_ld(slots + 2 * step + 1); ik_st(cellpos + h.cell, h.pos); }
            ik_st(&st->pos, h.pos); ik_st(&st->cell, h.cell); ik_st(&st->n_kv, h.n_kv);
            ik_st(&st->step, step + 1);
            ik_st(&st->best_key, 0ull);                       // a flush-only call leaves the key for the next generate call
        }
    }
    __syncthreads();
    if (!do_embed) return;
    // the cos / sin row of this step's position at a fixed address (bamd_attn_args::rope_cur)
    if (rope_cur && (int) threadIdx.x < hd) ik_st(rope_cur + threadIdx.x, rope[(size_t) pos_s * hd + threadIdx.x]);
    embed_row(embd, embd_type, E, tok_s, x);
}


// ===========================================================================================================
// launchers
// ===========================================================================================================
int bamd_timing_enabled(void) {
#ifdef BAMD_TIMING
    return 1;
#else
    return 0;
#endif
}
void bamd_launch_repack(const void * raw, void * dst, int type, int nrows, int K, hipStream_t s) {
    const int nb = K >> 8;
    const int64_t n = (int64_t) nrows * nb;
    BAMD_LAUNCH(repack_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, (const uint8_t *) raw, (uint8_t *) dst, type, nrows, nb);
}

void bamd_launch_quantize_q8k_test(const float * x, const float * nw, float eps, int K, int norm, void * out, hipStream_t s) {
    BAMD_LAUNCH(quantize_q8k_test_kernel, dim3(1), dim3(512), act_lds_bytes(K), s, x, nw, eps, K, norm, (uint8_t *) out);
}

template <int PRO>
static void launch_mv_epi(const bamd_mv_args & a, int epi, int grid, hipStream_t s) {
    const size_t lds = act_lds_bytes(a.K);
    switch (epi) {
        case BAMD_EPI_STORE:    BAMD_LAUNCH((matvec_kernel<PRO, BAMD_EPI_STORE>),    dim3(grid), dim3(512), lds, s, a); break;
        case BAMD_EPI_ADD:      BAMD_LAUNCH((matvec_kernel<PRO, BAMD_EPI_ADD>),      dim3(grid), dim3(512), lds, s, a); break;
        case BAMD_EPI_SILU_MUL: BAMD_LAUNCH((matvec_kernel<PRO, BAMD_EPI_SILU_MUL>), dim3(grid), dim3(512), lds, s, a); break;
        case BAMD_EPI_ARGMAX:   BAMD_LAUNCH((matvec_kernel<PRO, BAMD_EPI_ARGMAX>),   dim3(grid), dim3(512), lds, s, a); break;
    }
}
template <int PRO>
static void launch_mv_split(const bamd_mv_args & a, int epi, int grid, hipStream_t s) {
    const int nb = a.K >> 8;
    const int nbw = nb >> 3;
    const int M = (nb & 7) ? (nbw == 2 ? 2 : 1) : nbw == 2 ? 4 : nbw == 7 ? 1 : nbw == 4 ? 2 : 8, NBUF = 2;   // must match split_dispatch
    const size_t lds = act_lds_bytes(a.K) + 16 + (size_t) NBUF * M * nb * 256 * 4;   // 112..128 KiB of term buffers
    if (epi == BAMD_EPI_ADD) BAMD_LAUNCH((matvec_split_kernel<PRO, BAMD_EPI_ADD>),   dim3(grid), dim3(512), lds, s, a);
    else                     BAMD_LAUNCH((matvec_split_kernel<PRO, BAMD_EPI_STORE>), dim3(grid), dim3(512), lds, s, a);
}

// K / 256 a multiple of 8 with 1, 2, 4 or 7 records per wave; or uneven shares of 2-3, 5-6, 6-7 records (17..23, 41..47, 49..55 super-blocks:
// Llama-2-13B's n_embd 5120, Llama-2-7B's n_ff 11008, Llama-2-13B's n_ff 13824).  (9..15 super-blocks — Llama-3.2-3B's n_embd 3072 — measured no
// faster split than with one wave per row-group: 6.9 / 4.8 us against 6.2 / 5.1 for its QKV / wo.)
static bool split_supported(int nb) { const int nbw = nb >> 3; return (nb & 7) == 0 ? (nbw == 1 || nbw == 2 || nbw == 4 || nbw == 7) : (nbw == 2 || nbw == 5 || nbw == 6); }

static const bool g_mv_generic = [] { const char * e = getenv("BAMD_MV_GENERIC"); return e && e[0] == '1'; }();

void bamd_launch_matvec(const bamd_mv_args & a, int pro, int epi, int n_cu, hipStream_t s) {
    int nrg = 0;
    if (epi == BAMD_EPI_SILU_MUL) nrg = a.seg[0].nrows >> 3;
    else for (int i = 0; i < a.nseg; ++i) nrg += a.seg[i].nrows >> 3;
    const int cus = n_cu > 0 ? n_cu : 256;
    // few row-groups: split K over the 8 waves of a workgroup (mode B); otherwise one wave per row-group (mode A)
    const bool can_split = (epi == BAMD_EPI_STORE || epi == BAMD_EPI_ADD) && split_supported(a.K >> 8);
    // differently typed segments (wq|wk Q4_K + wv Q6_K): a split-K workgroup would stream them one after the other, each with its
    // own ring fill; with one wave per row-group every wave has a single row-group of a single type
    const bool mixed = a.nseg > 1 && epi == BAMD_EPI_STORE && nrg <= 8 * cus;
    const int fmode = a.mode & 15;
    const bool split = fmode == 2 ? can_split : fmode == 1 ? false : (can_split && nrg < 8 * cus && !mixed);
    int grid = cus;                                          // one 8-wave workgroup per CU
    if (grid > nrg) grid = nrg;
    if (grid < 1) grid = 1;
    const bool generic = g_mv_generic || a.mode >= 16;       // mode bit 4: force the generic kernels (tests)
    // K = 28672 (the 70B ffn_down): a fast split-K instance only (bamd_matvec_fast_b.hip); the generic split kernel has no table entry for 14 records per wave
    if (!generic && fmode != 1 && nrg < 8 * cus && bamd_launch_fast_b112_supported(a.K, pro, epi, a.nseg, a.seg[0].type) && bamd_launch_fast_b(a, pro, epi, grid, s)) return;
    if (!generic && mixed && fmode == 0 && can_split && bamd_launch_fast_mixed(a, pro, epi, grid, s)) return;
    if (split) {
        if (!generic && bamd_launch_fast_b(a, pro, epi, grid, s)) return;
        if (pro == BAMD_PRO_NORM) launch_mv_split<BAMD_PRO_NORM>(a, epi, grid, s);
        else                      launch_mv_split<BAMD_PRO_PLAIN>(a, epi, grid, s);
        return;
    }
    if (!generic && bamd_launch_fast_a(a, pro, epi, grid, s)) return;
    if (pro == BAMD_PRO_NORM) launch_mv_epi<BAMD_PRO_NORM>(a, epi, grid, s);
    else                      launch_mv_epi<BAMD_PRO_PLAIN>(a, epi, grid, s);
}

void bamd_launch_step_begin(bamd_step_state * st, const int32_t * forced, int n_forced, int32_t * out_tokens, const void * embd,
                            int embd_type, int E, int V, float * x, int do_embed, hipStream_t s, const int32_t * slots, int32_t * cellpos,
                            const float * rope, float * rope_cur, int hd, const bamd_step_state * inbox) {
    BAMD_LAUNCH(step_begin_kernel, dim3(1), dim3(1024), 0, s, st, forced, n_forced, out_tokens, (const uint8_t *) embd, embd_type, E, V, x, do_embed, slots, cellpos,
                rope, rope_cur, hd, inbox);
}


