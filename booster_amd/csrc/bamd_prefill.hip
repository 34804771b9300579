// bamd_prefill.hip — batched prefill: per-token Q8_K quantisation into global blobs, the exact MFMA mat-mul kernels (Q4_K, Q6_K),
// the integer-dot batched mat-mul (any K-quant), batched embedding, silu*up.  Helpers: bamd_device.h.
#include "bamd_device.h"
#include "bamd_mfma_common.h"
#include <type_traits>

// ===========================================================================================================
// Batched prefill (T > 1 tokens per call; reference: llama_decode with a micro-batch, ggml_compute_forward_mul_mat with
// ne11 = T, ggml.c:12277-12492).  Per (row, token) the arithmetic is EXACTLY the single-token chain above — the reference
// quantises each activation row to Q8_K and runs the same vec_dot per (row, column).  Two implementations, bit-identical:
//   matmul_mfma_q4k_kernel / matmul_mfma_q6k_kernel (default): the integer sums of a super-block on the matrix cores, exactly — f16
//     A = scale x quant, f16 B = the int8 activations, one 32-deep MFMA per SIMD lane e of the reference — and the f32 chains on the VALU;
//   matmul_batch_kernel (BAMD_PREFILL_MFMA=0; the second implementation the first is tested against): block_terms / chain_step /
//     finish_row of the decode path, the weights of a record unpacked once for BAMD_TT tokens whose Q8_K activations sit in LDS.
// ===========================================================================================================

// one workgroup per token: RMSNorm (optional) + Q8_K of row t of x[T][K] -> blob[t]
// f16 copy of a token's Q8_K row for the MFMA path.  Per super-block BAMD_B16_REC = 608 B (560 used): 8 (e) x 4 (g) groups of 8 halves — group
// (e, g) = the int8 of sub-blocks 2g and 2g+1, chunk e, as exact f16: one 16-byte B operand of v_mfma_f32_16x16x32_f16 per lane —
// then, per pair l of sub-blocks, the four halves {S_h(2l), S_h(2l+1), S_l(2l), S_l(2l+1)} of the block sums split as S = 2 S_h + S_l
// (the B operand of the Q4_K min-term MFMA), then the four i16 pairs (S_2l, S_2l+1) (Q5_K), then the block scale d_y (f32, byte 560); after the nb super-blocks, yd[nb] f32 again.
template <bool NORM>
__global__ void __launch_bounds__(512) quantize_batch_kernel(const float * __restrict__ x, const float * __restrict__ nw, float eps, int K,
                                                             uint8_t * __restrict__ blob, uint8_t * __restrict__ blob16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nb = K >> 8, t = blockIdx.x;
    uint32_t * q8 = (uint32_t *) smem; int * S = (int *) (q8 + nb * 64); float * yd = (float *) (S + nb * 8);
    double * red = (double *) (smem + BAMD_ACT_RED_OFF(nb));
    const float * xt = x + (size_t) t * K;
    ActPro<NORM> ap; ap.issue(xt, nw, K, wave_id()); ap.finish(xt, nw, eps, K, q8, S, yd, red);
    const size_t bb = BAMD_BLOB_BYTES(nb);
    if (blob) {
        const uint4 * src = (const uint4 *) smem; uint4 * dst = (uint4 *) (blob + (size_t) t * bb);
        for (int i = threadIdx.x; i < (int) (bb / 16); i += blockDim.x) dst[i] = src[i];
    }
    if (blob16) {
        uint8_t * o = blob16 + (size_t) t * BAMD_BLOB16_BYTES(nb);
        for (int i = threadIdx.x; i < nb * 64; i += blockDim.x) {          // q8[ci*64 + e*8 + c] = sub-block c, chunk e, 4 int8
            const int ci = i >> 6, e = (i >> 3) & 7, c = i & 7;
            const uint32_t w = q8[i];
            const unsigned short h0 = f2h((float) (int8_t) (w)), h1 = f2h((float) (int8_t) (w >> 8)), h2 = f2h((float) (int8_t) (w >> 16)), h3 = f2h((float) (int8_t) (w >> 24));
            uint2 v; v.x = (uint32_t) h0 | ((uint32_t) h1 << 16); v.y = (uint32_t) h2 | ((uint32_t) h3 << 16);
            *(uint2 *) (o + (size_t) ci * BAMD_B16_REC + (size_t) (e * 4 + (c >> 1)) * 16 + (c & 1) * 8) = v;
        }
        for (int i = threadIdx.x; i < nb * 4; i += blockDim.x) {
            const int ci = i >> 2, l = i & 3;
            const int sa = S[ci * 8 + 2 * l], sb = S[ci * 8 + 2 * l + 1];          // block sums of sub-blocks 2l, 2l+1: |S| <= 32 * 127
            // operands of the min-term MFMA (Q4_K): S = 2 S_h + S_l with S_h = S >> 1 (|.| <= 2048) and S_l = S & 1 both exact in f16
            uint2 mf;
            mf.x = (uint32_t) f2h((float) (sa >> 1)) | ((uint32_t) f2h((float) (sb >> 1)) << 16);
            mf.y = (uint32_t) f2h((float) (sa & 1)) | ((uint32_t) f2h((float) (sb & 1)) << 16);
            *(uint2 *) (o + (size_t) ci * BAMD_B16_REC + 512 + l * 8) = mf;
            // (S_2l, S_2l+1) as i16 pairs (Q5_K: v_dot2_i32_i16)
            *(uint32_t *) (o + (size_t) ci * BAMD_B16_REC + 544 + l * 4) = ((uint32_t) sa & 0xffffu) | ((uint32_t) sb << 16);
        }
        float * oyd = (float *) (o + (size_t) nb * BAMD_B16_REC);
        for (int i = threadIdx.x; i < nb; i += blockDim.x) { oyd[i] = yd[i]; *(float *) (o + (size_t) i * BAMD_B16_REC + 560) = yd[i]; }   // d_y also inside the record (bamd_prefill2.hip: no separate copy)
    }
}


template <int TYPE, typename REC, int D, int EPI>
__device__ __forceinline__ void batch_segment(const uint8_t * __restrict__ wA, const uint8_t * __restrict__ wB, int nb, int first, int count, int stride,
                                              float * __restrict__ out, const float * __restrict__ res, int ldo, int t0, int nt,
                                              const unsigned char * acts, size_t bb, int nvalid) {
    constexpr int RECB = TYPE == BAMD_Q4_K ? BAMD_RECB_Q4K : TYPE == BAMD_Q5_K ? BAMD_RECB_Q5K : 1680;     // bamd_record_bytes
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
    constexpr int NPARTS = PAIR ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const bamd_rsrc rsA = weight_rsrc(wA), rsB = PAIR ? weight_rsrc(wB) : rsA;
    const int rgb = nb * RECB, rg_step = stride * rgb;
    const int chunks = nb / D;
    REC ring[D];
#pragma unroll
    for (int s = 0; s < D; ++s) load_rec(ring[s], rsA, first * rgb + s * RECB, lane);
    for (int r = 0; r < count; ++r) {
        const int rg = first + r * stride;
        const int row = rg * 8 + (lane >> 3);
        const int rowoff = rg * rgb;
        float gate_val[BAMD_TT];
#pragma unroll
        for (int part = 0; part < NPARTS; ++part) {
            const bool last = !(PAIR && part == 0) && r + 1 >= count;
            const bool after_b = (PAIR && part == 0) || (last && part == 1);
            const int after_off = (PAIR && part == 0) ? rowoff : (last ? rowoff + (nb - 1) * RECB : rowoff + rg_step);
            RowAcc A[BAMD_TT];
#pragma unroll
            for (int u = 0; u < BAMD_TT; ++u) { A[u].acc = 0.f; A[u].accm = 0.f; }
            for (int c = 0; c < chunks; ++c) {
                const bool inrow = c + 1 < chunks;
                const bamd_rsrc nrs = (inrow ? part == 1 : after_b) ? rsB : rsA;
                const int nxt = inrow ? rowoff + (c + 1) * (D * RECB) : after_off;
                const int step = (inrow || !last) ? RECB : 0;
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    pin_rec(ring[s]);
#pragma unroll
                    for (int u = 0; u < BAMD_TT; ++u) {          // tokens beyond nt read stale-but-valid LDS and are never stored
                        const unsigned char * au = acts + (size_t) u * bb;
                        const uint32_t * q8 = (const uint32_t *) au; const int * S = (const int *) (q8 + nb * 64); const float * yd = (const float *) (S + nb * 8);
                        const Terms T = block_terms(ring[s], c * D + s, lane, q8, S, yd);
                        chain_step<TYPE>(A[u], T.d, T.fs, T.dmin, T.pm);
                    }
                    load_rec(ring[s], nrs, nxt + s * step, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int u = 0; u < BAMD_TT; ++u) {
                const float val = finish_row<TYPE>(A[u]);
                if (PAIR && part == 0) { gate_val[u] = val; continue; }
                if ((lane & 7) == 0 && row < nvalid && u < nt) {
                    const size_t o = (size_t) (t0 + u) * ldo + row;
                    float y = val;
                    if (PAIR) y = v_silu(gate_val[u]) * val;
                    if (EPI == BAMD_EPI_ADD) y = val + res[o];
                    out[o] = y;
                }
            }
        }
    }
}

// ---- Q4_K x Q8_K on the matrix cores, exact ---------------------------------------------------------------------------
// The reference's per-lane integer sums  isum_e = sum_j sc_j * sum_u w[j,e,u] * x[j,e,u]  (e = SIMD lane, j = 32-element sub-block,
// u = 0..3) are 32-term dot products per (row, token, super-block, e).  With A = sc_j * w (<= 63 * 15 = 945: exact in f16), B = x
// (int8: exact in f16) and f32 accumulation of integers < 2^24, ONE v_mfma_f32_16x16x32_f16 per e yields the sixteen-by-sixteen
// (row, token) tile of isum_e exactly; the f32 chains acc_e = fma(d_x * d_y, isum_e, acc_e), the min terms and the final hsum tree
// then run on the VALU in the reference's order (ggml-quants.c:6937-6978) — bit-identical to the integer-dot kernels above.
// MFMA lane l = (m = l & 15, g = l >> 4): A row m, B token m, k-slots (g, i) = (sub-block 2g + (i >> 2), u = i & 3);
// C/D rows 4g + i, token m (cdna_hip_programming.md, fragment layout).  One wave = 16 rows x 16 tokens over the whole K.
// four i16-pair dot products -> float in one asm block: VOP3P v_dot2_i32_i16 d, a, b, 0 (the builtin becomes v_dot2c + a zero-init move per
// product), the conversions four instructions behind their dots (a DOT result needs 3 wait states before a VALU read; inline asm is
// not hazard-checked)
__device__ __forceinline__ void dot2x4_f32(const uint32_t (&m)[4], const uint32_t (&sv)[4], float (&p)[4]) {
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %4, %8, %12, 0\n\t" "v_dot2_i32_i16 %5, %9, %13, 0\n\t" "v_dot2_i32_i16 %6, %10, %14, 0\n\t" "v_dot2_i32_i16 %7, %11, %15, 0\n\t"
        "v_cvt_f32_i32 %0, %4\n\t" "v_cvt_f32_i32 %1, %5\n\t" "v_cvt_f32_i32 %2, %6\n\t" "v_cvt_f32_i32 %3, %7"
        : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(sv[0]), "v"(sv[1]), "v"(sv[2]), "v"(sv[3]));
}
struct bamd_mma_args {
    const uint8_t * w; float * out; const float * res;      // Q4_K wave-stream; out / res [T][ldo]
    const uint8_t * blob16; int K, T, nrows, nrows_pad, ldo;
};
__device__ __forceinline__ void unpack_k4_(uint32_t u0, uint32_t u1, uint32_t u2, uint32_t & sc03, uint32_t & sc47, uint32_t & mn03, uint32_t & mn47) {
    sc03 = u0 & 0x3f3f3f3fu; mn03 = u1 & 0x3f3f3f3fu;                                    // ggml-quants.c:6928-6933
    sc47 = (u2 & 0x0f0f0f0fu) | (((u0 >> 6) & 0x03030303u) << 4);
    mn47 = ((u2 >> 4) & 0x0f0f0f0fu) | (((u1 >> 6) & 0x03030303u) << 4);
}
// Workgroup = 8 waves = 8 consecutive row tiles (128 rows) x one tile of 32 tokens (each wave: 16 rows x 2 x 16 tokens, so every A
// fragment is built once for two MFMAs).  Per super-block:
//   - the 32 tokens' B records (528 B each) are staged in LDS by the whole workgroup, double-buffered (one barrier per super-block);
//   - each wave loads its two weight records in the wave-stream layout (two coalesced 16-byte-per-lane loads, prefetched one
//     super-block ahead), transposes them into the MFMA A layout through a private, padded LDS tile, and lanes 0..15 unpack the
//     16 row headers ONCE (d, dmin, scales, mins as i16 pairs) into LDS for the other lanes;
//   - 8 (e) x 2 (token tiles) MFMAs; chains, min terms (v_dot2_i32_i16) and the final trees on the VALU.
#ifndef BAMD_MMA_NT                                  /* -DBAMD_MMA_NT=1: one 16-token tile per wave at <= 128 VGPRs, two workgroups per CU — the occupancy experiment of DESIGN 7b */
#define BAMD_MMA_NT 2
#endif
#if BAMD_MMA_NT == 1
#define BAMD_MMA_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define BAMD_MMA_OCC
#endif
#ifndef BAMD_MMA_NTLOADS
#define BAMD_MMA_NTLOADS 0                                                       /* weight loads of the MFMA kernels: default cache policy — the 16 token tiles of a row block re-read
                                                                                    the same records through L2 (nt: 246 vs 253 TFLOP/s at 512 tokens) */
#endif
template <typename T> __device__ __forceinline__ T ldw(const uint8_t * rec, uint32_t off) {
#if BAMD_MMA_NTLOADS
    return ldnt<T>(rec, off);
#else
    return *(const T *) (rec + off);
#endif
}
#define BAMD_MMA_TOK (16 * BAMD_MMA_NT)
#define BAMD_MMA_STAGE (BAMD_MMA_TOK * BAMD_B16_REC + BAMD_MMA_TOK * 4)          /* B records + yd */
#define BAMD_MMA_NSTAGE 4                                                        /* stage buffers of the Q4_K / Q5_K kernel (one barrier per two super-blocks) */
#define BAMD_MMA_WAVE_LDS (2 * 288 * 4 + 16 * 32 + 2 * 72 * 4)                   /* transposed A tile + row headers + Q5_K high-bit tile */
// Q5 = true: Q5_K records (1408 B: + one dword of high bits per lane).  The fifth bit joins the nibble before the f16 build
// (values <= 31, scale x value <= 1953: exact); (1024 + n) * s would overflow f16 at s = 63, so the bias is subtracted first (exact)
// and the product takes one more packed instruction; the min terms follow ggml_vec_dot_q5_K_q8_K: ONE float per row,
// summs = summs + dmin * (float) sum_j m_j S_j (multiply, then add: ggml-quants.c:7515-7518), added after the hsum tree.
template <int EPI, bool Q5>
__global__ void __launch_bounds__(512) BAMD_MMA_OCC matmul_mfma_q4k_kernel(bamd_mma_args a) {
    constexpr uint32_t RECB = Q5 ? BAMD_RECB_Q5K : BAMD_RECB_Q4K, HDRO = Q5 ? 1280u : 1024u;      // bamd_record_bytes; header {d|dmin, sc[0..3], sc[4..7], mn[0..3]} + mn[4..7] at HDRO + 128
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef short s2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), m = lane & 15, g = lane >> 4;
    const int nb = a.K >> 8;
    const int rt = blockIdx.y * 8 + wave;                    // row tile: rows rt*16 .. rt*16+15 = record groups 2rt, 2rt+1
    const bool live = rt * 16 < a.nrows_pad;                 // dead waves still take part in the staging and the barriers
    const int t0 = blockIdx.x * BAMD_MMA_TOK;
    const size_t b16 = BAMD_BLOB16_BYTES(nb);
    unsigned char * stage = smem;                                            // [2][BAMD_MMA_STAGE]
    uint32_t * wl = (uint32_t *) (smem + BAMD_MMA_NSTAGE * BAMD_MMA_STAGE + wave * BAMD_MMA_WAVE_LDS);   // this wave's A tile [2][288] dwords
    uint32_t * hl = wl + 2 * 288;                                            // this wave's row headers [16][8] dwords
    uint32_t * qht = hl + 16 * 8;                                            // Q5_K: high-bit dwords [2][8 rows][9] (row stride padded)
    // staging plan: BAMD_B16_Q uint4 per token record, BAMD_MMA_TOK tokens; tokens past T repeat the last one (never stored)
    // Per-thread source offsets, fixed over the K loop.  The records go global -> LDS directly (global_load_lds_dwordx4: each wave's 64
    // lanes fill 1 KiB of consecutive LDS, the source address is per lane), so the stage costs no registers and no ds_write pass; the
    // copies of super-block ci+1 are issued at the top of iteration ci into the buffer every wave left at the previous barrier, and
    // the barrier at the end of the iteration (vmcnt(0) inside) publishes them.  (An indexed register array as the staging buffer is
    // placed in scratch memory by the compiler, with a full s_waitcnt after every load: +20 % kernel time.)
    const bool third = tid + 1024 < BAMD_MMA_TOK * BAMD_B16_Q, second = BAMD_MMA_TOK * BAMD_B16_Q >= 1024 || tid + 512 < BAMD_MMA_TOK * BAMD_B16_Q;
    auto stage_src = [&](int idx) {                           // 32-bit byte offsets into the blob (T <= 512 tokens: a few MB)
        const int tok = idx / BAMD_B16_Q, q = idx - tok * BAMD_B16_Q;
        const int tg = t0 + tok < a.T ? t0 + tok : a.T - 1;
        return (uint32_t) ((size_t) tg * b16 + (size_t) q * 16);
    };
    const uint32_t ssrc0 = stage_src(tid), ssrc1 = second ? stage_src(tid + 512) : stage_src(tid), ssrc2 = third ? stage_src(tid + 1024) : ssrc1;
    uint32_t ysrc;
    { const int tk = tid & (BAMD_MMA_TOK - 1); const int tg = t0 + tk < a.T ? t0 + tk : a.T - 1; ysrc = (uint32_t) ((size_t) tg * b16 + (size_t) nb * BAMD_B16_REC); }
#define BAMD_STAGE_ISSUE(ci_, buf_) do { const uint8_t * sb_ = a.blob16 + (size_t) (ci_) * BAMD_B16_REC; \
        unsigned char * st_ = stage + (size_t) (buf_) * BAMD_MMA_STAGE + (size_t) (wave * 64) * 16; \
        lds_dma16(sb_ + ssrc0, st_); if (second) lds_dma16(sb_ + ssrc1, st_ + 512 * 16); \
        if (third) lds_dma16(sb_ + ssrc2, st_ + 1024 * 16); \
        if (tid < BAMD_MMA_TOK)                       /* the 32 block scales d_y: 4 bytes per lane, lanes 0..31 of wave 0 */ \
            lds_dma4(a.blob16 + ysrc + (size_t) (ci_) * 4, stage + (size_t) (buf_) * BAMD_MMA_STAGE + BAMD_MMA_TOK * BAMD_B16_REC); } while (0)
    const int rtc = live ? rt : 0;
    // record groups of rows 0-7 / 8-15 of the tile; a last tile with only 8 (padded) rows reads its first group twice (rows 8-15 are never stored)
    const uint8_t * rec0 = a.w + (size_t) (rtc * 2) * nb * RECB, * rec1 = (rtc * 2 + 1) * 8 < a.nrows_pad ? rec0 + (size_t) nb * RECB : rec0;
    const uint8_t * hdrm = (m < 8 ? rec0 : rec1) + HDRO + (m & 7) * 16;                                    // header of row m (lanes g == 0)
    bamd_f4 acc[BAMD_MMA_NT][8], accm[BAMD_MMA_NT][4];
#pragma unroll
    for (int n = 0; n < BAMD_MMA_NT; ++n) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[n][e] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int l = 0; l < 4; ++l) accm[n][l] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
    }
    // prologue: stage super-blocks 0 and 1, prefetch the weights of super-block 0.  FOUR stage buffers, one workgroup barrier per
    // TWO super-blocks: at the top of an even iteration ci the records of ci + 2 and ci + 3 are requested into the two buffers every
    // wave left at the last barrier; the barrier at the end of ci + 1 publishes them.  (The wave-private tiles wl / hl need no
    // barrier: one wave's LDS operations execute in order.)
    BAMD_STAGE_ISSUE(0, 0);
    BAMD_STAGE_ISSUE(nb > 1 ? 1 : 0, 1);
    const uint8_t * hdrm2 = (m < 8 ? rec0 : rec1) + HDRO + 128 + (m & 7) * 4;                              // its mins 4..7
    uint4 wa = ldw<uint4>(rec0, (uint32_t) lane * 16u), wb = ldw<uint4>(rec1, (uint32_t) lane * 16u), hd = *(const uint4 *) hdrm;
    uint32_t hd2 = BAMD_XSCALES ? *(const uint32_t *) hdrm2 : 0u;
    uint32_t qha = 0u, qhb = 0u;
    if (Q5) { qha = ldw<uint32_t>(rec0, 1024u + (uint32_t) lane * 4u); qhb = ldw<uint32_t>(rec1, 1024u + (uint32_t) lane * 4u); }
    lds_dma_wait();
    __syncthreads();
    auto step = [&](const int ci, auto even_tag) {
        constexpr bool EVEN = decltype(even_tag)::value;      // compile-time: the even half issues the next two stages, the odd half ends with the barrier
        const unsigned char * st = stage + (size_t) (ci & 3) * BAMD_MMA_STAGE;
        const bool more = ci + 1 < nb;
        // ---- weights of this super-block: transpose into the A layout, unpack the row headers once ----
        // Q4_K min terms on the matrix core: pm_l = m_2l S_2l + m_2l+1 S_2l+1 with S = 2 S_h + S_l is the 4-term dot product
        // {2 m_2l, 2 m_2l+1, m_2l, m_2l+1} . {S_h(2l), S_h(2l+1), S_l(2l), S_l(2l+1)}: every factor and every partial sum an exact f16 /
        // f32 integer.  A operand of v_mfma_f32_16x16x16f16: lanes g == 0 (k = 0..3) carry row m's four halves, the other k-groups zeros
        // — and lane (m, 0) is the lane that unpacks row m's header anyway.  (Was 16 v_dot2_i32_i16 + 16 conversions per token tile.)
        uint2 amin[4] = { { 0u, 0u }, { 0u, 0u }, { 0u, 0u }, { 0u, 0u } };
        {
            const int r = lane >> 3, e = lane & 7;           // wave-stream lane' = (row r of its record group, chunk e)
            *(uint4 *) (wl + 0 * 288 + r * 36 + e * 4) = wa;
            *(uint4 *) (wl + 1 * 288 + r * 36 + e * 4) = wb;
            if (Q5) { qht[0 * 72 + r * 9 + e] = qha; qht[1 * 72 + r * 9 + e] = qhb; }
            if (g == 0) {                                    // lanes 0..15: row m
#if BAMD_XSCALES
                const uint32_t sc03 = hd.y, sc47 = hd.z, mn03 = hd.w, mn47 = hd2;     // unpacked by the load-time repack
#else
                uint32_t sc03, sc47, mn03, mn47; unpack_k4_(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
#endif
                uint4 h0, h1;
                h0.x = hd.x; h0.y = sc03; h0.z = sc47; h0.w = 0u;
                h1.x = __builtin_amdgcn_perm(0u, mn03, 0x0c010c00u); h1.y = __builtin_amdgcn_perm(0u, mn03, 0x0c030c02u);
                h1.z = __builtin_amdgcn_perm(0u, mn47, 0x0c010c00u); h1.w = __builtin_amdgcn_perm(0u, mn47, 0x0c030c02u);
                *(uint4 *) (hl + m * 8) = h0; *(uint4 *) (hl + m * 8 + 4) = h1;
                if (!Q5) {
                    const h2_t k1024 = { (_Float16) 1024.f, (_Float16) 1024.f };
                    const uint32_t sel[2] = { 0x04010400u, 0x04030402u };
#pragma unroll
                    for (int l = 0; l < 4; ++l) {           // (1024 + m_a, 1024 + m_b) by byte permute, minus 1024: the pair as exact f16
                        union { uint32_t u; h2_t h; } c, one, two;
                        c.u = __builtin_amdgcn_perm(0x64646464u, l < 2 ? mn03 : mn47, sel[l & 1]);
                        one.h = c.h - k1024; two.h = one.h + one.h;
                        amin[l].x = two.u; amin[l].y = one.u;
                    }
                }
            }
        }
        // the next stage and the next weights in flight during the math (at the end: this super-block again, the stage into the idle
        // buffer).  Issued AFTER the registers of the previous prefetch were consumed: with a global_load_lds in flight the compiler
        // waits vmcnt(0) at the next use of an ordinary load result, which would otherwise sit right behind the issue.
        if (EVEN) {
            BAMD_STAGE_ISSUE(ci + 2 < nb ? ci + 2 : nb - 1, (ci + 2) & 3);
            BAMD_STAGE_ISSUE(ci + 3 < nb ? ci + 3 : nb - 1, (ci + 3) & 3);
        }
        {
            const uint32_t ro = (uint32_t) (more ? ci + 1 : ci) * RECB;
            wa = ldw<uint4>(rec0, ro + (uint32_t) lane * 16u); wb = ldw<uint4>(rec1, ro + (uint32_t) lane * 16u); hd = *(const uint4 *) (hdrm + ro); if (BAMD_XSCALES) hd2 = *(const uint32_t *) (hdrm2 + ro);
            if (Q5) { qha = ldw<uint32_t>(rec0, ro + 1024u + (uint32_t) lane * 4u); qhb = ldw<uint32_t>(rec1, ro + 1024u + (uint32_t) lane * 4u); }
        }
        __builtin_amdgcn_sched_barrier(0);                   // keep the prefetch ahead of the math (the scheduler would sink it to the loop end)
        // headers of the four C rows 4g + i; d products per token tile
        float D[BAMD_MMA_NT][4], Dm[BAMD_MMA_NT][4]; uint4 mp[4];
        {
            float dw[4], dmw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t dd = hl[(4 * g + i) * 8];
                dw[i] = h2f(dd & 0xffffu); dmw[i] = h2f(dd >> 16);
                mp[i] = *(const uint4 *) (hl + (4 * g + i) * 8 + 4);
            }
#pragma unroll
            for (int n = 0; n < BAMD_MMA_NT; ++n) {
                const float ydv = *(const float *) (st + BAMD_MMA_TOK * BAMD_B16_REC + (n * 16 + m) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { D[n][i] = ydv * dw[i]; Dm[n][i] = (-ydv) * dmw[i]; }
            }
        }
        // A fragment of row m per e (scales of sub-blocks 2g, 2g+1 as f16; A = (1024 + n) * s - 1024 * s = n * s exactly), used for
        // both token tiles, then the chains
        {
            const uint32_t scw = hl[m * 8 + 1 + (g >> 1)] >> (16 * (g & 1));
            const _Float16 s_lo = (_Float16) (float) (scw & 0xffu), s_hi = (_Float16) (float) ((scw >> 8) & 0xffu);
            const h2_t slo2 = { s_lo, s_lo }, shi2 = { s_hi, s_hi };
            const h2_t nlo2 = { (_Float16) -1024.f * s_lo, (_Float16) -1024.f * s_lo }, nhi2 = { (_Float16) -1024.f * s_hi, (_Float16) -1024.f * s_hi };
            const h2_t shi16 = { (_Float16) 0.0625f * s_hi, (_Float16) 0.0625f * s_hi }, nhi16 = { (_Float16) -64.f * s_hi, (_Float16) -64.f * s_hi };
            const uint32_t * wrow = wl + (m >> 3) * 288 + (m & 7) * 36 + g;
            // software pipeline over e, written out: the LDS operands of e + 2 are requested at the top of iteration e and the MFMA results
            // of e - 1 are folded into the chains in iteration e; a scheduling barrier per iteration keeps that order (left alone, the
            // scheduler requests an operand one iteration ahead — ~70 cycles for an LDS latency of 130+ under eight waves — and every
            // iteration of every wave waits).
            uint32_t Wq[3]; bamd_h8 Bq[3][BAMD_MMA_NT]; bamd_f4 sprev[BAMD_MMA_NT];
#define BAMD_LDB(e_, n_) (*(const bamd_h8 *) (st + (size_t) ((n_) * 16 + m) * BAMD_B16_REC + ((e_) * 4 + g) * 16))
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                Wq[e] = wrow[e * 4];
#pragma unroll
                for (int n = 0; n < BAMD_MMA_NT; ++n) Bq[e][n] = BAMD_LDB(e, n);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e + 2 < 8) {
                    Wq[(e + 2) % 3] = wrow[(e + 2) * 4];
#pragma unroll
                    for (int n = 0; n < BAMD_MMA_NT; ++n) Bq[(e + 2) % 3][n] = BAMD_LDB(e + 2, n);
                }
                const uint32_t wq = Wq[e % 3];
                // Q4_K: the high nibbles stay where they are (16 n in the f16 image: the scale operand below is s / 16, exact) — one shift less per e
                uint32_t lo = wq & 0x0f0f0f0fu, hi = Q5 ? (wq >> 4) & 0x0f0f0f0fu : wq & 0xf0f0f0f0u;
                if (Q5) {                                    // bit c of byte u of the row's high-bit dword e: element 4e+u of sub-block c
                    const uint32_t qh = qht[(m >> 3) * 72 + (m & 7) * 9 + e];
                    lo |= ((qh >> (2 * g)) & 0x01010101u) << 4; hi |= ((qh >> (2 * g + 1)) & 0x01010101u) << 4;
                }
                union { uint32_t u; h2_t h; } c0, c1, c2, c3;
                c0.u = __builtin_amdgcn_perm(0x64646464u, lo, 0x04010400u); c1.u = __builtin_amdgcn_perm(0x64646464u, lo, 0x04030402u);
                c2.u = __builtin_amdgcn_perm(0x64646464u, hi, 0x04010400u); c3.u = __builtin_amdgcn_perm(0x64646464u, hi, 0x04030402u);
                h2_t a0, a1, a2, a3;
                if (Q5) {
                    const h2_t k1024 = { (_Float16) 1024.f, (_Float16) 1024.f };
                    a0 = (c0.h - k1024) * slo2; a1 = (c1.h - k1024) * slo2; a2 = (c2.h - k1024) * shi2; a3 = (c3.h - k1024) * shi2;
                } else {
                    a0 = __builtin_elementwise_fma(c0.h, slo2, nlo2); a1 = __builtin_elementwise_fma(c1.h, slo2, nlo2);
                    a2 = __builtin_elementwise_fma(c2.h, shi16, nhi16); a3 = __builtin_elementwise_fma(c3.h, shi16, nhi16);   // (1024 + 16 n) s/16 - 64 s = n s
                }
                const bamd_h8 av = { a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y };
                bamd_f4 si[BAMD_MMA_NT];
#pragma unroll
                for (int n = 0; n < BAMD_MMA_NT; ++n) {
                    const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                    si[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, Bq[e % 3][n], z, 0, 0, 0);
                    if (e > 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[n][e - 1][i] = fmaf(D[n][i], sprev[n][i], acc[n][e - 1][i]);
                    }
                }
#pragma unroll
                for (int n = 0; n < BAMD_MMA_NT; ++n) sprev[n] = si[n];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int n = 0; n < BAMD_MMA_NT; ++n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[n][7][i] = fmaf(D[n][i], sprev[n][i], acc[n][7][i]);
            }
#undef BAMD_LDB
        }
        // min terms: pm_l = m_2l S_2l + m_2l+1 S_2l+1; accm_l = fma(dmin, pm_l, accm_l)   (:6937-6941)
#pragma unroll
        for (int n = 0; n < BAMD_MMA_NT; ++n) {
            if (Q5) {
                const uint4 sp = *(const uint4 *) (st + (size_t) (n * 16 + m) * BAMD_B16_REC + 544);
                const uint32_t spl[4] = { sp.x, sp.y, sp.z, sp.w };
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t mpl[4] = { mp[i].x, mp[i].y, mp[i].z, mp[i].w };
                    int hs = 0;
#pragma unroll
                    for (int l = 0; l < 4; ++l) { union { uint32_t u; s2_t v; } ma, sb; ma.u = mpl[l]; sb.u = spl[l]; hs = __builtin_amdgcn_sdot2(ma.v, sb.v, hs, false); }
                    const float t = Dm[n][i] * (float) hs;
                    accm[n][0][i] = accm[n][0][i] + t;
                }
            } else {
                const uint4 sfa = *(const uint4 *) (st + (size_t) (n * 16 + m) * BAMD_B16_REC + 512), sfb = *(const uint4 *) (st + (size_t) (n * 16 + m) * BAMD_B16_REC + 528);
                const uint2 bl[4] = { { sfa.x, sfa.y }, { sfa.z, sfa.w }, { sfb.x, sfb.y }, { sfb.z, sfb.w } };
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    union { uint2 u; bamd_h4 h; } av4, bv4; av4.u = amin[l]; bv4.u = bl[l];
                    const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                    const bamd_f4 pm = __builtin_amdgcn_mfma_f32_16x16x16f16(av4.h, bv4.h, z, 0, 0, 0);     // rows 4g + i, token m: exact integers
#pragma unroll
                    for (int i = 0; i < 4; ++i) accm[n][l][i] = fmaf(Dm[n][i], pm[i], accm[n][l][i]);
                }
            }
        }
        if (!EVEN) {
            lds_dma_wait();
            __syncthreads();                                 // the next two stages visible; the two just read free again
        }
    };
    for (int ci = 0; ci < nb; ci += 2) {
        step(ci, std::true_type());
        if (ci + 1 < nb) step(ci + 1, std::false_type());    // (an odd K / 256 ends on an even step: nothing reads the stages after it)
    }
    if (!live) return;
    // hsum_float_8 over e and the acc_m folds, in the reference's order (finish_row), then the epilogue
#pragma unroll
    for (int n = 0; n < BAMD_MMA_NT; ++n) {
        const int t = t0 + n * 16 + m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = ((acc[n][0][i] + acc[n][4][i]) + (acc[n][2][i] + acc[n][6][i])) + ((acc[n][1][i] + acc[n][5][i]) + (acc[n][3][i] + acc[n][7][i]));
            const float mm = Q5 ? accm[n][0][i] : (accm[n][0][i] + accm[n][2][i]) + (accm[n][1][i] + accm[n][3][i]);
            const float val = v + mm;
            const int row = rt * 16 + 4 * g + i;
            if (t < a.T && row < a.nrows) {
                const size_t o = (size_t) t * a.ldo + row;
                a.out[o] = EPI == BAMD_EPI_ADD ? val + a.res[o] : EPI == BAMD_EPI_SILU_MUL ? v_silu(a.res[o]) * val : val;   // SILU_MUL: res = the gate projection
            }
        }
    }
}
// ---- Q6_K x Q8_K on the matrix cores, exact: same skeleton as matmul_mfma_q4k_kernel -----------------------------------------
// scale (int8) x (q6 - 32) reaches 4096 in magnitude: not every such integer is an f16.  The SCALE is split, sc = 16 s_h + s_l with
// s_l = sc & 15 in [0, 15] and s_h = sc >> 4 in [-8, 7]: A_1 = s_h * v (|.| <= 256) and A_2 = s_l * v (<= 480), v = q6 - 32, are exact f16,
// TWO MFMAs per e give S_1, S_2 (< 2^24), and isum = 16 S_1 + S_2 (|isum| <= 32 x 128 x 32 x 127 < 2^24) is exact as fmaf(16, S_1, S_2).
// Both fragments come from ONE f16 image of the quants, c = 1024 + q6 (byte permute), as fma(c, s, -1056 s) = (q6 - 32) s: the products and the
// constants 1056 s are exact, the fma rounds once and its result is representable.  (The first form split the VALUE, q6 - 32 = 2 vh + vl: a
// second f16 image for the low bit and 50 instead of 32 vector instructions per e.)  Scales are per 16 elements: for SIMD lane e the sub-block c
// uses scales[2c + (e >= 4)] (ggml-quants.c:8145-8216); no min terms.  Wave-stream Q6_K record: bamd_formats.h.
#define BAMD_MMA6_WAVE_LDS ((2 * 288 + 2 * 144 + 16 * 8) * 4)                    /* ql tile + qh tile + row headers */
template <int EPI>
__global__ void __launch_bounds__(512) BAMD_MMA_OCC matmul_mfma_q6k_kernel(bamd_mma_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), m = lane & 15, g = lane >> 4;
    const int nb = a.K >> 8;
    const int rt = blockIdx.y * 8 + wave;
    const bool live = rt * 16 < a.nrows_pad;
    const int t0 = blockIdx.x * BAMD_MMA_TOK;
    const size_t b16 = BAMD_BLOB16_BYTES(nb);
    unsigned char * stage = smem;
    uint32_t * wl = (uint32_t *) (smem + 2 * BAMD_MMA_STAGE + wave * BAMD_MMA6_WAVE_LDS);    // ql tile [2][288]
    uint32_t * ql2 = wl + 2 * 288;                                                           // qh tile [2][144]
    uint32_t * hl = ql2 + 2 * 144;                                                           // row headers [16][8]
    // Per-thread source offsets, fixed over the K loop.  The records go global -> LDS directly (global_load_lds_dwordx4: each wave's 64
    // lanes fill 1 KiB of consecutive LDS, the source address is per lane), so the stage costs no registers and no ds_write pass; the
    // copies of super-block ci+1 are issued at the top of iteration ci into the buffer every wave left at the previous barrier, and
    // the barrier at the end of the iteration (vmcnt(0) inside) publishes them.  (An indexed register array as the staging buffer is
    // placed in scratch memory by the compiler, with a full s_waitcnt after every load: +20 % kernel time.)
    const bool third = tid + 1024 < BAMD_MMA_TOK * BAMD_B16_Q, second = BAMD_MMA_TOK * BAMD_B16_Q >= 1024 || tid + 512 < BAMD_MMA_TOK * BAMD_B16_Q;
    auto stage_src = [&](int idx) {                           // 32-bit byte offsets into the blob (T <= 512 tokens: a few MB)
        const int tok = idx / BAMD_B16_Q, q = idx - tok * BAMD_B16_Q;
        const int tg = t0 + tok < a.T ? t0 + tok : a.T - 1;
        return (uint32_t) ((size_t) tg * b16 + (size_t) q * 16);
    };
    const uint32_t ssrc0 = stage_src(tid), ssrc1 = second ? stage_src(tid + 512) : stage_src(tid), ssrc2 = third ? stage_src(tid + 1024) : ssrc1;
    uint32_t ysrc;
    { const int tk = tid & (BAMD_MMA_TOK - 1); const int tg = t0 + tk < a.T ? t0 + tk : a.T - 1; ysrc = (uint32_t) ((size_t) tg * b16 + (size_t) nb * BAMD_B16_REC); }
#undef BAMD_STAGE_ISSUE
#define BAMD_STAGE_ISSUE(ci_, buf_) do { const uint8_t * sb_ = a.blob16 + (size_t) (ci_) * BAMD_B16_REC; \
        unsigned char * st_ = stage + (size_t) (buf_) * BAMD_MMA_STAGE + (size_t) (wave * 64) * 16; \
        lds_dma16(sb_ + ssrc0, st_); if (second) lds_dma16(sb_ + ssrc1, st_ + 512 * 16); \
        if (third) lds_dma16(sb_ + ssrc2, st_ + 1024 * 16); \
        if (tid < BAMD_MMA_TOK)                       /* the 32 block scales d_y: 4 bytes per lane, lanes 0..31 of wave 0 */ \
            lds_dma4(a.blob16 + ysrc + (size_t) (ci_) * 4, stage + (size_t) (buf_) * BAMD_MMA_STAGE + BAMD_MMA_TOK * BAMD_B16_REC); } while (0)
    const int rtc = live ? rt : 0;
    const uint8_t * rec0 = a.w + (size_t) (rtc * 2) * nb * 1680, * rec1 = (rtc * 2 + 1) * 8 < a.nrows_pad ? rec0 + (size_t) nb * 1680 : rec0;     // (a last tile of 8 rows: its first group twice)
    const uint8_t * recm = m < 8 ? rec0 : rec1;                                             // record group of row m (lanes g == 0)
    bamd_f4 acc[BAMD_MMA_NT][8];
#pragma unroll
    for (int n = 0; n < BAMD_MMA_NT; ++n) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[n][e] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
    }
    BAMD_STAGE_ISSUE(0, 0);
    uint4 wa = ldw<uint4>(rec0, (uint32_t) lane * 16u), wb = ldw<uint4>(rec1, (uint32_t) lane * 16u);
    uint2 qa = ldw<uint2>(rec0, 1024u + (uint32_t) lane * 8u), qb = ldw<uint2>(rec1, 1024u + (uint32_t) lane * 8u);
    uint4 hsc = *(const uint4 *) (recm + 1536 + (m & 7) * 16); uint32_t hd = *(const unsigned short *) (recm + 1664 + (m & 7) * 2);
    lds_dma_wait();
    __syncthreads();
    for (int ci = 0; ci < nb; ++ci) {
        const unsigned char * st = stage + (size_t) (ci & 1) * BAMD_MMA_STAGE;
        const bool more = ci + 1 < nb;
        {
            const int r = lane >> 3, e = lane & 7;
            *(uint4 *) (wl + 0 * 288 + r * 36 + e * 4) = wa;
            *(uint4 *) (wl + 1 * 288 + r * 36 + e * 4) = wb;
            *(uint2 *) (ql2 + 0 * 144 + r * 18 + e * 2) = qa;
            *(uint2 *) (ql2 + 1 * 144 + r * 18 + e * 2) = qb;
            if (g == 0) { *(uint4 *) (hl + m * 8) = hsc; hl[m * 8 + 4] = hd; }
        }
        BAMD_STAGE_ISSUE(more ? ci + 1 : ci, (ci + 1) & 1);  // after the previous prefetch was consumed (see the Q4_K kernel)
        {
            const uint32_t ro = (uint32_t) (more ? ci + 1 : ci) * 1680u;
            wa = ldw<uint4>(rec0, ro + (uint32_t) lane * 16u); wb = ldw<uint4>(rec1, ro + (uint32_t) lane * 16u);
            qa = ldw<uint2>(rec0, ro + 1024u + (uint32_t) lane * 8u); qb = ldw<uint2>(rec1, ro + 1024u + (uint32_t) lane * 8u);
            hsc = *(const uint4 *) (recm + ro + 1536 + (m & 7) * 16); hd = *(const unsigned short *) (recm + ro + 1664 + (m & 7) * 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        float D[BAMD_MMA_NT][4];
        {
            float dw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dw[i] = h2f(hl[(4 * g + i) * 8 + 4]);
#pragma unroll
            for (int n = 0; n < BAMD_MMA_NT; ++n) {
                const float ydv = *(const float *) (st + BAMD_MMA_TOK * BAMD_B16_REC + (n * 16 + m) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) D[n][i] = ydv * dw[i];
            }
        }
        {
            // int8 scales of sub-blocks c = 2g, 2g+1 for the two e-halves (header byte hi*8 + c), split 16 s_h + s_l; with each the constant -1056 s
            h2_t sAh[2], sAl[2], sBh[2], sBl[2], nAh[2], nAl[2], nBh[2], nBl[2];
            const h2_t k1056 = { (_Float16) -1056.f, (_Float16) -1056.f };
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const uint32_t w = hl[m * 8 + hi * 2 + (g >> 1)] >> (16 * (g & 1));
                const int s0 = (int) (int8_t) (w & 0xffu), s1 = (int) (int8_t) ((w >> 8) & 0xffu);
                const _Float16 s0h = (_Float16) (float) (s0 >> 4), s0l = (_Float16) (float) (s0 & 15), s1h = (_Float16) (float) (s1 >> 4), s1l = (_Float16) (float) (s1 & 15);
                sAh[hi] = (h2_t) { s0h, s0h }; sAl[hi] = (h2_t) { s0l, s0l }; sBh[hi] = (h2_t) { s1h, s1h }; sBl[hi] = (h2_t) { s1l, s1l };
                nAh[hi] = k1056 * sAh[hi]; nAl[hi] = k1056 * sAl[hi]; nBh[hi] = k1056 * sBh[hi]; nBl[hi] = k1056 * sBl[hi];
            }
            const int sh = 4 * (g & 1);
            const uint32_t * wq = wl + (m >> 3) * 288 + (m & 7) * 36 + 2 * (g >> 1);
            const uint32_t * hq = ql2 + (m >> 3) * 144 + (m & 7) * 18 + (g >> 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint2 ab = *(const uint2 *) (wq + e * 4);
                const uint32_t h = hq[e * 2];
                // the two high bits of a quant sit at bits sh, sh + 1 (sub-block 2g) / sh + 2, sh + 3 (2g + 1) of their byte of h and belong at bits 4, 5:
                // a ROTATION of the dword by 4 - sh / 2 - sh (what wraps around lands outside the mask 0x30 of every byte), then one and-or
                const uint32_t uA = (__builtin_amdgcn_alignbit(h, h, (uint32_t) (28 + sh) & 31u) & 0x30303030u) | ((ab.x >> sh) & 0x0f0f0f0fu);      // q6 of sub-block 2g, chunk e
                const uint32_t uB = (__builtin_amdgcn_alignbit(h, h, (uint32_t) (30 + sh) & 31u) & 0x30303030u) | ((ab.y >> sh) & 0x0f0f0f0fu);      // sub-block 2g + 1
                union { uint32_t u; h2_t h; } c0, c1, c2, c3;
                c0.u = __builtin_amdgcn_perm(0x64646464u, uA, 0x04010400u); c1.u = __builtin_amdgcn_perm(0x64646464u, uA, 0x04030402u);   // 1024 + q6
                c2.u = __builtin_amdgcn_perm(0x64646464u, uB, 0x04010400u); c3.u = __builtin_amdgcn_perm(0x64646464u, uB, 0x04030402u);
                const int eh = e >> 2;
                const h2_t ah0 = __builtin_elementwise_fma(c0.h, sAh[eh], nAh[eh]), ah1 = __builtin_elementwise_fma(c1.h, sAh[eh], nAh[eh]);
                const h2_t ah2 = __builtin_elementwise_fma(c2.h, sBh[eh], nBh[eh]), ah3 = __builtin_elementwise_fma(c3.h, sBh[eh], nBh[eh]);
                const h2_t al0 = __builtin_elementwise_fma(c0.h, sAl[eh], nAl[eh]), al1 = __builtin_elementwise_fma(c1.h, sAl[eh], nAl[eh]);
                const h2_t al2 = __builtin_elementwise_fma(c2.h, sBl[eh], nBl[eh]), al3 = __builtin_elementwise_fma(c3.h, sBl[eh], nBl[eh]);
                const bamd_h8 avh = { ah0.x, ah0.y, ah1.x, ah1.y, ah2.x, ah2.y, ah3.x, ah3.y };
                const bamd_h8 avl = { al0.x, al0.y, al1.x, al1.y, al2.x, al2.y, al3.x, al3.y };
#pragma unroll
                for (int n = 0; n < BAMD_MMA_NT; ++n) {
                    const bamd_h8 bv = *(const bamd_h8 *) (st + (size_t) (n * 16 + m) * BAMD_B16_REC + (e * 4 + g) * 16);
                    const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                    const bamd_f4 sh_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(avh, bv, z, 0, 0, 0);
                    const bamd_f4 sl_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(avl, bv, z, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[n][e][i] = fmaf(D[n][i], fmaf(16.0f, sh_[i], sl_[i]), acc[n][e][i]);
                }
            }
        }
        lds_dma_wait();
        __syncthreads();
    }
    if (!live) return;
#pragma unroll
    for (int n = 0; n < BAMD_MMA_NT; ++n) {
        const int t = t0 + n * 16 + m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float val = ((acc[n][0][i] + acc[n][4][i]) + (acc[n][2][i] + acc[n][6][i])) + ((acc[n][1][i] + acc[n][5][i]) + (acc[n][3][i] + acc[n][7][i]));
            const int row = rt * 16 + 4 * g + i;
            if (t < a.T && row < a.nrows) {
                const size_t o = (size_t) t * a.ldo + row;
                a.out[o] = EPI == BAMD_EPI_ADD ? val + a.res[o] : EPI == BAMD_EPI_SILU_MUL ? v_silu(a.res[o]) * val : val;   // SILU_MUL: res = the gate projection
            }
        }
    }
}
// grid (token tiles, row slots): consecutive workgroups share the weights (L2) and differ in the token tile
template <int EPI>
__global__ void __launch_bounds__(512) matmul_batch_kernel(bamd_mm_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nb = a.K >> 8;
    const size_t bb = BAMD_BLOB_BYTES(nb);
    const int t0 = blockIdx.x * BAMD_TT;
    const int nt = a.T - t0 < BAMD_TT ? a.T - t0 : BAMD_TT;
    {   // this tile's activations -> LDS (rows past T: repeat the last token; results discarded)
        const int n16 = (int) (bb / 16);
        for (int i = threadIdx.x; i < n16 * BAMD_TT; i += blockDim.x) {
            const int u = i / n16, k = i - u * n16;
            const int tu = t0 + (u < nt ? u : nt - 1);
            ((uint4 *) smem)[(size_t) u * n16 + k] = ((const uint4 *) (a.blob + (size_t) tu * bb))[k];
        }
    }
    __syncthreads();
    const int wave = wave_id(), nwaves = blockDim.x >> 6;
    const int slot = blockIdx.y + gridDim.y * wave, stride = gridDim.y * nwaves;
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
    int off = 0;
    const int nseg = PAIR ? 1 : a.nseg;
    for (int s = 0; s < nseg; ++s) {
        const int nrg = a.seg[s].nrows >> 3;
        const int k0 = off <= slot ? 0 : (off - slot + stride - 1) / stride;
        const int g0 = slot + k0 * stride;
        const int count = g0 < off + nrg ? (off + nrg - 1 - g0) / stride + 1 : 0;
        if (count > 0) {
            const int t = a.seg[s].type;
            const uint8_t * wA = (const uint8_t *) a.seg[s].w;
            const uint8_t * wB = PAIR ? (const uint8_t *) a.seg[1].w : wA;
            const int nv = a.seg[s].nvalid > 0 ? a.seg[s].nvalid : a.seg[s].nrows;
            // ring depth 4 when it divides the row (it does for every K % 1024 == 0), else 1
            if ((nb & 3) == 0) {
                if (t == BAMD_Q4_K)      batch_segment<BAMD_Q4_K, RecQ4K, 4, EPI>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
                else if (t == BAMD_Q5_K) batch_segment<BAMD_Q5_K, RecQ5K, 4, EPI>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
                else                     batch_segment<BAMD_Q6_K, RecQ6K, 4, EPI>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
            } else {
                if (t == BAMD_Q4_K)      batch_segment<BAMD_Q4_K, RecQ4K, 1, EPI>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
                else if (t == BAMD_Q5_K) batch_segment<BAMD_Q5_K, RecQ5K, 1, EPI>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
                else                     batch_segment<BAMD_Q6_K, RecQ6K, 1, EPI>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
            }
        }
        off += nrg;
    }
}

// batched prefill: one workgroup per token of the micro-batch
__global__ void __launch_bounds__(256) embed_batch_kernel(const int32_t * __restrict__ tokens, const uint8_t * embd, int embd_type, int E, int V, float * x) {
    int tok = tokens[blockIdx.x];
    if (tok < 0 || tok >= V) tok = 0;
    embed_row(embd, embd_type, E, tok, x + (size_t) blockIdx.x * E);
}


// ===========================================================================================================
// launchers
// ===========================================================================================================
// ---- batched prefill launchers ------------------------------------------------------------------------------------------
size_t bamd_blob_bytes(int K) { return BAMD_BLOB_BYTES(K >> 8); }
size_t bamd_blob16_bytes(int K) { return BAMD_BLOB16_BYTES(K >> 8); }
void bamd_launch_quantize_batch(const float * x, const float * nw, float eps, int K, int T, void * blob, void * blob16, hipStream_t s) {
    if (nw) hipLaunchKernelGGL((quantize_batch_kernel<true>),  dim3(T), dim3(512), act_lds_bytes(K), s, x, nw, eps, K, (uint8_t *) blob, (uint8_t *) blob16);
    else    hipLaunchKernelGGL((quantize_batch_kernel<false>), dim3(T), dim3(512), act_lds_bytes(K), s, x, nw, eps, K, (uint8_t *) blob, (uint8_t *) blob16);
}
int bamd_launch_matmul_batch(const bamd_mm_args & a, int epi, int n_cu, hipStream_t s) {
    int nrg = 0;
    if (epi == BAMD_EPI_SILU_MUL) nrg = a.seg[0].nrows >> 3; else for (int i = 0; i < a.nseg; ++i) nrg += a.seg[i].nrows >> 3;
    const size_t lds = (size_t) BAMD_TT * BAMD_BLOB_BYTES(a.K >> 8);
    if (lds > 160 * 1024) return 1;                           // K > 17920: would need a K-split of the activation tile
    const int tiles = (a.T + BAMD_TT - 1) / BAMD_TT;
    // row slots: enough workgroups to fill the chip a few times over, at least one row-group per wave
    int gy = (4 * (n_cu > 0 ? n_cu : 256) + tiles - 1) / tiles;
    if (gy * 8 > nrg) gy = (nrg + 7) / 8;
    if (gy < 1) gy = 1;
    dim3 grid(tiles, gy);
    switch (epi) {
        case BAMD_EPI_STORE:    hipLaunchKernelGGL((matmul_batch_kernel<BAMD_EPI_STORE>),    grid, dim3(512), lds, s, a); break;
        case BAMD_EPI_ADD:      hipLaunchKernelGGL((matmul_batch_kernel<BAMD_EPI_ADD>),      grid, dim3(512), lds, s, a); break;
        case BAMD_EPI_SILU_MUL: hipLaunchKernelGGL((matmul_batch_kernel<BAMD_EPI_SILU_MUL>), grid, dim3(512), lds, s, a); break;
        default: return 1;
    }
    return 0;
}
int bamd_launch_matmul_mfma(const void * w_stream, int type, int nrows, int nrows_pad, int K, const void * blob16, int T, float * out, const float * res, int epi,
                            int ldo, hipStream_t s) {
    if ((type != BAMD_Q4_K && type != BAMD_Q5_K && type != BAMD_Q6_K) || (nrows_pad & 7) || (K & 255)) return 1;
    if (epi != BAMD_EPI_STORE && epi != BAMD_EPI_ADD && epi != BAMD_EPI_SILU_MUL) return 1;
    if ((epi != BAMD_EPI_STORE) != (res != nullptr)) return 1;
    bamd_mma_args a; a.w = (const uint8_t *) w_stream; a.out = out; a.res = res; a.blob16 = (const uint8_t *) blob16; a.K = K; a.T = T; a.nrows = nrows; a.nrows_pad = nrows_pad; a.ldo = ldo;
    dim3 grid((T + BAMD_MMA_TOK - 1) / BAMD_MMA_TOK, (nrows_pad / 16 + (nrows_pad % 16 ? 1 : 0) + 7) / 8);
#define BAMD_MMA_LAUNCH(KERNEL, LDS, ...) do { \
        if (epi == BAMD_EPI_ADD)           hipLaunchKernelGGL((KERNEL<BAMD_EPI_ADD __VA_ARGS__>),      grid, dim3(512), LDS, s, a); \
        else if (epi == BAMD_EPI_SILU_MUL) hipLaunchKernelGGL((KERNEL<BAMD_EPI_SILU_MUL __VA_ARGS__>), grid, dim3(512), LDS, s, a); \
        else                               hipLaunchKernelGGL((KERNEL<BAMD_EPI_STORE __VA_ARGS__>),    grid, dim3(512), LDS, s, a); } while (0)
    const size_t lds = BAMD_MMA_NSTAGE * BAMD_MMA_STAGE + 8 * BAMD_MMA_WAVE_LDS;
    if (type == BAMD_Q6_K) { const size_t lds6 = 2 * BAMD_MMA_STAGE + 8 * BAMD_MMA6_WAVE_LDS; BAMD_MMA_LAUNCH(matmul_mfma_q6k_kernel, lds6); }
    else if (type == BAMD_Q5_K) BAMD_MMA_LAUNCH(matmul_mfma_q4k_kernel, lds, , true);
    else                        BAMD_MMA_LAUNCH(matmul_mfma_q4k_kernel, lds, , false);
#undef BAMD_MMA_LAUNCH
    return 0;
}
void bamd_launch_embed_batch(const int32_t * tokens, int T, const void * embd, int embd_type, int E, int V, float * x, hipStream_t s) {
    hipLaunchKernelGGL(embed_batch_kernel, dim3(T), dim3(256), 0, s, tokens, (const uint8_t *) embd, embd_type, E, V, x);
}
