// bamd_prefill2.hip — round 5: the exact matrix-core prefill mat-mul with the A fragments BUILT ONCE PER WORKGROUP (Q4_K / Q5_K / Q6_K x Q8_K).
//
// Reference: ggml_compute_forward_mul_mat with ne11 = T (ggml.c:12277-12492) over ggml_vec_dot_q{4,5,6}_K_q8_K (ggml-quants.c:6832 / :7400 / :8037);
// arithmetic contract and operand layouts as in bamd_prefill.hip (whose kernels stay as the second implementation, BAMD_PREFILL_V=1): ONE
// v_mfma_f32_16x16x32_f16 per SIMD lane e of the reference gives the exact integer sums isum_e of a 16-row x 16-token tile, the f32 chains
// acc_e = fma(d_x d_y, isum_e, acc_e), the min terms and the final hsum tree follow on the VALU in the reference's order.
//
// What changed, and why: in bamd_prefill.hip every wave expands the nibbles of ITS 16 rows into f16 scale x quant
// fragments and uses each fragment for two MFMAs — 365 issued instructions per wave and super-block of which ~150 are that expansion and
// the header unpacking in front of it, so the kernel is bound by what one wave must issue (MFMA pipe 19 % busy, VALU 50 %).  Here
//   * a workgroup is 64 rows x 64 tokens: eight waves = four row tiles x two token pairs (each wave still 16 rows x 32 tokens: 96
//     accumulator registers, two MFMAs per A operand held in registers);
//   * the two waves of a row tile build its eight fragments ONCE, half each, one super-block AHEAD, into an LDS ring in MFMA operand
//     layout (ds_write_b128, read back by both with ds_read_b128: 1 KiB per two MFMAs), from the raw nibble dwords (4 B per lane and
//     fragment, straight from the wave-stream records: lane (m, g) of fragment e wants dword g of stream lane (m & 7, e)) — the build of
//     super-block ci + 1 is independent of the MFMAs of ci, so it is interleaved with them, half a fragment per e;
//   * everything the header unpacking produced per step is PRECOMPUTED AT LOAD TIME into a per-matrix side table ("prefill aux", 104 B per
//     row and super-block for Q4_K / Q5_K, 72 B for Q6_K: the MI355X has the HBM for it): the builder's per-lane scale operands
//     {s, -1024 s, s/16, -64 s} as packed f16 pairs (one 16-byte load per lane and step), and the consumers' {d, dmin} as f32 plus the
//     min-term MFMA operands {2 m_a, 2 m_b, m_a, m_b}, which travel global -> LDS by DMA with the activation records.
// Per wave and super-block: ~215 issued instructions instead of 365, same 24 MFMAs, same bits.
#include "bamd_device.h"
#include "bamd_mfma_common.h"
#include <type_traits>

typedef _Float16 bamd_h2 __attribute__((ext_vector_type(2)));
union bamd_h2u { uint32_t u; bamd_h2 h; };

struct bamd_mma2_args {
    const uint8_t * w;               // wave-stream records of the matrix (bamd_formats.h)
    const uint8_t * ph;              // prefill aux, builder part:  [row tile][super-block][64 lanes][16 B]
    const uint8_t * ch;              // prefill aux, consumer part: [row block of 64][super-block][4 row tiles][X_CH_RT B]
    float * out; const float * res;  // [T][ldo]
    const uint8_t * blob16;          // f16 activation records (quantize_batch_kernel)
    int K, T, nrows, nrows_pad, ldo;
};

// LDS map (byte offsets).  Both copies of every region lie within 64 KiB of the region's first copy, so one base register per region
// serves both buffers through the 16-bit offset field of the DS instructions (the buffer in use is a compile-time parity of the step).
#define X_AF_BYTES 32768                                   /* A fragments of one super-block: 4 row tiles x 8 e x 1 KiB */
#define X_BS_BYTES (64 * BAMD_B16_REC)                     /* 64 token records of one super-block (38 912 B) */
#define X_CH4_RT 640                                       /* Q4_K / Q5_K consumer header of a row tile: 16 x {d, dmin} f32 + 16 x 4 x 8 B of min operands */
#define X_CH6_RT 64                                        /* Q6_K: 16 x d f32 */
#define X_CH_MAX (4 * X_CH4_RT)
#define X_BLK (X_BS_BYTES + X_CH_MAX + 256 + 32)           /* records | headers | 64 d_y | 32 bytes of zeros: 41 760 B */
#define X_AF0 0
#define X_BLK0 (2 * X_AF_BYTES)
#define X_LDS_BYTES (X_BLK0 + 2 * X_BLK)                   /* 149 056 B */
#define X_CH_OFF X_BS_BYTES
#define X_YD_OFF (X_BS_BYTES + X_CH_MAX)
#define X_Z_OFF (X_YD_OFF + 256)

// ---- load-time side tables ---------------------------------------------------------------------------------------------------
// grid (super-blocks, row tiles incl. the padding tiles of the last row block), 64 threads = the MFMA lanes (m = row of the tile, g)
template <bool Q5>
__global__ void __launch_bounds__(64) prefill_aux_q4k_kernel(const uint8_t * __restrict__ w, int nrows_pad, int nb, uint8_t * __restrict__ ph, uint8_t * __restrict__ ch) {
    constexpr uint32_t RECB = Q5 ? BAMD_RECB_Q5K : BAMD_RECB_Q4K, HDRO = Q5 ? 1280u : 1024u;
    const int ci = blockIdx.x, rtile = blockIdx.y, lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    const int row = rtile * 16 + m;
    uint4 hd = { 0u, 0u, 0u, 0u };
    if (row < nrows_pad) hd = *(const uint4 *) (w + ((size_t) (row >> 3) * nb + ci) * RECB + HDRO + (row & 7) * 16);
    const uint32_t u0 = hd.y, u1 = hd.z, u2 = hd.w;                                               // ggml-quants.c:6928-6933
    const uint32_t sc03 = u0 & 0x3f3f3f3fu, mn03 = u1 & 0x3f3f3f3fu;
    const uint32_t sc47 = (u2 & 0x0f0f0f0fu) | (((u0 >> 6) & 0x03030303u) << 4);
    const uint32_t mn47 = ((u2 >> 4) & 0x0f0f0f0fu) | (((u1 >> 6) & 0x03030303u) << 4);
    // builder operands of lane (m, g): scales of sub-blocks 2g, 2g+1 of row m, every value an exact f16
    const uint32_t scw = ((g >> 1) ? sc47 : sc03) >> (16 * (g & 1));
    const _Float16 s_lo = (_Float16) (float) (scw & 0xffu), s_hi = (_Float16) (float) ((scw >> 8) & 0xffu);
    bamd_h2u o0, o1, o2, o3;
    o0.h = (bamd_h2) { s_lo, s_lo }; o1.h = (bamd_h2) { (_Float16) -1024.f * s_lo, (_Float16) -1024.f * s_lo };
    if (Q5) { o2.h = (bamd_h2) { s_hi, s_hi }; o3.h = (bamd_h2) { (_Float16) -1024.f * s_hi, (_Float16) -1024.f * s_hi }; }
    else    { o2.h = (bamd_h2) { (_Float16) 0.0625f * s_hi, (_Float16) 0.0625f * s_hi }; o3.h = (bamd_h2) { (_Float16) -64.f * s_hi, (_Float16) -64.f * s_hi }; }
    *(uint4 *) (ph + ((size_t) rtile * nb + ci) * 1024 + lane * 16) = (uint4) { o0.u, o1.u, o2.u, o3.u };
    uint8_t * c = ch + (((size_t) (rtile >> 2) * nb + ci) * 4 + (rtile & 3)) * X_CH4_RT;
    if (g == 0) { float2 dd; dd.x = h2f(hd.x & 0xffffu); dd.y = h2f(hd.x >> 16); *(float2 *) (c + m * 8) = dd; }
    {   // min operands of pair l = g: {2 m_2l, 2 m_2l+1, m_2l, m_2l+1}
        const uint32_t mw = ((g >> 1) ? mn47 : mn03) >> (16 * (g & 1));
        const _Float16 ma = (_Float16) (float) (mw & 0xffu), mb = (_Float16) (float) ((mw >> 8) & 0xffu);
        bamd_h2u two, one; one.h = (bamd_h2) { ma, mb }; two.h = one.h + one.h;
        *(uint2 *) (c + 128 + m * 32 + g * 8) = (uint2) { two.u, one.u };
    }
}
// Q6_K: per lane (m, g) and e-half h the int8 scales of sub-blocks 2g, 2g+1 (record byte h*8 + c = file scales[2c + h]) split sc = sa + sl with
// sa = sc & ~15 (a multiple of 16 in [-128, 112]) and sl = sc & 15: sa x (q - 32) and sl x (q - 32) are exact f16 (|.| <= 4096 in steps of 16, <= 480)
__global__ void __launch_bounds__(64) prefill_aux_q6k_kernel(const uint8_t * __restrict__ w, int nrows_pad, int nb, uint8_t * __restrict__ ph, uint8_t * __restrict__ ch) {
    const int ci = blockIdx.x, rtile = blockIdx.y, lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    const int row = rtile * 16 + m;
    uint4 sc = { 0u, 0u, 0u, 0u }; uint32_t d16 = 0u;
    if (row < nrows_pad) {
        const uint8_t * rec = w + ((size_t) (row >> 3) * nb + ci) * 1680;
        sc = *(const uint4 *) (rec + 1536 + (row & 7) * 16); d16 = *(const unsigned short *) (rec + 1664 + (row & 7) * 2);
    }
    const uint32_t scd[4] = { sc.x, sc.y, sc.z, sc.w };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t wv = scd[h * 2 + (g >> 1)] >> (16 * (g & 1));
        const int s0 = (int) (int8_t) (wv & 0xffu), s1 = (int) (int8_t) ((wv >> 8) & 0xffu);
        const _Float16 a0 = (_Float16) (float) (s0 & ~15), l0 = (_Float16) (float) (s0 & 15), a1 = (_Float16) (float) (s1 & ~15), l1 = (_Float16) (float) (s1 & 15);
        bamd_h2u o0, o1, o2, o3; o0.h = (bamd_h2) { a0, a0 }; o1.h = (bamd_h2) { l0, l0 }; o2.h = (bamd_h2) { a1, a1 }; o3.h = (bamd_h2) { l1, l1 };
        *(uint4 *) (ph + (((size_t) rtile * nb + ci) * 2 + h) * 1024 + lane * 16) = (uint4) { o0.u, o1.u, o2.u, o3.u };
    }
    if (g == 0) *(float *) (ch + (((size_t) (rtile >> 2) * nb + ci) * 4 + (rtile & 3)) * X_CH6_RT + m * 4) = h2f(d16);
}

// ---- the activation / header stage: global -> LDS by DMA --------------------------------------------------------------------------
// chunk list of a step (16 B each): 2432 chunks of token records (64 tokens x 38), then the NCH chunks of the row block's consumer headers;
// wave-instruction k = 64 consecutive chunks; round r: instruction 8 r + wave.  Rounds 0..3 (and round 4 for waves 0..5) are records; round
// 4 of waves 6, 7 and the tail instructions are headers (NCH = 160 for Q4_K / Q5_K: a half instruction on wave 0; 16 for Q6_K: lanes 0..15 of
// wave 6); the 64 block scales d_y go as one 4-byte instruction on wave 1.
template <int NCH>
struct XStage {
    uint32_t off[5]; uint32_t offx; bool x_on;
    __device__ __forceinline__ void plan(int tid, int wave, int lane, int t0, int T, size_t b16, int nb) {
        auto rec_off = [&](int idx) { const int tok = idx / BAMD_B16_Q, q = idx - tok * BAMD_B16_Q; const int tg = t0 + tok < T ? t0 + tok : T - 1; return (uint32_t) ((size_t) tg * b16 + (size_t) q * 16); };
#pragma unroll
        for (int r = 0; r < 4; ++r) off[r] = rec_off(r * 512 + tid);
        off[4] = wave < 6 ? rec_off(2048 + tid) : (uint32_t) (tid - 384) * 16u;
        if (NCH > 128) { offx = wave == 0 ? (uint32_t) (128 + lane) * 16u : 0u; x_on = wave == 0 && lane < NCH - 128; } else { offx = 0u; x_on = false; }
        if (wave == 1) { const int tg = t0 + lane < T ? t0 + lane : T - 1; offx = (uint32_t) ((size_t) tg * b16 + (size_t) nb * BAMD_B16_REC); }
    }
    // recs = blob16 + ci * BAMD_B16_REC, hdrs = headers of (row block, ci), yds = blob16 + ci * 4; blk = LDS byte address of the block to fill
    __device__ __forceinline__ void issue(const uint8_t * recs, const uint8_t * hdrs, const uint8_t * yds, uint32_t blk, int wave, int lane) const {
        const uint32_t wdst = blk + (uint32_t) wave * 1024u;
#pragma unroll
        for (int r = 0; r < 4; ++r) lds_dma16_s(recs, off[r], wdst + (uint32_t) r * 8192u);
        if (wave < 6) lds_dma16_s(recs, off[4], wdst + 32768u);
        else if (NCH >= 128 || (wave == 6 && lane < NCH)) lds_dma16_s(hdrs, off[4], wdst + 32768u);
        if (NCH > 128 && x_on) lds_dma16_s(hdrs, offx, blk + X_CH_OFF + 2048u);
        if (wave == 1) lds_dma4_s(yds, offx, blk + X_YD_OFF);
    }
};

// ---- Q4_K / Q5_K ----------------------------------------------------------------------------------------------------------------------
// MFMA lane l = (m = l & 15, g = l >> 4): A row m, B token m, k-slots (g, i) = (sub-block 2g + (i >> 2), u = i & 3); C/D rows 4g + i, token m.
template <int EPI, bool Q5>
__global__ void __launch_bounds__(512) matmul_mfma2_q4k_kernel(bamd_mma2_args a) {
    constexpr uint32_t RECB = Q5 ? BAMD_RECB_Q5K : BAMD_RECB_Q4K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), m = lane & 15, g = lane >> 4;
    const int rt = wave >> 1, tp = wave & 1;
    const int nb = a.K >> 8;
    const int rb = blockIdx.y, t0 = blockIdx.x * 64;
    const int rtg = rb * 4 + rt;                                    // row tile: rows rtg*16 .. +15 = record groups 2 rtg, 2 rtg + 1
    const bool live = rtg * 16 < a.nrows_pad;
    const size_t b16 = BAMD_BLOB16_BYTES(nb);
    const uint32_t lds0 = (uint32_t) (size_t) (bamd_lds_vp) smem;
    XStage<160> stg; stg.plan(tid, wave, lane, t0, a.T, b16, nb);
    const uint8_t * chb = a.ch + (size_t) rb * nb * (4 * X_CH4_RT);
#define X_STAGE(ci_, b_) stg.issue(a.blob16 + (size_t) (ci_) * BAMD_B16_REC, chb + (size_t) (ci_) * (4 * X_CH4_RT), a.blob16 + (size_t) (ci_) * 4, lds0 + X_BLK0 + (uint32_t) (b_) * X_BLK, wave, lane)
    // builder: raw nibble dwords of fragments e = 4 tp + j (j = 0..3) of row tile rt, and the lane's scale operands
    const int rg0 = (live ? rtg : rb * 4) * 2;
    const bool two = (rg0 + 1) * 8 < a.nrows_pad;                   // a last tile of 8 (padded) rows reads its first record group twice (rows 8..15 are never stored)
    const uint8_t * wrec = a.w + (size_t) rg0 * nb * RECB;
    const uint32_t vraw = ((m >= 8 && two) ? (uint32_t) nb * RECB : 0u) + (uint32_t) ((m & 7) * 8 + 4 * tp) * 16u + (uint32_t) g * 4u;
    const uint32_t vqh = ((m >= 8 && two) ? (uint32_t) nb * RECB : 0u) + 1024u + (uint32_t) ((m & 7) * 8 + 4 * tp) * 4u;
    const uint8_t * phb = a.ph + (size_t) rtg * nb * 1024 + (size_t) lane * 16;
    uint32_t raw[2][4], qh[2][4]; uint4 sc[2];
    auto load_set = [&](int ci, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        const uint8_t * r = wrec + (size_t) ci * RECB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            raw[S][j] = *(const uint32_t *) (r + vraw + j * 16);
            if (Q5) qh[S][j] = *(const uint32_t *) (r + vqh + j * 4);
        }
        sc[S] = *(const uint4 *) (phb + (size_t) ci * 1024);
    };
    // one fragment = two halves of work: (1) nibbles -> f16 images 1024 + n, (2) x scale, store
    bamd_h2u c0, c1, c2, c3;
    const uint32_t qs0 = 2u * (uint32_t) g, qs1 = qs0 + 1u;
    auto build_a = [&](uint32_t wq, uint32_t q) {
        uint32_t lo = wq & 0x0f0f0f0fu, hi = Q5 ? (wq >> 4) & 0x0f0f0f0fu : wq & 0xf0f0f0f0u;      // Q4_K: the high nibbles stay in place (16 n; the scale operand is s / 16)
        if (Q5) { lo |= ((q >> qs0) & 0x01010101u) << 4; hi |= ((q >> qs1) & 0x01010101u) << 4; }  // bit c of byte u of the row's high-bit dword e: element 4e+u of sub-block c
        c0.u = __builtin_amdgcn_perm(0x64646464u, lo, 0x04010400u); c1.u = __builtin_amdgcn_perm(0x64646464u, lo, 0x04030402u);
        c2.u = __builtin_amdgcn_perm(0x64646464u, hi, 0x04010400u); c3.u = __builtin_amdgcn_perm(0x64646464u, hi, 0x04030402u);
    };
    auto build_b = [&](const uint4 & s, unsigned char * dst) {
        bamd_h2u s0, n0, s1, n1, a0, a1, a2, a3; s0.u = s.x; n0.u = s.y; s1.u = s.z; n1.u = s.w;
        a0.h = __builtin_elementwise_fma(c0.h, s0.h, n0.h); a1.h = __builtin_elementwise_fma(c1.h, s0.h, n0.h);     // (1024 + n) s - 1024 s = n s, one rounding, exact
        a2.h = __builtin_elementwise_fma(c2.h, s1.h, n1.h); a3.h = __builtin_elementwise_fma(c3.h, s1.h, n1.h);
        *(uint4 *) dst = (uint4) { a0.u, a1.u, a2.u, a3.u };
    };
    // per-lane LDS addresses (first copy of each region)
    unsigned char * afw = smem + X_AF0 + rt * 8192 + (4 * tp) * 1024 + lane * 16;                  // where this wave writes its fragments
    const unsigned char * afr = smem + X_AF0 + rt * 8192 + lane * 16;                              // where it reads the tile's eight
    const unsigned char * bop = smem + X_BLK0 + (size_t) ((2 * tp) * 16 + m) * BAMD_B16_REC + g * 16;      // B operands of token tile 0 (tile 1: + 16 records)
    const unsigned char * bmn = smem + X_BLK0 + (size_t) ((2 * tp) * 16 + m) * BAMD_B16_REC + 512 + (Q5 ? g * 8 : 0);
    const unsigned char * chd = smem + X_BLK0 + X_CH_OFF + rt * X_CH4_RT + g * 32;                 // {d, dmin} of rows 4g .. 4g+3
    const unsigned char * cmn = Q5 ? smem + X_BLK0 + X_CH_OFF + rt * X_CH4_RT + 128 + m * 32 + g * 8
                                   : (g == 0 ? smem + X_BLK0 + X_CH_OFF + rt * X_CH4_RT + 128 + m * 32 : smem + X_BLK0 + X_Z_OFF);
    const unsigned char * ydp = smem + X_BLK0 + X_YD_OFF + ((2 * tp) * 16 + m) * 4;
    bamd_f4 acc[2][8], accm[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[n][e] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int l = 0; l < 4; ++l) accm[n][l] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
    }
    // prologue: stage 0 in flight, fragments of super-block 0 built, operands of super-block 1 requested
    if (tid < 16) *(uint32_t *) (smem + X_BLK0 + X_Z_OFF + (tid >> 3) * X_BLK + (tid & 7) * 4) = 0u;
    X_STAGE(0, 0);
    load_set(0, std::integral_constant<int, 0>());
    load_set(nb > 1 ? 1 : 0, std::integral_constant<int, 1>());
#pragma unroll
    for (int j = 0; j < 4; ++j) { build_a(raw[0][j], Q5 ? qh[0][j] : 0u); build_b(sc[0], afw + j * 1024); }
    lds_dma_wait();
    __syncthreads();
    auto step = [&](const int ci, auto cur_tag) {
        constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
        // the next stage (records, headers, d_y of ci + 1) and the builder operands of ci + 2; at the end: the last super-block again
        X_STAGE(ci + 1 < nb ? ci + 1 : nb - 1, NXT);
        load_set(ci + 2 < nb ? ci + 2 : nb - 1, std::integral_constant<int, CUR>());
        __builtin_amdgcn_sched_barrier(0);
        float D[2][4], Dm[2][4];
        {
            const bamd_f4 h0 = *(const bamd_f4 *) (chd + CUR * X_BLK), h1 = *(const bamd_f4 *) (chd + CUR * X_BLK + 16);   // {d, dmin} x 2 rows each
            const float dw[4] = { h0[0], h0[2], h1[0], h1[2] }, dmw[4] = { h0[1], h0[3], h1[1], h1[3] };
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float ydv = *(const float *) (ydp + CUR * X_BLK + n * 64);
#pragma unroll
                for (int i = 0; i < 4; ++i) { D[n][i] = ydv * dw[i]; Dm[n][i] = (-ydv) * dmw[i]; }
            }
        }
        {
            // software pipeline over e: the LDS operands of e + 2 are requested at the top of iteration e, the MFMA results of e - 1 are folded
            // into the chains in iteration e, and half a fragment of the NEXT super-block is built in every iteration
            bamd_h8 Aq[3], Bq[3][2]; bamd_f4 sprev[2];
#define X_LDA(e_) (*(const bamd_h8 *) (afr + CUR * X_AF_BYTES + (e_) * 1024))
#define X_LDB(e_, n_) (*(const bamd_h8 *) (bop + CUR * X_BLK + (n_) * (16 * BAMD_B16_REC) + (e_) * 64))
#pragma unroll
            for (int e = 0; e < 2; ++e) { Aq[e] = X_LDA(e); Bq[e][0] = X_LDB(e, 0); Bq[e][1] = X_LDB(e, 1); }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e + 2 < 8) { Aq[(e + 2) % 3] = X_LDA(e + 2); Bq[(e + 2) % 3][0] = X_LDB(e + 2, 0); Bq[(e + 2) % 3][1] = X_LDB(e + 2, 1); }
                if ((e & 1) == 0) build_a(raw[NXT][e >> 1], Q5 ? qh[NXT][e >> 1] : 0u);
                bamd_f4 si[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                    si[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[e % 3], Bq[e % 3][n], z, 0, 0, 0);
                    if (e > 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[n][e - 1][i] = fmaf(D[n][i], sprev[n][i], acc[n][e - 1][i]);
                    }
                }
                if (e & 1) build_b(sc[NXT], afw + NXT * X_AF_BYTES + (e >> 1) * 1024);
                sprev[0] = si[0]; sprev[1] = si[1];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[n][7][i] = fmaf(D[n][i], sprev[n][i], acc[n][7][i]);
            }
#undef X_LDA
#undef X_LDB
        }
        // min terms (ggml-quants.c:6937-6941 / :7515-7518): exact integer products on the matrix core, operands precomputed
        if (Q5) {
            union { uint2 u; bamd_h4 h; } av; av.u = *(const uint2 *) (cmn + CUR * X_BLK);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                union { uint2 u; bamd_h4 h; } bv; bv.u = *(const uint2 *) (bmn + CUR * X_BLK + n * (16 * BAMD_B16_REC));
                const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                const bamd_f4 pm = __builtin_amdgcn_mfma_f32_16x16x16f16(av.h, bv.h, z, 0, 0, 0);              // sum_j m_j S_j of rows 4g + i, token m
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float t = Dm[n][i] * pm[i]; accm[n][0][i] = accm[n][0][i] + t; }
            }
        } else {
            const uint4 ma = *(const uint4 *) (cmn + CUR * X_BLK), mb = *(const uint4 *) (cmn + CUR * X_BLK + 16);
            const uint2 al[4] = { { ma.x, ma.y }, { ma.z, ma.w }, { mb.x, mb.y }, { mb.z, mb.w } };
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const uint4 sfa = *(const uint4 *) (bmn + CUR * X_BLK + n * (16 * BAMD_B16_REC)), sfb = *(const uint4 *) (bmn + CUR * X_BLK + n * (16 * BAMD_B16_REC) + 16);
                const uint2 bl[4] = { { sfa.x, sfa.y }, { sfa.z, sfa.w }, { sfb.x, sfb.y }, { sfb.z, sfb.w } };
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    union { uint2 u; bamd_h4 h; } av4, bv4; av4.u = al[l]; bv4.u = bl[l];
                    const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                    const bamd_f4 pm = __builtin_amdgcn_mfma_f32_16x16x16f16(av4.h, bv4.h, z, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) accm[n][l][i] = fmaf(Dm[n][i], pm[i], accm[n][l][i]);
                }
            }
        }
        lds_dma_wait();
        __syncthreads();                                     // stage and fragments of ci + 1 visible; those of ci free
    };
    for (int ci = 0; ci < nb; ci += 2) {
        step(ci, std::integral_constant<int, 0>());
        if (ci + 1 < nb) step(ci + 1, std::integral_constant<int, 1>());
    }
#undef X_STAGE
    if (!live) return;
    // hsum_float_8 over e and the acc_m folds, in the reference's order (finish_row), then the epilogue
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int t = t0 + (2 * tp + n) * 16 + m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = ((acc[n][0][i] + acc[n][4][i]) + (acc[n][2][i] + acc[n][6][i])) + ((acc[n][1][i] + acc[n][5][i]) + (acc[n][3][i] + acc[n][7][i]));
            const float mm = Q5 ? accm[n][0][i] : (accm[n][0][i] + accm[n][2][i]) + (accm[n][1][i] + accm[n][3][i]);
            const float val = v + mm;
            const int row = rtg * 16 + 4 * g + i;
            if (t < a.T && row < a.nrows) {
                const size_t o = (size_t) t * a.ldo + row;
                a.out[o] = EPI == BAMD_EPI_ADD ? val + a.res[o] : EPI == BAMD_EPI_SILU_MUL ? v_silu(a.res[o]) * val : val;   // SILU_MUL: res = the gate projection
            }
        }
    }
}

// ---- Q6_K: a step is HALF a super-block (four e: 16 MFMAs per wave, two per e and token tile) -----------------------------------------------
// scale x (q - 32) reaches 4096: the scale is split sc = sa + sl (prefill_aux_q6k_kernel), both fragments come from one f16 image v = (1024 + q) - 1056
// of the quants as v x sa and v x sl (exact), and the two MFMAs of an (e, token tile) are CHAINED through the accumulator: S = A_l.B + (A_a.B + 0)
// is the exact integer isum (|isum| < 2^24, every partial sum an integer below that bound) — the fmaf(16, S_1, S_2) of bamd_prefill.hip is gone.
// Scales are per 16 elements: for SIMD lane e the sub-block c uses scales[2c + (e >= 4)] (ggml-quants.c:8145-8216), i.e. the builder operands of a
// half step are those of e-half h.  Fragment ring: two half steps x 4 row tiles x 4 e x 2 fragments x 1 KiB = 2 x 32 KiB; the activation stage
// (whole super-blocks) is refilled every second half step.
template <int EPI>
__global__ void __launch_bounds__(512) matmul_mfma2_q6k_kernel(bamd_mma2_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), m = lane & 15, g = lane >> 4;
    const int rt = wave >> 1, tp = wave & 1;
    const int nb = a.K >> 8;
    const int rb = blockIdx.y, t0 = blockIdx.x * 64;
    const int rtg = rb * 4 + rt;
    const bool live = rtg * 16 < a.nrows_pad;
    const size_t b16 = BAMD_BLOB16_BYTES(nb);
    const uint32_t lds0 = (uint32_t) (size_t) (bamd_lds_vp) smem;
    XStage<16> stg; stg.plan(tid, wave, lane, t0, a.T, b16, nb);
    const uint8_t * chb = a.ch + (size_t) rb * nb * (4 * X_CH6_RT);
#define X_STAGE(ci_, b_) stg.issue(a.blob16 + (size_t) (ci_) * BAMD_B16_REC, chb + (size_t) (ci_) * (4 * X_CH6_RT), a.blob16 + (size_t) (ci_) * 4, lds0 + X_BLK0 + (uint32_t) (b_) * X_BLK, wave, lane)
    // builder: fragments e = 4 h + 2 tp + j (j = 0, 1) of half step h: per lane and fragment the ql dwords of sub-blocks 2g / 2g+1 (stream lane
    // (m & 7, e), dwords 2 (g >> 1) and + 1) and the qh dword (g >> 1); the shift 4 (g & 1) picks the nibble, the rotations the two high bits
    const int rg0 = (live ? rtg : rb * 4) * 2;
    const bool two = (rg0 + 1) * 8 < a.nrows_pad;
    const uint8_t * wrec = a.w + (size_t) rg0 * nb * 1680;
    const uint32_t vrow = (m >= 8 && two) ? (uint32_t) nb * 1680u : 0u;
    const uint32_t vql = vrow + (uint32_t) ((m & 7) * 8 + 2 * tp) * 16u + (uint32_t) (g >> 1) * 8u;
    const uint32_t vqh = vrow + 1024u + (uint32_t) ((m & 7) * 8 + 2 * tp) * 8u + (uint32_t) (g >> 1) * 4u;
    const uint8_t * phb = a.ph + (size_t) rtg * nb * 2048 + (size_t) lane * 16;
    uint2 ql[2][2]; uint32_t qh[2][2]; uint4 sc[2];
    auto load_set = [&](int hs, auto set_tag) {                // operands of half step hs = 2 ci + h
        constexpr int S = decltype(set_tag)::value;
        const int ci = hs >> 1, h = hs & 1;
        const uint8_t * r = wrec + (size_t) ci * 1680 + h * 64;                       // e = 4h + ...: 4 stream lanes = 64 B of ql, 32 B of qh further on
#pragma unroll
        for (int j = 0; j < 2; ++j) { ql[S][j] = *(const uint2 *) (r + vql + j * 16); qh[S][j] = *(const uint32_t *) (r - h * 32 + vqh + j * 8); }
        sc[S] = *(const uint4 *) (phb + (size_t) hs * 1024);
    };
    const uint32_t sh = 4u * (uint32_t) (g & 1);
    const uint32_t rotA = (28u + sh) & 31u, rotB = (30u + sh) & 31u;
    const bamd_h2 k1056 = { (_Float16) -1056.f, (_Float16) -1056.f };
    bamd_h2u v0, v1, v2, v3;
    auto build_a = [&](const uint2 & q, uint32_t hq) {
        // the two high bits of a quant sit at bits sh, sh + 1 (sub-block 2g) / sh + 2, sh + 3 (2g + 1) of their byte of hq and belong at bits 4, 5:
        // a ROTATION of the dword (what wraps around lands outside the mask 0x30 of every byte), then one and-or
        const uint32_t uA = (__builtin_amdgcn_alignbit(hq, hq, rotA) & 0x30303030u) | ((q.x >> sh) & 0x0f0f0f0fu);
        const uint32_t uB = (__builtin_amdgcn_alignbit(hq, hq, rotB) & 0x30303030u) | ((q.y >> sh) & 0x0f0f0f0fu);
        bamd_h2u c;
        c.u = __builtin_amdgcn_perm(0x64646464u, uA, 0x04010400u); v0.h = c.h + k1056;      // (1024 + q) - 1056 = q - 32, exact
        c.u = __builtin_amdgcn_perm(0x64646464u, uA, 0x04030402u); v1.h = c.h + k1056;
        c.u = __builtin_amdgcn_perm(0x64646464u, uB, 0x04010400u); v2.h = c.h + k1056;
        c.u = __builtin_amdgcn_perm(0x64646464u, uB, 0x04030402u); v3.h = c.h + k1056;
    };
    auto build_b = [&](const uint4 & s, unsigned char * dst) {  // dst: the A_a fragment; A_l 1 KiB behind it
        bamd_h2u sa0, sl0, sa1, sl1, x0, x1, x2, x3; sa0.u = s.x; sl0.u = s.y; sa1.u = s.z; sl1.u = s.w;
        x0.h = v0.h * sa0.h; x1.h = v1.h * sa0.h; x2.h = v2.h * sa1.h; x3.h = v3.h * sa1.h;
        *(uint4 *) dst = (uint4) { x0.u, x1.u, x2.u, x3.u };
        x0.h = v0.h * sl0.h; x1.h = v1.h * sl0.h; x2.h = v2.h * sl1.h; x3.h = v3.h * sl1.h;
        *(uint4 *) (dst + 1024) = (uint4) { x0.u, x1.u, x2.u, x3.u };
    };
    unsigned char * afw = smem + X_AF0 + rt * 8192 + (2 * tp) * 2048 + lane * 16;                  // [row tile][e' = 0..3][a | l][1 KiB]
    const unsigned char * afr = smem + X_AF0 + rt * 8192 + lane * 16;
    const unsigned char * bop = smem + X_BLK0 + (size_t) ((2 * tp) * 16 + m) * BAMD_B16_REC + g * 16;
    const unsigned char * chd = smem + X_BLK0 + X_CH_OFF + rt * X_CH6_RT + g * 16;                 // d of rows 4g .. 4g+3
    const unsigned char * ydp = smem + X_BLK0 + X_YD_OFF + ((2 * tp) * 16 + m) * 4;
    bamd_f4 acc[2][8];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[n][e] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
    }
    X_STAGE(0, 0);
    load_set(0, std::integral_constant<int, 0>());
    load_set(1, std::integral_constant<int, 1>());
#pragma unroll
    for (int j = 0; j < 2; ++j) { build_a(ql[0][j], qh[0][j]); build_b(sc[0], afw + j * 2048); }
    lds_dma_wait();
    __syncthreads();
    const int nhs = 2 * nb;
    // half step hs = 2 ci + H: fragments in ring slot H, activation block ci & 1 = BLK
    auto step = [&](const int hs, auto h_tag, auto blk_tag) {
        constexpr int H = decltype(h_tag)::value, BLK = decltype(blk_tag)::value, NXT = H ^ 1;
        const int ci = hs >> 1;
        if (H == 0) X_STAGE(ci + 1 < nb ? ci + 1 : nb - 1, BLK ^ 1);                               // the next super-block's records: a whole step ahead
        load_set(hs + 2 < nhs ? hs + 2 : nhs - 2 + H, std::integral_constant<int, H>());
        __builtin_amdgcn_sched_barrier(0);
        float D[2][4];
        {
            const bamd_f4 dw = *(const bamd_f4 *) (chd + BLK * X_BLK);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float ydv = *(const float *) (ydp + BLK * X_BLK + n * 64);
#pragma unroll
                for (int i = 0; i < 4; ++i) D[n][i] = ydv * dw[i];
            }
        }
        bamd_h8 Aa[2], Al[2], Bq[2][2];
#define X_LDA(e_, p_) (*(const bamd_h8 *) (afr + H * X_AF_BYTES + (e_) * 2048 + (p_) * 1024))
#define X_LDB(e_, n_) (*(const bamd_h8 *) (bop + BLK * X_BLK + (n_) * (16 * BAMD_B16_REC) + (4 * H + (e_)) * 64))
        Aa[0] = X_LDA(0, 0); Al[0] = X_LDA(0, 1); Bq[0][0] = X_LDB(0, 0); Bq[0][1] = X_LDB(0, 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e + 1 < 4) { Aa[(e + 1) & 1] = X_LDA(e + 1, 0); Al[(e + 1) & 1] = X_LDA(e + 1, 1); Bq[(e + 1) & 1][0] = X_LDB(e + 1, 0); Bq[(e + 1) & 1][1] = X_LDB(e + 1, 1); }
            if ((e & 1) == 0) build_a(ql[NXT][e >> 1], qh[NXT][e >> 1]);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                const bamd_f4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aa[e & 1], Bq[e & 1][n], z, 0, 0, 0);
                const bamd_f4 s2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[e & 1], Bq[e & 1][n], s1, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[n][4 * H + e][i] = fmaf(D[n][i], s2[i], acc[n][4 * H + e][i]);
            }
            if (e & 1) build_b(sc[NXT], afw + NXT * X_AF_BYTES + (e >> 1) * 2048);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef X_LDA
#undef X_LDB
        if (H == 1) lds_dma_wait();
        __syncthreads();
    };
    for (int ci = 0; ci < nb; ci += 2) {
        step(2 * ci, std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
        step(2 * ci + 1, std::integral_constant<int, 1>(), std::integral_constant<int, 0>());
        if (ci + 1 < nb) {
            step(2 * ci + 2, std::integral_constant<int, 0>(), std::integral_constant<int, 1>());
            step(2 * ci + 3, std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
        }
    }
#undef X_STAGE
    if (!live) return;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int t = t0 + (2 * tp + n) * 16 + m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float val = ((acc[n][0][i] + acc[n][4][i]) + (acc[n][2][i] + acc[n][6][i])) + ((acc[n][1][i] + acc[n][5][i]) + (acc[n][3][i] + acc[n][7][i]));
            const int row = rtg * 16 + 4 * g + i;
            if (t < a.T && row < a.nrows) {
                const size_t o = (size_t) t * a.ldo + row;
                a.out[o] = EPI == BAMD_EPI_ADD ? val + a.res[o] : EPI == BAMD_EPI_SILU_MUL ? v_silu(a.res[o]) * val : val;
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------------------
static inline int x_row_blocks(int nrows_pad) { return (nrows_pad + 63) / 64; }
// bytes of the side table of a K-quant matrix [nrows_pad][K]: builder part first, the consumer part behind it (both 16-byte aligned)
size_t bamd_prefill_aux_bytes(int type, int nrows_pad, int K) {
    if ((type != BAMD_Q4_K && type != BAMD_Q5_K && type != BAMD_Q6_K) || (K & 255) || (nrows_pad & 7)) return 0;
    const size_t nb = (size_t) (K >> 8), rbk = (size_t) x_row_blocks(nrows_pad);
    return type == BAMD_Q6_K ? rbk * 4 * nb * 2048 + rbk * nb * 4 * X_CH6_RT : rbk * 4 * nb * 1024 + rbk * nb * 4 * X_CH4_RT;
}
static inline size_t x_ph_bytes(int type, int nrows_pad, int K) { return (size_t) x_row_blocks(nrows_pad) * 4 * (size_t) (K >> 8) * (type == BAMD_Q6_K ? 2048 : 1024); }
void bamd_launch_prefill_aux(const void * w_stream, int type, int nrows_pad, int K, void * aux, hipStream_t s) {
    const int nb = K >> 8, nrt = x_row_blocks(nrows_pad) * 4;
    uint8_t * ph = (uint8_t *) aux, * ch = ph + x_ph_bytes(type, nrows_pad, K);
    const dim3 grid(nb, nrt);
    if (type == BAMD_Q6_K)      hipLaunchKernelGGL(prefill_aux_q6k_kernel, grid, dim3(64), 0, s, (const uint8_t *) w_stream, nrows_pad, nb, ph, ch);
    else if (type == BAMD_Q5_K) hipLaunchKernelGGL((prefill_aux_q4k_kernel<true>), grid, dim3(64), 0, s, (const uint8_t *) w_stream, nrows_pad, nb, ph, ch);
    else                        hipLaunchKernelGGL((prefill_aux_q4k_kernel<false>), grid, dim3(64), 0, s, (const uint8_t *) w_stream, nrows_pad, nb, ph, ch);
}
int bamd_launch_matmul_mfma2(const void * w_stream, const void * aux, int type, int nrows, int nrows_pad, int K, const void * blob16, int T, float * out, const float * res,
                             int epi, int ldo, hipStream_t s) {
    if ((type != BAMD_Q4_K && type != BAMD_Q5_K && type != BAMD_Q6_K) || (nrows_pad & 7) || (K & 255) || !aux) return 1;
    if (epi != BAMD_EPI_STORE && epi != BAMD_EPI_ADD && epi != BAMD_EPI_SILU_MUL) return 1;
    if ((epi != BAMD_EPI_STORE) != (res != nullptr)) return 1;
    bamd_mma2_args a; a.w = (const uint8_t *) w_stream; a.ph = (const uint8_t *) aux; a.ch = a.ph + x_ph_bytes(type, nrows_pad, K);
    a.out = out; a.res = res; a.blob16 = (const uint8_t *) blob16; a.K = K; a.T = T; a.nrows = nrows; a.nrows_pad = nrows_pad; a.ldo = ldo;
    const dim3 grid((T + 63) / 64, x_row_blocks(nrows_pad));
#define X_LAUNCH(KERNEL, ...) do { \
        if (epi == BAMD_EPI_ADD)           hipLaunchKernelGGL((KERNEL<BAMD_EPI_ADD __VA_ARGS__>),      grid, dim3(512), X_LDS_BYTES, s, a); \
        else if (epi == BAMD_EPI_SILU_MUL) hipLaunchKernelGGL((KERNEL<BAMD_EPI_SILU_MUL __VA_ARGS__>), grid, dim3(512), X_LDS_BYTES, s, a); \
        else                               hipLaunchKernelGGL((KERNEL<BAMD_EPI_STORE __VA_ARGS__>),    grid, dim3(512), X_LDS_BYTES, s, a); } while (0)
    if (type == BAMD_Q6_K)      X_LAUNCH(matmul_mfma2_q6k_kernel);
    else if (type == BAMD_Q5_K) X_LAUNCH(matmul_mfma2_q4k_kernel, , true);
    else                        X_LAUNCH(matmul_mfma2_q4k_kernel, , false);
#undef X_LAUNCH
    return 0;
}
