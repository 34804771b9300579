// bamd_prefill2.hip — round 5: the exact matrix-core prefill mat-mul with the A fragments BUILT ONCE PER WORKGROUP (Q4_K / Q5_K / Q6_K x Q8_K).
//
// Reference: ggml_compute_forward_mul_mat with ne11 = T (ggml.c:12277-12492) over ggml_vec_dot_q{4,5,6}_K_q8_K (ggml-quants.c:6832 / :7400 / :8037);
// the second implementation it is tested against is the integer-dot kernel of bamd_prefill.hip (matmul_batch_kernel): ONE
// v_mfma_f32_16x16x32_f16 per SIMD lane e of the reference gives the exact integer sums isum_e of a 16-row x 16-token tile, the f32 chains
// acc_e = fma(d_x d_y, isum_e, acc_e), the min terms and the final hsum tree follow on the VALU in the reference's order.
//
// What changed against bamd_prefill.hip, and why: there every wave expands the nibbles of ITS 16 rows into f16 scale x quant
// fragments and uses each fragment for two MFMAs — 365 issued instructions per wave and super-block, ~150 of them that expansion and the header
// unpacking in front of it.  Here
//   * a workgroup is 64 rows x 64 tokens and every fragment is built ONCE per workgroup, one super-block AHEAD, into an LDS ring in MFMA operand layout
//     (ds_write_b128, read back with ds_read_b128), by builders that work in the wave-stream's own lane order (lane (r, e) of a record group holds the
//     halves of MFMA lanes (row, g = 0..3) of fragment e: one coalesced request per wave); the build of super-block ci + 1 is independent of the MFMAs of
//     ci and interleaved with them;
//   * everything the header unpacking produced per step is PRECOMPUTED AT LOAD TIME into a per-matrix side table ("prefill aux", 104 B per row and
//     super-block for Q4_K / Q5_K, 72 B for Q6_K: the MI355X has the HBM for it): the builders' per-(row, sub-block pair) scale operands
//     {s, -1024 s, s'/16, -64 s'} as packed f16 pairs, and the consumers' d, dmin as f32 plus the min-term MFMA operands {2 m_a, 2 m_b, m_a, m_b}, which
//     travel global -> LDS by DMA with the activation records;
//   * two layouts: sixteen waves with one 16 x 16 tile each (matmul_mfma3_q4k_kernel: Q4_K / Q5_K), eight waves with 16 x 32 each (matmul_mfma2_q6k_kernel: Q6_K).
// Same bits as the round-2 kernels (removed in round 6), a third fewer vector instructions per MFMA — and the same time: DESIGN 4c has the measurements of what
// bounds it, profiles/r06_prefill_ceiling.txt the timing-only ceiling builds (-DBAMD_PREFILL_CEILING).
#include "bamd_device.h"
#include "bamd_mfma_common.h"
#include <type_traits>
#include <stdlib.h>

typedef _Float16 bamd_h2 __attribute__((ext_vector_type(2)));
union bamd_h2u { uint32_t u; bamd_h2 h; };

struct bamd_mma2_args {
    const uint8_t * w;               // wave-stream records of the matrix (bamd_formats.h)
    const uint8_t * ph;              // prefill aux, builder part:  [row block of 64][super-block]([e-half], Q6_K)[4 row tiles][64 MFMA lanes][16 B]
    const uint8_t * ch;              // prefill aux, consumer part: [row block of 64][super-block][4 row tiles][X_CH_RT B]
    float * out; const float * res;  // [T][ldo]
    const uint8_t * blob16;          // f16 activation records (quantize_batch_kernel)
    int K, T, nrows, nrows_pad, ldo;
    unsigned long long * dbg;        // -DX_TIMING builds: per-wave phase clocks of workgroup (0, 0) (tools/prefill_phase.py); else unused
};

// LDS map (byte offsets).  Both copies of every region lie within 64 KiB of the region's first copy, so one base register per region
// serves both buffers through the 16-bit offset field of the DS instructions (the buffer in use is a compile-time parity of the step).
#define X_FR 1040                                          /* bytes between fragments: 1 KiB + 16 — the builder's 16-byte stores of eight lanes (one row, e = 0..7) then fall on eight
                                                              different bank quads (260 dwords = 4 mod 64); a fragment itself stays contiguous (lane * 16) for the consumers' ds_read_b128 */
#define X_AF_BYTES (32 * X_FR)                             /* A fragments of one super-block: 4 row tiles x 8 fragments */
#define X_BS_BYTES (64 * BAMD_B16_REC)                     /* 64 token records of one super-block (38 912 B) */
#define X_CH4_RT 640                                       /* Q4_K / Q5_K consumer header of a row tile: 16 x d f32 | 16 x dmin f32 | 16 x 4 x 8 B of min operands */
#define X_CH6_RT 64                                        /* Q6_K: 16 x d f32 */
#define X_CH_MAX (4 * X_CH4_RT)
#define X3_CHS 3072                                        /* Q4_K / Q5_K consumer headers of a row block and super-block in the side table: 4 x 640 B, padded to three DMA instructions */
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
    *(uint4 *) (ph + (((size_t) (rtile >> 2) * nb + ci) * 4 + (rtile & 3)) * 1024 + lane * 16) = (uint4) { o0.u, o1.u, o2.u, o3.u };
    uint8_t * c = ch + ((size_t) (rtile >> 2) * nb + ci) * X3_CHS + (rtile & 3) * X_CH4_RT;
    if (g == 0) { *(float *) (c + m * 4) = h2f(hd.x & 0xffffu); *(float *) (c + 64 + m * 4) = h2f(hd.x >> 16); }     // d of the 16 rows, then dmin of the 16 rows
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
        *(uint4 *) (ph + ((((size_t) (rtile >> 2) * nb + ci) * 2 + h) * 4 + (rtile & 3)) * 1024 + lane * 16) = (uint4) { o0.u, o1.u, o2.u, o3.u };
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

// ---- Q4_K / Q5_K: the sixteen-wave kernel further down (matmul_mfma3_q4k_kernel).  MFMA lane l = (m = l & 15, g = l >> 4): A row m, B token m, k-slots (g, i) =
//      (sub-block 2g + (i >> 2), u = i & 3); C/D rows 4g + i, token m.  (An eight-wave layout of the same arithmetic — 16 x 32 per wave, 96 accumulators, two waves
//      per SIMD — lived here in round 5: 27.4 vs 25.4 ms per 512-token micro-batch; removed in round 6, profiles/r06_prefill_ceiling.txt.)

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
    // builder (as in the Q4_K kernel: the stream's own rows, one record group per wave): wave (rt, q = tp), lane (p = lane >> 5, r = (lane >> 2) & 7, el = lane & 3)
    // takes, per half step, chunk e = 4 H + el of row 8q + r and builds the pieces of MFMA lanes (8q + r, g = 2p + j), j = 0, 1: the ql dwords 2p, 2p + 1
    // (sub-blocks 2g / 2g+1 sit in the low (j = 0) or high (j = 1) nibbles) and the qh dword p of stream lane (r, e).  Eight consecutive lanes = two rows x
    // four fragments: their 16-byte stores fall on eight different bank quads (fragment pairs 2 x X_FR bytes apart).
    const int bp = lane >> 5, br = (lane >> 2) & 7, bel = lane & 3;
    const int rgq = 2 * (live ? rtg : rb * 4) + tp;
    const int rgc = rgq * 8 < a.nrows_pad ? rgq : rgq - 1;
    const uint8_t * wql = a.w + (size_t) rgc * nb * 1680 + (size_t) ((br * 8 + bel) * 16 + bp * 8);
    const uint8_t * wqh = a.w + (size_t) rgc * nb * 1680 + 1024 + (size_t) ((br * 8 + bel) * 8 + bp * 4);
    const uint8_t * phb = a.ph + (size_t) rb * nb * 8192 + (size_t) rt * 1024 + (size_t) ((2 * bp) * 256 + (8 * tp + br) * 16);
    uint2 ql[2]; uint32_t qh[2]; uint4 sc[2][2];
    auto load_set = [&](int hs, auto set_tag) {                // operands of half step hs = 2 ci + h
        constexpr int S = decltype(set_tag)::value;
        const int ci = hs >> 1, h = hs & 1;
        ql[S] = *(const uint2 *) (wql + (size_t) ci * 1680 + h * 64);
        qh[S] = *(const uint32_t *) (wqh + (size_t) ci * 1680 + h * 32);
        sc[S][0] = *(const uint4 *) (phb + (size_t) hs * 4096); sc[S][1] = *(const uint4 *) (phb + (size_t) hs * 4096 + 256);
    };
    const bamd_h2 k1056 = { (_Float16) -1056.f, (_Float16) -1056.f };
    bamd_h2u v0, v1, v2, v3;
    auto build_a = [&](const uint2 & q, uint32_t hq, int j) {
        // the two high bits of a quant sit at bits sh, sh + 1 (sub-block 2g) / sh + 2, sh + 3 (2g + 1) of their byte of hq (sh = 4 j) and belong at bits 4, 5:
        // a ROTATION of the dword (what wraps around lands outside the mask 0x30 of every byte), then one and-or
        const uint32_t sh = 4u * (uint32_t) j;
        const uint32_t uA = (__builtin_amdgcn_alignbit(hq, hq, (28u + sh) & 31u) & 0x30303030u) | ((q.x >> sh) & 0x0f0f0f0fu);
        const uint32_t uB = (__builtin_amdgcn_alignbit(hq, hq, (30u + sh) & 31u) & 0x30303030u) | ((q.y >> sh) & 0x0f0f0f0fu);
        bamd_h2u c;
        c.u = __builtin_amdgcn_perm(0x64646464u, uA, 0x04010400u); v0.h = c.h + k1056;      // (1024 + q) - 1056 = q - 32, exact
        c.u = __builtin_amdgcn_perm(0x64646464u, uA, 0x04030402u); v1.h = c.h + k1056;
        c.u = __builtin_amdgcn_perm(0x64646464u, uB, 0x04010400u); v2.h = c.h + k1056;
        c.u = __builtin_amdgcn_perm(0x64646464u, uB, 0x04030402u); v3.h = c.h + k1056;
    };
    auto build_b = [&](const uint4 & s, unsigned char * dst) {  // dst: the A_a fragment; A_l X_FR bytes behind it
        bamd_h2u sa0, sl0, sa1, sl1, x0, x1, x2, x3; sa0.u = s.x; sl0.u = s.y; sa1.u = s.z; sl1.u = s.w;
        x0.h = v0.h * sa0.h; x1.h = v1.h * sa0.h; x2.h = v2.h * sa1.h; x3.h = v3.h * sa1.h;
        *(uint4 *) dst = (uint4) { x0.u, x1.u, x2.u, x3.u };
        x0.h = v0.h * sl0.h; x1.h = v1.h * sl0.h; x2.h = v2.h * sl1.h; x3.h = v3.h * sl1.h;
        *(uint4 *) (dst + X_FR) = (uint4) { x0.u, x1.u, x2.u, x3.u };
    };
    unsigned char * afw = smem + X_AF0 + (rt * 4 + bel) * (2 * X_FR) + ((2 * bp) * 16 + 8 * tp + br) * 16;     // [row tile][e' = 0..3][a | l][X_FR]: MFMA lane (8 tp + br, 2 bp + j) at + j * 256
    const unsigned char * afr = smem + X_AF0 + rt * 8 * X_FR + lane * 16;
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
    for (int j = 0; j < 2; ++j) { build_a(ql[0], qh[0], j); build_b(sc[0][j], afw + j * 256); }
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
#define X_LDA(e_, p_) (*(const bamd_h8 *) (afr + H * X_AF_BYTES + (e_) * (2 * X_FR) + (p_) * X_FR))
#define X_LDB(e_, n_) (*(const bamd_h8 *) (bop + BLK * X_BLK + (n_) * (16 * BAMD_B16_REC) + (4 * H + (e_)) * 64))
        Aa[0] = X_LDA(0, 0); Al[0] = X_LDA(0, 1); Bq[0][0] = X_LDB(0, 0); Bq[0][1] = X_LDB(0, 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e + 1 < 4) { Aa[(e + 1) & 1] = X_LDA(e + 1, 0); Al[(e + 1) & 1] = X_LDA(e + 1, 1); Bq[(e + 1) & 1][0] = X_LDB(e + 1, 0); Bq[(e + 1) & 1][1] = X_LDB(e + 1, 1); }
            if ((e & 1) == 0) build_a(ql[NXT], qh[NXT], e >> 1);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                const bamd_f4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aa[e & 1], Bq[e & 1][n], z, 0, 0, 0);
                const bamd_f4 s2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[e & 1], Bq[e & 1][n], s1, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[n][4 * H + e][i] = fmaf(D[n][i], s2[i], acc[n][4 * H + e][i]);
            }
            if (e & 1) build_b(sc[NXT][e >> 1], afw + NXT * X_AF_BYTES + (e >> 1) * 256);
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

// ==== sixteen waves ===========================================================================================================================
// The eight-wave kernels above trade vector instructions for LDS traffic and end where the round-2 kernels were (two waves per SIMD: a step is a chain of
// dependent latencies).  This kernel — the default for Q4_K / Q5_K — puts four waves on every SIMD:
//   * a workgroup is SIXTEEN waves: four row tiles x four token tiles, ONE 16 x 16 tile per wave (48 accumulator registers, <= 118 VGPRs);
//     every A fragment is built once and read by four waves, every token record staged once and read by four;
//   * staging is three full-wave DMA instructions per wave and step, the same straight-line code on every wave: sources are fixed per-wave scalar pointers
//     plus per-lane offsets that advance by a per-wave stride (records 608 B, consumer headers 3 KiB, builder operands 4 KiB per super-block), destinations
//     per-wave constants; no clamping at the end of K (the last copies read one super-block past the tables — allocated with that slack — into a block
//     nobody reads again); the copies follow the MFMAs of e = 0..2 instead of opening the step (all sixteen waves' copies at once queue in the vector
//     memory path: every wave sat 500-1200 clocks in its issue stage);
//   * the builder's scale operands travel by DMA as well, two steps ahead, into the operand slot of the block that is being read (its previous content
//     was consumed one step earlier), so a wave's own loads are 8 bytes of nibbles per lane and step through a buffer descriptor (scalar offset);
//   * d_y sits inside the token record; the chain FMAs of e follow the MFMA of e + 2 (no MFMA -> VALU wait states); the epilogue stores 16 bytes per lane.
// 112 issued instructions per wave and step (the round-2 kernel: 365 per two tiles).  What that bought and what it did not: DESIGN 4c.
#define X3_BLK (X_BS_BYTES + X3_CHS + 4096 + 32)           /* records (38 KiB) | headers (3 KiB) | operands (4 KiB) | 32 bytes of zeros: 46 112 B */
#define X3_BLK0 (2 * X_AF_BYTES)
#define X3_LDS_BYTES (X3_BLK0 + 2 * X3_BLK)                /* 158 784 B */
#define X3_CH_OFF X_BS_BYTES
#define X3_PH_OFF (X_BS_BYTES + X3_CHS)
#define X3_Z_OFF (X3_PH_OFF + 4096)
// one DMA instruction: 1 KiB from (scalar base + per-lane offset) to the LDS address lds_dst (m0 is saved and restored around it: the compiler reserves the register)
__device__ __forceinline__ void x3_dma(const void * sbase, uint32_t voff, uint32_t lds_dst) { lds_dma16_s(sbase, voff, lds_dst); }
#ifndef X_TIMING
#define X_TIMING 0
#endif
#ifndef BAMD_PREFILL_CEILING
#define BAMD_PREFILL_CEILING 0
#endif
#if X_TIMING
#define X_T(i_) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i_] += now_ - tlast; tlast = now_; } while (0)
#else
#define X_T(i_) do { } while (0)
#endif
template <int EPI, bool Q5>
__global__ void __launch_bounds__(1024) matmul_mfma3_q4k_kernel(bamd_mma2_args a) {
    constexpr uint32_t RECB = Q5 ? BAMD_RECB_Q5K : BAMD_RECB_Q4K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), m = lane & 15, g = lane >> 4;
    const int rt = wave >> 2, tt = wave & 3;
    const int nb = a.K >> 8;
    const int rb = blockIdx.y, t0 = blockIdx.x * 64;
    const int rtg = rb * 4 + rt;
    const bool live = rtg * 16 < a.nrows_pad;
    const size_t b16 = BAMD_BLOB16_BYTES(nb);
    const uint32_t lds0 = (uint32_t) (size_t) (bamd_lds_vp) smem;
    // ---- stage plan.  A block = 45 KiB-instructions: k = 0..37 records (chunk 64 k + lane of 64 tokens x 38), 38..40 consumer headers, 41..44 builder
    //      operands.  Wave w issues k = w, 16 + w and k2 = 32 + w (w <= 12; waves 13..15 repeat the operand copies 41..43: same bytes, same place).
    auto rec_off = [&](int idx) { const int tok = idx / BAMD_B16_Q, q = idx - tok * BAMD_B16_Q; const int tg = t0 + tok < a.T ? t0 + tok : a.T - 1; return (uint32_t) ((size_t) tg * b16 + (size_t) q * 16); };
    const int k2 = wave <= 12 ? 32 + wave : 28 + wave;
    const bool rec2 = k2 <= 37, hdr2 = k2 >= 38 && k2 <= 40;
    uint32_t so0 = rec_off(tid), so1 = rec_off(1024 + tid);          // advance by one super-block per step (sources stay fixed scalar pointers)
    uint32_t so2 = rec2 ? rec_off(k2 * 64 + lane) : (uint32_t) (((hdr2 ? k2 - 38 : k2 - 41) * 64 + lane) * 16);
    // sources of the NEXT stage call (the first call stages super-block 0 and the operands of super-block 1)
    const uint8_t * p01 = a.blob16;
    const uint8_t * p2 = rec2 ? a.blob16 : hdr2 ? a.ch + (size_t) rb * nb * X3_CHS : a.ph + (size_t) rb * nb * 4096 + 4096;
    const uint32_t st2 = rec2 ? (uint32_t) BAMD_B16_REC : hdr2 ? (uint32_t) X3_CHS : 4096u;
    // destinations by the parity of the step that issues the copies: records / headers of ci + 1 -> the OTHER block, operands of ci + 2 -> the block in use
    const uint32_t blk_a = lds0 + X3_BLK0, blk_b = blk_a + X3_BLK;
    const uint32_t d0_even = blk_b + (uint32_t) wave * 1024u, d0_odd = blk_a + (uint32_t) wave * 1024u;
    const uint32_t d2_even = (k2 <= 40 ? blk_b : blk_a) + (uint32_t) k2 * 1024u, d2_odd = (k2 <= 40 ? blk_a : blk_b) + (uint32_t) k2 * 1024u;
#define X3_STAGE(PAR) do { x3_dma(p01, so0, (PAR) ? d0_odd : d0_even); x3_dma(p01, so1, ((PAR) ? d0_odd : d0_even) + 16384u); x3_dma(p2, so2, (PAR) ? d2_odd : d2_even); \
                           so0 += BAMD_B16_REC; so1 += BAMD_B16_REC; so2 += st2; } while (0)
    // ---- builder: wave (rt, tt) = record group q = tt & 1 of the row tile, pieces g = 2 gh + j (gh = tt >> 1, j = 0, 1): lane (r, e) holds dwords 2 gh, 2 gh + 1
    //      of stream lane (r, e) = the halves of MFMA lanes (8 q + r, 2 gh + j) of fragment e
    const int br = lane >> 3, be = lane & 7, bq = tt & 1, gh = tt >> 1;
    const int rgq = 2 * (live ? rtg : rb * 4) + bq;
    const int rgc = rgq * 8 < a.nrows_pad ? rgq : rgq - 1;            // a last tile of 8 (padded) rows: its first record group twice (rows 8..15 are never stored)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *) uniform_ptr(a.w + (size_t) rgc * nb * RECB), 0, 0x7fffffff, 0x00020000);
    const int vraw = lane * 16 + gh * 8, vqh = 1024 + lane * 4;
    uint2 raw[2]; uint32_t qh[2];
    auto load_set = [&](int ci, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(wrs, vraw, ci * (int) RECB, 0); raw[S] = make_uint2(v.x, v.y);
        if (Q5) qh[S] = __builtin_amdgcn_raw_buffer_load_b32(wrs, vqh, ci * (int) RECB, 0);
    };
    bamd_h2u c0, c1, c2, c3;
    const uint32_t qs0 = 4u * (uint32_t) gh;
    auto build_a = [&](uint32_t wq, uint32_t q, int j) {
        uint32_t lo = wq & 0x0f0f0f0fu, hi = Q5 ? (wq >> 4) & 0x0f0f0f0fu : wq & 0xf0f0f0f0u;      // Q4_K: the high nibbles stay in place (16 n; the scale operand is s / 16)
        if (Q5) { lo |= ((q >> (qs0 + 2 * j)) & 0x01010101u) << 4; hi |= ((q >> (qs0 + 2 * j + 1)) & 0x01010101u) << 4; }
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
    unsigned char * afw = smem + X_AF0 + (rt * 8 + be) * X_FR + ((2 * gh) * 16 + 8 * bq + br) * 16;
    const unsigned char * afr = smem + X_AF0 + rt * 8 * X_FR + lane * 16;
    const unsigned char * phl = smem + X3_BLK0 + X3_PH_OFF + rt * 1024 + (2 * gh) * 256 + (8 * bq + br) * 16;
    const unsigned char * tok = smem + X3_BLK0 + (size_t) (tt * 16 + m) * BAMD_B16_REC;                  // this lane's token record
    const unsigned char * bop = tok + g * 16;
    const unsigned char * bmn = tok + 512 + (Q5 ? g * 8 : 0);
    const unsigned char * chd = smem + X3_BLK0 + X3_CH_OFF + rt * X_CH4_RT + g * 16;                // d of rows 4g .. 4g+3; their dmin 64 bytes on
    const unsigned char * cmn = Q5 ? smem + X3_BLK0 + X3_CH_OFF + rt * X_CH4_RT + 128 + m * 32 + g * 8
                                   : (g == 0 ? smem + X3_BLK0 + X3_CH_OFF + rt * X_CH4_RT + 128 + m * 32 : smem + X3_BLK0 + X3_Z_OFF);
    bamd_f4 acc[8], accm[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int l = 0; l < 4; ++l) accm[l] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
    // prologue = the staging half of a step "-1" of odd parity: records / headers of super-block 0 into block a, operands of super-block 1 into block b;
    // the fragments of super-block 0 are built with operands straight from memory
    if (tid < 16) *(uint32_t *) (smem + X3_BLK0 + X3_Z_OFF + (tid >> 3) * X3_BLK + (tid & 7) * 4) = 0u;
    X3_STAGE(1);
    load_set(0, std::integral_constant<int, 0>());
    load_set(1, std::integral_constant<int, 1>());               // (no clamping at the end of K: the wave-stream copies are allocated with 4 KiB of slack, bamd_engine.cpp dev_alloc)
    {
        const uint8_t * p0 = a.ph + (size_t) rb * nb * 4096 + (size_t) rt * 1024 + (size_t) ((2 * gh) * 256 + (8 * bq + br) * 16);
        const uint4 s0 = *(const uint4 *) p0, s1 = *(const uint4 *) (p0 + 256);
        build_a(raw[0].x, Q5 ? qh[0] : 0u, 0); build_b(s0, afw);
        build_a(raw[0].y, Q5 ? qh[0] : 0u, 1); build_b(s1, afw + 256);
    }
    lds_dma_wait();
    __syncthreads();
#if X_TIMING
    unsigned long long tacc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tlast = __builtin_readcyclecounter();
#endif
    auto step = [&](const int ci, auto cur_tag) {
        constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
        X_T(0);
        float D[4], Dm[4];
        const bamd_f4 h0 = *(const bamd_f4 *) (chd + CUR * X3_BLK), h1 = *(const bamd_f4 *) (chd + CUR * X3_BLK + 64);
        const float ydv = *(const float *) (tok + CUR * X3_BLK + 560);
        D[0] = ydv * h0[0]; D[1] = ydv * h0[1]; D[2] = ydv * h0[2]; D[3] = ydv * h0[3];
        Dm[0] = (-ydv) * h1[0]; Dm[1] = (-ydv) * h1[1]; Dm[2] = (-ydv) * h1[2]; Dm[3] = (-ydv) * h1[3];
        {
            // The step opens with operand reads and MFMAs; the three staging copies and the nibble load of this wave follow the MFMAs of e = 0..3 one by
            // one (issued together at the top they queue behind the other fifteen waves' copies — the vector memory path takes 64 B per clock — and every
            // wave sits in its issue stage for 500-1200 clocks before its first MFMA: tools/prefill_phase.py); the fragment pieces of the next
            // super-block are built beside e = 4..7
            // the chain FMAs of e follow the MFMA of e + 2 (one iteration behind, the compiler pads every MFMA -> VALU read with s_nop 1..6: ten issue slots per step)
            bamd_h8 Aq[3], Bq[3]; bamd_f4 sp[2];
#if BAMD_PREFILL_CEILING
            bamd_f4 csum = { 0.f, 0.f, 0.f, 0.f }, csum2 = { 0.f, 0.f, 0.f, 0.f };
            const unsigned char * afr2 = smem + X_AF0 + (rt ^ 1) * 8 * X_FR + lane * 16;
            (void) afr2; (void) csum2;
#endif
#define X_LDA(e_) (*(const bamd_h8 *) (afr + CUR * X_AF_BYTES + (e_) * X_FR))
#define X_LDB(e_) (*(const bamd_h8 *) (bop + CUR * X3_BLK + (e_) * 64))
            Aq[0] = X_LDA(0); Bq[0] = X_LDB(0); Aq[1] = X_LDA(1); Bq[1] = X_LDB(1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e + 2 < 8) { Aq[(e + 2) % 3] = X_LDA(e + 2); Bq[(e + 2) % 3] = X_LDB(e + 2); }
                if (e >= 4 && (e & 1) == 0) build_a(e == 4 ? raw[NXT].x : raw[NXT].y, Q5 ? qh[NXT] : 0u, (e >> 1) & 1);
                const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
#if BAMD_PREFILL_CEILING
                // TIMING-ONLY ceiling builds (tools/prefill_ceiling.sh; never shipped, results are garbage): what the exact arithmetic's EIGHT separate per-lane sums per
                // super-block cost.  1: the eight MFMAs of a step accumulate into ONE accumulator (K = 256) and ONE chain FMA per super-block and tile follows — same
                // operand traffic, same staging, same fragment build.  2: the same, and every B operand read serves the MFMAs of TWO row tiles (this wave's and its
                // neighbour's A fragments: 1.5 KB of LDS reads per MFMA instead of 2; the launcher halves the row blocks so that the MFMA count stays what it was)
                csum = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[e % 3], Bq[e % 3], e == 0 ? z : csum, 0, 0, 0);
                if (BAMD_PREFILL_CEILING == 2) csum2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const bamd_h8 *) (afr2 + CUR * X_AF_BYTES + e * X_FR), Bq[e % 3], e == 0 ? z : csum2, 0, 0, 0);
                const bamd_f4 si = z;
#else
                const bamd_f4 si = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[e % 3], Bq[e % 3], z, 0, 0, 0);
#endif
                if (e == 0) x3_dma(p01, so0, CUR ? d0_odd : d0_even);
                if (e == 1) x3_dma(p01, so1, (CUR ? d0_odd : d0_even) + 16384u);
                if (e == 2) { x3_dma(p2, so2, CUR ? d2_odd : d2_even); so0 += BAMD_B16_REC; so1 += BAMD_B16_REC; so2 += st2; }
                if (e == 3) load_set(ci + 2, std::integral_constant<int, CUR>());
#if !BAMD_PREFILL_CEILING
                if (e > 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[e - 2][i] = fmaf(D[i], sp[e & 1][i], acc[e - 2][i]);
                }
#endif
                if (e >= 4 && (e & 1)) build_b(*(const uint4 *) (phl + NXT * X3_BLK + ((e >> 1) & 1) * 256), afw + NXT * X_AF_BYTES + ((e >> 1) & 1) * 256);
                sp[e & 1] = si;
                __builtin_amdgcn_sched_barrier(0);
                if (e == 3) X_T(1);
            }
#if BAMD_PREFILL_CEILING
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[0][i] = fmaf(D[i], csum[i], acc[0][i]); if (BAMD_PREFILL_CEILING == 2) acc[1][i] = fmaf(D[i], csum2[i], acc[1][i]); }
            (void) sp;
#else
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[6][i] = fmaf(D[i], sp[0][i], acc[6][i]); acc[7][i] = fmaf(D[i], sp[1][i], acc[7][i]); }
#endif
#undef X_LDA
#undef X_LDB
        }
        X_T(2);
        if (Q5) {
            union { uint2 u; bamd_h4 h; } av, bv; av.u = *(const uint2 *) (cmn + CUR * X3_BLK); bv.u = *(const uint2 *) (bmn + CUR * X3_BLK);
            const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
            const bamd_f4 pm = __builtin_amdgcn_mfma_f32_16x16x16f16(av.h, bv.h, z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float t = Dm[i] * pm[i]; accm[0][i] = accm[0][i] + t; }
        } else {
            const uint4 ma = *(const uint4 *) (cmn + CUR * X3_BLK), mb = *(const uint4 *) (cmn + CUR * X3_BLK + 16);
            const uint4 sfa = *(const uint4 *) (bmn + CUR * X3_BLK), sfb = *(const uint4 *) (bmn + CUR * X3_BLK + 16);
            const uint2 al[4] = { { ma.x, ma.y }, { ma.z, ma.w }, { mb.x, mb.y }, { mb.z, mb.w } };
            const uint2 bl[4] = { { sfa.x, sfa.y }, { sfa.z, sfa.w }, { sfb.x, sfb.y }, { sfb.z, sfb.w } };
            bamd_f4 pm[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                union { uint2 u; bamd_h4 h; } av4, bv4; av4.u = al[l]; bv4.u = bl[l];
                const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                pm[l] = __builtin_amdgcn_mfma_f32_16x16x16f16(av4.h, bv4.h, z, 0, 0, 0);
            }
#pragma unroll
            for (int l = 0; l < 4; ++l) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accm[l][i] = fmaf(Dm[i], pm[l][i], accm[l][i]);
            }
        }
        X_T(3);
        lds_dma_wait();
        X_T(4);
        __syncthreads();
        X_T(5);
    };
    for (int ci = 0; ci < nb; ci += 2) {
        step(ci, std::integral_constant<int, 0>());
        if (ci + 1 < nb) step(ci + 1, std::integral_constant<int, 1>());
    }
#undef X3_STAGE
#if X_TIMING
    if (a.dbg && blockIdx.x == 0 && blockIdx.y == (gridDim.y >> 1) && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.dbg[wave * 8 + i] = tacc[i];
        a.dbg[wave * 8 + 6] = (unsigned long long) nb;
    }
#endif
    if (!live) return;
    // hsum_float_8 over e and the acc_m folds, in the reference's order (finish_row), then the epilogue: rows 4g .. 4g+3 of token t are 16 consecutive bytes
    const int t = t0 + tt * 16 + m;
    float val[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = ((acc[0][i] + acc[4][i]) + (acc[2][i] + acc[6][i])) + ((acc[1][i] + acc[5][i]) + (acc[3][i] + acc[7][i]));
        const float mm = Q5 ? accm[0][i] : (accm[0][i] + accm[2][i]) + (accm[1][i] + accm[3][i]);
        val[i] = v + mm;
    }
    const int row0 = rtg * 16 + 4 * g;
    if (t < a.T && row0 < a.nrows) {
        const size_t o = (size_t) t * a.ldo + row0;
        if (row0 + 3 < a.nrows && (a.ldo & 3) == 0) {
            bamd_f4 y = { val[0], val[1], val[2], val[3] };
            if (EPI != BAMD_EPI_STORE) {
                const bamd_f4 r = *(const bamd_f4 *) (a.res + o);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = EPI == BAMD_EPI_ADD ? val[i] + r[i] : v_silu(r[i]) * val[i];
            }
            *(bamd_f4 *) (a.out + o) = y;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (row0 + i < a.nrows) a.out[o + i] = EPI == BAMD_EPI_ADD ? val[i] + a.res[o + i] : EPI == BAMD_EPI_SILU_MUL ? v_silu(a.res[o + i]) * val[i] : val[i];
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------------------
static inline int x_row_blocks(int nrows_pad) { return (nrows_pad + 63) / 64; }
static unsigned long long * g_prefill_dbg = nullptr;      // -DX_TIMING builds: where the kernels of the next launches leave their phase clocks (bamd_prefill_dbg)
extern "C" __attribute__((visibility("default"))) void bamd_prefill_dbg(void * dev_buf) { g_prefill_dbg = (unsigned long long *) dev_buf; }
// bytes of the side table of a K-quant matrix [nrows_pad][K]: builder part first, the consumer part behind it (both 16-byte aligned)
size_t bamd_prefill_aux_bytes(int type, int nrows_pad, int K) {
    if ((type != BAMD_Q4_K && type != BAMD_Q5_K && type != BAMD_Q6_K) || (K & 255) || (nrows_pad & 7)) return 0;
    const size_t nb = (size_t) (K >> 8), rbk = (size_t) x_row_blocks(nrows_pad);
    // + two super-blocks of slack behind each table: the sixteen-wave kernel's last two staging calls read past the end of K
    return type == BAMD_Q6_K ? (rbk * nb + 2) * 8192 + (rbk * nb + 2) * 4 * X_CH6_RT : (rbk * nb + 2) * 4096 + (rbk * nb + 2) * X3_CHS;
}
static inline size_t x_ph_bytes(int type, int nrows_pad, int K) { return ((size_t) x_row_blocks(nrows_pad) * (size_t) (K >> 8) + 2) * (type == BAMD_Q6_K ? 8192 : 4096); }
void bamd_launch_prefill_aux(const void * w_stream, int type, int nrows_pad, int K, void * aux, hipStream_t s) {
    const int nb = K >> 8, nrt = x_row_blocks(nrows_pad) * 4;
    uint8_t * ph = (uint8_t *) aux, * ch = ph + x_ph_bytes(type, nrows_pad, K);
    const dim3 grid(nb, nrt);
    if (type == BAMD_Q6_K)      hipLaunchKernelGGL(prefill_aux_q6k_kernel, grid, dim3(64), 0, s, (const uint8_t *) w_stream, nrows_pad, nb, ph, ch);
    else if (type == BAMD_Q5_K) hipLaunchKernelGGL((prefill_aux_q4k_kernel<true>), grid, dim3(64), 0, s, (const uint8_t *) w_stream, nrows_pad, nb, ph, ch);
    else                        hipLaunchKernelGGL((prefill_aux_q4k_kernel<false>), grid, dim3(64), 0, s, (const uint8_t *) w_stream, nrows_pad, nb, ph, ch);
}
// 1 when the current device takes the matrix-core kernels' launches (158 784 B of dynamic LDS at 1024 threads): asked once per model load, before any side table is
// built (a rejected launch would otherwise leave stale output behind a success code)
int bamd_prefill_mfma_supported(void) {
    int dev = 0, lds = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void) hipGetLastError(); return 0; }
    if ((size_t) lds < (size_t) X3_LDS_BYTES || (size_t) lds < (size_t) X_LDS_BYTES) return 0;
    if (hipFuncSetAttribute((const void *) matmul_mfma3_q4k_kernel<BAMD_EPI_STORE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) X3_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void *) matmul_mfma2_q6k_kernel<BAMD_EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) X_LDS_BYTES) != hipSuccess) { (void) hipGetLastError(); return 0; }
    return 1;
}
int bamd_launch_matmul_mfma2(const void * w_stream, const void * aux, int type, int nrows, int nrows_pad, int K, const void * blob16, int T, float * out, const float * res,
                             int epi, int ldo, hipStream_t s) {
    if ((type != BAMD_Q4_K && type != BAMD_Q5_K && type != BAMD_Q6_K) || (nrows_pad & 7) || (K & 255) || !aux) return 1;
    if (epi != BAMD_EPI_STORE && epi != BAMD_EPI_ADD && epi != BAMD_EPI_SILU_MUL) return 1;
    if ((epi != BAMD_EPI_STORE) != (res != nullptr)) return 1;
    bamd_mma2_args a; a.w = (const uint8_t *) w_stream; a.ph = (const uint8_t *) aux; a.ch = a.ph + x_ph_bytes(type, nrows_pad, K);
    a.out = out; a.res = res; a.blob16 = (const uint8_t *) blob16; a.K = K; a.T = T; a.nrows = nrows; a.nrows_pad = nrows_pad; a.ldo = ldo;
    a.dbg = g_prefill_dbg;
    const dim3 grid((T + 63) / 64, x_row_blocks(nrows_pad));
#define X_LAUNCH(KERNEL, ...) do { \
        if (epi == BAMD_EPI_ADD)           hipLaunchKernelGGL((KERNEL<BAMD_EPI_ADD __VA_ARGS__>),      grid, dim3(512), X_LDS_BYTES, s, a); \
        else if (epi == BAMD_EPI_SILU_MUL) hipLaunchKernelGGL((KERNEL<BAMD_EPI_SILU_MUL __VA_ARGS__>), grid, dim3(512), X_LDS_BYTES, s, a); \
        else                               hipLaunchKernelGGL((KERNEL<BAMD_EPI_STORE __VA_ARGS__>),    grid, dim3(512), X_LDS_BYTES, s, a); } while (0)
    const dim3 grid3(grid.x, BAMD_PREFILL_CEILING == 2 ? (grid.y + 1) / 2 : grid.y);     // (ceiling build 2: every workgroup issues the MFMAs of two; timing only)
#define X3_LAUNCH(KERNEL, ...) do { \
        if (epi == BAMD_EPI_ADD)           hipLaunchKernelGGL((KERNEL<BAMD_EPI_ADD __VA_ARGS__>),      grid3, dim3(1024), X3_LDS_BYTES, s, a); \
        else if (epi == BAMD_EPI_SILU_MUL) hipLaunchKernelGGL((KERNEL<BAMD_EPI_SILU_MUL __VA_ARGS__>), grid3, dim3(1024), X3_LDS_BYTES, s, a); \
        else                               hipLaunchKernelGGL((KERNEL<BAMD_EPI_STORE __VA_ARGS__>),    grid3, dim3(1024), X3_LDS_BYTES, s, a); } while (0)
    if (type == BAMD_Q6_K)      X_LAUNCH(matmul_mfma2_q6k_kernel);
    else if (type == BAMD_Q5_K) X3_LAUNCH(matmul_mfma3_q4k_kernel, , true);
    else                        X3_LAUNCH(matmul_mfma3_q4k_kernel, , false);
#undef X3_LAUNCH
#undef X_LAUNCH
    return 0;
}
