// bamd_attention_mfma.hip — batched-prefill attention on the matrix cores, bit for bit the reference's arithmetic.
//
// v_mfma_f32_16x16x4_f32 computes D = C + a0 b0 + a1 b1 + a2 b2 + a3 b3 as the SEQUENTIAL fmaf chain fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))
// — measured on the MI355X against every other candidate order on 2^20 outputs (tools/mfma_f32_probe.hip) — and that is exactly the shape of the
// reference's f32 chains in the attention of a micro-batch (T > 1):
//   scores  ggml_vec_dot_f16 (cpp/ggml/src/ggml.c:2038-2079): 4 accumulators x 8 SIMD lanes, accumulator (j, e) = the chain over the steps s of
//           k[32 s + 8 j + e] * q[32 s + 8 j + e] (f16 operands widened to f32: exact products); head_dim 128 = FOUR steps = one MFMA per (j, e)
//           with A = the K rows of 16 positions, B = the q vectors of 16 (token, head) columns, C = 0; then the reference's reduction tree on the
//           32 result tiles, element-wise;
//   P.V     tinyBLAS (cpp/ggml/src/llamafile/sgemm.cpp:405-431): for each SIMD lane e a sequential chain over l of V^T[d][8 l + e] * p[8 l + e]
//           (V f16 widened): four steps of the chain per MFMA, chained through C, eight accumulator tiles (one per e) per 16 rows d; then the
//           reference's horizontal sum across e, element-wise.
// The softmax in between is the reference's (8-wide f32 partial sums in its tree, double total, guard for the summation order: bamd_device.h).
// One workgroup = one KV head x 16 (token, query head) columns (16 / gq tokens x the gq heads that share the KV head): every K row and V^T row
// fetched serves 16 columns, the matrix pipe does the multiply-adds (the VALU kernel, attn_batch_kernel, spent ~3 vector instructions per
// multiply-add and was 55 % of a long prompt), and the workgroup -> KV head mapping keeps one head's K / V in one XCD's L2.
// head_dim 128; the score / probability rows of the 16 columns live in LDS up to BAMD_AM_MAXPOS cached positions, in a global scratch block beyond (LONG);
// other shapes: attn_batch_kernel.
#include "bamd_device.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define BAMD_AM_NW 4                           /* waves per workgroup: FOUR, at <= 256 VGPRs and <= 56 KB of LDS, so that TWO workgroups share a CU — one workgroup is a chain of
                                                  dependent phases (q / K requests, scores, softmax, V requests, P.V: ~19 us of latency even for a short sequence, measured
                                                  phase by phase), and only a second resident workgroup fills the matrix pipe while the first one waits */
#define BAMD_AM_DT (8 / BAMD_AM_NW)            /* 16-row tiles of V^T (of the 128 rows of a head) per wave in the P.V pass */
#define BAMD_AM_VROW 144                       /* bytes per staged V^T row (64 positions x f16 = 128 B, padded: 36 dwords -> 16 rows on 16 distinct banks) */
#define BAMD_AM_QBYTES (16 * 128 * 2)          /* q of the 16 columns, f16, chain-major */
#define BAMD_AM_VBYTES (BAMD_AM_NW * BAMD_AM_DT * 16 * BAMD_AM_VROW)   /* one staged block per wave and row tile */
#define BAMD_AM_RBYTES 1024                    /* column maxima per wave + 1 / sum per column */
#define BAMD_AM_MAXPOS 512                     /* positions whose score rows stay in LDS (64 B each: 32 KB); beyond: global scratch + chunks (LONG) */
#define BAMD_AM_CHUNK 512                      /* LONG: positions staged through LDS at a time in the P.V pass */
#ifndef BAMD_AM_KD
#define BAMD_AM_KD 2                           /* tiles of 16 K rows in flight per wave (pass 1) */
#endif

// score / probability storage [position][column] in LDS, laid out for the three access patterns (64 banks of 4 bytes):
//   row(p) = the positions of a group of 32 reordered as [p & 7][(p >> 3) & 3]: the four positions p, p + 8, p + 16, p + 24 that the four k-groups of a
//            P.V MFMA read together sit in rows with different (row & 3) = different quarters of the banks;
//   column n of a row is stored at n ^ ((row >> 2) & 15): consecutive positions of one column (the softmax pass) hit distinct banks.
__device__ __forceinline__ int am_sidx(int p, int n) {
    const int row = (p & ~31) | ((p & 7) << 2) | ((p >> 3) & 3);
    return row * 16 + (n ^ ((row >> 2) & 15));
}
// LONG: the scratch block of a workgroup holds its 16 score rows as [position / 4][column][4]: the four consecutive positions a lane of pass 1 produces for
// its column are one 16-byte store, and the 16 columns of a position quad are 256 contiguous bytes
__device__ __forceinline__ size_t am_gidx(int p, int n) { return ((size_t) (p >> 2) * 16 + n) * 4 + (p & 3); }
__device__ __forceinline__ float h2f_lo(uint32_t w) { return __half2float(__ushort_as_half((unsigned short) (w & 0xffffu))); }
__device__ __forceinline__ float h2f_hi(uint32_t w) { return __half2float(__ushort_as_half((unsigned short) (w >> 16))); }

// LONG: more positions than BAMD_AM_MAXPOS — the scores / exp values of the 16 columns live in a global scratch block of this workgroup
// ([16 columns][ld] f32: written by pass 1, exponentiated in place by pass 2) and pass 3 walks them in chunks of BAMD_AM_CHUNK positions staged
// through LDS; the P.V chains simply continue from chunk to chunk.
template <int GQ, bool LONG>
__global__ void __launch_bounds__(64 * BAMD_AM_NW) __attribute__((amdgpu_waves_per_eu(2, 2))) attn_batch_mfma_kernel(bamd_attn_args a, int T, int dbg_exit) {
    constexpr int hd = 128, L = 16, TT = 16 / GQ, NW = BAMD_AM_NW, NT = 64 * NW, DT = BAMD_AM_DT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short * q16 = (unsigned short *) smem;                            // [16 columns][128], chain-major (kperm)
    unsigned char * vst = smem + BAMD_AM_QBYTES;                               // [NW waves][DT row tiles][16 rows][BAMD_AM_VROW]
    float * cmaxs = (float *) (smem + BAMD_AM_QBYTES + BAMD_AM_VBYTES);        // [NW waves][16 columns] score maxima; fsv [16]: 1 / sum of every column
    float * fsv = cmaxs + 128;
    float * S = (float *) (smem + BAMD_AM_QBYTES + BAMD_AM_VBYTES + BAMD_AM_RBYTES);   // [positions][16]: scores, then exp values (the probabilities are formed in pass 3)
    const int sld = a.lds_ld ? a.lds_ld : a.n_ctx;                              // LONG: floats per column of this workgroup's scratch block
    float * scr = LONG ? a.batch_scratch + (size_t) blockIdx.x * 16 * sld : nullptr;
    const bamd_step_state * st = a.st;
    const int Hkv = a.Hkv, Ekv = Hkv * hd, n_ctx = a.n_ctx;
    const int hk = (int) blockIdx.x % Hkv, tile = (int) blockIdx.x / Hkv;      // consecutive workgroups: different KV heads (= different XCDs for Hkv = 8)
    const int t0 = tile * TT;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int P0 = a.batch_pos0p1 > 0 ? a.batch_pos0p1 - 1 : st->pos;          // (the host knows the micro-batch's first position: no dependent load in front of everything)
    const int tlast = (t0 + TT - 1 < T ? t0 + TT - 1 : T - 1);
    const int pmax = P0 + tlast;                                               // highest position any column of this tile attends
    int npos = (pmax + 1 + 63) & ~63; npos = npos < n_ctx ? npos : n_ctx;       // positions past a column's own are masked: exact no-ops (attn_batch_kernel)
    const int mrow = lane & 15, kq = lane >> 4;                                // MFMA operand roles of this lane: A[m = mrow][k = kq], B[k = kq][n = mrow]
    // a ring of BAMD_AM_KD tiles of K rows per wave in flight (64 bytes per lane and tile: the 8-byte piece of chain steps 4 kq .. 4 kq + 3 of every lane e, kperm order), requested
    // unconditionally (a tile past the end: the last one again) so that the waits stay counted
    const unsigned short * kbase = a.kc + (size_t) hk * hd + (kq >> 1) * BAMD_KGRP + (kq & 1) * 4;      // chain steps l = 4 kq .. 4 kq + 3 of every lane e (kperm)
    const int ntile = npos >> 4;
    uint2 ring[BAMD_AM_KD][8];
    auto kload = [&](uint2 (&dst)[8], int pt) {
        const int row = (pt < ntile ? pt : ntile - 1) * 16 + mrow;
        const unsigned short * kp = kbase + (size_t) row * Ekv;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = *(const uint2 *) (kp + e * 8);
    };
    // ---- RoPE of the 16 query vectors -> f16, chain-major (rope_heads' arithmetic, ggml.c:14130-14143).  16 x 64 pairs: 1024 / NT per thread, all
    //      requested (8-byte loads) before anything else, the K ring right behind them ----
    constexpr int RP = 1024 / NT;
    float2 qx[RP], cs[RP];
#pragma unroll
    for (int it = 0; it < RP; ++it) {
        const int i = tid + it * NT, n = i >> 6, p = i & 63;
        int tok = t0 + n / GQ; tok = tok < T ? tok : T - 1;                     // a tile past the batch end repeats the last token (never stored)
        const int h = hk * GQ + n % GQ;
        qx[it] = *(const float2 *) (a.q + (size_t) tok * a.ld_qkv + (size_t) h * hd + 2 * p);
        cs[it] = *(const float2 *) (a.rope + (size_t) (P0 + tok) * hd + 2 * p);
    }
#pragma unroll
    for (int d = 0; d < BAMD_AM_KD; ++d) kload(ring[d], wave + NW * d);
#pragma unroll
    for (int it = 0; it < RP; ++it) {
        const int i = tid + it * NT, n = i >> 6, p = i & 63;
        const float x0 = qx[it].x, x1 = qx[it].y, c = cs[it].x, sn = cs[it].y;
        const float u0 = x0 * c, u1 = x1 * sn, u2 = x0 * sn, u3 = x1 * c;
        q16[n * hd + kperm(2 * p, L)] = f2h(u0 - u1); q16[n * hd + kperm(2 * p + 1, L)] = f2h(u2 + u3);
    }
    __syncthreads();
    if (dbg_exit == 1) return;
    // ================= pass 1: scores (and the running maximum of every column) =================
    {
        // B fragments, once: column mrow, elements 32 s + 8 j + e for s = kq — stored at (kq >> 1) * 64 + e * 8 + (kq & 1) * 4 + j (kperm): four consecutive halves per e
        float B[8][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint2 w = *(const uint2 *) (q16 + mrow * hd + (kq >> 1) * BAMD_KGRP + e * 8 + (kq & 1) * 4);
            B[e][0] = h2f_lo(w.x); B[e][1] = h2f_hi(w.x); B[e][2] = h2f_lo(w.y); B[e][3] = h2f_hi(w.y);
        }
        const int pcol = P0 + ((t0 + mrow / GQ) < T ? (t0 + mrow / GQ) : T - 1);   // the position of this lane's column (D layout: column = lane & 15 as well)
        float cmax = -INFINITY;
        for (int pt0 = wave; pt0 < ntile; pt0 += NW * BAMD_AM_KD) {
#pragma unroll
            for (int d = 0; d < BAMD_AM_KD; ++d) {
                const int pt = pt0 + NW * d;
                if (pt < ntile) {                                              // wave-uniform; no request inside (short sequences have fewer tiles than ring turns)
                    f32x4_t Se[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const f32x4_t z = { 0.f, 0.f, 0.f, 0.f };
                        const f32x4_t d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(h2f_lo(ring[d][e].x), B[e][0], z, 0, 0, 0);
                        const f32x4_t d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(h2f_hi(ring[d][e].x), B[e][1], z, 0, 0, 0);
                        const f32x4_t d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(h2f_lo(ring[d][e].y), B[e][2], z, 0, 0, 0);
                        const f32x4_t d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(h2f_hi(ring[d][e].y), B[e][3], z, 0, 0, 0);
                        const f32x4_t s02 = d0 + d2, s13 = d1 + d3;             // GGML_F16_VEC_REDUCE: sum[0] += sum[2]; sum[1] += sum[3]; sum[0] += sum[1]
                        Se[e] = s02 + s13;
                    }
                    // the 8-lane horizontal sum of ggml_vec_dot_f16 (hsum8_vecdot): (lo + hi), then two hadd_ps
                    const f32x4_t t0_ = Se[0] + Se[4], t1_ = Se[1] + Se[5], t2_ = Se[2] + Se[6], t3_ = Se[3] + Se[7];
                    const f32x4_t u0 = t0_ + t1_, u2 = t2_ + t3_;
                    const f32x4_t sc = u0 + u2;
                    float vv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int p = pt * 16 + 4 * kq + r;                    // D layout: register r of lane l = row 4 (l / 16) + r, column l % 16
                        const float v = p <= pcol ? sc[r] : -INFINITY;         // KQ_mask (llama.cpp:14152-14200)
                        if (!LONG) S[am_sidx(p, mrow)] = v;
                        vv[r] = v;
                        cmax = v > cmax ? v : cmax;
                    }
                    if (LONG) *(float4 *) (scr + am_gidx(pt * 16 + 4 * kq, mrow)) = make_float4(vv[0], vv[1], vv[2], vv[3]);   // four consecutive positions of column mrow
                }
                kload(ring[d], pt + NW * BAMD_AM_KD);
            }
        }
        // column maxima: across the four k-groups of the wave, then across the waves through LDS
        { const float o = __shfl_xor(cmax, 16); cmax = o > cmax ? o : cmax; }
        { const float o = __shfl_xor(cmax, 32); cmax = o > cmax ? o : cmax; }
        if (lane < 16) cmaxs[wave * 16 + lane] = cmax;
    }
    if (dbg_exit == 2) return;
    // the first blocks of this wave's V^T rows are requested now: they land while the softmax pass runs.  Staging: lane i fetches 16 bytes of row (i >> 3)
    // [+ 8] of each of the wave's DT row tiles, at byte (i & 7) * 16 of the 128-byte block; two blocks in flight in named registers
    const int srow = lane >> 3, sby = (lane & 7) * 16;
    const unsigned char * gv = (const unsigned char *) (a.vc + (size_t) (hk * hd + wave * DT * 16 + srow) * n_ctx) + sby;
    const size_t row8 = (size_t) 8 * n_ctx * 2, tile16 = (size_t) 16 * n_ctx * 2;        // bytes between rows r and r + 8, between row tiles
    const int nblk = npos >> 6;
    static_assert(DT == 2, "the V^T ring is written out for two row tiles per wave");
    uint4 v0a, v0b, v0c, v0d, v1a, v1b, v1c, v1d;                              // (named registers: an array ring ends up in scratch memory)
#define BAMD_AM_VLOAD(A_, B_, C_, D_, blk_) do { const int bb_ = ((blk_) < nblk ? (blk_) : nblk - 1) * 128; \
        A_ = *(const uint4 *) (gv + bb_); B_ = *(const uint4 *) (gv + row8 + bb_); C_ = *(const uint4 *) (gv + tile16 + bb_); D_ = *(const uint4 *) (gv + tile16 + row8 + bb_); } while (0)
    BAMD_AM_VLOAD(v0a, v0b, v0c, v0d, 0); BAMD_AM_VLOAD(v1a, v1b, v1c, v1d, 1);
    __syncthreads();
    // ================= pass 2: softmax, 16 / NW columns per wave (64 NW / 16 lanes each), ggml.c:13682-13778 + :2619-2671 =================
    {
        constexpr int LPC = 4 * NW;                                            // lanes per column: 16 (a DPP row: two groups of eight positions)
        const int n = (16 / NW) * wave + lane / LPC, pl = lane % LPC, qb = lane - pl;
        const float scale = a.kq_scale;
        float smax = cmaxs[n];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) { const float o = cmaxs[w2 * 16 + n]; smax = o > smax ? o : smax; }
        const float mx = smax * scale;                                         // max_i (s_i * scale): the product is monotonic in s (scale > 0)
        double sum = 0.0;
        float * colv = scr;
        for (int p = pl; p < npos; p += LPC) {                                 // npos % 64 == 0: every 8-lane group is all-active
            const float w = (LONG ? colv[am_gidx(p, n)] : S[am_sidx(p, n)]) * scale;
            const float val = v_expf(w - mx);
            if (LONG) colv[am_gidx(p, n)] = val; else S[am_sidx(p, n)] = val;
            const float c = hsum8_tinyblas(val);                               // the reference's 8-wide partial sum (same tree shape), valid in lane & 7 == 0
            if ((lane & 7) == 0) sum += (double) c;
        }
        // the partial sums of the column's lanes in a fixed order; the reference's order is sequential over the 8-groups: f32_rounding_safe
        // decides whether the order can matter for (float) (1 / sum), and if it can one lane redoes the sum in the reference's order
        double tot = __shfl(sum, qb);
#pragma unroll
        for (int g = 8; g < LPC; g += 8) tot += __shfl(sum, qb + g);
        double rs = 1.0 / tot;
        if (!f32_rounding_safe(rs, BAMD_F64_GUARD_ULPS(npos / 8))) {           // uniform per column; rare
            double sq = 0.0;
            if (LONG) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }   // the column's own stores, then one lane's loads
            if (pl == 0) {
                for (int p = 0; p < npos; p += 8) {
                    float x0, x1, x2, x3, x4, x5, x6, x7;
                    if (LONG) { const float4 qa = *(const float4 *) (colv + am_gidx(p, n)), qb = *(const float4 *) (colv + am_gidx(p + 4, n));
                                x0 = qa.x; x1 = qa.y; x2 = qa.z; x3 = qa.w; x4 = qb.x; x5 = qb.y; x6 = qb.z; x7 = qb.w; }
                    else { x0 = S[am_sidx(p, n)]; x1 = S[am_sidx(p + 1, n)]; x2 = S[am_sidx(p + 2, n)]; x3 = S[am_sidx(p + 3, n)];
                           x4 = S[am_sidx(p + 4, n)]; x5 = S[am_sidx(p + 5, n)]; x6 = S[am_sidx(p + 6, n)]; x7 = S[am_sidx(p + 7, n)]; }
                    const float a0 = x0 + x4, a1 = x1 + x5, a2 = x2 + x6, a3 = x3 + x7;
                    const float b0 = a0 + a2, b1 = a1 + a3;
                    sq += (double) (b0 + b1);
                }
            }
            rs = 1.0 / __shfl(sq, qb);
        }
        if (pl == 0) fsv[n] = (float) rs;                                       // the probabilities are formed where they are used: p = e * fs in pass 3
    }
    __syncthreads();
    if (dbg_exit == 3) return;
    // ================= pass 3: P.V — wave w owns the row tiles DT w .. DT w + DT - 1 (16 rows of V^T each), all 16 columns =================
    {
        unsigned char * vs = vst + wave * (DT * 16 * BAMD_AM_VROW);
        const float fs = fsv[mrow];                                            // ggml_vec_scale_f32 by (float) (1 / sum): one multiply per probability
        // where this lane's B operands sit (am_sidx of p = 64 b + 8 (l0 + kq) + e, column mrow): row = 64 b + 32 (l0 / 4) + 4 e + kq and the column
        // swizzle (8 (l0 / 4) + e) & 15 do not depend on b — sixteen lane constants, everything else is an immediate offset of the DS read
        int bcol[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) bcol[c] = kq * 16 + (mrow ^ c);
        f32x4_t acc[DT][8];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[dt][e] = (f32x4_t) { 0.f, 0.f, 0.f, 0.f };
#define BAMD_AM_VSTEP(A_, B_, C_, D_, b_) do { \
            const int b = (b_); \
            if (b < cb1) {                                                     /* wave-uniform; no request inside */ \
                /* (DS operations of one wave execute in order: these stores land behind the previous block's reads) */ \
                *(uint4 *) (vs + srow * BAMD_AM_VROW + sby) = A_; *(uint4 *) (vs + (srow + 8) * BAMD_AM_VROW + sby) = B_; \
                *(uint4 *) (vs + (16 + srow) * BAMD_AM_VROW + sby) = C_; *(uint4 *) (vs + (16 + srow + 8) * BAMD_AM_VROW + sby) = D_; \
                const float * Sb = S + (b - cb0) * 64 * 16; \
                const unsigned short * vr0 = (const unsigned short *) (vs + mrow * BAMD_AM_VROW) + kq, * vr1 = (const unsigned short *) (vs + (16 + mrow) * BAMD_AM_VROW) + kq; \
                _Pragma("unroll") for (int l0 = 0; l0 < 8; l0 += 4) { \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) { \
                        const float bv = Sb[(32 * (l0 / 4) + 4 * e) * 16 + bcol[(8 * (l0 / 4) + e) & 15]] * fs;   /* B[k = kq][n = mrow]: p[64 b + 8 (l0 + kq) + e] of column mrow */ \
                        const float av0 = __half2float(__ushort_as_half(vr0[8 * e + l0])), av1 = __half2float(__ushort_as_half(vr1[8 * e + l0]));   /* A[m = mrow][k = kq]: V^T[d][64 b + 8 (l0 + kq) + e] sits at 8 e + l0 + kq */ \
                        acc[0][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv, acc[0][e], 0, 0, 0); \
                        acc[1][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv, acc[1][e], 0, 0, 0); \
                    } \
                } \
            } \
            BAMD_AM_VLOAD(A_, B_, C_, D_, b + 2); } while (0)
        for (int cb0 = 0; cb0 < nblk; cb0 += (LONG ? BAMD_AM_CHUNK / 64 : nblk)) {     // !LONG: one turn, everything is in LDS already
            const int cb1 = LONG ? (cb0 + BAMD_AM_CHUNK / 64 < nblk ? cb0 + BAMD_AM_CHUNK / 64 : nblk) : nblk;
            if (LONG) {
                if (cb0) __syncthreads();                                      // every wave is done with the previous chunk
                const int cn = (cb1 - cb0) * 64;                               // positions of this chunk: column tid / 16 walks them 16 at a time, stores them swizzled
                const float * src = scr + (size_t) cb0 * 64 * 16;                   // position quads of the chunk: one 16-byte load per (quad, column)
                for (int i = tid; i < cn * 4; i += NT) {
                    const int quad = i >> 4, n = i & 15;
                    const float4 v = *(const float4 *) (src + (size_t) i * 4);
                    S[am_sidx(4 * quad, n)] = v.x; S[am_sidx(4 * quad + 1, n)] = v.y; S[am_sidx(4 * quad + 2, n)] = v.z; S[am_sidx(4 * quad + 3, n)] = v.w;
                }
                __syncthreads();
            }
            for (int b0 = cb0; b0 < cb1; b0 += 2) { BAMD_AM_VSTEP(v0a, v0b, v0c, v0d, b0); BAMD_AM_VSTEP(v1a, v1b, v1c, v1d, b0 + 1); }     // (chunks hold an even number of blocks, except possibly the last)
        }
#undef BAMD_AM_VSTEP
#undef BAMD_AM_VLOAD
        const int tok = t0 + mrow / GQ;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            // hsum8_tinyblas across the eight e-tiles, element-wise: (a_e + a_{e+4}), then + the tile two over, then + the tile one over
            const f32x4_t t0_ = acc[dt][0] + acc[dt][4], t1_ = acc[dt][1] + acc[dt][5], t2_ = acc[dt][2] + acc[dt][6], t3_ = acc[dt][3] + acc[dt][7];
            const f32x4_t u0 = t0_ + t2_, u1 = t1_ + t3_;
            const f32x4_t o = u0 + u1;
            if (tok < T) {
                float * out = a.out + (size_t) tok * a.ld_out + (size_t) (hk * GQ + mrow % GQ) * hd + (wave * DT + dt) * 16 + 4 * kq;      // rows d = 16 (DT w + dt) + 4 kq + r
                *(float4 *) out = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// 0 = launched (after the KV store); 1 = shape not covered: the caller takes attn_batch_kernel
size_t bamd_attention_batch_mfma_scratch(int Hkv, int gq, int T, int ld) {       // bytes of a.batch_scratch a micro-batch of T tokens needs when ld > BAMD_AM_MAXPOS (else 0)
    if (ld <= BAMD_AM_MAXPOS || (gq != 1 && gq != 2 && gq != 4 && gq != 8 && gq != 16)) return 0;
    const int tt = 16 / gq;
    return (size_t) Hkv * ((T + tt - 1) / tt) * 16 * (size_t) ld * 4;
}
int bamd_launch_attention_batch_mfma(const bamd_attn_args & a, int gq, int T, hipStream_t s) {
    static const bool on = [] { const char * e = getenv("BAMD_ATTN_MFMA"); return !(e && e[0] == '0'); }();
    if (!on || a.hd != 128 || !a.batch || T < 2) return 1;
    if (gq != 1 && gq != 2 && gq != 4 && gq != 8 && gq != 16) return 1;
    const int npos = a.lds_ld ? a.lds_ld : a.n_ctx;                            // the caller's bound on the padded sequence length of this micro-batch (multiple of 64)
    if ((npos & 63) || npos > a.n_ctx) return 1;
    const bool lng = npos > BAMD_AM_MAXPOS;
    if (lng && !a.batch_scratch) return 1;
    static const int dbg = [] { const char * e = getenv("BAMD_AM_EXIT"); return e ? atoi(e) : 0; }();     // timing experiments only: leave the kernel after phase 1 / 2 / 3
    const size_t lds = BAMD_AM_QBYTES + BAMD_AM_VBYTES + BAMD_AM_RBYTES + (size_t) (lng ? BAMD_AM_CHUNK : npos) * 64;
    const int tt = 16 / gq;
    const dim3 grid(a.Hkv * ((T + tt - 1) / tt)), block(64 * BAMD_AM_NW);
#define BAMD_AM_GO(GQ_) do { if (lng) hipLaunchKernelGGL((attn_batch_mfma_kernel<GQ_, true>), grid, block, lds, s, a, T, dbg); \
                             else     hipLaunchKernelGGL((attn_batch_mfma_kernel<GQ_, false>), grid, block, lds, s, a, T, dbg); } while (0)
    switch (gq) {
        case 1: BAMD_AM_GO(1); break;
        case 2: BAMD_AM_GO(2); break;
        case 4: BAMD_AM_GO(4); break;
        case 8: BAMD_AM_GO(8); break;
        default: BAMD_AM_GO(16); break;
    }
#undef BAMD_AM_GO
    return 0;
}
