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
#define BAMD_AM_VROW 144                       /* bytes per staged V^T row (64 positions x f16 = 128 B, padded: 36 dwords -> 16 rows on 16 distinct banks) */
#define BAMD_AM_QBYTES (16 * 128 * 2)          /* q of the 16 columns, f16, chain-major */
#define BAMD_AM_VBYTES (8 * 16 * BAMD_AM_VROW) /* one staged block per wave */
#define BAMD_AM_RBYTES 1024                    /* column maxima per wave + 1 / sum per column */
#define BAMD_AM_MAXPOS 2176                    /* 64 B of LDS per position: 136 KB + q + V stage + reduction scratch <= 160 KB */
#define BAMD_AM_CHUNK 1024                     /* LONG: positions staged through LDS at a time in pass 3 (64 KB) */
#ifndef BAMD_AM_KD
#define BAMD_AM_KD 2                           /* tiles of 16 K rows in flight per wave (pass 1) */
#endif

// score / probability storage [position][column] in LDS, laid out for the three access patterns (64 banks of 4 bytes):
//   row(p) = the positions of a group of 32 reordered as [p & 7][(p >> 3) & 3]: the four positions p, p + 8, p + 16, p + 24 that the four k-groups of a
//            P.V MFMA read together sit in rows with different (row & 3) = different quarters of the banks;
//   column n of a row is stored at n ^ ((row >> 2) & 15): the 32 consecutive positions a half-wave walks in the softmax pass hit 32 distinct banks.
__device__ __forceinline__ int am_sidx(int p, int n) {
    const int row = (p & ~31) | ((p & 7) << 2) | ((p >> 3) & 3);
    return row * 16 + (n ^ ((row >> 2) & 15));
}
__device__ __forceinline__ float h2f_lo(uint32_t w) { return __half2float(__ushort_as_half((unsigned short) (w & 0xffffu))); }
__device__ __forceinline__ float h2f_hi(uint32_t w) { return __half2float(__ushort_as_half((unsigned short) (w >> 16))); }

// LONG: more positions than the LDS holds rows for — the scores / exp values of the 16 columns live in a global scratch block of this workgroup
// ([16 columns][ld] f32: written by pass 1, exponentiated in place by pass 2) and pass 3 walks them in chunks of BAMD_AM_CHUNK positions staged
// through LDS; the P.V chains simply continue from chunk to chunk.
template <int GQ, bool LONG>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) attn_batch_mfma_kernel(bamd_attn_args a, int T, int dbg_exit) {
    constexpr int hd = 128, L = 16, TT = 16 / GQ;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short * q16 = (unsigned short *) smem;                            // [16 columns][128], chain-major (kperm)
    unsigned char * vst = smem + BAMD_AM_QBYTES;                               // [8 waves][16 rows][BAMD_AM_VROW]
    float * cmaxs = (float *) (smem + BAMD_AM_QBYTES + BAMD_AM_VBYTES);        // [8 waves][16 columns] score maxima; fsv [16]: 1 / sum of every column
    float * fsv = cmaxs + 128;
    float * S = (float *) (smem + BAMD_AM_QBYTES + BAMD_AM_VBYTES + BAMD_AM_RBYTES);   // [positions][16]: scores, then exp values (the probabilities are formed in pass 3)
    const int sld = a.lds_ld ? a.lds_ld : a.n_ctx;                              // LONG: floats per column of this workgroup's scratch block
    float * scr = LONG ? a.batch_scratch + (size_t) blockIdx.x * 16 * sld : nullptr;
    const bamd_step_state * st = a.st;
    const int Hkv = a.Hkv, Ekv = Hkv * hd, n_ctx = a.n_ctx;
    const int hk = (int) blockIdx.x % Hkv, tile = (int) blockIdx.x / Hkv;      // consecutive workgroups: different KV heads (= different XCDs for Hkv = 8)
    const int t0 = tile * TT;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int P0 = a.batch_pos0p1 > 0 ? a.batch_pos0p1 - 1 : st->pos;                 // (the host knows the micro-batch's first position: no dependent load in front of everything)
    const int tlast = (t0 + TT - 1 < T ? t0 + TT - 1 : T - 1);
    const int pmax = P0 + tlast;                                               // highest position any column of this tile attends
    int npos = (pmax + 1 + 63) & ~63; npos = npos < n_ctx ? npos : n_ctx;       // positions past a column's own are masked: exact no-ops (attn_batch_kernel)
    const int mrow = lane & 15, kq = lane >> 4;                                // MFMA operand roles of this lane: A[m = mrow][k = kq], B[k = kq][n = mrow]
    // a ring of BAMD_AM_KD tiles of K rows per wave in flight (64 bytes per lane and tile: the 8-byte piece e * 16 + 4 kq .. + 3 of every e), requested
    // unconditionally (a tile past the end: the last one again) so that the waits stay counted.  The first ring goes out BEFORE the RoPE prologue:
    // with one workgroup per CU nothing else hides that latency
    const unsigned short * kbase = a.kc + (size_t) hk * hd + 4 * kq;
    const int ntile = npos >> 4;
    uint2 ring[BAMD_AM_KD][8];
    auto kload = [&](uint2 (&dst)[8], int pt) {
        const int row = (pt < ntile ? pt : ntile - 1) * 16 + mrow;
        const unsigned short * kp = kbase + (size_t) row * Ekv;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = *(const uint2 *) (kp + e * L);
    };
    // ---- RoPE of the 16 query vectors -> f16, chain-major (rope_heads' arithmetic, ggml.c:14130-14143).  16 x 64 pairs on 512 threads: two per thread,
    //      both requested (8-byte loads) before anything else, the K ring right behind them ----
    float2 qx[2], cs[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = tid + it * 512, n = i >> 6, p = i & 63;
        int tok = t0 + n / GQ; tok = tok < T ? tok : T - 1;                     // a tile past the batch end repeats the last token (never stored)
        const int h = hk * GQ + n % GQ;
        qx[it] = *(const float2 *) (a.q + (size_t) tok * a.ld_qkv + (size_t) h * hd + 2 * p);
        cs[it] = *(const float2 *) (a.rope + (size_t) (P0 + tok) * hd + 2 * p);
    }
#pragma unroll
    for (int d = 0; d < BAMD_AM_KD; ++d) kload(ring[d], wave + 8 * d);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = tid + it * 512, n = i >> 6, p = i & 63;
        const float x0 = qx[it].x, x1 = qx[it].y, c = cs[it].x, sn = cs[it].y;
        const float u0 = x0 * c, u1 = x1 * sn, u2 = x0 * sn, u3 = x1 * c;
        q16[n * hd + kperm(2 * p, L)] = f2h(u0 - u1); q16[n * hd + kperm(2 * p + 1, L)] = f2h(u2 + u3);
    }
    __syncthreads();
    if (dbg_exit == 1) return;
    // ================= pass 1: scores (and the running maximum of every column) =================
    {
        // B fragments, once: column mrow, elements 32 s + 8 j + e for s = kq — stored at e * 16 + 4 kq + j: four consecutive halves per e
        float B[8][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint2 w = *(const uint2 *) (q16 + mrow * hd + e * L + 4 * kq);
            B[e][0] = h2f_lo(w.x); B[e][1] = h2f_hi(w.x); B[e][2] = h2f_lo(w.y); B[e][3] = h2f_hi(w.y);
        }
        const int pcol = P0 + ((t0 + mrow / GQ) < T ? (t0 + mrow / GQ) : T - 1);   // the position of this lane's column (D layout: column = lane & 15 as well)
        float cmax = -INFINITY;
        for (int pt0 = wave; pt0 < ntile; pt0 += 8 * BAMD_AM_KD) {
#pragma unroll
            for (int d = 0; d < BAMD_AM_KD; ++d) {
                const int pt = pt0 + 8 * d;
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
                    if (LONG) *(float4 *) (scr + (size_t) mrow * sld + pt * 16 + 4 * kq) = make_float4(vv[0], vv[1], vv[2], vv[3]);   // four consecutive positions of column mrow
                }
                kload(ring[d], pt + 8 * BAMD_AM_KD);
            }
        }
        // column maxima: across the four k-groups of the wave, then across the waves through LDS
        { const float o = __shfl_xor(cmax, 16); cmax = o > cmax ? o : cmax; }
        { const float o = __shfl_xor(cmax, 32); cmax = o > cmax ? o : cmax; }
        if (lane < 16) cmaxs[wave * 16 + lane] = cmax;
    }
    if (dbg_exit == 2) return;
    // the first four blocks of this wave's V^T rows are requested now: they land while the softmax pass runs
    const unsigned short * vrow0 = a.vc + (size_t) (hk * hd + wave * 16) * n_ctx;
    const int srow = lane >> 3, sby = (lane & 7) * 16;                         // staging: lane i fetches 16 bytes of row (i >> 3) [+ 8] at byte (i & 7) * 16 of the 128-byte block
    const unsigned char * ga = (const unsigned char *) (vrow0 + (size_t) srow * n_ctx) + sby, * gb = (const unsigned char *) (vrow0 + (size_t) (srow + 8) * n_ctx) + sby;
    const int nblk = npos >> 6;
    uint4 va0, vb0, va1, vb1, va2, vb2, va3, vb3;                              // four blocks in flight, in named registers (an array ring ended up in scratch memory)
#define BAMD_AM_VLOAD(A_, B_, blk_) do { const int bb_ = ((blk_) < nblk ? (blk_) : nblk - 1) * 128; A_ = *(const uint4 *) (ga + bb_); B_ = *(const uint4 *) (gb + bb_); } while (0)
    BAMD_AM_VLOAD(va0, vb0, 0); BAMD_AM_VLOAD(va1, vb1, 1); BAMD_AM_VLOAD(va2, vb2, 2); BAMD_AM_VLOAD(va3, vb3, 3);
    __syncthreads();
    // ================= pass 2: softmax, two columns per wave (one per half-wave), ggml.c:13682-13778 + :2619-2671 =================
    {
        const int n = 2 * wave + (lane >> 5), pl = lane & 31, hb = lane & 32;
        const float scale = a.kq_scale;
        float smax = cmaxs[n];
#pragma unroll
        for (int w2 = 1; w2 < 8; ++w2) { const float o = cmaxs[w2 * 16 + n]; smax = o > smax ? o : smax; }
        const float mx = smax * scale;                                         // max_i (s_i * scale): the product is monotonic in s (scale > 0)
        double sum = 0.0;
        float * colv = LONG ? scr + (size_t) n * sld : nullptr;
        for (int p = pl; p < npos; p += 32) {                                  // npos % 64 == 0: every 8-lane group is all-active
            const float w = (LONG ? colv[p] : S[am_sidx(p, n)]) * scale;
            const float val = v_expf(w - mx);
            if (LONG) colv[p] = val; else S[am_sidx(p, n)] = val;
            const float c = hsum8_tinyblas(val);                               // the reference's 8-wide partial sum (same tree shape), valid in lane & 7 == 0
            if ((lane & 7) == 0) sum += (double) c;
        }
        // the four partial sums of the half-wave in a fixed order; the reference's order is sequential over the 8-groups: f32_rounding_safe
        // decides whether the order can matter for (float) (1 / sum), and if it can one lane redoes the sum in the reference's order
        double tot = ((__shfl(sum, hb) + __shfl(sum, hb + 8)) + __shfl(sum, hb + 16)) + __shfl(sum, hb + 24);
        double rs = 1.0 / tot;
        if (!f32_rounding_safe(rs, BAMD_F64_GUARD_ULPS(npos / 8))) {           // uniform per half-wave; rare
            double sq = 0.0;
            if (LONG) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }   // the half-wave's own stores, then lane 0's loads
            if (pl == 0) {
                for (int p = 0; p < npos; p += 8) {
                    float v0, v1, v2, v3, v4, v5, v6, v7;
                    if (LONG) { v0 = colv[p]; v1 = colv[p + 1]; v2 = colv[p + 2]; v3 = colv[p + 3]; v4 = colv[p + 4]; v5 = colv[p + 5]; v6 = colv[p + 6]; v7 = colv[p + 7]; }
                    else { v0 = S[am_sidx(p, n)]; v1 = S[am_sidx(p + 1, n)]; v2 = S[am_sidx(p + 2, n)]; v3 = S[am_sidx(p + 3, n)];
                           v4 = S[am_sidx(p + 4, n)]; v5 = S[am_sidx(p + 5, n)]; v6 = S[am_sidx(p + 6, n)]; v7 = S[am_sidx(p + 7, n)]; }
                    const float a0 = v0 + v4, a1 = v1 + v5, a2 = v2 + v6, a3 = v3 + v7;
                    const float b0 = a0 + a2, b1 = a1 + a3;
                    sq += (double) (b0 + b1);
                }
            }
            rs = 1.0 / __shfl(sq, hb);
        }
        if (pl == 0) fsv[n] = (float) rs;                                       // the probabilities are formed where they are used: p = e * fs in pass 3
    }
    __syncthreads();
    if (dbg_exit == 3) return;
    // ================= pass 3: P.V — wave w owns the 16 rows d = 16 w .. 16 w + 15 of V^T, all 16 columns =================
    {
        unsigned char * vs = vst + wave * (16 * BAMD_AM_VROW);
        const float fs = fsv[mrow];                                            // ggml_vec_scale_f32 by (float) (1 / sum): one multiply per probability
        // where this lane's B operands sit (am_sidx of p = 64 b + 8 (l0 + kq) + e, column mrow): row = 64 b + 32 (l0 / 4) + 4 e + kq and the column
        // swizzle (8 (l0 / 4) + e) & 15 do not depend on b — sixteen lane constants, everything else is an immediate offset of the DS read
        int bcol[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) bcol[c] = kq * 16 + (mrow ^ c);
        f32x4_t acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = (f32x4_t) { 0.f, 0.f, 0.f, 0.f };
#define BAMD_AM_VSTEP(A_, B_, b_) do { \
            const int b = (b_), bc = b; \
            if (b < cb1) {                                                     /* wave-uniform; no request inside */ \
                /* (DS operations of one wave execute in order: these stores land behind the previous block's reads) */ \
                *(uint4 *) (vs + srow * BAMD_AM_VROW + sby) = A_; \
                *(uint4 *) (vs + (srow + 8) * BAMD_AM_VROW + sby) = B_; \
                const unsigned short * vr = (const unsigned short *) (vs + mrow * BAMD_AM_VROW) + kq;   /* A[m = mrow][k = kq]: V^T[d][64 b + 8 (l0 + kq) + e] sits at 8 e + l0 + kq */ \
                const float * Sb = S + (bc - cb0) * 64 * 16; const float fsb = fs; \
                _Pragma("unroll") for (int l0 = 0; l0 < 8; l0 += 4) { \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) { \
                        const float av = __half2float(__ushort_as_half(vr[8 * e + l0])); \
                        const float bv = Sb[(32 * (l0 / 4) + 4 * e) * 16 + bcol[(8 * (l0 / 4) + e) & 15]] * fsb;   /* B[k = kq][n = mrow]: p[64 b + 8 (l0 + kq) + e] of column mrow */ \
                        acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[e], 0, 0, 0); \
                    } \
                } \
            } \
            BAMD_AM_VLOAD(A_, B_, b + 4); } while (0)
        for (int cb0 = 0; cb0 < nblk; cb0 += (LONG ? BAMD_AM_CHUNK / 64 : nblk)) {     // !LONG: one turn, everything is in LDS already
            const int cb1 = LONG ? (cb0 + BAMD_AM_CHUNK / 64 < nblk ? cb0 + BAMD_AM_CHUNK / 64 : nblk) : nblk;
            if (LONG) {
                if (cb0) __syncthreads();                                      // every wave is done with the previous chunk
                const int cn = (cb1 - cb0) * 64;                               // positions of this chunk: column n = tid / 32 walks them 32 at a time (coalesced), stores them swizzled
                const float * src = scr + (size_t) (tid >> 5) * sld + cb0 * 64;
                for (int p = tid & 31; p < cn; p += 32) S[am_sidx(p, tid >> 5)] = src[p];
                __syncthreads();
            }
            for (int b0 = cb0; b0 < cb1; b0 += 4) {                            // (chunks hold a multiple of four blocks, except possibly the last)
                BAMD_AM_VSTEP(va0, vb0, b0); BAMD_AM_VSTEP(va1, vb1, b0 + 1); BAMD_AM_VSTEP(va2, vb2, b0 + 2); BAMD_AM_VSTEP(va3, vb3, b0 + 3);
            }
        }
#undef BAMD_AM_VSTEP
#undef BAMD_AM_VLOAD
        // hsum8_tinyblas across the eight e-tiles, element-wise: (a_e + a_{e+4}), then + the tile two over, then + the tile one over
        const f32x4_t t0_ = acc[0] + acc[4], t1_ = acc[1] + acc[5], t2_ = acc[2] + acc[6], t3_ = acc[3] + acc[7];
        const f32x4_t u0 = t0_ + t2_, u1 = t1_ + t3_;
        const f32x4_t o = u0 + u1;
        const int tok = t0 + mrow / GQ;
        if (tok < T) {
            float * out = a.out + (size_t) tok * a.ld_out + (size_t) (hk * GQ + mrow % GQ) * hd + wave * 16 + 4 * kq;      // rows d = 16 w + 4 kq + r
            *(float4 *) out = make_float4(o[0], o[1], o[2], o[3]);
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
    const dim3 grid(a.Hkv * ((T + tt - 1) / tt)), block(512);
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
