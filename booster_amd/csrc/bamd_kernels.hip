// bamd_kernels.hip — hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64): the Llama decode hot path
// behind Booster's bridge ABI.  Written from scratch for this chip; not derived from ggml-cuda.
//
// NUMERICS CONTRACT.  Every kernel reproduces, operation for operation, the IEEE-754 arithmetic of the
// reference's CPU path as built for x86 AVX2+FMA+F16C with GGML_USE_LLAMAFILE (what Booster ships):
// integer block dot products are exact; every f32 operation (which products are fused, which sums are
// sequential chains over super-blocks, the shape of each horizontal reduction tree) is the one the
// reference's 256-bit code performs, with one wave lane standing for one SIMD lane.  Compiled with
// -ffp-contract=off; fused multiply-adds are explicit fmaf().  Do NOT build with -ffast-math.
// The only deliberate deviation: the two double-precision sums (RMSNorm sum of squares, softmax denominator)
// are tree-reduced in a fixed order instead of sequentially; their result is rounded to f32 right after, so
// this cannot be observed except with probability ~1e-8 per reduction.
//
// Reference functions restated here (cpp/ = /root/reference/cpp):
//   quantize_row_q8_K_ref            ggml/src/ggml-quants.c:3593-3630
//   ggml_vec_dot_q4_K_q8_K (AVX2)    ggml/src/ggml-quants.c:6914-6978
//   ggml_vec_dot_q5_K_q8_K (AVX2)    ggml/src/ggml-quants.c:7487-7564
//   ggml_vec_dot_q6_K_q8_K (AVX2)    ggml/src/ggml-quants.c:8145-8222
//   ggml_compute_forward_rms_norm    ggml/src/ggml.c:11850-11896
//   ggml_compute_forward_rope_f32    ggml/src/ggml.c:14043-14167 (NORM mode)
//   ggml_compute_forward_soft_max    ggml/src/ggml.c:13682-13778, ggml_v_expf :2490-2522
//   ggml_v_silu / ggml_vec_silu_f32  ggml/src/ggml.c:2524-2531, :2595-2617
//   tinyBLAS<8,..,fp16,float,float>  ggml/src/llamafile/sgemm.cpp:405-431 (KQ at T=1, KQV always)
//   ggml_vec_dot_f16                 ggml/src/ggml.c:2038-2079 (KQ at T>1)
//   dequantize_row_q{4,5,6}_K        ggml/src/ggml-quants.c:2548, :2756, :2970 (embedding get_rows)
//   CUDA counterparts replaced       ggml/src/ggml-cuda/mmvq.cu:50-130, quantize.cu:4-38, norm.cu:101-131,
//                                    rope.cu:31-69, softmax.cu:14-116, cpy.cu:33-59, unary.cu:25-32
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "bamd_formats.h"
#include "bamd_kernels.h"

#define WAVE 64

__device__ __forceinline__ float h2f(uint32_t bits16) { return __half2float(__ushort_as_half((unsigned short) bits16)); }
__device__ __forceinline__ unsigned short f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }
__device__ __forceinline__ int sdot4(uint32_t a, uint32_t b) { return __builtin_amdgcn_sdot4((int) a, (int) b, 0, false); }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)); }

// ggml-quants.c:1632-1637
__device__ __forceinline__ int nearest_int(float fval) {
    float val = fval + 12582912.f;
    return (__float_as_int(val) & 0x007fffff) - 0x00400000;
}

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
    uint8_t * rec = dst + ((int64_t) rg * nb + i) * (8 * bb);
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
        for (int t = 0; t < 16; ++t) rec[hdr_off + r * 16 + t] = src[t];
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

// ===========================================================================================================
// Activation prologue: f32 vector [K] -> Q8_K in LDS, optionally RMSNorm * weight first.
//   q8[i*64 + e*8 + c] : dword = the 4 int8 of elements 32c+4e..32c+4e+3 of super-block i  (lane e reads 32 B)
//   S [i*8 + c]        : int   = sum of the 32 int8 of chunk c  (= bsums[2c] + bsums[2c+1])
//   yd[i]              : f32   = block scale d
// ===========================================================================================================
template <bool NORM>
__device__ __forceinline__ void build_act(const float * __restrict__ x, const float * __restrict__ nw, float eps, int K,
                                          uint32_t * q8, int * S, float * yd, double * red) {
    const int lane = threadIdx.x & 63, wave = wave_id(), nwaves = blockDim.x >> 6;
    const int nb = K >> 8;
    float scale = 1.0f;
    if (NORM) {
        double s = 0.0;
        for (int i = wave; i < nb; i += nwaves) {
            const float4 v = *(const float4 *) (x + i * 256 + lane * 4);
            s += (double) (v.x * v.x); s += (double) (v.y * v.y); s += (double) (v.z * v.z); s += (double) (v.w * v.w);
        }
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < nwaves; ++w) tot += red[w];
        const float mean = (float) (tot / (double) K);
        scale = 1.0f / sqrtf(mean + eps);
    }
    for (int i = wave; i < nb; i += nwaves) {
        const float4 v = *(const float4 *) (x + i * 256 + lane * 4);
        float e0 = v.x, e1 = v.y, e2 = v.z, e3 = v.w;
        if (NORM) {
            const float4 w = *(const float4 *) (nw + i * 256 + lane * 4);
            e0 = (e0 * scale) * w.x; e1 = (e1 * scale) * w.y; e2 = (e2 * scale) * w.z; e3 = (e3 * scale) * w.w;
        }
        float amax = 0.f, mx = 0.f;
        { float a = fabsf(e0); if (a > amax) { amax = a; mx = e0; } }
        { float a = fabsf(e1); if (a > amax) { amax = a; mx = e1; } }
        { float a = fabsf(e2); if (a > amax) { amax = a; mx = e2; } }
        { float a = fabsf(e3); if (a > amax) { amax = a; mx = e3; } }
        int idx = lane;
        for (int o = 32; o; o >>= 1) {
            const float oa = __shfl_xor(amax, o), om = __shfl_xor(mx, o);
            const int oi = __shfl_xor(idx, o);
            if (oa > amax || (oa == amax && oi < idx)) { amax = oa; mx = om; idx = oi; }
        }
        uint32_t packed = 0; int s4 = 0; float d = 0.f;
        if (amax != 0.f) {
            const float iscale = -127.f / mx;
            int q0 = nearest_int(iscale * e0), q1 = nearest_int(iscale * e1), q2 = nearest_int(iscale * e2), q3 = nearest_int(iscale * e3);
            q0 = q0 < 127 ? q0 : 127; q1 = q1 < 127 ? q1 : 127; q2 = q2 < 127 ? q2 : 127; q3 = q3 < 127 ? q3 : 127;
            packed = (uint32_t) (q0 & 0xff) | ((uint32_t) (q1 & 0xff) << 8) | ((uint32_t) (q2 & 0xff) << 16) | ((uint32_t) (q3 & 0xff) << 24);
            s4 = q0 + q1 + q2 + q3;
            d = 1.0f / iscale;
        }
        s4 += __shfl_xor(s4, 1); s4 += __shfl_xor(s4, 2); s4 += __shfl_xor(s4, 4);
        q8[i * 64 + (lane & 7) * 8 + (lane >> 3)] = packed;
        if ((lane & 7) == 0) S[i * 8 + (lane >> 3)] = s4;
        if (lane == 0) yd[i] = d;
    }
    __syncthreads();
}

// test entry: standard block_q8_K bytes out of the prologue (for parity tests against quantize_row_q8_K)
__global__ void __launch_bounds__(512) quantize_q8k_test_kernel(const float * x, const float * nw, float eps, int K, int norm, uint8_t * out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nb = K >> 8;
    uint32_t * q8 = (uint32_t *) smem; int * S = (int *) (q8 + nb * 64); float * yd = (float *) (S + nb * 8);
    double * red = (double *) (((uintptr_t) (yd + nb) + 15) & ~(uintptr_t) 15);
    if (norm) build_act<true>(x, nw, eps, K, q8, S, yd, red); else build_act<false>(x, nw, eps, K, q8, S, yd, red);
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

// ===========================================================================================================
// Quantised mat-vec: y = W . Q8_K(x).  One wave = 8 rows at a time (lane = r*8+e), rows streamed sequentially
// over super-blocks so each lane carries exactly the f32 chain of SIMD lane e of the reference.
// ===========================================================================================================
struct RowAcc { float acc, accm; };

// ---- per-record arithmetic -----------------------------------------------------------------------------
struct RecQ4K { uint4 qs, hd; };
struct RecQ5K { uint4 qs, hd; uint32_t qh; };
struct RecQ6K { uint4 ql; uint2 qh, sc; uint32_t d; };

__device__ __forceinline__ void load_rec(RecQ4K & R, const uint8_t * rec, int lane) {
    R.qs = *(const uint4 *) (rec + lane * 16);
    R.hd = *(const uint4 *) (rec + 1024 + (lane >> 3) * 16);
}
__device__ __forceinline__ void load_rec(RecQ5K & R, const uint8_t * rec, int lane) {
    R.qs = *(const uint4 *) (rec + lane * 16);
    R.qh = *(const uint32_t *) (rec + 1024 + lane * 4);
    R.hd = *(const uint4 *) (rec + 1280 + (lane >> 3) * 16);
}
__device__ __forceinline__ void load_rec(RecQ6K & R, const uint8_t * rec, int lane) {
    R.ql = *(const uint4 *) (rec + lane * 16);
    R.qh = *(const uint2 *) (rec + 1024 + lane * 8);
    R.sc = *(const uint2 *) (rec + 1536 + (lane >> 3) * 16 + ((lane >> 2) & 1) * 8);
    R.d  = *(const unsigned short *) (rec + 1664 + (lane >> 3) * 2);
}

// 6-bit scale/min unpack, ggml-quants.c:6928-6933
__device__ __forceinline__ void unpack_k4(const uint4 & hd, uint32_t & sc03, uint32_t & sc47, uint32_t & mn03, uint32_t & mn47) {
    const uint32_t u0 = hd.y, u1 = hd.z, u2 = hd.w;
    sc03 = u0 & 0x3f3f3f3fu; mn03 = u1 & 0x3f3f3f3fu;
    sc47 = (u2 & 0x0f0f0f0fu) | (((u0 >> 6) & 0x03030303u) << 4);
    mn47 = ((u2 >> 4) & 0x0f0f0f0fu) | (((u1 >> 6) & 0x03030303u) << 4);
}
#define BYTE(w, k) (int) (((w) >> (8 * (k))) & 0xffu)

__device__ __forceinline__ void consume(const RecQ4K & R, int ci, int lane, const uint32_t * q8, const int * S, const float * yd, RowAcc & A) {
    const int e = lane & 7, l = e & 3;
    const float ydv = yd[ci];
    const float d = ydv * h2f(R.hd.x & 0xffffu);
    const float dmin = (-ydv) * h2f(R.hd.x >> 16);
    uint32_t sc03, sc47, mn03, mn47; unpack_k4(R.hd, sc03, sc47, mn03, mn47);
    const uint4 a0 = *(const uint4 *) (q8 + ci * 64 + e * 8), a1 = *(const uint4 *) (q8 + ci * 64 + e * 8 + 4);
    int sumi = 0;
    sumi += BYTE(sc03, 0) * sdot4(R.qs.x & 0x0f0f0f0fu, a0.x) + BYTE(sc03, 1) * sdot4((R.qs.x >> 4) & 0x0f0f0f0fu, a0.y);
    sumi += BYTE(sc03, 2) * sdot4(R.qs.y & 0x0f0f0f0fu, a0.z) + BYTE(sc03, 3) * sdot4((R.qs.y >> 4) & 0x0f0f0f0fu, a0.w);
    sumi += BYTE(sc47, 0) * sdot4(R.qs.z & 0x0f0f0f0fu, a1.x) + BYTE(sc47, 1) * sdot4((R.qs.z >> 4) & 0x0f0f0f0fu, a1.y);
    sumi += BYTE(sc47, 2) * sdot4(R.qs.w & 0x0f0f0f0fu, a1.z) + BYTE(sc47, 3) * sdot4((R.qs.w >> 4) & 0x0f0f0f0fu, a1.w);
    A.acc = fmaf(d, (float) sumi, A.acc);
    const uint32_t mw = (l < 2) ? mn03 : mn47;
    const int sh = (l & 1) * 16;
    const int ma = (int) ((mw >> sh) & 0xffu), mb = (int) ((mw >> (sh + 8)) & 0xffu);
    const int2 sp = *(const int2 *) (S + ci * 8 + 2 * l);
    A.accm = fmaf(dmin, (float) (ma * sp.x + mb * sp.y), A.accm);
}

__device__ __forceinline__ void consume(const RecQ5K & R, int ci, int lane, const uint32_t * q8, const int * S, const float * yd, RowAcc & A) {
    const int e = lane & 7;
    const float ydv = yd[ci];
    const float d = ydv * h2f(R.hd.x & 0xffffu);
    const float dmin = (-ydv) * h2f(R.hd.x >> 16);
    uint32_t sc03, sc47, mn03, mn47; unpack_k4(R.hd, sc03, sc47, mn03, mn47);
    const uint4 a0 = *(const uint4 *) (q8 + ci * 64 + e * 8), a1 = *(const uint4 *) (q8 + ci * 64 + e * 8 + 4);
    const uint32_t qh = R.qh;
#define Q5(w, shift, c) ((((w) >> (shift)) & 0x0f0f0f0fu) | (((qh >> (c)) & 0x01010101u) << 4))
    int sumi = 0;
    sumi += BYTE(sc03, 0) * sdot4(Q5(R.qs.x, 0, 0), a0.x) + BYTE(sc03, 1) * sdot4(Q5(R.qs.x, 4, 1), a0.y);
    sumi += BYTE(sc03, 2) * sdot4(Q5(R.qs.y, 0, 2), a0.z) + BYTE(sc03, 3) * sdot4(Q5(R.qs.y, 4, 3), a0.w);
    sumi += BYTE(sc47, 0) * sdot4(Q5(R.qs.z, 0, 4), a1.x) + BYTE(sc47, 1) * sdot4(Q5(R.qs.z, 4, 5), a1.y);
    sumi += BYTE(sc47, 2) * sdot4(Q5(R.qs.w, 0, 6), a1.z) + BYTE(sc47, 3) * sdot4(Q5(R.qs.w, 4, 7), a1.w);
#undef Q5
    A.acc = fmaf(d, (float) sumi, A.acc);
    // summs += dmin * hsum(mins . q8sums)   (:7515-7518) — integer sum over all 8 sub-blocks, then mul, then add
    const uint32_t mw = (e < 4) ? mn03 : mn47;
    int hs = (int) ((mw >> (8 * (e & 3))) & 0xffu) * S[ci * 8 + e];
    hs += __shfl_xor(hs, 1); hs += __shfl_xor(hs, 2); hs += __shfl_xor(hs, 4);
    const float t = dmin * (float) hs;
    A.accm = A.accm + t;
}

__device__ __forceinline__ void consume(const RecQ6K & R, int ci, int lane, const uint32_t * q8, const int * S, const float * yd, RowAcc & A) {
    (void) S;
    const int e = lane & 7;
    const float d = yd[ci] * h2f(R.d);
    const uint4 a0 = *(const uint4 *) (q8 + ci * 64 + e * 8), a1 = *(const uint4 *) (q8 + ci * 64 + e * 8 + 4);
    // (q6 - 32) as int8: q6 in [0,63] -> (q6 + 0x60) ^ 0x80 per byte, no inter-byte carry
#define Q6(lo, hb) ((((lo) | ((hb) << 4)) + 0x60606060u) ^ 0x80808080u)
#define SB(w, k) ((int) (int8_t) ((w) >> (8 * (k))))
    int sumi = 0;
    {
        const uint32_t A_ = R.ql.x, B_ = R.ql.y, h = R.qh.x, s = R.sc.x;
        sumi += SB(s, 0) * sdot4(Q6(A_ & 0x0f0f0f0fu, h & 0x03030303u), a0.x);
        sumi += SB(s, 1) * sdot4(Q6(B_ & 0x0f0f0f0fu, (h >> 2) & 0x03030303u), a0.y);
        sumi += SB(s, 2) * sdot4(Q6((A_ >> 4) & 0x0f0f0f0fu, (h >> 4) & 0x03030303u), a0.z);
        sumi += SB(s, 3) * sdot4(Q6((B_ >> 4) & 0x0f0f0f0fu, (h >> 6) & 0x03030303u), a0.w);
    }
    {
        const uint32_t A_ = R.ql.z, B_ = R.ql.w, h = R.qh.y, s = R.sc.y;
        sumi += SB(s, 0) * sdot4(Q6(A_ & 0x0f0f0f0fu, h & 0x03030303u), a1.x);
        sumi += SB(s, 1) * sdot4(Q6(B_ & 0x0f0f0f0fu, (h >> 2) & 0x03030303u), a1.y);
        sumi += SB(s, 2) * sdot4(Q6((A_ >> 4) & 0x0f0f0f0fu, (h >> 4) & 0x03030303u), a1.z);
        sumi += SB(s, 3) * sdot4(Q6((B_ >> 4) & 0x0f0f0f0fu, (h >> 6) & 0x03030303u), a1.w);
    }
#undef Q6
#undef SB
    A.acc = fmaf(d, (float) sumi, A.acc);
}

// horizontal reductions at the end of a row (hsum_float_8, ggml-quants.c:47-53, and the acc_m folds)
template <int TYPE>
__device__ __forceinline__ float finish_row(const RowAcc & A) {
    float v = A.acc;
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    if (TYPE == BAMD_Q4_K) {
        float m = A.accm;
        m += __shfl_xor(m, 2); m += __shfl_xor(m, 1);
        return v + m;
    }
    if (TYPE == BAMD_Q5_K) return v + A.accm;
    return v;
}

// ggml_v_expf (AVX2), one lane — ggml.c:2490-2522
__device__ __forceinline__ float v_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = fmaf(-n, 0x1.7f7d1cp-20f, fmaf(-n, 0x1.62e4p-1f, x));
    const uint32_t e = __float_as_uint(z) << 23;
    const float k = __uint_as_float(e + __float_as_uint(1.0f));
    const bool c = fabsf(n) > 126.0f;
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u, 0x1.ffffecp-1f * b);
    if (!c) return fmaf(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    const float s1 = __uint_as_float(g + 0x7f000000u), s2 = __uint_as_float(e - g);
    if (fabsf(n) > 192.0f) return s1 * s1;
    return fmaf(s2, j, s2) * s1;
}
__device__ __forceinline__ float v_silu(float x) {
    const float neg_x = 0.0f - x;
    const float one_plus = 1.0f + v_expf(neg_x);
    return x / one_plus;
}

__device__ __forceinline__ unsigned long long argmax_key(float v, int row) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long) u << 32) | (unsigned long long) (0xffffffffu - (uint32_t) row);
}

// ---- the stream over (row-group, super-block) records for one segment -----------------------------------
// The wave walks row-groups rg = first, first+stride, ... (count of them) as ONE flattened record stream, so the
// register prefetch ring (depth D records) never drains between row-groups.  With PAIR each row-group is
// streamed twice back to back — gate (wA) then up (wB) — and the epilogue fuses silu(gate)*up.
template <int TYPE, typename REC, int D, int EPI>
__device__ __forceinline__ void stream_segment(const uint8_t * __restrict__ wA, const uint8_t * __restrict__ wB, int nb,
                                               int first, int count, int stride, float * __restrict__ out,
                                               const float * __restrict__ res, const uint32_t * q8, const int * S, const float * yd,
                                               unsigned long long & best) {
    constexpr int RECB = TYPE == BAMD_Q4_K ? 1152 : TYPE == BAMD_Q5_K ? 1408 : 1680;
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
    const int lane = threadIdx.x & 63;
    const int total = count * nb * (PAIR ? 2 : 1);       // D divides nb (chosen by the dispatcher below)
    const size_t rgb = (size_t) nb * RECB;
    // loader cursor (wave-uniform)
    int lrg = first, lpart = 0, li = 0, lt = 0;
    REC ring[D];
#define BAMD_LOAD_NEXT(slot) do { \
        const uint8_t * base_ = (PAIR && lpart) ? wB : wA; \
        load_rec(ring[slot], base_ + (size_t) lrg * rgb + (size_t) li * RECB, lane); \
        if (++lt < total) { if (++li == nb) { li = 0; if (PAIR && lpart == 0) lpart = 1; else { lpart = 0; lrg += stride; } } } \
    } while (0)
#pragma unroll
    for (int s = 0; s < D; ++s) BAMD_LOAD_NEXT(s);
    RowAcc A = { 0.f, 0.f };
    float gate_val = 0.f;
    int crg = first, cpart = 0, ci = 0;
    for (int t0 = 0; t0 < total; t0 += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const REC R = ring[s];
            // refill this slot; at the tail the cursor stays on the last record (a redundant, branch-free reload)
            BAMD_LOAD_NEXT(s);
            consume(R, ci, lane, q8, S, yd, A);
            if (++ci == nb) {
                const float val = finish_row<TYPE>(A);
                const int row = crg * 8 + (lane >> 3);
                if (PAIR) {
                    if (cpart == 0) gate_val = val;
                    else if ((lane & 7) == 0) out[row] = v_silu(gate_val) * val;
                } else if ((lane & 7) == 0) {
                    float o = val;
                    if (EPI == BAMD_EPI_ADD) o = val + res[row];
                    out[row] = o;
                    if (EPI == BAMD_EPI_ARGMAX) { const unsigned long long k = argmax_key(o, row); best = k > best ? k : best; }
                }
                A.acc = 0.f; A.accm = 0.f; ci = 0;
                if (PAIR && cpart == 0) cpart = 1; else { cpart = 0; crg += stride; }
            }
        }
    }
#undef BAMD_LOAD_NEXT
}

template <int TYPE, typename REC, int EPI>
__device__ __forceinline__ void stream_dispatch_depth(const uint8_t * wA, const uint8_t * wB, int nb, int first, int count, int stride,
                                                      float * out, const float * res, const uint32_t * q8, const int * S, const float * yd,
                                                      unsigned long long & best) {
    if ((nb & 7) == 0)      stream_segment<TYPE, REC, 8, EPI>(wA, wB, nb, first, count, stride, out, res, q8, S, yd, best);
    else if ((nb & 3) == 0) stream_segment<TYPE, REC, 4, EPI>(wA, wB, nb, first, count, stride, out, res, q8, S, yd, best);
    else if ((nb & 1) == 0) stream_segment<TYPE, REC, 2, EPI>(wA, wB, nb, first, count, stride, out, res, q8, S, yd, best);
    else                    stream_segment<TYPE, REC, 1, EPI>(wA, wB, nb, first, count, stride, out, res, q8, S, yd, best);
}

template <int PRO, int EPI>
__global__ void __launch_bounds__(512) matvec_kernel(bamd_mv_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nb = a.K >> 8;
    uint32_t * q8 = (uint32_t *) smem; int * S = (int *) (q8 + nb * 64); float * yd = (float *) (S + nb * 8);
    double * red = (double *) (((uintptr_t) (yd + nb) + 15) & ~(uintptr_t) 15);
    const float * x = a.x;
    if (PRO == BAMD_PRO_NORM) build_act<true>(x, a.normw, a.eps, a.K, q8, S, yd, red);
    else                      build_act<false>(x, nullptr, 0.f, a.K, q8, S, yd, red);

    const int wave = wave_id(), nwaves = blockDim.x >> 6;
    const int slot = blockIdx.x + gridDim.x * wave;          // consecutive row-groups land on different CUs
    const int stride = gridDim.x * nwaves;
    unsigned long long best = 0ull;
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
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
            if (t == BAMD_Q4_K)      stream_dispatch_depth<BAMD_Q4_K, RecQ4K, EPI>(wA, wB, nb, g0 - off, count, stride, out, res, q8, S, yd, best);
            else if (t == BAMD_Q5_K) stream_dispatch_depth<BAMD_Q5_K, RecQ5K, EPI>(wA, wB, nb, g0 - off, count, stride, out, res, q8, S, yd, best);
            else                     stream_dispatch_depth<BAMD_Q6_K, RecQ6K, EPI>(wA, wB, nb, g0 - off, count, stride, out, res, q8, S, yd, best);
        }
        off += nrg;
    }
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
}

// ===========================================================================================================
// Step begin: pick the token of this step (forced prompt token, or the arg-max of the previous step's logits),
// advance the position, and dequantise its embedding row into the residual stream.
// ===========================================================================================================
__device__ __forceinline__ void get_scale_min_k4(int j, const uint8_t * q, int & d, int & m) {
    if (j < 4) { d = q[j] & 63; m = q[j + 4] & 63; }
    else { d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4); }
}

__global__ void __launch_bounds__(256) step_begin_kernel(bamd_step_state * st, const int32_t * forced, int n_forced,
                                                         int32_t * out_tokens, const uint8_t * embd, int embd_type, int E, int V,
                                                         float * x, int do_embed) {
    __shared__ int tok_s;
    if (threadIdx.x == 0) {
        int step = st->step;
        int tok;
        const unsigned long long key = st->best_key;         // arg-max of the previous lm_head, 0 = none ran
        if (key != 0ull) {
            tok = (int) (0xffffffffu - (uint32_t) (key & 0xffffffffull));
            out_tokens[st->n_out] = tok; st->n_out += 1;
        } else tok = 0;
        if (step < n_forced) tok = forced[step];
        if (tok < 0 || tok >= V) tok = 0;
        st->token = tok;
        if (do_embed) {
            st->pos = st->pos_base + step;
            int n_kv = (st->pos + 1 + 31) / 32 * 32;
            if (n_kv > st->n_ctx) n_kv = st->n_ctx;
            st->n_kv = n_kv;
            st->step = step + 1;
        }
        if (do_embed) st->best_key = 0ull;                   // a flush-only call leaves the key for the next generate call
        tok_s = tok;
    }
    __syncthreads();
    if (!do_embed) return;
    const int tok = tok_s;
    // get_rows: ggml.c:13186-13228 -> dequantize_row_*
    if (embd_type == BAMD_F32) {
        const float * src = (const float *) embd + (size_t) tok * E;
        for (int i = threadIdx.x; i < E; i += blockDim.x) x[i] = src[i];
    } else if (embd_type == BAMD_F16) {
        const unsigned short * src = (const unsigned short *) embd + (size_t) tok * E;
        for (int i = threadIdx.x; i < E; i += blockDim.x) x[i] = h2f(src[i]);
    } else {
        const int nb = E >> 8;
        const int bb = bamd_block_bytes(embd_type);
        const uint8_t * row = embd + (size_t) tok * nb * bb;
        for (int i = threadIdx.x; i < E; i += blockDim.x) {
            const uint8_t * b = row + (size_t) (i >> 8) * bb;
            const int n = i & 255;
            float y;
            if (embd_type == BAMD_Q4_K || embd_type == BAMD_Q5_K) {
                const float d = h2f(*(const unsigned short *) b), mn = h2f(*(const unsigned short *) (b + 2));
                const int c = n >> 5, l = n & 31;           // chunk c: sub-block scale index c
                int sc, m; get_scale_min_k4(c, b + 4, sc, m);
                const float d1 = d * (float) sc, m1 = mn * (float) m;
                int q;
                if (embd_type == BAMD_Q4_K) {
                    const uint8_t v = b[16 + 32 * (c >> 1) + l];
                    q = (c & 1) ? (v >> 4) : (v & 0xF);
                } else {
                    const uint8_t v = b[48 + 32 * (c >> 1) + l];
                    q = ((c & 1) ? (v >> 4) : (v & 0xF)) + (((b[16 + l] >> c) & 1) ? 16 : 0);
                }
                const float t = d1 * (float) q;
                y = t - m1;
            } else {
                const float d = h2f(*(const unsigned short *) (b + 208));
                const int half = n >> 7, nn = n & 127, cc = nn >> 5, l = nn & 31;
                const uint8_t * ql = b + 64 * half, * qh = b + 128 + 32 * half;
                const int8_t * sc = (const int8_t *) (b + 192 + 8 * half);
                const int lo = (cc & 1) ? ql[l + 32] : ql[l];
                const int nib = (cc & 2) ? (lo >> 4) : (lo & 0xF);
                const int q = (int) (int8_t) (nib | (((qh[l] >> (2 * cc)) & 3) << 4)) - 32;
                const int is = l / 16;
                const float t = d * (float) sc[is + 2 * cc];
                y = t * (float) q;
            }
            x[i] = y;
        }
    }
}

// ===========================================================================================================
// Attention (decode): RoPE + KV store + scores | softmax | P.V        (reference: llm_build_kv, llama.cpp:8318)
// ===========================================================================================================
// K cache [n_ctx][Hkv*hd] f16, V cache transposed [Hkv*hd][n_ctx] f16 — the reference's layouts (llama.cpp:7845-7875).
// grid (Hkv, tiles of 64 positions), block 512 = 8 waves x (8 positions x 8 lanes).
template <int GQ>
__global__ void __launch_bounds__(512) attn_qk_kernel(bamd_attn_args a) {
    __shared__ float q_s[GQ * 256];
    __shared__ unsigned short q16_s[GQ * 256];
    __shared__ unsigned short k16_s[256];
    const bamd_step_state * st = a.st;
    const int pos = st->pos, n_kv = st->n_kv;
    const int hd = a.hd, Hkv = a.Hkv, Ekv = Hkv * hd, n_ctx = a.n_ctx;
    const int hk = blockIdx.x;
    const float * rope = a.rope + (size_t) pos * hd;          // (cos, sin) pairs, host-built (ggml_rope_cache_init)
    // RoPE (NORM mode, adjacent pairs) — ggml.c:14130-14143
    for (int i = threadIdx.x; i < (GQ + 1) * (hd / 2); i += blockDim.x) {
        const int hh = i / (hd / 2), p = i % (hd / 2);
        const float c = rope[2 * p], s = rope[2 * p + 1];
        if (hh < GQ) {
            const float * src = a.q + (size_t) (hk * GQ + hh) * hd + 2 * p;
            const float x0 = src[0], x1 = src[1];
            const float t0 = x0 * c, t1 = x1 * s, t2 = x0 * s, t3 = x1 * c;
            const float r0 = t0 - t1, r1 = t2 + t3;
            q_s[hh * hd + 2 * p] = r0; q_s[hh * hd + 2 * p + 1] = r1;
            q16_s[hh * hd + 2 * p] = f2h(r0); q16_s[hh * hd + 2 * p + 1] = f2h(r1);
        } else {
            const float * src = a.k + (size_t) hk * hd + 2 * p;
            const float x0 = src[0], x1 = src[1];
            const float t0 = x0 * c, t1 = x1 * s, t2 = x0 * s, t3 = x1 * c;
            k16_s[2 * p] = f2h(t0 - t1); k16_s[2 * p + 1] = f2h(t2 + t3);
        }
    }
    __syncthreads();
    const int tiles = (n_kv + 63) >> 6;
    // KV store by the block that owns the tile of `pos` — llm_build_kv_store, llama.cpp:7830-7875
    if ((int) blockIdx.y == ((pos >> 6) % (int) gridDim.y)) {
        for (int i = threadIdx.x; i < hd; i += blockDim.x) {
            a.kc[(size_t) pos * Ekv + hk * hd + i] = k16_s[i];
            a.vc[(size_t) (hk * hd + i) * n_ctx + pos] = f2h(a.v[hk * hd + i]);
        }
    }
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int e = lane & 7;
    for (int tile = blockIdx.y; tile < tiles; tile += gridDim.y) {
        const int i = tile * 64 + wave * 8 + (lane >> 3);        // position
        if (i >= n_kv) continue;
        float sc[GQ];
        if (i > pos) {
#pragma unroll
            for (int g = 0; g < GQ; ++g) sc[g] = -INFINITY;      // masked (KQ_mask, llama.cpp:14152-14200)
        } else {
            const unsigned short * krow = (i == pos) ? k16_s : a.kc + (size_t) i * Ekv + hk * hd;
            if (!a.prefill_mode) {
                // tinyBLAS F16 x F32, KN = 8: Cv[e] = fma(K[l+e], q[l+e], Cv[e]) ; then hsum   (sgemm.cpp:405-431)
                float acc[GQ];
#pragma unroll
                for (int g = 0; g < GQ; ++g) acc[g] = 0.f;
                for (int l = 0; l < hd; l += 8) {
                    const float kv = h2f(krow[l + e]);
#pragma unroll
                    for (int g = 0; g < GQ; ++g) acc[g] = fmaf(kv, q_s[g * hd + l + e], acc[g]);
                }
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    float v = acc[g];
                    v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
                    sc[g] = v;
                }
            } else {
                // T > 1: q rounded to f16, ggml_vec_dot_f16 with 4 accumulators x 8 lanes (ggml.c:2038, :1285-1305)
                float acc[GQ][4];
#pragma unroll
                for (int g = 0; g < GQ; ++g) { acc[g][0] = acc[g][1] = acc[g][2] = acc[g][3] = 0.f; }
                for (int l = 0; l < hd; l += 32) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float kv = h2f(krow[l + 8 * j + e]);
#pragma unroll
                        for (int g = 0; g < GQ; ++g) acc[g][j] = fmaf(kv, h2f(q16_s[g * hd + l + 8 * j + e]), acc[g][j]);
                    }
                }
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    const float s02 = acc[g][0] + acc[g][2], s13 = acc[g][1] + acc[g][3];
                    float v = s02 + s13;
                    v += __shfl_xor(v, 4); v += __shfl_xor(v, 1); v += __shfl_xor(v, 2);   // lo+hi, then two hadd_ps
                    sc[g] = v;
                }
            }
        }
        if (e == 0) {
#pragma unroll
            for (int g = 0; g < GQ; ++g) a.scores[(size_t) (hk * GQ + g) * n_ctx + i] = sc[g];
        }
    }
}

// softmax over n_kv scores of one head: grid (H), block 256.  ggml.c:13682-13778 + :2619-2671 (AVX2 branch).
__global__ void __launch_bounds__(256) attn_softmax_kernel(bamd_attn_args a) {
    __shared__ float redf[4];
    __shared__ double redd[4];
    const bamd_step_state * st = a.st;
    const int n_kv = st->n_kv, n_ctx = a.n_ctx;
    const int h = blockIdx.x;
    float * s = a.scores + (size_t) h * n_ctx;
    const float scale = a.kq_scale;
    const int lane = threadIdx.x & 63, wave = wave_id();
    // wp = s*scale + mask  (mask already folded in as -inf scores); max
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n_kv; i += blockDim.x) { const float w = s[i] * scale; mx = w > mx ? w : mx; }
    for (int o = 32; o; o >>= 1) { const float om = __shfl_xor(mx, o); mx = om > mx ? om : mx; }
    if (lane == 0) redf[wave] = mx;
    __syncthreads();
    mx = redf[0]; for (int w = 1; w < 4; ++w) mx = redf[w] > mx ? redf[w] : mx;
    // exp, 8-element chunk sums (f32 tree of the reference), double accumulation of the chunk sums
    double sum = 0.0;
    for (int i = threadIdx.x; i < n_kv; i += blockDim.x) {
        const float w = s[i] * scale;
        const float val = v_expf(w - mx);
        s[i] = val;
        float c = val;
        c += __shfl_xor(c, 4); c += __shfl_xor(c, 2); c += __shfl_xor(c, 1);
        if ((lane & 7) == 0) sum += (double) c;
    }
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) redd[wave] = sum;
    __syncthreads();
    double tot = 0.0; for (int w = 0; w < 4; ++w) tot += redd[w];
    const float fs = (float) (1.0 / tot);
    for (int i = threadIdx.x; i < n_kv; i += blockDim.x) s[i] = s[i] * fs;
}

// P.V: grid (Hkv, hd/8), block 64: lane = d_local*8 + e carries the tinyBLAS chain Cv[e] of output (h, d) for the
// GQ heads that share this KV head.  sgemm.cpp:405-431 with A = V^T rows (f16), B = p (f32).
template <int GQ>
__global__ void __launch_bounds__(64) attn_pv_kernel(bamd_attn_args a) {
    const bamd_step_state * st = a.st;
    const int n_kv = st->n_kv, n_ctx = a.n_ctx, hd = a.hd;
    const int hk = blockIdx.x;
    const int lane = threadIdx.x, e = lane & 7;
    const int d = blockIdx.y * 8 + (lane >> 3);
    const unsigned short * vrow = a.vc + (size_t) (hk * hd + d) * n_ctx;
    const float * p = a.scores + (size_t) (hk * GQ) * n_ctx;
    float acc[GQ];
#pragma unroll
    for (int g = 0; g < GQ; ++g) acc[g] = 0.f;
    for (int l = 0; l < n_kv; l += 8) {
        const float vv = h2f(vrow[l + e]);
#pragma unroll
        for (int g = 0; g < GQ; ++g) acc[g] = fmaf(vv, p[(size_t) g * n_ctx + l + e], acc[g]);
    }
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
        float v = acc[g];
        v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        if (e == 0) a.out[(size_t) (hk * GQ + g) * hd + d] = v;
    }
}

// ===========================================================================================================
// launchers
// ===========================================================================================================
static size_t act_lds_bytes(int K) {
    const int nb = K >> 8;
    size_t b = (size_t) nb * (256 + 32 + 4);
    b = (b + 15) & ~(size_t) 15;
    return b + 16 * sizeof(double) + 16 * sizeof(unsigned long long);
}

void bamd_launch_repack(const void * raw, void * dst, int type, int nrows, int K, hipStream_t s) {
    const int nb = K >> 8;
    const int64_t n = (int64_t) nrows * nb;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, (const uint8_t *) raw, (uint8_t *) dst, type, nrows, nb);
}

void bamd_launch_quantize_q8k_test(const float * x, const float * nw, float eps, int K, int norm, void * out, hipStream_t s) {
    hipLaunchKernelGGL(quantize_q8k_test_kernel, dim3(1), dim3(512), act_lds_bytes(K), s, x, nw, eps, K, norm, (uint8_t *) out);
}

template <int PRO>
static void launch_mv_epi(const bamd_mv_args & a, int epi, int grid, hipStream_t s) {
    const size_t lds = act_lds_bytes(a.K);
    switch (epi) {
        case BAMD_EPI_STORE:    hipLaunchKernelGGL((matvec_kernel<PRO, BAMD_EPI_STORE>),    dim3(grid), dim3(512), lds, s, a); break;
        case BAMD_EPI_ADD:      hipLaunchKernelGGL((matvec_kernel<PRO, BAMD_EPI_ADD>),      dim3(grid), dim3(512), lds, s, a); break;
        case BAMD_EPI_SILU_MUL: hipLaunchKernelGGL((matvec_kernel<PRO, BAMD_EPI_SILU_MUL>), dim3(grid), dim3(512), lds, s, a); break;
        case BAMD_EPI_ARGMAX:   hipLaunchKernelGGL((matvec_kernel<PRO, BAMD_EPI_ARGMAX>),   dim3(grid), dim3(512), lds, s, a); break;
    }
}

void bamd_launch_matvec(const bamd_mv_args & a, int pro, int epi, int n_cu, hipStream_t s) {
    int nrg = 0;
    if (epi == BAMD_EPI_SILU_MUL) nrg = a.seg[0].nrows >> 3;
    else for (int i = 0; i < a.nseg; ++i) nrg += a.seg[i].nrows >> 3;
    int grid = n_cu > 0 ? n_cu : 256;                // one 8-wave workgroup per CU (160 VGPRs -> 3 waves/SIMD)
    if (grid > nrg) grid = nrg;
    if (grid < 1) grid = 1;
    if (pro == BAMD_PRO_NORM) launch_mv_epi<BAMD_PRO_NORM>(a, epi, grid, s);
    else                      launch_mv_epi<BAMD_PRO_PLAIN>(a, epi, grid, s);
}

void bamd_launch_step_begin(bamd_step_state * st, const int32_t * forced, int n_forced, int32_t * out_tokens, const void * embd,
                            int embd_type, int E, int V, float * x, int do_embed, hipStream_t s) {
    hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(256), 0, s, st, forced, n_forced, out_tokens, (const uint8_t *) embd, embd_type, E, V, x, do_embed);
}

int bamd_launch_attention(const bamd_attn_args & a, int gq, int max_tiles, hipStream_t s) {
    if (a.hd > 256 || (a.hd & 31)) return 1;
    int ty = max_tiles < 1 ? 1 : max_tiles;
    dim3 g1(a.Hkv, ty), g3(a.Hkv, a.hd / 8);
    switch (gq) {
#define CASE(G) case G: \
        hipLaunchKernelGGL((attn_qk_kernel<G>), g1, dim3(512), 0, s, a); \
        hipLaunchKernelGGL(attn_softmax_kernel, dim3(a.Hkv * G), dim3(256), 0, s, a); \
        hipLaunchKernelGGL((attn_pv_kernel<G>), g3, dim3(64), 0, s, a); break;
        CASE(1) CASE(2) CASE(4) CASE(8)
#undef CASE
        default: return 1;
    }
    return 0;
}
