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
#include <stdlib.h>
#include "bamd_formats.h"
#include "bamd_kernels.h"

#define WAVE 64
#ifndef BAMD_SCHED_GROUP
#define BAMD_SCHED_GROUP 1      /* records the scheduler may interleave between barriers (power of two) */
#endif

// optional in-kernel phase stamps (build with -DBAMD_TIMING): block 0 / lane 0 of each wave writes s_memtime
#ifdef BAMD_TIMING
__device__ unsigned long long g_stamps[64 * 16];
#define STAMP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_stamps[(threadIdx.x >> 6) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ float h2f(uint32_t bits16) { return __half2float(__ushort_as_half((unsigned short) bits16)); }
__device__ __forceinline__ unsigned short f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }
__device__ __forceinline__ int sdot4(uint32_t a, uint32_t b) { return __builtin_amdgcn_sdot4((int) a, (int) b, 0, false); }
// sum_j scale_j * dot4(w_j, a_j) over the 8 sub-blocks a lane covers in one super-block: the 8 scale bytes are the bytes of (s0, s1),
// unsigned (Q4_K/Q5_K 6-bit scales) or signed (Q6_K int8 scales).  One asm block: 8 VOP3P dots, 8 SDWA multiplies that pick their
// scale byte directly (no extraction instructions), 4 adds.  Every product is >= 8 instructions behind its dot: no wait states needed.
// hipcc selects v_dot4c (accumulate-into-destination) for __builtin_amdgcn_sdot4(a, b, 0) and spends a v_mov 0 per product and a
// v_bfe per scale byte; the VOP3P form takes the zero as an inline constant (checked against the builtin by tools/dot4_probe.cpp; a
// DOT result needs 3 wait states before another VALU instruction reads it, which the 8-instruction distance provides).
// Exact integer arithmetic: any association gives the reference's int32 (ggml-quants.c:6950-6968, :8190-8216).
template <bool SIGNED>
__device__ __forceinline__ int dotscale8(const uint32_t (&a)[8], const uint32_t (&b)[8], uint32_t s0, uint32_t s1) {
    int t0, t1, t2, t3, t4, t5, t6, t7, sum;
#define BAMD_SDWA_MUL(k, sreg, byte) "v_mul_i32_i24_sdwa %" #k ", %" #k ", " sreg " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #byte "\n\t"
    if (SIGNED) {
        asm("v_dot4_i32_i8 %0, %9, %17, 0\n\t" "v_dot4_i32_i8 %1, %10, %18, 0\n\t" "v_dot4_i32_i8 %2, %11, %19, 0\n\t" "v_dot4_i32_i8 %3, %12, %20, 0\n\t"
            "v_dot4_i32_i8 %4, %13, %21, 0\n\t" "v_dot4_i32_i8 %5, %14, %22, 0\n\t" "v_dot4_i32_i8 %6, %15, %23, 0\n\t" "v_dot4_i32_i8 %7, %16, %24, 0\n\t"
            BAMD_SDWA_MUL(0, "sext(%25)", 0) BAMD_SDWA_MUL(1, "sext(%25)", 1) BAMD_SDWA_MUL(2, "sext(%25)", 2) BAMD_SDWA_MUL(3, "sext(%25)", 3)
            BAMD_SDWA_MUL(4, "sext(%26)", 0) BAMD_SDWA_MUL(5, "sext(%26)", 1) BAMD_SDWA_MUL(6, "sext(%26)", 2) BAMD_SDWA_MUL(7, "sext(%26)", 3)
            "v_add3_u32 %8, %0, %1, %2\n\t" "v_add3_u32 %8, %8, %3, %4\n\t" "v_add3_u32 %8, %8, %5, %6\n\t" "v_add_u32 %8, %8, %7"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(sum)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
              "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(s0), "v"(s1));
    } else {
        asm("v_dot4_i32_i8 %0, %9, %17, 0\n\t" "v_dot4_i32_i8 %1, %10, %18, 0\n\t" "v_dot4_i32_i8 %2, %11, %19, 0\n\t" "v_dot4_i32_i8 %3, %12, %20, 0\n\t"
            "v_dot4_i32_i8 %4, %13, %21, 0\n\t" "v_dot4_i32_i8 %5, %14, %22, 0\n\t" "v_dot4_i32_i8 %6, %15, %23, 0\n\t" "v_dot4_i32_i8 %7, %16, %24, 0\n\t"
            BAMD_SDWA_MUL(0, "%25", 0) BAMD_SDWA_MUL(1, "%25", 1) BAMD_SDWA_MUL(2, "%25", 2) BAMD_SDWA_MUL(3, "%25", 3)
            BAMD_SDWA_MUL(4, "%26", 0) BAMD_SDWA_MUL(5, "%26", 1) BAMD_SDWA_MUL(6, "%26", 2) BAMD_SDWA_MUL(7, "%26", 3)
            "v_add3_u32 %8, %0, %1, %2\n\t" "v_add3_u32 %8, %8, %3, %4\n\t" "v_add3_u32 %8, %8, %5, %6\n\t" "v_add_u32 %8, %8, %7"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(sum)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
              "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(s0), "v"(s1));
    }
#undef BAMD_SDWA_MUL
    return sum;
}
// scale (<= 8 bits) x block dot (<= 15 bits): full-rate 24-bit multiply instead of the quarter-rate v_mul_lo_u32
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
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
// ---- cross-lane helpers: DPP (no LDS-crossbar latency) for everything inside a row of 16 lanes, v_readlane across rows ----
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ int dpp_z(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }   // old = 0: foldable into add / umax
#define DPP_XOR1 0xB1          /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E          /* quad_perm [2,3,0,1] */
#define DPP_HALF_MIRROR 0x141  /* lane i <-> 7-i inside each group of 8 */
#define DPP_MIRROR 0x140       /* lane i <-> 15-i inside each row of 16 */
__device__ __forceinline__ uint32_t umax_(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {             // wave-uniform result
    v = umax_(v, (uint32_t) dpp_z<DPP_XOR1>((int) v)); v = umax_(v, (uint32_t) dpp_z<DPP_XOR2>((int) v));
    v = umax_(v, (uint32_t) dpp_z<DPP_HALF_MIRROR>((int) v)); v = umax_(v, (uint32_t) dpp_z<DPP_MIRROR>((int) v));
    const uint32_t r0 = (uint32_t) __builtin_amdgcn_readlane((int) v, 15), r1 = (uint32_t) __builtin_amdgcn_readlane((int) v, 31);
    const uint32_t r2 = (uint32_t) __builtin_amdgcn_readlane((int) v, 47), r3 = (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
    return umax_(umax_(r0, r1), umax_(r2, r3));
}
__device__ __forceinline__ int group8_sum(int v) {                          // sum over aligned groups of 8 lanes, in every lane
    v += dpp_z<DPP_XOR1>(v); v += dpp_z<DPP_XOR2>(v); v += dpp_z<DPP_HALF_MIRROR>(v);
    return v;
}
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
    const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum_f64(double s) {                  // fixed order; wave-uniform result
    s += dpp_d<DPP_XOR1>(s); s += dpp_d<DPP_XOR2>(s); s += dpp_d<DPP_HALF_MIRROR>(s); s += dpp_d<DPP_MIRROR>(s);
    return ((readlane_d(s, 15) + readlane_d(s, 31)) + readlane_d(s, 47)) + readlane_d(s, 63);
}

// One wave quantises BATCH super-blocks at a time (independent dependency chains interleave); lane l holds the 4
// consecutive elements 4l..4l+3 of a block.  quantize_row_q8_K_ref semantics (ggml-quants.c:3593-3630): the scale comes
// from the FIRST element of largest magnitude (strict > scan), so ties resolve to the lowest lane, lowest element.
#define BAMD_ACT_BATCH 4
// LDS layout of the quantised activations of one mat-vec: q8[nb][64] u32 | S[nb][8] i32 | yd[nb] f32 | (16-byte aligned) red[16] f64
#define BAMD_ACT_RED_OFF(nb) ((((size_t) (nb) * (256 + 32 + 4)) + 15) & ~(size_t) 15)
template <bool NORM>
struct ActPro {
    float4 v[BAMD_ACT_BATCH], w[BAMD_ACT_BATCH];

    // the loads of this wave's first batch of blocks: issued at kernel entry, AHEAD of the bulk weight prefetch, so the
    // (tiny, latency-critical) activation read is not queued behind megabytes of weight requests
    // blocks i0, i0 + bstride, ... below blimit (defaults: this wave's share of the whole vector, interleaved over the waves)
    __device__ __forceinline__ void issue(const float * __restrict__ x, const float * __restrict__ nw, int K, int i0, int bstride = 0, int blimit = 0) {
        const int lane = threadIdx.x & 63;
        if (bstride == 0) { bstride = blockDim.x >> 6; blimit = K >> 8; }
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) {
            const int i = i0 + b * bstride;
            v[b] = i < blimit ? *(const float4 *) (x + i * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (NORM) w[b] = i < blimit ? *(const float4 *) (nw + i * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // Issue-bound code (every CU quantises the whole activation vector: ~1/3 of a decode step's VALU work), so the instruction
    // count per block is what matters here:
    //   - the two IEEE divisions per block (iscale = -127/max, d = 1/iscale) run ONCE per batch: block b's operand sits in lane b;
    //   - nearest_int(v) & 0xff is the low byte of the bits of v + 12582912.f (ggml-quants.c:1632-1637: the mask and the
    //     0x400000 offset do not touch that byte), and MIN(127, .) (:3617) never binds for |iscale * x| <= 127(1 + 2^-23);
    //   - the sum of the four signed bytes is one v_dot4 against 0x01010101;
    //   - the four wave-max chains are interleaved step by step (DPP results need wait states); row_bcast leaves the result in lane 63.
    __device__ __forceinline__ void quantize_batch(float scale, int K, int i0, uint32_t * q8, int * S, float * yd, int bstride = 0, int blimit = 0) {
        const int lane = threadIdx.x & 63;
        const int nwaves = bstride ? bstride : (int) (blockDim.x >> 6), nb = bstride ? blimit : (K >> 8);
        uint32_t amaxb[BAMD_ACT_BATCH];
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) {
            if (NORM) {                                  // y = (x*scale)*w : ggml_vec_scale_f32 then ggml_mul (llama.cpp:7940-7950)
                v[b].x = (v[b].x * scale) * w[b].x; v[b].y = (v[b].y * scale) * w[b].y;
                v[b].z = (v[b].z * scale) * w[b].z; v[b].w = (v[b].w * scale) * w[b].w;
            }
            const float a = fmaxf(fmaxf(fmaxf(fabsf(v[b].x), fabsf(v[b].y)), fabsf(v[b].z)), fabsf(v[b].w));
            amaxb[b] = __float_as_uint(a);               // non-negative floats order like their bit patterns
        }
        uint32_t t[BAMD_ACT_BATCH], wmax[BAMD_ACT_BATCH];
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) t[b] = umax_(amaxb[b], (uint32_t) dpp_z<DPP_XOR1>((int) amaxb[b]));
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) t[b] = umax_(t[b], (uint32_t) dpp_z<DPP_XOR2>((int) t[b]));
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) t[b] = umax_(t[b], (uint32_t) dpp_z<DPP_HALF_MIRROR>((int) t[b]));
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) t[b] = umax_(t[b], (uint32_t) dpp_z<DPP_MIRROR>((int) t[b]));
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) t[b] = umax_(t[b], (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t[b], 0x142, 0xa, 0xf, false));   // row_bcast:15
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) t[b] = umax_(t[b], (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t[b], 0x143, 0xc, 0xf, false));   // row_bcast:31
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) wmax[b] = (uint32_t) __builtin_amdgcn_readlane((int) t[b], 63);
        // the scale comes from the FIRST element of largest magnitude (strict > scan of the reference): lowest lane, lowest element
        float mine[BAMD_ACT_BATCH];
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) {
            const float M = __uint_as_float(wmax[b]);
            const bool ex = fabsf(v[b].x) == M, ey = fabsf(v[b].y) == M, ez = fabsf(v[b].z) == M;
            float m = v[b].w;                            // branch-free selects, lowest element wins
            m = ez ? v[b].z : m; m = ey ? v[b].y : m; m = ex ? v[b].x : m;
            mine[b] = m;
        }
        int mxv = __float_as_int(1.0f);
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) {
            const unsigned long long who = __ballot(amaxb[b] == wmax[b]);
            const int first = __ffsll((long long) who) - 1;
            const int mxb = __builtin_amdgcn_readlane(__float_as_int(mine[b]), first);
            mxv = lane == b ? mxb : mxv;
        }
        const float isc = -127.f / __int_as_float(mxv);  // lane b: block b (other lanes: -127)
        const float dd = 1.0f / isc;
#pragma unroll
        for (int b = 0; b < BAMD_ACT_BATCH; ++b) {
            const int i = i0 + b * nwaves;
            if (i < nb) {                                // wave-uniform
                const float iscale = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(isc), b));
                const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dd), b));
                const bool nz = wmax[b] != 0u;           // all-zero block: q = 0, d = 0 (ggml-quants.c:3607-3612)
                const float t0 = iscale * v[b].x + 12582912.f, t1 = iscale * v[b].y + 12582912.f;
                const float t2 = iscale * v[b].z + 12582912.f, t3 = iscale * v[b].w + 12582912.f;
                const uint32_t p01 = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x0c0c0400u);
                const uint32_t p23 = __builtin_amdgcn_perm(__float_as_uint(t3), __float_as_uint(t2), 0x0c0c0400u);
                uint32_t packed = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                packed = nz ? packed : 0u;
                const int s4 = group8_sum(sdot4(packed, 0x01010101u));
                q8[i * 64 + (lane & 7) * 8 + (lane >> 3)] = packed;
                if ((lane & 7) == 0) S[i * 8 + (lane >> 3)] = s4;
                if (lane == 0) yd[i] = nz ? d : 0.f;
            }
        }
    }

    __device__ __forceinline__ void finish(const float * __restrict__ x, const float * __restrict__ nw, float eps, int K,
                                           uint32_t * q8, int * S, float * yd, double * red) {
        const int lane = threadIdx.x & 63, wave = wave_id(), nwaves = blockDim.x >> 6, nb = K >> 8;
        const int step = nwaves * BAMD_ACT_BATCH;
        float scale = 1.0f;
        if (NORM) {
            // sum of squares in double (ggml.c:11874-11877), fixed tree order instead of the reference's sequential order
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < BAMD_ACT_BATCH; ++b) {
                s += (double) (v[b].x * v[b].x); s += (double) (v[b].y * v[b].y); s += (double) (v[b].z * v[b].z); s += (double) (v[b].w * v[b].w);
            }
            for (int i0 = wave + step; i0 < nb; i0 += step) {          // only for K > 256 * 4 * nwaves
                ActPro<NORM> t; t.issue(x, nw, K, i0);
#pragma unroll
                for (int b = 0; b < BAMD_ACT_BATCH; ++b) {
                    s += (double) (t.v[b].x * t.v[b].x); s += (double) (t.v[b].y * t.v[b].y); s += (double) (t.v[b].z * t.v[b].z); s += (double) (t.v[b].w * t.v[b].w);
                }
            }
            s = wave_sum_f64(s);
            if (lane == 0) red[wave] = s;
            __syncthreads();
            double tot = 0.0;
            for (int w2 = 0; w2 < nwaves; ++w2) tot += red[w2];
            const float mean = (float) (tot / (double) K);
            scale = 1.0f / sqrtf(mean + eps);
        }
        quantize_batch(scale, K, wave, q8, S, yd);
        for (int i0 = wave + step; i0 < nb; i0 += step) {
            ActPro<NORM> t; t.issue(x, nw, K, i0);
            t.quantize_batch(scale, K, i0, q8, S, yd);
        }
        __syncthreads();
    }
};

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

// ===========================================================================================================
// Quantised mat-vec: y = W . Q8_K(x).  One wave = 8 rows at a time (lane = r*8+e), rows streamed sequentially
// over super-blocks so each lane carries exactly the f32 chain of SIMD lane e of the reference.
// ===========================================================================================================
struct RowAcc { float acc, accm; };

// ---- per-record arithmetic -----------------------------------------------------------------------------
struct RecQ4K { uint4 qs, hd; };
struct RecQ5K { uint4 qs, hd; uint32_t qh; };
struct RecQ6K { uint4 ql; uint2 qh, sc; uint32_t d; };

// Pin a loaded register at its point of use: without this, LLVM folds the first ALU op on a ring register into
// the loop PHI (i.e. executes it right after the load, one iteration early), which forces s_waitcnt vmcnt(0) at
// the loop tail and serialises the whole prefetch ring.
__device__ __forceinline__ void pin(uint32_t & x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(uint2 & x) { pin(x.x); pin(x.y); }
__device__ __forceinline__ void pin(uint4 & x) { pin(x.x); pin(x.y); pin(x.z); pin(x.w); }
__device__ __forceinline__ void pin_rec(RecQ4K & R) { pin(R.qs); pin(R.hd); }
__device__ __forceinline__ void pin_rec(RecQ5K & R) { pin(R.qs); pin(R.hd); pin(R.qh); }
__device__ __forceinline__ void pin_rec(RecQ6K & R) { pin(R.ql); pin(R.qh); pin(R.sc); pin(R.d); }

// `rec` is wave-uniform (SGPR pair); the per-lane part is a 32-bit offset, so the loads take the saddr form and need no
// 64-bit VALU address arithmetic.  Weights are read exactly once per token: non-temporal loads keep them out of the way
// of the L2-resident activations (MI355X_MICROARCH.md, row nt-weights).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T ldnt(const uint8_t * rec, uint32_t off) { return __builtin_nontemporal_load((const T *) (rec + off)); }
template <> __device__ __forceinline__ uint4 ldnt<uint4>(const uint8_t * rec, uint32_t off) {
    const u32x4_t v = __builtin_nontemporal_load((const u32x4_t *) (rec + off)); return make_uint4(v.x, v.y, v.z, v.w);
}
template <> __device__ __forceinline__ uint2 ldnt<uint2>(const uint8_t * rec, uint32_t off) {
    const u32x2_t v = __builtin_nontemporal_load((const u32x2_t *) (rec + off)); return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void load_rec(RecQ4K & R, const uint8_t * rec, int lane) {
    const uint32_t l = (uint32_t) lane;
    R.qs = ldnt<uint4>(rec, l * 16u);
    R.hd = ldnt<uint4>(rec, 1024u + (l >> 3) * 16u);
}
__device__ __forceinline__ void load_rec(RecQ5K & R, const uint8_t * rec, int lane) {
    const uint32_t l = (uint32_t) lane;
    R.qs = ldnt<uint4>(rec, l * 16u);
    R.qh = ldnt<uint32_t>(rec, 1024u + l * 4u);
    R.hd = ldnt<uint4>(rec, 1280u + (l >> 3) * 16u);
}
__device__ __forceinline__ void load_rec(RecQ6K & R, const uint8_t * rec, int lane) {
    const uint32_t l = (uint32_t) lane;
    R.ql = ldnt<uint4>(rec, l * 16u);
    R.qh = ldnt<uint2>(rec, 1024u + l * 8u);
    R.sc = ldnt<uint2>(rec, 1536u + (l >> 3) * 16u + ((l >> 2) & 1u) * 8u);
    R.d  = ldnt<unsigned short>(rec, 1664u + (l >> 3) * 2u);
}

// 6-bit scale/min unpack, ggml-quants.c:6928-6933
__device__ __forceinline__ void unpack_k4(const uint4 & hd, uint32_t & sc03, uint32_t & sc47, uint32_t & mn03, uint32_t & mn47) {
    const uint32_t u0 = hd.y, u1 = hd.z, u2 = hd.w;
    sc03 = u0 & 0x3f3f3f3fu; mn03 = u1 & 0x3f3f3f3fu;
    sc47 = (u2 & 0x0f0f0f0fu) | (((u0 >> 6) & 0x03030303u) << 4);
    mn47 = ((u2 >> 4) & 0x0f0f0f0fu) | (((u1 >> 6) & 0x03030303u) << 4);
}
#define BYTE(w, k) (int) (((w) >> (8 * (k))) & 0xffu)

// The terms one super-block contributes to the f32 chains of lane (r, e):
//   d, fs   : acc  = fma(d, fs, acc)                      (all types; fs = (float) of the exact int32 lane sum)
//   dmin, pm: Q4_K: accm = fma(dmin, pm, accm) for l = e&3 ; Q5_K: accm = accm + dmin*pm (pm = all-8 integer sum)
struct Terms { float d, fs, dmin, pm; };

__device__ __forceinline__ Terms block_terms(const RecQ4K & R, int ci, int lane, const uint32_t * q8, const int * S, const float * yd) {
    const int e = lane & 7, l = e & 3;
    const float ydv = yd[ci];
    Terms T;
    T.d = ydv * h2f(R.hd.x & 0xffffu);
    T.dmin = (-ydv) * h2f(R.hd.x >> 16);
    uint32_t sc03, sc47, mn03, mn47; unpack_k4(R.hd, sc03, sc47, mn03, mn47);
    const uint4 a0 = *(const uint4 *) (q8 + ci * 64 + e * 8), a1 = *(const uint4 *) (q8 + ci * 64 + e * 8 + 4);
    const uint32_t wq[8] = { R.qs.x & 0x0f0f0f0fu, (R.qs.x >> 4) & 0x0f0f0f0fu, R.qs.y & 0x0f0f0f0fu, (R.qs.y >> 4) & 0x0f0f0f0fu,
                             R.qs.z & 0x0f0f0f0fu, (R.qs.z >> 4) & 0x0f0f0f0fu, R.qs.w & 0x0f0f0f0fu, (R.qs.w >> 4) & 0x0f0f0f0fu };
    const uint32_t aq[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
    const int sumi = dotscale8<false>(wq, aq, sc03, sc47);
    T.fs = (float) sumi;
    const uint32_t mw = (l < 2) ? mn03 : mn47;
    const int sh = (l & 1) * 16;
    const int ma = (int) ((mw >> sh) & 0xffu), mb = (int) ((mw >> (sh + 8)) & 0xffu);
    const int2 sp = *(const int2 *) (S + ci * 8 + 2 * l);
    T.pm = (float) (mul24(ma, sp.x) + mul24(mb, sp.y));  // 6-bit min x sum of 32 int8
    return T;
}

__device__ __forceinline__ Terms block_terms(const RecQ5K & R, int ci, int lane, const uint32_t * q8, const int * S, const float * yd) {
    const int e = lane & 7;
    const float ydv = yd[ci];
    Terms T;
    T.d = ydv * h2f(R.hd.x & 0xffffu);
    T.dmin = (-ydv) * h2f(R.hd.x >> 16);
    uint32_t sc03, sc47, mn03, mn47; unpack_k4(R.hd, sc03, sc47, mn03, mn47);
    const uint4 a0 = *(const uint4 *) (q8 + ci * 64 + e * 8), a1 = *(const uint4 *) (q8 + ci * 64 + e * 8 + 4);
    const uint32_t qh = R.qh;
#define Q5(w, shift, c) ((((w) >> (shift)) & 0x0f0f0f0fu) | (((qh >> (c)) & 0x01010101u) << 4))
    const uint32_t wq[8] = { Q5(R.qs.x, 0, 0), Q5(R.qs.x, 4, 1), Q5(R.qs.y, 0, 2), Q5(R.qs.y, 4, 3), Q5(R.qs.z, 0, 4), Q5(R.qs.z, 4, 5), Q5(R.qs.w, 0, 6), Q5(R.qs.w, 4, 7) };
    const uint32_t aq[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
    const int sumi = dotscale8<false>(wq, aq, sc03, sc47);
#undef Q5
    T.fs = (float) sumi;
    // hsum(mins . q8sums) over all 8 sub-blocks (:7515-7518): exact integer, any order
    const uint32_t mw = (e < 4) ? mn03 : mn47;
    const int hs = group8_sum(mul24((int) ((mw >> (8 * (e & 3))) & 0xffu), S[ci * 8 + e]));
    T.pm = (float) hs;
    return T;
}

__device__ __forceinline__ Terms block_terms(const RecQ6K & R, int ci, int lane, const uint32_t * q8, const int * S, const float * yd) {
    (void) S;
    const int e = lane & 7;
    Terms T;
    T.d = yd[ci] * h2f(R.d);
    T.dmin = 0.f; T.pm = 0.f;
    const uint4 a0 = *(const uint4 *) (q8 + ci * 64 + e * 8), a1 = *(const uint4 *) (q8 + ci * 64 + e * 8 + 4);
    // (q6 - 32) as int8: q6 in [0,63] -> (q6 + 0x60) ^ 0x80 per byte, no inter-byte carry
#define Q6(lo, hb) ((((lo) | ((hb) << 4)) + 0x60606060u) ^ 0x80808080u)
#define SB(w, k) ((int) (int8_t) ((w) >> (8 * (k))))
    const uint32_t A0 = R.ql.x, B0 = R.ql.y, h0 = R.qh.x, A1 = R.ql.z, B1 = R.ql.w, h1 = R.qh.y;
    const uint32_t wq[8] = { Q6(A0 & 0x0f0f0f0fu, h0 & 0x03030303u), Q6(B0 & 0x0f0f0f0fu, (h0 >> 2) & 0x03030303u),
                             Q6((A0 >> 4) & 0x0f0f0f0fu, (h0 >> 4) & 0x03030303u), Q6((B0 >> 4) & 0x0f0f0f0fu, (h0 >> 6) & 0x03030303u),
                             Q6(A1 & 0x0f0f0f0fu, h1 & 0x03030303u), Q6(B1 & 0x0f0f0f0fu, (h1 >> 2) & 0x03030303u),
                             Q6((A1 >> 4) & 0x0f0f0f0fu, (h1 >> 4) & 0x03030303u), Q6((B1 >> 4) & 0x0f0f0f0fu, (h1 >> 6) & 0x03030303u) };
    const uint32_t aq[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
    const int sumi = dotscale8<true>(wq, aq, R.sc.x, R.sc.y);
#undef Q6
#undef SB
    T.fs = (float) sumi;
    return T;
}

// one step of the reference's per-lane f32 chains (the ONLY place their order is defined)
template <int TYPE>
__device__ __forceinline__ void chain_step(RowAcc & A, float d, float fs, float dmin, float pm) {
    A.acc = fmaf(d, fs, A.acc);
    if (TYPE == BAMD_Q4_K) A.accm = fmaf(dmin, pm, A.accm);                    // _mm_fmadd_ps(dmin, prod, acc_m)
    if (TYPE == BAMD_Q5_K) { const float t = dmin * pm; A.accm = A.accm + t; } // summs += dmin * hsum  (mul, then add)
}

// horizontal reductions at the end of a row (hsum_float_8, ggml-quants.c:47-53, and the acc_m folds), valid in lane e == 0 of
// each 8-lane group: (a_e + a_{e+4}) -> (+ lane e+2) -> (+ lane e+1), the reference's tree, by DPP row_shl:4 / quad_perm.
__device__ __forceinline__ float dpp_f_shl4(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0xf, false)); }
__device__ __forceinline__ float dpp_f_xor2(float v) { return __int_as_float(dpp_i<DPP_XOR2>(__float_as_int(v))); }
__device__ __forceinline__ float dpp_f_xor1(float v) { return __int_as_float(dpp_i<DPP_XOR1>(__float_as_int(v))); }
template <int TYPE>
__device__ __forceinline__ float finish_row(const RowAcc & A) {
    float v = A.acc;
    v = v + dpp_f_shl4(v); v = v + dpp_f_xor2(v); v = v + dpp_f_xor1(v);
    if (TYPE == BAMD_Q4_K) {
        float m = A.accm;
        m = m + dpp_f_xor2(m); m = m + dpp_f_xor1(m);
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

struct ProArgs { const float * x, * nw; float eps; int K; uint32_t * q8; int * S; float * yd; double * red; };
#define BAMD_PRO_ISSUE(ap, pa) (ap).issue((pa).x, (pa).nw, (pa).K, wave_id())
#define BAMD_PRO_FINISH(ap, pa) (ap).finish((pa).x, (pa).nw, (pa).eps, (pa).K, (pa).q8, (pa).S, (pa).yd, (pa).red)

// ---- MODE A: one wave per row-group --------------------------------------------------------------------------
// The wave walks row-groups rg = first, first+stride, ... (count of them).  A register ring of D records is kept
// in flight by a LOADER cursor that runs D records ahead of the consumer and crosses row-group boundaries by
// pure (branch-free, scalar) arithmetic, so the prefetch never drains and the compiler can keep counted
// s_waitcnt vmcnt(N) waits.  The ring is filled BEFORE the activation prologue (weights do not depend on it), so
// the first HBM round trip overlaps the RMSNorm/Q8_K work.  With PAIR each row-group is streamed twice back to
// back — gate (wA) then up (wB) — and the epilogue fuses silu(gate)*up.
template <int TYPE, typename REC, int D, int EPI, int PRO>
__device__ __forceinline__ void stream_segment(const uint8_t * __restrict__ wA, const uint8_t * __restrict__ wB, int nb,
                                               int first, int count, int stride, float * __restrict__ out,
                                               const float * __restrict__ res, const ProArgs & pa, bool do_pro,
                                               unsigned long long & best, int nvalid) {
    constexpr int RECB = TYPE == BAMD_Q4_K ? 1152 : TYPE == BAMD_Q5_K ? 1408 : 1680;
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
    constexpr int NPARTS = PAIR ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const long rgb = (long) nb * RECB;                   // D divides nb (chosen by the dispatcher below)
    const long rg_step = (long) stride * rgb;
    const int chunks = nb / D;
    ActPro<PRO == BAMD_PRO_NORM> ap;
    if (do_pro) BAMD_PRO_ISSUE(ap, pa);                  // activation loads go out FIRST (see ActPro::issue)
    REC ring[D];
    // The loader runs exactly one CHUNK (D records = the whole ring) ahead of the consumer: slot s is refilled, right after it
    // is consumed, with record s of the chunk that follows in this wave's sequence (next chunk of the row, else the other half
    // of a gate/up pair, else the next row-group).  One wave-uniform base address per chunk: the per-record cost of the cursor
    // is a constant offset, and the loads stay unconditional so the compiler keeps counted s_waitcnt vmcnt(N) waits.
    const uint8_t * rowA = wA + (long) first * rgb;
#pragma unroll
    for (int s = 0; s < D; ++s) load_rec(ring[s], rowA + s * RECB, lane);
    if (do_pro) BAMD_PRO_FINISH(ap, pa);
    const uint32_t * q8 = pa.q8; const int * S = pa.S; const float * yd = pa.yd;
    for (int r = 0; r < count; ++r) {
        const int rg = first + r * stride;
        const int row = rg * 8 + (lane >> 3);
        const long rowoff = (long) rg * rgb;
        float gate_val = 0.f;
#pragma unroll
        for (int part = 0; part < NPARTS; ++part) {
            const uint8_t * pbase = (part ? wB : wA) + rowoff;
            // after the last chunk of this row-part: the other half of the pair, the next row-group, or — at the very end of the
            // wave's stream — its own last record again, D times (step 0: one record of redundant traffic, never consumed; the
            // requests stay unconditional so that the waits stay counted)
            const bool last = !(PAIR && part == 0) && r + 1 >= count;
            const uint8_t * after = (PAIR && part == 0) ? wB + rowoff : (last ? pbase + (long) (nb - 1) * RECB : wA + rowoff + rg_step);
            // residual fetched at the START of the row: by the epilogue it is the oldest outstanding load
            float resv = 0.f;
            if (EPI == BAMD_EPI_ADD && row < nvalid) resv = res[row];
            RowAcc A = { 0.f, 0.f };
            for (int c = 0; c < chunks; ++c) {
                const bool inrow = c + 1 < chunks;
                const uint8_t * nxt = inrow ? pbase + (long) (c + 1) * (D * RECB) : after;
                const int step = (inrow || !last) ? RECB : 0;
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    pin_rec(ring[s]);
                    const Terms T = block_terms(ring[s], c * D + s, lane, q8, S, yd);
                    chain_step<TYPE>(A, T.d, T.fs, T.dmin, T.pm);
                    load_rec(ring[s], nxt + s * step, lane);
                    if ((s & (BAMD_SCHED_GROUP - 1)) == BAMD_SCHED_GROUP - 1)
                        __builtin_amdgcn_sched_barrier(0);   // keep hipcc from clustering the refills at the loop tail
                }
            }
            const float val = finish_row<TYPE>(A);
            if (PAIR) {
                if (part == 0) gate_val = val;
                else if ((lane & 7) == 0 && row < nvalid) out[row] = v_silu(gate_val) * val;
            } else if ((lane & 7) == 0 && row < nvalid) {
                float o = val;
                if (EPI == BAMD_EPI_ADD) o = val + resv;
                out[row] = o;
                if (EPI == BAMD_EPI_ARGMAX) { const unsigned long long k = argmax_key(o, row); best = k > best ? k : best; }
            }
        }
    }
}

template <int TYPE, typename REC, int EPI, int PRO>
__device__ __forceinline__ void stream_dispatch_depth(const uint8_t * wA, const uint8_t * wB, int nb, int first, int count, int stride,
                                                      float * out, const float * res, const ProArgs & pa, bool do_pro,
                                                      unsigned long long & best, int nvalid) {
    if ((nb & 7) == 0)      stream_segment<TYPE, REC, 8, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, do_pro, best, nvalid);
    else if ((nb & 3) == 0) stream_segment<TYPE, REC, 4, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, do_pro, best, nvalid);
    else if ((nb & 1) == 0) stream_segment<TYPE, REC, 2, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, do_pro, best, nvalid);
    else                    stream_segment<TYPE, REC, 1, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, do_pro, best, nvalid);
}

__device__ __forceinline__ ProArgs carve_lds(const bamd_mv_args & a, unsigned char * smem) {
    const int nb = a.K >> 8;
    ProArgs pa;
    pa.x = a.x; pa.nw = a.normw; pa.eps = a.eps; pa.K = a.K;
    pa.q8 = (uint32_t *) smem; pa.S = (int *) (pa.q8 + nb * 64); pa.yd = (float *) (pa.S + nb * 8);
    pa.red = (double *) (smem + BAMD_ACT_RED_OFF(nb));     // byte offsets, never a pointer->integer->pointer round trip: that loses
                                                           // the LDS address space and turns every access into a FLAT instruction
    return pa;
}

template <int PRO, int EPI>
__global__ void __launch_bounds__(512) matvec_kernel(bamd_mv_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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
}

// ---- MODE B: split-K, one 8-wave workgroup per row-group ------------------------------------------------------
// For matrices with few row-groups (wq/wk/wv/wo, ffn_down: 512..768 of them) one wave per row-group leaves the chip
// short of bytes in flight.  Here the 8 waves of a workgroup share a row-group: wave w streams super-blocks
// [w*nb/8, (w+1)*nb/8) and writes the per-block TERMS (d, fs, dmin, pm — exact integers already converted) to LDS;
// after a workgroup barrier ONE wave replays the reference's sequential f32 chain over all nb blocks in order.
// Same arithmetic, same order, 8x the parallelism.  Term buffers are double-buffered so the chain of row-group n
// overlaps the streaming of row-group n+1; the prefetch ring spans row-group boundaries (M row-groups per body).
// LDS term buffers: 2 (double buffer) x M (row-groups per batch) x nb x 64 lanes x float4 {d, fs, dmin, pm}
#define BAMD_TERM_FLOATS(nb) ((size_t) (nb) * 256)      /* one float4 {d, fs, dmin, pm} per lane per super-block */

template <int TYPE, typename REC, int NBW, int M, int NBUF, int EPI, int PRO>
__device__ __forceinline__ void split_stream(const uint8_t * __restrict__ w, int nb, int first, int count, int stride,
                                             float * __restrict__ out, const float * __restrict__ res, const ProArgs & pa, bool do_pro,
                                             float * part0, int & batchctr, int nvalid) {
    constexpr int RECB = TYPE == BAMD_Q4_K ? 1152 : TYPE == BAMD_Q5_K ? 1408 : 1680;
    constexpr int D = NBW * M;                               // ring depth = one batch (M row-groups) of this wave's records
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int r8 = lane >> 3, l4 = lane & 3;
    const long rgb = (long) nb * RECB;
    const long rg_step = (long) stride * rgb;
    const int i0 = wave * NBW;                               // this wave's first super-block inside a row
    const size_t rg_floats = BAMD_TERM_FLOATS(nb);
    // PLAIN prologue: wave w consumes only the activations of its own K-slice (blocks i0 .. i0+NBW-1), so it quantises exactly
    // those — no workgroup barrier, and a wave starts on its records as soon as ITS blocks are done.  (NORM needs the sum of
    // squares of the whole vector: shared prologue as in mode A.)
    constexpr bool OWN = PRO == BAMD_PRO_PLAIN;
    ActPro<PRO == BAMD_PRO_NORM> ap, ap2;
    if (do_pro) {
        if (OWN) { ap.issue(pa.x, pa.nw, pa.K, i0, 1, i0 + NBW); if (NBW > BAMD_ACT_BATCH) ap2.issue(pa.x, pa.nw, pa.K, i0 + BAMD_ACT_BATCH, 1, i0 + NBW); }
        else BAMD_PRO_ISSUE(ap, pa);                         // activation loads go out FIRST
    }
    // ring slot (m, j) holds record i0+j of row-group r0+m; after it is consumed it is refilled with the same record of row-group
    // r0+M+m, i.e. a constant M*rg_step further on: the loader needs one wave-uniform base per batch and nothing per record
    const uint8_t * bbase = w + (long) first * rgb + (long) i0 * RECB;
    REC ring[D];
    STAMP(0);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        if (m < count) {                                     // no redundant requests when the stream is short
#pragma unroll
            for (int j = 0; j < NBW; ++j) load_rec(ring[m * NBW + j], bbase + (long) m * rg_step + j * RECB, lane);
        }
    }
    STAMP(1);
    if (do_pro) {
        if (OWN) {
            static_assert(NBW <= 2 * BAMD_ACT_BATCH, "own-slice prologue handles two batches");
            ap.quantize_batch(1.0f, pa.K, i0, pa.q8, pa.S, pa.yd, 1, i0 + NBW);
            if (NBW > BAMD_ACT_BATCH) ap2.quantize_batch(1.0f, pa.K, i0 + BAMD_ACT_BATCH, pa.q8, pa.S, pa.yd, 1, i0 + NBW);
        } else BAMD_PRO_FINISH(ap, pa);
    }
    STAMP(2);
    const uint32_t * q8 = pa.q8; const int * S = pa.S; const float * yd = pa.yd;
    for (int r0 = 0; r0 < count; r0 += M) {
        const int nbatch = count - r0 < M ? count - r0 : M;  // workgroup-uniform
        float * B0 = part0 + (NBUF == 2 ? (size_t) (batchctr & 1) * M * rg_floats : (size_t) 0);
        // the wave that will run the chain of row-group r0+wave fetches its residual now (old by chain time)
        const int crow = (first + (r0 + (wave < nbatch ? wave : 0)) * stride) * 8 + r8;
        float resv = 0.f;
        if (EPI == BAMD_EPI_ADD && crow < nvalid) resv = res[crow];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (m < nbatch) {
                float4 * P = (float4 *) (B0 + (size_t) m * rg_floats);
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const int s = m * NBW + j;
                    const int ci = i0 + j;
                    pin_rec(ring[s]);
                    const Terms T = block_terms(ring[s], ci, lane, q8, S, yd);
                    P[ci * 64 + lane] = make_float4(T.d, T.fs, T.dmin, T.pm);   // every lane owns the terms of its chain: one 16-byte store
                    if (r0 + M + m < count) load_rec(ring[s], bbase + (long) (M + m) * rg_step + j * RECB, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        STAMP(3);
        __syncthreads();
        STAMP(4);
        if (wave < nbatch) {
            // the reference's chains, in order, for lane (r, e)   (ggml-quants.c:6937-6941, :6970, :7518, :8219)
            const float4 * P = (const float4 *) (B0 + (size_t) wave * rg_floats);
            RowAcc A = { 0.f, 0.f };
            for (int i = 0; i < nb; i += 8) {                // nb % 8 == 0 here; the 16-byte LDS reads of 8 blocks issued together
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = P[(i + u) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) chain_step<TYPE>(A, t[u].x, t[u].y, t[u].z, t[u].w);
            }
            const float val = finish_row<TYPE>(A);
            if ((lane & 7) == 0 && crow < nvalid) out[crow] = EPI == BAMD_EPI_ADD ? val + resv : val;
            STAMP(5);
        }
        batchctr += 1;
        bbase += (long) M * rg_step;
        if (NBUF == 1 && r0 + M < count) __syncthreads();    // single term buffer: the chains must be done before the next batch writes
    }
}

template <int TYPE, typename REC, int EPI, int PRO>
__device__ __forceinline__ void split_dispatch(const uint8_t * w, int nb, int first, int count, int stride, float * out, const float * res,
                                               const ProArgs & pa, bool do_pro, float * part0, int & rgctr, int nvalid) {
    const int nbw = nb >> 3;
    // (records per wave per row-group, row-groups per batch, term buffers): the batch is the prefetch depth.  K = 14336 with M = 2
    // (all of ffn_down's work per workgroup in flight from the first instruction, single-buffered) measured no better for Q4_K and
    // 14 % worse for Q6_K than M = 1: the kernel is instruction-issue bound, not latency bound.
    if (nbw == 2)       split_stream<TYPE, REC, 2, 4, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, do_pro, part0, rgctr, nvalid);
    else if (nbw == 7)  split_stream<TYPE, REC, 7, 1, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, do_pro, part0, rgctr, nvalid);
    else if (nbw == 4)  split_stream<TYPE, REC, 4, 2, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, do_pro, part0, rgctr, nvalid);
    else if (nbw == 1)  split_stream<TYPE, REC, 1, 8, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, do_pro, part0, rgctr, nvalid);
    else __builtin_trap();                               // the launcher only picks this kernel for the shapes above
}

// host must check bamd_split_supported(nb) before choosing this kernel
template <int PRO, int EPI>
__global__ void __launch_bounds__(512) matvec_split_kernel(bamd_mv_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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
}

// ===========================================================================================================
// Step begin: pick the token of this step (forced prompt token, or the arg-max of the previous step's logits),
// advance the position, and dequantise its embedding row into the residual stream.
// ===========================================================================================================
__device__ __forceinline__ void get_scale_min_k4(int j, const uint8_t * q, int & d, int & m) {
    if (j < 4) { d = q[j] & 63; m = q[j + 4] & 63; }
    else { d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4); }
}

// get_rows of one token (ggml.c:13186-13228 -> dequantize_row_*): row `tok` of the embedding matrix (GGUF layout) -> x[E]
__device__ __forceinline__ void embed_row(const uint8_t * embd, int embd_type, int E, int tok, float * x) {
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

__global__ void __launch_bounds__(1024) step_begin_kernel(bamd_step_state * st, const int32_t * forced, int n_forced,
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
    embed_row(embd, embd_type, E, tok_s, x);
}

// batched prefill: one workgroup per token of the micro-batch
__global__ void __launch_bounds__(256) embed_batch_kernel(const int32_t * __restrict__ tokens, const uint8_t * embd, int embd_type, int E, int V, float * x) {
    int tok = tokens[blockIdx.x];
    if (tok < 0 || tok >= V) tok = 0;
    embed_row(embd, embd_type, E, tok, x + (size_t) blockIdx.x * E);
}

// ===========================================================================================================
// Attention (single token): RoPE + KV store + scores + softmax + P.V         (reference: llm_build_kv, llama.cpp:8318)
// ===========================================================================================================
// KV cache, "chain-major" physical order (logically the reference's K [n_ctx][Hkv*hd] f16 and V^T [Hkv*hd][n_ctx] f16,
// llama.cpp:7845-7875; bamd_op_attention converts at the boundary):
//   K : inside each head row, element n = 8l + e is stored at index e*(hd/8) + l
//   V^T: inside each row, position p = 64B + 8l + e is stored at index 64B + 8e + l
// The reference's attention mat-muls (tinyBLAS, sgemm.cpp:405-431) keep 8 SIMD lanes e, each a sequential f32 chain over
// the steps l.  With this order the wave lane that stands for SIMD lane e finds the operands of consecutive steps
// CONTIGUOUS: 16-byte loads straight from HBM/L2, no LDS staging, no gather.
__device__ __forceinline__ int kperm(int n, int L) { return (n & 7) * L + (n >> 3); }
__device__ __forceinline__ int vperm(int p) { return (p & ~63) + ((p & 7) << 3) + ((p & 63) >> 3); }

// dot of up to 32 steps for lane e: k8 = this lane's L halves of the K row (L <= 32), q = this lane's L floats / halves
template <bool PREFILL>
__device__ __forceinline__ float kq_chain(const uint4 (&kv)[4], int L, const float * qf, const unsigned short * qh) {
    if (!PREFILL) {
        float acc = 0.f;                                           // tinyBLAS F16 x F32, KN = 8 (sgemm.cpp:405-431)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g * 8 < L) {
                const uint32_t w[4] = { kv[g].x, kv[g].y, kv[g].z, kv[g].w };
                const float4 qa = *(const float4 *) (qf + g * 8), qb = *(const float4 *) (qf + g * 8 + 4);
                const float qv[8] = { qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w };
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fmaf(h2f((w[u >> 1] >> (16 * (u & 1))) & 0xffffu), qv[u], acc);
            }
        }
        return acc;
    } else {
        float a4[4] = { 0.f, 0.f, 0.f, 0.f };                      // ggml_vec_dot_f16: 4 accumulators x 8 lanes (ggml.c:2038)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g * 8 < L) {
                const uint32_t w[4] = { kv[g].x, kv[g].y, kv[g].z, kv[g].w };
                const uint4 qq = *(const uint4 *) (qh + g * 8);
                const uint32_t qw[4] = { qq.x, qq.y, qq.z, qq.w };
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    a4[u & 3] = fmaf(h2f((w[u >> 1] >> (16 * (u & 1))) & 0xffffu), h2f((qw[u >> 1] >> (16 * (u & 1))) & 0xffffu), a4[u & 3]);
            }
        }
        const float s02 = a4[0] + a4[2], s13 = a4[1] + a4[3];
        return s02 + s13;
    }
}
__device__ __forceinline__ float hsum8_tinyblas(float v) { v = v + dpp_f_shl4(v); v = v + dpp_f_xor2(v); v = v + dpp_f_xor1(v); return v; }
__device__ __forceinline__ float hsum8_vecdot(float v) { v = v + dpp_f_shl4(v); v = v + dpp_f_xor1(v); v = v + dpp_f_xor2(v); return v; }   // lo+hi, then two hadd_ps

// RoPE (NORM mode, adjacent pairs; ggml.c:14130-14143) of `nheads` consecutive heads of src into chain-major LDS copies
__device__ __forceinline__ void rope_heads(const float * src, const float * rope, int hd, int nheads, float * qt, unsigned short * q16t,
                                           unsigned short * k16t) {
    const int L = hd >> 3;
    for (int i = threadIdx.x; i < nheads * (hd / 2); i += blockDim.x) {
        const int hh = i / (hd / 2), p = i - hh * (hd / 2);
        const float c = rope[2 * p], s = rope[2 * p + 1];
        const float x0 = src[hh * hd + 2 * p], x1 = src[hh * hd + 2 * p + 1];
        const float t0 = x0 * c, t1 = x1 * s, t2 = x0 * s, t3 = x1 * c;
        const float r0 = t0 - t1, r1 = t2 + t3;
        const int i0 = hh * hd + kperm(2 * p, L), i1 = hh * hd + kperm(2 * p + 1, L);
        if (qt) { qt[i0] = r0; qt[i1] = r1; q16t[i0] = f2h(r0); q16t[i1] = f2h(r1); }
        else { k16t[i0] = f2h(r0); k16t[i1] = f2h(r1); }
    }
}

// ---- long contexts: three launches (scores | softmax | P.V), positions / rows spread over many workgroups ------------------
// grid (Hkv, tiles of 64 positions), block 512 = 8 waves x (8 positions x 8 lanes); the GQ query heads of a KV head share K
template <int GQ>
__global__ void __launch_bounds__(512) attn_qk_kernel(bamd_attn_args a) {
    __shared__ __attribute__((aligned(16))) float qt[GQ * 256];
    __shared__ __attribute__((aligned(16))) unsigned short q16t[GQ * 256];
    __shared__ __attribute__((aligned(16))) unsigned short k16t[256];
    const bamd_step_state * st = a.st;
    const int pos = st->pos, n_kv = st->n_kv;
    const int hd = a.hd, Hkv = a.Hkv, Ekv = Hkv * hd, n_ctx = a.n_ctx, L = hd >> 3;
    const int hk = blockIdx.x;
    const float * rope = a.rope + (size_t) pos * hd;
    rope_heads(a.q + (size_t) hk * GQ * hd, rope, hd, GQ, qt, q16t, nullptr);
    rope_heads(a.k + (size_t) hk * hd, rope, hd, 1, nullptr, nullptr, k16t);
    __syncthreads();
    // KV store by the block that owns the tile of `pos` — llm_build_kv_store, llama.cpp:7830-7875
    if ((int) blockIdx.y == ((pos >> 6) % (int) gridDim.y)) {
        for (int i = threadIdx.x; i < hd; i += blockDim.x) {
            a.kc[(size_t) pos * Ekv + hk * hd + i] = k16t[i];
            a.vc[(size_t) (hk * hd + i) * n_ctx + vperm(pos)] = f2h(a.v[hk * hd + i]);
        }
    }
    const int lane = threadIdx.x & 63, wave = wave_id(), e = lane & 7;
    const int tiles = (n_kv + 63) >> 6;
    for (int tile = blockIdx.y; tile < tiles; tile += gridDim.y) {
        const int i = tile * 64 + wave * 8 + (lane >> 3);        // position
        if (i >= n_kv) continue;
        float sc[GQ];
#pragma unroll
        for (int g = 0; g < GQ; ++g) sc[g] = -INFINITY;          // masked (KQ_mask, llama.cpp:14152-14200)
        if (i <= pos) {
            uint4 kreg[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                kreg[g] = make_uint4(0, 0, 0, 0);
                if (g * 8 < L) kreg[g] = i == pos ? *(const uint4 *) (k16t + e * L + g * 8) : *(const uint4 *) (a.kc + (size_t) i * Ekv + hk * hd + e * L + g * 8);
            }
#pragma unroll
            for (int g = 0; g < GQ; ++g) {
                const float v = a.prefill_mode ? kq_chain<true>(kreg, L, nullptr, q16t + g * hd + e * L) : kq_chain<false>(kreg, L, qt + g * hd + e * L, nullptr);
                sc[g] = a.prefill_mode ? hsum8_vecdot(v) : hsum8_tinyblas(v);
            }
        }
        if (e == 0) {
#pragma unroll
            for (int g = 0; g < GQ; ++g) a.scores[(size_t) (hk * GQ + g) * n_ctx + i] = sc[g];
        }
    }
}

// softmax over n_kv scores of one head: grid (H), block 256.  ggml.c:13682-13778 + :2619-2671 (AVX2 branch).
// Probabilities are written back in the V^T position order (vperm) so the P.V lanes read them contiguously.
__global__ void __launch_bounds__(256) attn_softmax_kernel(bamd_attn_args a) {
    __shared__ float redf[4];
    __shared__ double redd[4];
    const bamd_step_state * st = a.st;
    const int n_kv = st->n_kv, n_ctx = a.n_ctx;
    const int h = blockIdx.x;
    float * s = a.scores + (size_t) h * n_ctx;
    const float scale = a.kq_scale;
    const int lane = threadIdx.x & 63, wave = wave_id();
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n_kv; i += blockDim.x) { const float w = s[i] * scale; mx = w > mx ? w : mx; }
    for (int o = 32; o; o >>= 1) { const float om = __shfl_xor(mx, o); mx = om > mx ? om : mx; }
    if (lane == 0) redf[wave] = mx;
    __syncthreads();
    mx = redf[0]; for (int w = 1; w < 4; ++w) mx = redf[w] > mx ? redf[w] : mx;
    double sum = 0.0;
    for (int i = threadIdx.x; i < n_kv; i += blockDim.x) {
        const float w = s[i] * scale;
        const float val = v_expf(w - mx);
        s[i] = val;                                              // same index this thread just read: no hazard
        const float c = hsum8_tinyblas(val);                     // the reference's 8-wide partial sum (same tree shape)
        if ((lane & 7) == 0) sum += (double) c;
    }
    sum = wave_sum_f64(sum);
    if (lane == 0) redd[wave] = sum;
    __syncthreads();
    double tot = 0.0; for (int w = 0; w < 4; ++w) tot += redd[w];
    const float fs = (float) (1.0 / tot);
    for (int i = threadIdx.x; i < n_kv; i += blockDim.x) a.probs[(size_t) h * n_ctx + vperm(i)] = s[i] * fs;
}

// P.V: grid (Hkv, hd/8), block 64: lane = d_local*8 + e carries the tinyBLAS chain Cv[e] of output (h, d) for the GQ heads
// that share this KV head.  sgemm.cpp:405-431 with A = V^T rows (f16), B = p (f32).
template <int GQ>
__global__ void __launch_bounds__(64) attn_pv_kernel(bamd_attn_args a) {
    const bamd_step_state * st = a.st;
    const int n_kv = st->n_kv, n_ctx = a.n_ctx, hd = a.hd;
    const int hk = blockIdx.x;
    const int lane = threadIdx.x, e = lane & 7;
    const int d = blockIdx.y * 8 + (lane >> 3);
    const unsigned short * vrow = a.vc + (size_t) (hk * hd + d) * n_ctx;
    const float * p = a.probs + (size_t) (hk * GQ) * n_ctx;
    float acc[GQ];
#pragma unroll
    for (int g = 0; g < GQ; ++g) acc[g] = 0.f;
    for (int b0 = 0; b0 < n_kv; b0 += 64) {                      // one 64-position block = 8 chain steps per lane
        const uint4 vv = *(const uint4 *) (vrow + b0 + e * 8);
        const uint32_t w[4] = { vv.x, vv.y, vv.z, vv.w };
        const int nstep = n_kv - b0 >= 64 ? 8 : (n_kv - b0) >> 3;
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
            const float4 pa = *(const float4 *) (p + (size_t) g * n_ctx + b0 + e * 8), pb = *(const float4 *) (p + (size_t) g * n_ctx + b0 + e * 8 + 4);
            const float pv[8] = { pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w };
#pragma unroll
            for (int u = 0; u < 8; ++u) if (u < nstep) acc[g] = fmaf(h2f((w[u >> 1] >> (16 * (u & 1))) & 0xffffu), pv[u], acc[g]);
        }
    }
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
        const float v = hsum8_tinyblas(acc[g]);
        if (e == 0) a.out[(size_t) (hk * GQ + g) * hd + d] = v;
    }
}

// ---- ONE launch per layer, one workgroup per QUERY head (and per token of a prefill micro-batch) ---------------------------
// scores and probabilities live in dynamic LDS (2 x n_ctx floats).  Single-token decode uses this kernel up to
// BAMD_ATTN_FUSED_MAX positions (beyond that one workgroup per head no longer has the bandwidth: three-kernel path);
// batched prefill, with T x H workgroups, up to BAMD_ATTN_BATCH_MAX.
#define BAMD_ATTN_FUSED_MAX 2048
#define BAMD_ATTN_BATCH_MAX 8192
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) attn_fused_kernel(bamd_attn_args a, int gq) {
    __shared__ __attribute__((aligned(16))) float qt[256];
    __shared__ __attribute__((aligned(16))) unsigned short q16t[256];
    __shared__ __attribute__((aligned(16))) unsigned short k16t[256];
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_dyn[];
    float * sc = (float *) attn_dyn;                                         // [n_ctx] scores, then exp values (natural order)
    float * pt = sc + a.n_ctx;                                               // [n_ctx] probabilities in V^T position order
    __shared__ float redf[8];
    __shared__ double redd[8];
    const bamd_step_state * st = a.st;
    // batched prefill (a.batch): blockIdx.y = token of the micro-batch; its K/V rows and those of the earlier tokens of the batch
    // were stored by kv_store_batch_kernel, and masked positions are exact no-ops, so each token uses its own padded length
    const int tokb = a.batch ? (int) blockIdx.y : 0;
    const int pos = st->pos + tokb;
    int n_kv = st->n_kv;
    if (a.batch) { n_kv = (pos + 1 + 31) / 32 * 32; n_kv = n_kv < st->n_ctx ? n_kv : st->n_ctx; }
    a.q += (size_t) tokb * a.ld_qkv; a.k += (size_t) tokb * a.ld_qkv; a.v += (size_t) tokb * a.ld_qkv; a.out += (size_t) tokb * a.ld_out;
    const int hd = a.hd, Hkv = a.Hkv, Ekv = Hkv * hd, n_ctx = a.n_ctx, L = hd >> 3;
    const int h = blockIdx.x, hk = h / gq;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), e = lane & 7;
    const float * rope = a.rope + (size_t) pos * hd;
    STAMP(0);
    // requests that do not depend on RoPE go out first: this lane's K chunks of the first 4 x 64 positions and its V^T chunks
    const int r_pos = wave * 8 + (lane >> 3);                     // position inside a 64-tile (scores) / d inside a 64-block (P.V)
    uint4 kreg[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = t * 64 + r_pos;
#pragma unroll
        for (int g = 0; g < 4; ++g) kreg[t][g] = (i < n_kv && i < pos && g * 8 < L) ? *(const uint4 *) (a.kc + (size_t) i * Ekv + hk * hd + e * L + g * 8) : make_uint4(0, 0, 0, 0);
    }
    // ... and its V^T chunks: the first 4 blocks of 64 positions of rows d = r_pos and r_pos + 64 (consumed after the softmax)
    uint4 vreg[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
            const int d = r_pos + 64 * dd;
            vreg[t][dd] = (t * 64 < n_kv && d < hd) ? *(const uint4 *) (a.vc + (size_t) (hk * hd + d) * n_ctx + t * 64 + e * 8) : make_uint4(0, 0, 0, 0);
        }
    }
    rope_heads(a.q + (size_t) h * hd, rope, hd, 1, qt, q16t, nullptr);
    rope_heads(a.k + (size_t) hk * hd, rope, hd, 1, nullptr, nullptr, k16t);
    __syncthreads();
    // KV store by the first query head of each KV head — llm_build_kv_store, llama.cpp:7830-7875
    if (h == hk * gq && !a.batch) {
        for (int i = tid; i < hd; i += blockDim.x) {
            a.kc[(size_t) pos * Ekv + hk * hd + i] = k16t[i];
            a.vc[(size_t) (hk * hd + i) * n_ctx + vperm(pos)] = f2h(a.v[hk * hd + i]);
        }
    }
    STAMP(1);
    // ---- scores ----
#define BAMD_SCORE_TILE(t0_, KL_) do { \
        const int i = (t0_) + r_pos; \
        float v = -INFINITY;                                       /* masked (KQ_mask, llama.cpp:14152-14200) */ \
        if (i < n_kv && i <= pos) { \
            if (i == pos) {                                        /* this token's K row is not visible in the cache yet */ \
                _Pragma("unroll") for (int g = 0; g < 4; ++g) if (g * 8 < L) KL_[g] = *(const uint4 *) (k16t + e * L + g * 8); \
            } \
            v = a.prefill_mode ? hsum8_vecdot(kq_chain<true>(KL_, L, nullptr, q16t + e * L)) : hsum8_tinyblas(kq_chain<false>(KL_, L, qt + e * L, nullptr)); \
        } \
        if (e == 0 && i < n_kv) sc[i] = v; \
    } while (0)
#pragma unroll
    for (int t = 0; t < 4; ++t) { if (t * 64 < n_kv) BAMD_SCORE_TILE(t * 64, kreg[t]); }
    for (int t0 = 256; t0 < n_kv; t0 += 64) {
        const int i2 = t0 + r_pos;
        uint4 kl[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) kl[g] = (i2 < n_kv && i2 < pos && g * 8 < L) ? *(const uint4 *) (a.kc + (size_t) i2 * Ekv + hk * hd + e * L + g * 8) : make_uint4(0, 0, 0, 0);
        BAMD_SCORE_TILE(t0, kl);
    }
#undef BAMD_SCORE_TILE
    __syncthreads();
    STAMP(2);
    // ---- softmax (ggml.c:13682-13778 + :2619-2671): wp = s*scale (+mask), max, exp, 8-chunk f32 sums, double total ----
    const float scale = a.kq_scale;
    float mx = -INFINITY;
    for (int i = tid; i < n_kv; i += blockDim.x) { const float w = sc[i] * scale; mx = w > mx ? w : mx; }
    {   // -inf..inf floats: order-preserving key for an unsigned max
        uint32_t u = __float_as_uint(mx); u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        u = wave_max_u32(u);
        if (lane == 0) redf[wave] = __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
    }
    __syncthreads();
    mx = redf[0];
    for (int w = 1; w < 8; ++w) mx = redf[w] > mx ? redf[w] : mx;
    double sum = 0.0;
    for (int i = tid; i < n_kv; i += blockDim.x) {                 // n_kv % 32 == 0: 8-lane groups are all-active or all-idle
        const float w = sc[i] * scale;
        const float val = v_expf(w - mx);
        sc[i] = val;
        const float c = hsum8_tinyblas(val);
        if (e == 0) sum += (double) c;
    }
    sum = wave_sum_f64(sum);
    if (lane == 0) redd[wave] = sum;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < 8; ++w) tot += redd[w];
    const float fs = (float) (1.0 / tot);
    for (int i = tid; i < n_kv; i += blockDim.x) pt[vperm(i)] = sc[i] * fs;
    // half-filled last block (n_kv % 64 == 32): p = 0 for the missing positions, so the chain steps there are exact no-ops
    for (int i = n_kv + tid; i < ((n_kv + 63) & ~63); i += blockDim.x) pt[vperm(i)] = 0.f;
    __syncthreads();
    STAMP(3);
    // ---- P.V: lane (d, e) carries the tinyBLAS chain Cv[e] of output d (A = V^T row, B = p), up to 4 rows d per lane ----
    const unsigned short vcur[4] = { f2h(a.v[hk * hd + (r_pos < hd ? r_pos : 0)]), f2h(a.v[hk * hd + (r_pos + 64 < hd ? r_pos + 64 : 0)]),
                                     f2h(a.v[hk * hd + (r_pos + 128 < hd ? r_pos + 128 : 0)]), f2h(a.v[hk * hd + (r_pos + 192 < hd ? r_pos + 192 : 0)]) };
    float acc4[4] = { 0.f, 0.f, 0.f, 0.f };
    const int pblk = pos & ~63, pe = pos & 7, pl = (pos & 63) >> 3;   // where this token's own V element sits
#define BAMD_PV_BLOCK(b0_, dd_, VV_) do { \
        const float4 pa = *(const float4 *) (pt + (b0_) + e * 8), pb = *(const float4 *) (pt + (b0_) + e * 8 + 4); \
        const float pv[8] = { pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w }; \
        uint32_t w[4] = { (VV_).x, (VV_).y, (VV_).z, (VV_).w }; \
        if ((b0_) == pblk && e == pe) {                            /* column `pos` is being written by another workgroup: splice it in */ \
            const uint32_t keep = (pl & 1) ? 0x0000ffffu : 0xffff0000u, ins = (pl & 1) ? (uint32_t) vcur[dd_] << 16 : (uint32_t) vcur[dd_]; \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) if (j == (pl >> 1)) w[j] = (w[j] & keep) | ins; \
        } \
        float acc = acc4[dd_]; \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) acc = fmaf(h2f((w[u >> 1] >> (16 * (u & 1))) & 0xffffu), pv[u], acc); \
        acc4[dd_] = acc; \
    } while (0)
#pragma unroll
    for (int t = 0; t < 4; ++t) {                                  // blocks whose V chunks were requested at kernel entry
        if (t * 64 < n_kv) {
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) if (r_pos + 64 * dd < hd) BAMD_PV_BLOCK(t * 64, dd, vreg[t][dd]);
#pragma unroll
            for (int dd = 2; dd < 4; ++dd) if (r_pos + 64 * dd < hd) {  // hd > 128
                const uint4 vv = *(const uint4 *) (a.vc + (size_t) (hk * hd + r_pos + 64 * dd) * n_ctx + t * 64 + e * 8);
                BAMD_PV_BLOCK(t * 64, dd, vv);
            }
        }
    }
    for (int b0 = 256; b0 < n_kv; b0 += 64) {
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) if (r_pos + 64 * dd < hd) {
            const uint4 vv = *(const uint4 *) (a.vc + (size_t) (hk * hd + r_pos + 64 * dd) * n_ctx + b0 + e * 8);
            BAMD_PV_BLOCK(b0, dd, vv);
        }
    }
#undef BAMD_PV_BLOCK
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
        const int d = r_pos + 64 * dd;
        if (d < hd) { const float v = hsum8_tinyblas(acc4[dd]); if (e == 0) a.out[(size_t) h * hd + d] = v; }
    }
    STAMP(4);
}

// ===========================================================================================================
// Batched prefill (T > 1 tokens per call; reference: llama_decode with a micro-batch, ggml_compute_forward_mul_mat with
// ne11 = T, ggml.c:12277-12492).  Per (row, token) the arithmetic is EXACTLY the single-token chain above — the reference
// quantises each activation row to Q8_K and runs the same vec_dot per (row, column) — so the batched kernels reuse
// block_terms / chain_step / finish_row unchanged and differ only in data movement: the weights of a record are unpacked once
// and used for BAMD_TT tokens whose Q8_K activations sit in LDS.  (An MFMA formulation that keeps the per-lane chains exact —
// f16 A = scale x quant, one 32-deep MFMA per SIMD lane e — is the next step; see DESIGN.md.)
// ===========================================================================================================
#define BAMD_TT 8                       /* tokens per workgroup tile */
#define BAMD_BLOB_BYTES(nb) (BAMD_ACT_RED_OFF(nb))   /* one token's Q8_K activations in the LDS layout: q8[nb][64] | S[nb][8] | yd[nb], 16-byte padded */

// one workgroup per token: RMSNorm (optional) + Q8_K of row t of x[T][K] -> blob[t]
// f16 copy of a token's Q8_K row for the MFMA path.  Per super-block 528 B: 8 (e) x 4 (g) groups of 8 halves — group (e, g) = the
// int8 of sub-blocks 2g and 2g+1, chunk e, as exact f16: one 16-byte B operand of v_mfma_f32_16x16x32_f16 per lane — followed by
// the four i16 pairs (S_2l, S_2l+1) of the block sums; after the nb super-blocks, yd[nb] f32.  (528 B = 132 dwords: the MFMA
// kernel stages these records in LDS, and 132 = 4 mod 64 makes its 16-byte reads bank-conflict free.)
#define BAMD_B16_REC 528
#define BAMD_BLOB16_BYTES(nb) ((size_t) (nb) * (BAMD_B16_REC + 4))
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
        for (int i = threadIdx.x; i < nb * 4; i += blockDim.x) {           // (S_2l, S_2l+1) as i16 pairs: |S| <= 32 * 127
            const int ci = i >> 2, l = i & 3;
            *(uint32_t *) (o + (size_t) ci * BAMD_B16_REC + 512 + l * 4) = ((uint32_t) S[ci * 8 + 2 * l] & 0xffffu) | ((uint32_t) S[ci * 8 + 2 * l + 1] << 16);
        }
        float * oyd = (float *) (o + (size_t) nb * BAMD_B16_REC);
        for (int i = threadIdx.x; i < nb; i += blockDim.x) oyd[i] = yd[i];
    }
}


template <int TYPE, typename REC, int D, int EPI>
__device__ __forceinline__ void batch_segment(const uint8_t * __restrict__ wA, const uint8_t * __restrict__ wB, int nb, int first, int count, int stride,
                                              float * __restrict__ out, const float * __restrict__ res, int ldo, int t0, int nt,
                                              const unsigned char * acts, size_t bb, int nvalid) {
    constexpr int RECB = TYPE == BAMD_Q4_K ? 1152 : TYPE == BAMD_Q5_K ? 1408 : 1680;
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
    constexpr int NPARTS = PAIR ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const long rgb = (long) nb * RECB, rg_step = (long) stride * rgb;
    const int chunks = nb / D;
    REC ring[D];
    const uint8_t * rowA = wA + (long) first * rgb;
#pragma unroll
    for (int s = 0; s < D; ++s) load_rec(ring[s], rowA + s * RECB, lane);
    for (int r = 0; r < count; ++r) {
        const int rg = first + r * stride;
        const int row = rg * 8 + (lane >> 3);
        const long rowoff = (long) rg * rgb;
        float gate_val[BAMD_TT];
#pragma unroll
        for (int part = 0; part < NPARTS; ++part) {
            const uint8_t * pbase = (part ? wB : wA) + rowoff;
            const bool last = !(PAIR && part == 0) && r + 1 >= count;
            const uint8_t * after = (PAIR && part == 0) ? wB + rowoff : (last ? pbase + (long) (nb - 1) * RECB : wA + rowoff + rg_step);
            RowAcc A[BAMD_TT];
#pragma unroll
            for (int u = 0; u < BAMD_TT; ++u) { A[u].acc = 0.f; A[u].accm = 0.f; }
            for (int c = 0; c < chunks; ++c) {
                const bool inrow = c + 1 < chunks;
                const uint8_t * nxt = inrow ? pbase + (long) (c + 1) * (D * RECB) : after;
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
                    load_rec(ring[s], nxt + s * step, lane);
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
typedef _Float16 bamd_h8 __attribute__((ext_vector_type(8)));
typedef float bamd_f4 __attribute__((ext_vector_type(4)));
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
#define BAMD_MMA_NT 2
#define BAMD_MMA_TOK (16 * BAMD_MMA_NT)
#define BAMD_MMA_STAGE (BAMD_MMA_TOK * BAMD_B16_REC + BAMD_MMA_TOK * 4)          /* B records + yd */
#define BAMD_MMA_WAVE_LDS (2 * 288 * 4 + 16 * 32)                                /* transposed A tile + row headers */
template <int EPI>
__global__ void __launch_bounds__(512) matmul_mfma_q4k_kernel(bamd_mma_args a) {
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
    uint32_t * wl = (uint32_t *) (smem + 2 * BAMD_MMA_STAGE + wave * BAMD_MMA_WAVE_LDS);   // this wave's A tile [2][288] dwords
    uint32_t * hl = wl + 2 * 288;                                            // this wave's row headers [16][8] dwords
    // staging plan: 33 uint4 per token record, BAMD_MMA_TOK tokens; tokens past T repeat the last one (never stored)
    auto stage_issue = [&](int ci, uint4 (&r)[3], float & y) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = tid + k * 512;
            if (idx < BAMD_MMA_TOK * 33) {
                const int tok = idx / 33, q = idx - tok * 33;
                const int tg = t0 + tok < a.T ? t0 + tok : a.T - 1;
                r[k] = *(const uint4 *) (a.blob16 + (size_t) tg * b16 + (size_t) ci * BAMD_B16_REC + q * 16);
            }
        }
        if (tid < BAMD_MMA_TOK) { const int tg = t0 + tid < a.T ? t0 + tid : a.T - 1; y = *(const float *) (a.blob16 + (size_t) tg * b16 + (size_t) nb * BAMD_B16_REC + ci * 4); }
    };
    auto stage_commit = [&](int buf, const uint4 (&r)[3], float y) {
        unsigned char * st = stage + (size_t) buf * BAMD_MMA_STAGE;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int idx = tid + k * 512; if (idx < BAMD_MMA_TOK * 33) *(uint4 *) (st + (size_t) idx * 16) = r[k]; }
        if (tid < BAMD_MMA_TOK) *(float *) (st + BAMD_MMA_TOK * BAMD_B16_REC + tid * 4) = y;
    };
    const int rtc = live ? rt : 0;
    const uint8_t * rec0 = a.w + (size_t) (rtc * 2) * nb * 1152, * rec1 = rec0 + (size_t) nb * 1152;     // record groups of rows 0-7 / 8-15
    const uint8_t * hdrm = (m < 8 ? rec0 : rec1) + 1024 + (m & 7) * 16;                                    // header of row m (lanes g == 0)
    bamd_f4 acc[BAMD_MMA_NT][8], accm[BAMD_MMA_NT][4];
#pragma unroll
    for (int n = 0; n < BAMD_MMA_NT; ++n) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[n][e] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int l = 0; l < 4; ++l) accm[n][l] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
    }
    // prologue: stage super-block 0, prefetch the weights of super-block 0
    uint4 sr[3]; float sy = 0.f;
    stage_issue(0, sr, sy);
    uint4 wa = ldnt<uint4>(rec0, (uint32_t) lane * 16u), wb = ldnt<uint4>(rec1, (uint32_t) lane * 16u), hd = *(const uint4 *) hdrm;
    stage_commit(0, sr, sy);
    __syncthreads();
    for (int ci = 0; ci < nb; ++ci) {
        const unsigned char * st = stage + (size_t) (ci & 1) * BAMD_MMA_STAGE;
        const bool more = ci + 1 < nb;
        if (more) stage_issue(ci + 1, sr, sy);               // global loads of the next stage in flight during the math
        // ---- weights of this super-block: transpose into the A layout, unpack the row headers once ----
        {
            const int r = lane >> 3, e = lane & 7;           // wave-stream lane' = (row r of its record group, chunk e)
            *(uint4 *) (wl + 0 * 288 + r * 36 + e * 4) = wa;
            *(uint4 *) (wl + 1 * 288 + r * 36 + e * 4) = wb;
            if (g == 0) {                                    // lanes 0..15: row m
                uint32_t sc03, sc47, mn03, mn47; unpack_k4_(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
                uint4 h0, h1;
                h0.x = hd.x; h0.y = sc03; h0.z = sc47; h0.w = 0u;
                h1.x = __builtin_amdgcn_perm(0u, mn03, 0x0c010c00u); h1.y = __builtin_amdgcn_perm(0u, mn03, 0x0c030c02u);
                h1.z = __builtin_amdgcn_perm(0u, mn47, 0x0c010c00u); h1.w = __builtin_amdgcn_perm(0u, mn47, 0x0c030c02u);
                *(uint4 *) (hl + m * 8) = h0; *(uint4 *) (hl + m * 8 + 4) = h1;
            }
        }
        if (more) {                                          // prefetch the next super-block's weights
            const uint32_t ro = (uint32_t) (ci + 1) * 1152u;
            wa = ldnt<uint4>(rec0, ro + (uint32_t) lane * 16u); wb = ldnt<uint4>(rec1, ro + (uint32_t) lane * 16u); hd = *(const uint4 *) (hdrm + ro);
        }
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
            const uint32_t * wrow = wl + (m >> 3) * 288 + (m & 7) * 36 + g;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t wq = wrow[e * 4];
                const uint32_t lo = wq & 0x0f0f0f0fu, hi = (wq >> 4) & 0x0f0f0f0fu;
                union { uint32_t u; h2_t h; } c0, c1, c2, c3;
                c0.u = __builtin_amdgcn_perm(0x64646464u, lo, 0x04010400u); c1.u = __builtin_amdgcn_perm(0x64646464u, lo, 0x04030402u);
                c2.u = __builtin_amdgcn_perm(0x64646464u, hi, 0x04010400u); c3.u = __builtin_amdgcn_perm(0x64646464u, hi, 0x04030402u);
                const h2_t a0 = __builtin_elementwise_fma(c0.h, slo2, nlo2), a1 = __builtin_elementwise_fma(c1.h, slo2, nlo2);
                const h2_t a2 = __builtin_elementwise_fma(c2.h, shi2, nhi2), a3 = __builtin_elementwise_fma(c3.h, shi2, nhi2);
                const bamd_h8 av = { a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y };
#pragma unroll
                for (int n = 0; n < BAMD_MMA_NT; ++n) {
                    const bamd_h8 bv = *(const bamd_h8 *) (st + (size_t) (n * 16 + m) * BAMD_B16_REC + (e * 4 + g) * 16);
                    const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                    const bamd_f4 si = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, z, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[n][e][i] = fmaf(D[n][i], si[i], acc[n][e][i]);
                }
            }
        }
        // min terms: pm_l = m_2l S_2l + m_2l+1 S_2l+1 (one v_dot2_i32_i16); accm_l = fma(dmin, pm_l, accm_l)   (:6937-6941)
#pragma unroll
        for (int n = 0; n < BAMD_MMA_NT; ++n) {
            const uint4 sp = *(const uint4 *) (st + (size_t) (n * 16 + m) * BAMD_B16_REC + 512);
            const uint32_t spl[4] = { sp.x, sp.y, sp.z, sp.w };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t mpl[4] = { mp[i].x, mp[i].y, mp[i].z, mp[i].w };
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    union { uint32_t u; s2_t v; } ma, sb; ma.u = mpl[l]; sb.u = spl[l];
                    const float pm = (float) __builtin_amdgcn_sdot2(ma.v, sb.v, 0, false);
                    accm[n][l][i] = fmaf(Dm[n][i], pm, accm[n][l][i]);
                }
            }
        }
        if (more) stage_commit((ci + 1) & 1, sr, sy);
        __syncthreads();                                     // next stage visible; this stage and the wave tiles free again
    }
    if (!live) return;
    // hsum_float_8 over e and the acc_m folds, in the reference's order (finish_row), then the epilogue
#pragma unroll
    for (int n = 0; n < BAMD_MMA_NT; ++n) {
        const int t = t0 + n * 16 + m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = ((acc[n][0][i] + acc[n][4][i]) + (acc[n][2][i] + acc[n][6][i])) + ((acc[n][1][i] + acc[n][5][i]) + (acc[n][3][i] + acc[n][7][i]));
            const float mm = (accm[n][0][i] + accm[n][2][i]) + (accm[n][1][i] + accm[n][3][i]);
            const float val = v + mm;
            const int row = rt * 16 + 4 * g + i;
            if (t < a.T && row < a.nrows) {
                const size_t o = (size_t) t * a.ldo + row;
                a.out[o] = EPI == BAMD_EPI_ADD ? val + a.res[o] : val;
            }
        }
    }
}
// ---- Q6_K x Q8_K on the matrix cores, exact: same skeleton as matmul_mfma_q4k_kernel -----------------------------------------
// scale (int8) x (q6 - 32) reaches 4096 in magnitude: not every such integer is an f16.  With u = q6 in [0, 63]:
//   q6 - 32 = 2 (u >> 1) - 32 + (u & 1) = 2 vh + vl,  vh = (u >> 1) - 16 in [-16, 15],  vl = u & 1,
// so A_h = scale * vh (|.| <= 2048) and A_l = scale * vl (|.| <= 128) are exact f16, TWO MFMAs per e give S_h, S_l (< 2^24), and
// isum = 2 S_h + S_l (< 2^24) is exact as fmaf(2, S_h, S_l).  Scales are per 16 elements: for SIMD lane e the sub-block c uses
// scales[2c + (e >= 4)] (ggml-quants.c:8145-8216); no min terms.  Wave-stream Q6_K record: bamd_formats.h.
#define BAMD_MMA6_WAVE_LDS ((2 * 288 + 2 * 144 + 16 * 8) * 4)                    /* ql tile + qh tile + row headers */
template <int EPI>
__global__ void __launch_bounds__(512) matmul_mfma_q6k_kernel(bamd_mma_args a) {
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
    auto stage_issue = [&](int ci, uint4 (&r)[3], float & y) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = tid + k * 512;
            if (idx < BAMD_MMA_TOK * 33) {
                const int tok = idx / 33, q = idx - tok * 33;
                const int tg = t0 + tok < a.T ? t0 + tok : a.T - 1;
                r[k] = *(const uint4 *) (a.blob16 + (size_t) tg * b16 + (size_t) ci * BAMD_B16_REC + q * 16);
            }
        }
        if (tid < BAMD_MMA_TOK) { const int tg = t0 + tid < a.T ? t0 + tid : a.T - 1; y = *(const float *) (a.blob16 + (size_t) tg * b16 + (size_t) nb * BAMD_B16_REC + ci * 4); }
    };
    auto stage_commit = [&](int buf, const uint4 (&r)[3], float y) {
        unsigned char * st = stage + (size_t) buf * BAMD_MMA_STAGE;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int idx = tid + k * 512; if (idx < BAMD_MMA_TOK * 33) *(uint4 *) (st + (size_t) idx * 16) = r[k]; }
        if (tid < BAMD_MMA_TOK) *(float *) (st + BAMD_MMA_TOK * BAMD_B16_REC + tid * 4) = y;
    };
    const int rtc = live ? rt : 0;
    const uint8_t * rec0 = a.w + (size_t) (rtc * 2) * nb * 1680, * rec1 = rec0 + (size_t) nb * 1680;
    const uint8_t * recm = m < 8 ? rec0 : rec1;                                             // record group of row m (lanes g == 0)
    bamd_f4 acc[BAMD_MMA_NT][8];
#pragma unroll
    for (int n = 0; n < BAMD_MMA_NT; ++n) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[n][e] = (bamd_f4) { 0.f, 0.f, 0.f, 0.f };
    }
    uint4 sr[3]; float sy = 0.f;
    stage_issue(0, sr, sy);
    uint4 wa = ldnt<uint4>(rec0, (uint32_t) lane * 16u), wb = ldnt<uint4>(rec1, (uint32_t) lane * 16u);
    uint2 qa = ldnt<uint2>(rec0, 1024u + (uint32_t) lane * 8u), qb = ldnt<uint2>(rec1, 1024u + (uint32_t) lane * 8u);
    uint4 hsc = *(const uint4 *) (recm + 1536 + (m & 7) * 16); uint32_t hd = *(const unsigned short *) (recm + 1664 + (m & 7) * 2);
    stage_commit(0, sr, sy);
    __syncthreads();
    for (int ci = 0; ci < nb; ++ci) {
        const unsigned char * st = stage + (size_t) (ci & 1) * BAMD_MMA_STAGE;
        const bool more = ci + 1 < nb;
        if (more) stage_issue(ci + 1, sr, sy);
        {
            const int r = lane >> 3, e = lane & 7;
            *(uint4 *) (wl + 0 * 288 + r * 36 + e * 4) = wa;
            *(uint4 *) (wl + 1 * 288 + r * 36 + e * 4) = wb;
            *(uint2 *) (ql2 + 0 * 144 + r * 18 + e * 2) = qa;
            *(uint2 *) (ql2 + 1 * 144 + r * 18 + e * 2) = qb;
            if (g == 0) { *(uint4 *) (hl + m * 8) = hsc; hl[m * 8 + 4] = hd; }
        }
        if (more) {
            const uint32_t ro = (uint32_t) (ci + 1) * 1680u;
            wa = ldnt<uint4>(rec0, ro + (uint32_t) lane * 16u); wb = ldnt<uint4>(rec1, ro + (uint32_t) lane * 16u);
            qa = ldnt<uint2>(rec0, ro + 1024u + (uint32_t) lane * 8u); qb = ldnt<uint2>(rec1, ro + 1024u + (uint32_t) lane * 8u);
            hsc = *(const uint4 *) (recm + ro + 1536 + (m & 7) * 16); hd = *(const unsigned short *) (recm + ro + 1664 + (m & 7) * 2);
        }
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
            // int8 scales of sub-blocks c = 2g, 2g+1 for the two e-halves: header byte hi*8 + c
            h2_t sA[2], sB[2];
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const uint32_t w = hl[m * 8 + hi * 2 + (g >> 1)] >> (16 * (g & 1));
                const _Float16 s0 = (_Float16) (float) (int) (int8_t) (w & 0xffu), s1 = (_Float16) (float) (int) (int8_t) ((w >> 8) & 0xffu);
                sA[hi] = (h2_t) { s0, s0 }; sB[hi] = (h2_t) { s1, s1 };
            }
            const h2_t k1040 = { (_Float16) 1040.f, (_Float16) 1040.f };
            const us2_t one16 = { 0x3c00, 0x3c00 };
            const int sh = 4 * (g & 1);
            const uint32_t * wq = wl + (m >> 3) * 288 + (m & 7) * 36 + 2 * (g >> 1);
            const uint32_t * hq = ql2 + (m >> 3) * 144 + (m & 7) * 18 + (g >> 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint2 ab = *(const uint2 *) (wq + e * 4);
                const uint32_t h = hq[e * 2];
                const uint32_t uA = ((ab.x >> sh) & 0x0f0f0f0fu) | (((h >> sh) & 0x03030303u) << 4);          // q6 of sub-block 2g, chunk e
                const uint32_t uB = ((ab.y >> sh) & 0x0f0f0f0fu) | (((h >> (sh + 2)) & 0x03030303u) << 4);    // sub-block 2g + 1
                const uint32_t hA = (uA >> 1) & 0x1f1f1f1fu, hB = (uB >> 1) & 0x1f1f1f1fu, lA = uA & 0x01010101u, lB = uB & 0x01010101u;
                const h2_t sa = sA[e >> 2], sb = sB[e >> 2];
                union { uint32_t u; h2_t h; us2_t s; } c0, c1, c2, c3, d0, d1, d2, d3;
                c0.u = __builtin_amdgcn_perm(0x64646464u, hA, 0x04010400u); c1.u = __builtin_amdgcn_perm(0x64646464u, hA, 0x04030402u);
                c2.u = __builtin_amdgcn_perm(0x64646464u, hB, 0x04010400u); c3.u = __builtin_amdgcn_perm(0x64646464u, hB, 0x04030402u);
                d0.u = __builtin_amdgcn_perm(0u, lA, 0x0c010c00u); d1.u = __builtin_amdgcn_perm(0u, lA, 0x0c030c02u);   // 0 / 1 as u16 pairs
                d2.u = __builtin_amdgcn_perm(0u, lB, 0x0c010c00u); d3.u = __builtin_amdgcn_perm(0u, lB, 0x0c030c02u);
                d0.s = d0.s * one16; d1.s = d1.s * one16; d2.s = d2.s * one16; d3.s = d3.s * one16;                       // -> f16 0.0 / 1.0
                const h2_t ah0 = (c0.h - k1040) * sa, ah1 = (c1.h - k1040) * sa, ah2 = (c2.h - k1040) * sb, ah3 = (c3.h - k1040) * sb;   // exact, |.| <= 2048
                const h2_t al0 = d0.h * sa, al1 = d1.h * sa, al2 = d2.h * sb, al3 = d3.h * sb;
                const bamd_h8 avh = { ah0.x, ah0.y, ah1.x, ah1.y, ah2.x, ah2.y, ah3.x, ah3.y };
                const bamd_h8 avl = { al0.x, al0.y, al1.x, al1.y, al2.x, al2.y, al3.x, al3.y };
#pragma unroll
                for (int n = 0; n < BAMD_MMA_NT; ++n) {
                    const bamd_h8 bv = *(const bamd_h8 *) (st + (size_t) (n * 16 + m) * BAMD_B16_REC + (e * 4 + g) * 16);
                    const bamd_f4 z = { 0.f, 0.f, 0.f, 0.f };
                    const bamd_f4 sh_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(avh, bv, z, 0, 0, 0);
                    const bamd_f4 sl_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(avl, bv, z, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[n][e][i] = fmaf(D[n][i], fmaf(2.0f, sh_[i], sl_[i]), acc[n][e][i]);
                }
            }
        }
        if (more) stage_commit((ci + 1) & 1, sr, sy);
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
                a.out[o] = EPI == BAMD_EPI_ADD ? val + a.res[o] : val;
            }
        }
    }
}
// h[t][i] = silu(gate[t][i]) * up[t][i] — the SILU_MUL epilogue of the mat-vec kernels as its own pass (ggml_v_silu op for op)
__global__ void __launch_bounds__(256) silu_mul_kernel(const float * __restrict__ gate, const float * __restrict__ up, float * __restrict__ h, size_t n) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) h[i] = v_silu(gate[i]) * up[i];
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

// ---- batched prefill attention: one workgroup per (KV head, token) computes ALL GQH query heads that share the KV head -------
// Same arithmetic per (token, head) as attn_fused_kernel in its T > 1 mode (q rounded to f16, ggml_vec_dot_f16 order for the scores,
// softmax with the reference's 8-wide partial sums, tinyBLAS chains for P.V), but every K row and V^T chunk is loaded once for the
// GQH heads, and the token's own K/V are already in the cache (kv_store_batch_kernel).  Dynamic LDS: GQH x 2 x n_ctx floats.
template <int GQH>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) attn_batch_kernel(bamd_attn_args a) {
    __shared__ __attribute__((aligned(16))) unsigned short q16t[GQH][256];
    __shared__ float redf[GQH][8];
    __shared__ double redd[GQH][8];
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_dyn[];
    float * sc = (float *) attn_dyn;                                         // [GQH][n_ctx] scores, then exp values
    float * pt = sc + (size_t) GQH * a.n_ctx;                                // [GQH][n_ctx] probabilities in V^T position order
    const bamd_step_state * st = a.st;
    const int tokb = blockIdx.y;
    const int pos = st->pos + tokb;
    int n_kv = (pos + 1 + 31) / 32 * 32; n_kv = n_kv < st->n_ctx ? n_kv : st->n_ctx;
    const int hd = a.hd, Hkv = a.Hkv, Ekv = Hkv * hd, n_ctx = a.n_ctx, L = hd >> 3;
    const int hk = blockIdx.x, h0 = hk * GQH;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), e = lane & 7;
    const int r_pos = wave * 8 + (lane >> 3);
    const float * q = a.q + (size_t) tokb * a.ld_qkv + (size_t) h0 * hd;
    const float * rope = a.rope + (size_t) pos * hd;
    // RoPE of the GQH query heads -> f16, chain-major (rope_heads arithmetic; only the f16 copy is needed at T > 1)
    for (int i = tid; i < GQH * (hd / 2); i += blockDim.x) {
        const int hh = i / (hd / 2), p = i - hh * (hd / 2);
        const float c = rope[2 * p], sn = rope[2 * p + 1];
        const float x0 = q[hh * hd + 2 * p], x1 = q[hh * hd + 2 * p + 1];
        const float t0 = x0 * c, t1 = x1 * sn, t2 = x0 * sn, t3 = x1 * c;
        q16t[hh][kperm(2 * p, L)] = f2h(t0 - t1); q16t[hh][kperm(2 * p + 1, L)] = f2h(t2 + t3);
    }
    __syncthreads();
    // ---- scores: K row i once, GQH chains ----
    for (int t0 = 0; t0 < n_kv; t0 += 64) {
        const int i = t0 + r_pos;
        const bool valid = i < n_kv && i <= pos;
        uint4 kl[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) kl[g] = (valid && g * 8 < L) ? *(const uint4 *) (a.kc + (size_t) i * Ekv + hk * hd + e * L + g * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int hh = 0; hh < GQH; ++hh) {
            float v = -INFINITY;                                   // masked (KQ_mask, llama.cpp:14152-14200)
            if (valid) v = hsum8_vecdot(kq_chain<true>(kl, L, nullptr, &q16t[hh][0] + e * L));
            if (e == 0 && i < n_kv) sc[(size_t) hh * n_ctx + i] = v;
        }
    }
    __syncthreads();
    // ---- softmax per head (ggml.c:13682-13778 + :2619-2671) ----
    const float scale = a.kq_scale;
#pragma unroll
    for (int hh = 0; hh < GQH; ++hh) {
        const float * s_ = sc + (size_t) hh * n_ctx;
        float mx = -INFINITY;
        for (int i = tid; i < n_kv; i += blockDim.x) { const float w = s_[i] * scale; mx = w > mx ? w : mx; }
        uint32_t u = __float_as_uint(mx); u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        u = wave_max_u32(u);
        if (lane == 0) redf[hh][wave] = __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
    }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < GQH; ++hh) {
        float * s_ = sc + (size_t) hh * n_ctx;
        float mx = redf[hh][0];
        for (int w = 1; w < 8; ++w) mx = redf[hh][w] > mx ? redf[hh][w] : mx;
        double sum = 0.0;
        for (int i = tid; i < n_kv; i += blockDim.x) {             // n_kv % 32 == 0: 8-lane groups are all-active or all-idle
            const float w = s_[i] * scale;
            const float val = v_expf(w - mx);
            s_[i] = val;
            const float c = hsum8_tinyblas(val);
            if (e == 0) sum += (double) c;
        }
        sum = wave_sum_f64(sum);
        if (lane == 0) redd[hh][wave] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < GQH; ++hh) {
        const float * s_ = sc + (size_t) hh * n_ctx; float * p_ = pt + (size_t) hh * n_ctx;
        double tot = 0.0;
        for (int w = 0; w < 8; ++w) tot += redd[hh][w];
        const float fs = (float) (1.0 / tot);
        for (int i = tid; i < n_kv; i += blockDim.x) p_[vperm(i)] = s_[i] * fs;
        for (int i = n_kv + tid; i < ((n_kv + 63) & ~63); i += blockDim.x) p_[vperm(i)] = 0.f;   // half-filled last block: exact no-ops
    }
    __syncthreads();
    // ---- P.V: V^T chunk once, GQH chains; lane (d, e) carries Cv[e] of output d, up to 4 rows d per lane ----
    float acc[GQH][4];
#pragma unroll
    for (int hh = 0; hh < GQH; ++hh) { acc[hh][0] = 0.f; acc[hh][1] = 0.f; acc[hh][2] = 0.f; acc[hh][3] = 0.f; }
    for (int b0 = 0; b0 < n_kv; b0 += 64) {
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            if (r_pos + 64 * dd < hd) {
                const uint4 vv = *(const uint4 *) (a.vc + (size_t) (hk * hd + r_pos + 64 * dd) * n_ctx + b0 + e * 8);
                const uint32_t w[4] = { vv.x, vv.y, vv.z, vv.w };
                float vf[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) vf[u] = h2f((w[u >> 1] >> (16 * (u & 1))) & 0xffffu);
#pragma unroll
                for (int hh = 0; hh < GQH; ++hh) {
                    const float * p_ = pt + (size_t) hh * n_ctx + b0 + e * 8;
                    const float4 pa = *(const float4 *) p_, pb = *(const float4 *) (p_ + 4);
                    const float pv[8] = { pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w };
                    float c = acc[hh][dd];
#pragma unroll
                    for (int u = 0; u < 8; ++u) c = fmaf(vf[u], pv[u], c);
                    acc[hh][dd] = c;
                }
            }
        }
    }
    float * out = a.out + (size_t) tokb * a.ld_out + (size_t) h0 * hd;
#pragma unroll
    for (int hh = 0; hh < GQH; ++hh) {
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            const int d = r_pos + 64 * dd;
            if (d < hd) { const float v = hsum8_tinyblas(acc[hh][dd]); if (e == 0) out[(size_t) hh * hd + d] = v; }
        }
    }
}

// batched prefill: RoPE(K) + KV store of every token of the micro-batch, before any of them attends (grid (Hkv, T))
__global__ void __launch_bounds__(256) kv_store_batch_kernel(bamd_attn_args a) {
    __shared__ __attribute__((aligned(16))) unsigned short k16t[256];
    const int hk = blockIdx.x, tokb = blockIdx.y;
    const int pos = a.st->pos + tokb;
    const int hd = a.hd, Ekv = a.Hkv * hd, n_ctx = a.n_ctx;
    const float * k = a.k + (size_t) tokb * a.ld_qkv, * v = a.v + (size_t) tokb * a.ld_qkv;
    rope_heads(k + (size_t) hk * hd, a.rope + (size_t) pos * hd, hd, 1, nullptr, nullptr, k16t);
    __syncthreads();
    for (int i = threadIdx.x; i < hd; i += blockDim.x) {
        a.kc[(size_t) pos * Ekv + hk * hd + i] = k16t[i];
        a.vc[(size_t) (hk * hd + i) * n_ctx + vperm(pos)] = f2h(v[hk * hd + i]);
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
template <int PRO>
static void launch_mv_split(const bamd_mv_args & a, int epi, int grid, hipStream_t s) {
    const int nb = a.K >> 8;
    const int nbw = nb >> 3;
    const int M = nbw == 2 ? 4 : nbw == 7 ? 1 : nbw == 4 ? 2 : 8, NBUF = 2;                  // must match split_dispatch
    const size_t lds = act_lds_bytes(a.K) + 16 + (size_t) NBUF * M * nb * 256 * 4;   // 112..128 KiB of term buffers
    if (epi == BAMD_EPI_ADD) hipLaunchKernelGGL((matvec_split_kernel<PRO, BAMD_EPI_ADD>),   dim3(grid), dim3(512), lds, s, a);
    else                     hipLaunchKernelGGL((matvec_split_kernel<PRO, BAMD_EPI_STORE>), dim3(grid), dim3(512), lds, s, a);
}

static bool split_supported(int nb) { const int nbw = nb >> 3; return (nb & 7) == 0 && (nbw == 1 || nbw == 2 || nbw == 4 || nbw == 7); }

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
    const bool split = a.mode == 2 ? can_split : a.mode == 1 ? false : (can_split && nrg < 8 * cus && !mixed);
    int grid = cus;                                          // one 8-wave workgroup per CU
    if (grid > nrg) grid = nrg;
    if (grid < 1) grid = 1;
    if (split) {
        if (pro == BAMD_PRO_NORM) launch_mv_split<BAMD_PRO_NORM>(a, epi, grid, s);
        else                      launch_mv_split<BAMD_PRO_PLAIN>(a, epi, grid, s);
        return;
    }
    if (pro == BAMD_PRO_NORM) launch_mv_epi<BAMD_PRO_NORM>(a, epi, grid, s);
    else                      launch_mv_epi<BAMD_PRO_PLAIN>(a, epi, grid, s);
}

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
int bamd_launch_matmul_mfma(const void * w_stream, int type, int nrows, int nrows_pad, int K, const void * blob16, int T, float * out, const float * res, int ldo,
                            hipStream_t s) {
    if ((type != BAMD_Q4_K && type != BAMD_Q6_K) || (nrows_pad & 7) || (K & 1023)) return 1;     // K % 1024: 16-byte alignment of the per-token f16 blobs
    bamd_mma_args a; a.w = (const uint8_t *) w_stream; a.out = out; a.res = res; a.blob16 = (const uint8_t *) blob16; a.K = K; a.T = T; a.nrows = nrows; a.nrows_pad = nrows_pad; a.ldo = ldo;
    dim3 grid((T + BAMD_MMA_TOK - 1) / BAMD_MMA_TOK, (nrows_pad / 16 + (nrows_pad % 16 ? 1 : 0) + 7) / 8);
    if (type == BAMD_Q6_K) {
        const size_t lds6 = 2 * BAMD_MMA_STAGE + 8 * BAMD_MMA6_WAVE_LDS;
        if (res) hipLaunchKernelGGL((matmul_mfma_q6k_kernel<BAMD_EPI_ADD>),   grid, dim3(512), lds6, s, a);
        else     hipLaunchKernelGGL((matmul_mfma_q6k_kernel<BAMD_EPI_STORE>), grid, dim3(512), lds6, s, a);
        return 0;
    }
    const size_t lds = 2 * BAMD_MMA_STAGE + 8 * BAMD_MMA_WAVE_LDS;
    if (res) hipLaunchKernelGGL((matmul_mfma_q4k_kernel<BAMD_EPI_ADD>),   grid, dim3(512), lds, s, a);
    else     hipLaunchKernelGGL((matmul_mfma_q4k_kernel<BAMD_EPI_STORE>), grid, dim3(512), lds, s, a);
    return 0;
}
void bamd_launch_silu_mul(const float * gate, const float * up, float * h, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, gate, up, h, n);
}
void bamd_launch_embed_batch(const int32_t * tokens, int T, const void * embd, int embd_type, int E, int V, float * x, hipStream_t s) {
    hipLaunchKernelGGL(embed_batch_kernel, dim3(T), dim3(256), 0, s, tokens, (const uint8_t *) embd, embd_type, E, V, x);
}
// attention of a micro-batch of T tokens (a.batch = 1, a.ld_qkv / a.ld_out set): KV store for all tokens, then (head, token) workgroups
int bamd_launch_attention_batch(const bamd_attn_args & a, int gq, int T, hipStream_t s) {
    if (a.hd > 256 || (a.hd & 63) || a.n_ctx > BAMD_ATTN_BATCH_MAX || !a.batch) return 1;
    if (gq != 1 && gq != 2 && gq != 4 && gq != 8) return 1;
    hipLaunchKernelGGL(kv_store_batch_kernel, dim3(a.Hkv, T), dim3(256), 0, s, a);
    // all query heads of a KV head in one workgroup while their score buffers fit the LDS; else one workgroup per query head
    const size_t lds_g = (size_t) gq * a.n_ctx * 8;
    if (lds_g <= 144 * 1024 && (gq == 2 || gq == 4 || gq == 8)) {
        if (gq == 2)      hipLaunchKernelGGL((attn_batch_kernel<2>), dim3(a.Hkv, T), dim3(512), lds_g, s, a);
        else if (gq == 4) hipLaunchKernelGGL((attn_batch_kernel<4>), dim3(a.Hkv, T), dim3(512), lds_g, s, a);
        else              hipLaunchKernelGGL((attn_batch_kernel<8>), dim3(a.Hkv, T), dim3(512), lds_g, s, a);
    } else hipLaunchKernelGGL(attn_fused_kernel, dim3(a.Hkv * gq, T), dim3(512), (size_t) a.n_ctx * 8, s, a, gq);
    return 0;
}

#ifdef BAMD_TIMING
void bamd_read_stamps(unsigned long long * host) { hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 16); }
#endif

void bamd_launch_step_begin(bamd_step_state * st, const int32_t * forced, int n_forced, int32_t * out_tokens, const void * embd,
                            int embd_type, int E, int V, float * x, int do_embed, hipStream_t s) {
    hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(1024), 0, s, st, forced, n_forced, out_tokens, (const uint8_t *) embd, embd_type, E, V, x, do_embed);
}

int bamd_launch_attention(const bamd_attn_args & a, int gq, int max_tiles, hipStream_t s) {
    if (a.hd > 256 || (a.hd & 63)) return 1;           // chain-major K rows are read in 16-byte (8-step) groups
    if (gq != 1 && gq != 2 && gq != 4 && gq != 8) return 1;
    if (a.n_ctx <= BAMD_ATTN_FUSED_MAX && max_tiles >= 0) {
        // context fits the LDS score buffer: one fused launch per layer, one workgroup per query head
        hipLaunchKernelGGL(attn_fused_kernel, dim3(a.Hkv * gq), dim3(512), (size_t) a.n_ctx * 8, s, a, gq);
        return 0;
    }
    int ty = max_tiles < 0 ? -max_tiles : max_tiles;
    if (ty < 1) ty = 1;
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
