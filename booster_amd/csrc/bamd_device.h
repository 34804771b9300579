// bamd_device.h — device-side building blocks shared by the kernel files (bamd_matvec.hip, bamd_attention.hip, bamd_prefill.hip):
// numerics helpers, the Q8_K activation prologue, wave-stream records and their block terms, the f32 chains, attention chain
// helpers.  Everything here is __device__ __forceinline__ (or a macro / type): including it in several translation units is safe.
//
//
// NUMERICS CONTRACT.  Every kernel reproduces, operation for operation, the IEEE-754 arithmetic of the
// reference's CPU path as built for x86 AVX2+FMA+F16C with GGML_USE_LLAMAFILE (what Booster ships):
// integer block dot products are exact; every f32 operation (which products are fused, which sums are
// sequential chains over super-blocks, the shape of each horizontal reduction tree) is the one the
// reference's 256-bit code performs, with one wave lane standing for one SIMD lane.  Compiled with
// -ffp-contract=off; fused multiply-adds are explicit fmaf().  Do NOT build with -ffast-math.
// The two double-precision sums (RMSNorm sum of squares, softmax denominator) are sequential in the reference and tree-reduced here;
// both are rounded to f32 right after (mean, 1/sum), so the order can only show when the tree sum lies within the worst-case reordering
// error of a rounding boundary of that f32.  f32_rounding_safe() checks exactly that; when it cannot rule a difference out (~1e-5 of the
// reductions) ONE lane redoes the sum in the reference's order, so the result is the reference's in every case (tests/test_f64_order.py).
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
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>
#include "bamd_formats.h"
#include "bamd_kernels.h"
#include "bamd_aql.h"

#define WAVE 64
#ifndef BAMD_SCHED_GROUP
#define BAMD_SCHED_GROUP 4      /* records the scheduler may interleave between barriers (power of two); 1 -> 4: 663 -> 670 tok/s */
#endif

// optional in-kernel phase stamps (build with -DBAMD_TIMING: booster_amd/lib/libbooster_amd_timing.so, tools/timeline.py): lane 0 of
// waves 0 and 7 of every workgroup writes the 100 MHz wall clock (s_memrealtime, one time base for the whole device and across launches)
// into the stamp block of its launch — [workgroup][wave 0 | wave 7][8 phases] u64, handed over in the kernel arguments (`tl`, null = off).
#ifdef BAMD_TIMING
#define TL_STAMP(tl, k) do { if ((tl) && (threadIdx.x & 63) == 0 && blockIdx.x < BAMD_TL_WG && blockIdx.y == 0) { const int w_ = (int) (threadIdx.x >> 6); \
        if (w_ == 0 || w_ == 7) (tl)[(size_t) blockIdx.x * BAMD_TL_WG_WORDS + (w_ ? 8 : 0) + (k)] = wall_clock64(); \
        if ((k) == 7 && w_ < 8) (tl)[(size_t) blockIdx.x * BAMD_TL_WG_WORDS + 16 + w_] = wall_clock64(); } } while (0)
#else
#define TL_STAMP(tl, k) do { } while (0)
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
__device__ __forceinline__ const uint8_t * uniform_ptr(const uint8_t * p) {      // a pointer the caller knows to be wave-uniform -> SGPR pair
    const uint64_t u = (uint64_t) p;
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) u), hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (u >> 32));
    return (const uint8_t *) (((uint64_t) hi << 32) | lo);
}
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)); }

// ===========================================================================================================
// Inter-kernel data.  Everything one launch of a decode step writes for another launch to read — activation vectors, residuals, q | k | v, the
// attention output, logits, the arg-max key, the device-side step state, the KV cache rows — is stored write-through (sc1) and loaded with sc1
// loads, which bypass this CU's vector L1 and are coherent across the eight XCD L2s.  A decode step replayed from the context's own AQL queue
// (bamd_aql.cpp) runs its packets with acquire / release fence scope NONE: no L2 write-back at a kernel's end, no L1 / L2 / scalar-cache invalidate
// at the next one's start (−0.4 us per launch, profiles/r03_aql_probe.txt), so nothing but these accesses carries data from launch to launch.
// Rules for a kernel of the decode graph: (1) mutable data is never read through a scalar load or a plain vector load — ik_* only; (2) every
// global store is ik_*; (3) constants (weights, norm weights, the RoPE table, token_embd) keep their plain / nt loads: nobody rewrites them.
// Under ordinary HIP launches (agent-scope fences at every boundary) the same instructions are merely redundant.  BAMD_IK_SC1=0 compiles the
// plain forms back in (A/B: profiles/r06_levers.txt).
// ===========================================================================================================
#ifndef BAMD_IK_SC1
#define BAMD_IK_SC1 1
#endif
#define BAMD_IK_AUX (BAMD_IK_SC1 ? 16 : 0)          /* aux bits of the raw buffer loads / stores: 16 = sc1 */
typedef __amdgpu_buffer_rsrc_t bamd_ik_rsrc;
typedef uint32_t ik_u32x4 __attribute__((ext_vector_type(4)));
// base must be wave-uniform; offsets are 32-bit byte offsets below 2 GiB (an activation vector, one layer's K or V^T cache)
__device__ __forceinline__ bamd_ik_rsrc ik_rsrc(const void * base) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) uniform_ptr((const uint8_t *) base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 ik_ld128(bamd_ik_rsrc r, uint32_t voff, int soff = 0) {
    const ik_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int) voff, soff, BAMD_IK_AUX); return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ik_ld128f(bamd_ik_rsrc r, uint32_t voff, int soff = 0) {
    const ik_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int) voff, soff, BAMD_IK_AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void ik_st128f(bamd_ik_rsrc r, uint32_t voff, const float4 v) {
    const ik_u32x4 u = { __float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w) };
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int) voff, 0, BAMD_IK_AUX);
}
#if BAMD_IK_SC1
template <typename T> __device__ __forceinline__ T ik_ld(const T * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void ik_st(T * p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
template <typename T> __device__ __forceinline__ T ik_ld(const T * p) { return *p; }
template <typename T> __device__ __forceinline__ void ik_st(T * p, T v) { *p = v; }
#endif
// experiment switches (timing under HIP launches only — a plain access is NOT coherent on the own queue): which class of access of the attention kernels
// costs what (profiles/r06_levers.txt)
#ifndef BAMD_IK_ST
#define BAMD_IK_ST BAMD_IK_SC1       /* step-state loads (pos, n_kv, serial, step) */
#endif
#ifndef BAMD_IK_QKV
#define BAMD_IK_QKV BAMD_IK_SC1      /* this token's q / k / v */
#endif
#ifndef BAMD_IK_KVLD
#define BAMD_IK_KVLD BAMD_IK_SC1     /* KV cache loads */
#endif
#ifndef BAMD_IK_KLD
#define BAMD_IK_KLD BAMD_IK_KVLD     /* K rows only */
#endif
#ifndef BAMD_IK_VLD
#define BAMD_IK_VLD BAMD_IK_KVLD     /* V^T rows only */
#endif
#ifndef BAMD_IK_KVST
#define BAMD_IK_KVST BAMD_IK_SC1     /* KV cache stores */
#endif
template <bool SC1, typename T> __device__ __forceinline__ T ik_ld_if(const T * p) { if (SC1) return ik_ld(p); return *p; }
template <bool SC1, typename T> __device__ __forceinline__ void ik_st_if(T * p, T v) { if (SC1) ik_st(p, v); else *p = v; }
template <bool SC1> __device__ __forceinline__ uint4 ik_ld128_if(bamd_ik_rsrc r, uint32_t voff) {
    const ik_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int) voff, 0, SC1 ? 16 : 0); return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 ik_ld2f(const float * p) {                // 8-byte aligned pair
    const unsigned long long u = ik_ld((const unsigned long long *) p);
    return make_float2(__uint_as_float((uint32_t) u), __uint_as_float((uint32_t) (u >> 32)));
}

// ggml-quants.c:1632-1637
__device__ __forceinline__ int nearest_int(float fval) {
    float val = fval + 12582912.f;
    return (__float_as_int(val) & 0x007fffff) - 0x00400000;
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

// ---- the f64 sums of the reference are SEQUENTIAL (ggml.c:11874-11877 RMSNorm, :2619-2671 softmax); ours are trees ----------------------------
// Both orders add the same n non-negative terms, so each of them is within (n - 1) u S of the exact sum (u = 2^-53) and they differ by less than
// 2 n u S; a division / reciprocal adds u more on each side.  If every double within that relative distance of our value rounds to the SAME f32,
// the reference's f32 is that one too.  Otherwise (the value sits next to a rounding boundary: ~1e-5 of the reductions at n = 4096) the caller
// recomputes sequentially.  v > 0 finite: a zero or non-finite sum is the same in any order.
// In integer terms: a double rounds to f32 by its low 29 mantissa bits D (the boundary is D = 2^28), and a relative distance of (2 n + 8) u is
// at most 2 n + 8 units of D: safe iff |D - 2^28| > 2 n + 8 (and the value is in the f32 normal range; anything else takes the slow path).
#define BAMD_F64_GUARD_ULPS(n) (2 * (n) + 8)
__device__ __forceinline__ bool f32_rounding_safe(double v, int ulps) {
#ifdef BAMD_NO_F64_GUARD                                                     /* experiment builds: shows that the constructed worst cases of tests/test_f64_order.py need the guard */
    return true;
#endif
    if (v == 0.0) return true;                                              // the same in any order (non-negative terms)
    const uint32_t hi = (uint32_t) __double2hiint(v), lo = (uint32_t) __double2loint(v);
    const uint32_t ex = (hi >> 20) & 0x7ffu;                                // sign bit is 0: sums of squares / of exponentials
    const int dist = (int) (lo & 0x1fffffffu) - 0x10000000;
    return ex >= 1023u - 126u && ex <= 1023u + 127u && (dist < 0 ? -dist : dist) > ulps;
}
// the reference's softmax denominator over exp values already stored at vals[0..n) (n % 8 == 0): 8-wide f32 partial sums in the AVX2 tree
// ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7)), added to a double one after the other (ggml.c:2635-2644)
__device__ __forceinline__ double seq_expsum8(const float * vals, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i += 8) {
        const float a0 = vals[i] + vals[i + 4], a1 = vals[i + 1] + vals[i + 5], a2 = vals[i + 2] + vals[i + 6], a3 = vals[i + 3] + vals[i + 7];
        const float b0 = a0 + a2, b1 = a1 + a3;
        s += (double) (b0 + b1);
    }
    return s;
}

// One wave quantises BATCH super-blocks at a time (independent dependency chains interleave); lane l holds the 4
// consecutive elements 4l..4l+3 of a block.  quantize_row_q8_K_ref semantics (ggml-quants.c:3593-3630): the scale comes
// from the FIRST element of largest magnitude (strict > scan), so ties resolve to the lowest lane, lowest element.
#define BAMD_ACT_BATCH 4
// LDS layout of the quantised activations of one mat-vec: q8[nb][64] u32 | S[nb][8] i32 | yd[nb] f32 | (16-byte aligned) red[16] f64
#ifndef BAMD_CEILING
#define BAMD_CEILING 0               /* timing-only builds (never shipped): 1 = mat-vec prologues without the Q8_K / RMSNorm statistics, 2 = split-K kernels without the
                                        barrier + sequential chain replay: what the reference's arithmetic costs per launch (profiles/r05_ceiling.txt) */
#endif
#define BAMD_ACT_RED_OFF(nb) ((((size_t) (nb) * (256 + 32 + 4)) + 15) & ~(size_t) 15)
template <bool NORM>
struct ActPro {
    float4 v[BAMD_ACT_BATCH], w[BAMD_ACT_BATCH];
    int okmask;                                          // bit b: batch slot b holds a block of the vector (wave-uniform)
    unsigned long long * tl = nullptr;                   // phase stamps (BAMD_TIMING builds)

    // the loads of this wave's first batch of blocks: issued at kernel entry, AHEAD of the bulk weight prefetch, so the
    // (tiny, latency-critical) activation read is not queued behind megabytes of weight requests
    // blocks i0, i0 + bstride, ... below blimit (defaults: this wave's share of the whole vector, interleaved over the waves)
    // NB: batch slots in use (compile time) — K = 4096 on eight waves fills two of the four; the others would request the last block again
    // and run the whole quantisation arithmetic on it for nothing
    template <int NB = BAMD_ACT_BATCH>
    __device__ __forceinline__ void issue(const float * __restrict__ x, const float * __restrict__ nw, int K, int i0, int bstride = 0, int blimit = 0) {
        const int lane = threadIdx.x & 63;
        if (bstride == 0) { bstride = blockDim.x >> 6; blimit = K >> 8; }
        // UNCONDITIONAL requests (a block index past the end is clamped to the last block and its values are
        // never used): a conditional load becomes a branch around the request with a full s_waitcnt at the join, which serialised the
        // batches into one memory round trip each — and held back the weight ring that is issued after them
        okmask = 0;
        const bamd_ik_rsrc rx = ik_rsrc(x);              // the activation vector is another launch's output: sc1 loads ("Inter-kernel data" above)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = i0 + b * bstride;
            const bool ok = i < blimit;
            okmask |= ok ? 1 << b : 0;
            const int ic = ok ? i : blimit - 1;          // past the end: the last block again; its values are never used (okmask, i < nb below)
            v[b] = ik_ld128f(rx, (uint32_t) lane * 16u, ic * 1024);
            if (NORM) w[b] = *(const float4 *) (nw + ic * 256 + lane * 4);
        }
    }

    // Issue-bound code (every CU quantises the whole activation vector: ~1/3 of a decode step's VALU work), so the instruction
    // count per block is what matters here:
    //   - the two IEEE divisions per block (iscale = -127/max, d = 1/iscale) run ONCE per batch: block b's operand sits in lane b;
    //   - nearest_int(v) & 0xff is the low byte of the bits of v + 12582912.f (ggml-quants.c:1632-1637: the mask and the
    //     0x400000 offset do not touch that byte), and MIN(127, .) (:3617) never binds for |iscale * x| <= 127(1 + 2^-23);
    //   - the sum of the four signed bytes is one v_dot4 against 0x01010101;
    //   - the four wave-max chains are interleaved step by step (DPP results need wait states); row_bcast leaves the result in lane 63.
    template <int NB = BAMD_ACT_BATCH>
    __device__ __forceinline__ void quantize_batch(float scale, int K, int i0, uint32_t * q8, int * S, float * yd, int bstride = 0, int blimit = 0) {
        const int lane = threadIdx.x & 63;
        const int nwaves = bstride ? bstride : (int) (blockDim.x >> 6), nb = bstride ? blimit : (K >> 8);
#if BAMD_CEILING & 1
        // TIMING-ONLY ceiling build (tools/ceiling.sh; never shipped, results are garbage): the activations arrive "already quantised" — no block maxima,
        // no divisions, no rounding: what a prologue costs that is a plain load + three LDS stores
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = i0 + b * nwaves;
            if (i < nb) {
                const uint32_t packed = __float_as_uint(v[b].x) ^ (__float_as_uint(v[b].y) >> 8);
                q8[i * 64 + (lane & 7) * 8 + (lane >> 3)] = packed;
                if ((lane & 7) == 0) S[i * 8 + (lane >> 3)] = (int) (packed & 0xff);
                if (lane == 0) yd[i] = scale;
            }
        }
        return;
#endif
        uint32_t amaxb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (NORM) {                                  // y = (x*scale)*w : ggml_vec_scale_f32 then ggml_mul (llama.cpp:7940-7950)
                v[b].x = (v[b].x * scale) * w[b].x; v[b].y = (v[b].y * scale) * w[b].y;
                v[b].z = (v[b].z * scale) * w[b].z; v[b].w = (v[b].w * scale) * w[b].w;
            }
            const float a = fmaxf(fmaxf(fmaxf(fabsf(v[b].x), fabsf(v[b].y)), fabsf(v[b].z)), fabsf(v[b].w));
            amaxb[b] = __float_as_uint(a);               // non-negative floats order like their bit patterns
        }
        uint32_t t[NB], wmax[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) t[b] = umax_(amaxb[b], (uint32_t) dpp_z<DPP_XOR1>((int) amaxb[b]));
#pragma unroll
        for (int b = 0; b < NB; ++b) t[b] = umax_(t[b], (uint32_t) dpp_z<DPP_XOR2>((int) t[b]));
#pragma unroll
        for (int b = 0; b < NB; ++b) t[b] = umax_(t[b], (uint32_t) dpp_z<DPP_HALF_MIRROR>((int) t[b]));
#pragma unroll
        for (int b = 0; b < NB; ++b) t[b] = umax_(t[b], (uint32_t) dpp_z<DPP_MIRROR>((int) t[b]));
#pragma unroll
        for (int b = 0; b < NB; ++b) t[b] = umax_(t[b], (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t[b], 0x142, 0xa, 0xf, false));   // row_bcast:15
#pragma unroll
        for (int b = 0; b < NB; ++b) t[b] = umax_(t[b], (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t[b], 0x143, 0xc, 0xf, false));   // row_bcast:31
#pragma unroll
        for (int b = 0; b < NB; ++b) wmax[b] = (uint32_t) __builtin_amdgcn_readlane((int) t[b], 63);
        // the scale comes from the FIRST element of largest magnitude (strict > scan of the reference): lowest lane, lowest element
        float mine[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float M = __uint_as_float(wmax[b]);
            const bool ex = fabsf(v[b].x) == M, ey = fabsf(v[b].y) == M, ez = fabsf(v[b].z) == M;
            float m = v[b].w;                            // branch-free selects, lowest element wins
            m = ez ? v[b].z : m; m = ey ? v[b].y : m; m = ex ? v[b].x : m;
            mine[b] = m;
        }
        int mxv = __float_as_int(1.0f);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const unsigned long long who = __ballot(amaxb[b] == wmax[b]);
            const int first = __ffsll((long long) who) - 1;
            const int mxb = __builtin_amdgcn_readlane(__float_as_int(mine[b]), first);
            mxv = lane == b ? mxb : mxv;
        }
        const float isc = -127.f / __int_as_float(mxv);  // lane b: block b (other lanes: -127)
        const float dd = 1.0f / isc;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
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

    // SMALLK: the caller guarantees K <= 256 * BAMD_ACT_BATCH * (waves per workgroup), so the first batch is the whole share of this
    // wave and the loops over further batches (whose in-loop requests force a full s_waitcnt at their exit — which would also wait for
    // the weight ring issued before this call) are compiled out
    struct NoMid { __device__ __forceinline__ void operator()() const { } };
    // mid: called once behind the first workgroup barrier (NORM) / at the start (plain).  The mode-A kernels request the second half of their
    // weight ring there: a CU's texture path takes ~1.5 us to accept the requests of eight full rings, every wave sits in its issue stage
    // for that long, and the barrier behind the sum of squares waited for the last of them
    template <bool SMALLK = false, typename MID = NoMid, int NB = BAMD_ACT_BATCH>
    __device__ __forceinline__ void finish(const float * __restrict__ x, const float * __restrict__ nw, float eps, int K,
                                           uint32_t * q8, int * S, float * yd, double * red, MID mid = MID()) {
        const int lane = threadIdx.x & 63, wave = wave_id(), nwaves = blockDim.x >> 6, nb = K >> 8;
        const int step = nwaves * BAMD_ACT_BATCH;
        float scale = 1.0f;
#if BAMD_CEILING & 1
        if (NORM) { mid(); } else
#endif
        if (NORM) {
            // sum of squares in double (ggml.c:11874-11877), fixed tree order; f32_rounding_safe() below decides whether the order can matter
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (okmask >> b & 1) { s += (double) (v[b].x * v[b].x); s += (double) (v[b].y * v[b].y); s += (double) (v[b].z * v[b].z); s += (double) (v[b].w * v[b].w); }
            }
            if (!SMALLK) for (int i0 = wave + step; i0 < nb; i0 += step) {          // only for K > 256 * 4 * nwaves
                ActPro<NORM> t; t.issue(x, nw, K, i0);
#pragma unroll
                for (int b = 0; b < BAMD_ACT_BATCH; ++b) {
                    if (t.okmask >> b & 1) { s += (double) (t.v[b].x * t.v[b].x); s += (double) (t.v[b].y * t.v[b].y); s += (double) (t.v[b].z * t.v[b].z); s += (double) (t.v[b].w * t.v[b].w); }
                }
            }
            s = wave_sum_f64(s);
            if (lane == 0) red[wave] = s;
            TL_STAMP(tl, 5);
            __syncthreads();
            mid();
            double tot = 0.0;
            for (int w2 = 0; w2 < nwaves; ++w2) tot += red[w2];
            // sum / n in double (ggml.c:11879).  For n a power of two (4096, 8192) the quotient is an exact scaling, so the product with the
            // exact reciprocal is the same double — without the ~30 dependent f64 instructions of an IEEE division
            double md = (K & (K - 1)) == 0 ? tot * (1.0 / (double) K) : tot / (double) K;
            float mean = (float) md;
            if (!f32_rounding_safe(md, BAMD_F64_GUARD_ULPS(K))) {     // workgroup-uniform (every thread holds the same tot); rare
                __syncthreads();                                            // everybody has read red[]
                if (threadIdx.x == 0) {                                     // the reference's order, one lane (ggml.c:11874-11877)
                    double sq = 0.0;
                    for (int i = 0; i < K; ++i) { const float xv = ik_ld(x + i); sq += (double) (xv * xv); }
                    red[0] = sq;
                }
                __syncthreads();
                md = red[0] / (double) K;
                mean = (float) md;
            }
            scale = 1.0f / sqrtf(mean + eps);
        } else mid();
        quantize_batch<NB>(scale, K, wave, q8, S, yd);
        if (!SMALLK) for (int i0 = wave + step; i0 < nb; i0 += step) {
            ActPro<NORM> t; t.issue(x, nw, K, i0);
            t.quantize_batch(scale, K, i0, q8, S, yd);
        }
        TL_STAMP(tl, 6);
        __syncthreads();
    }
};

// ===========================================================================================================
// Quantised mat-vec: y = W . Q8_K(x).  One wave = 8 rows at a time (lane = r*8+e), rows streamed sequentially
// over super-blocks so each lane carries exactly the f32 chain of SIMD lane e of the reference.
// ===========================================================================================================
struct RowAcc { float acc, accm; };

// ---- per-record arithmetic -----------------------------------------------------------------------------
struct RecQ4K { uint4 qs, hd; uint32_t mn47; };              // mn47: BAMD_XSCALES = 1 only (bamd_formats.h); dead otherwise
struct RecQ5K { uint4 qs, hd; uint32_t qh, mn47; };
struct RecQ6K { uint4 ql; uint2 qh, sc; uint32_t d; };

// Pin a loaded register at its point of use: without this, LLVM folds the first ALU op on a ring register into
// the loop PHI (i.e. executes it right after the load, one iteration early), which forces s_waitcnt vmcnt(0) at
// the loop tail and serialises the whole prefetch ring.
__device__ __forceinline__ void pin(uint32_t & x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(uint2 & x) { pin(x.x); pin(x.y); }
__device__ __forceinline__ void pin(uint4 & x) { pin(x.x); pin(x.y); pin(x.z); pin(x.w); }
#if BAMD_XSCALES
__device__ __forceinline__ void pin_rec(RecQ4K & R) { pin(R.qs); pin(R.hd); pin(R.mn47); }
__device__ __forceinline__ void pin_rec(RecQ5K & R) { pin(R.qs); pin(R.hd); pin(R.qh); pin(R.mn47); }
#else
__device__ __forceinline__ void pin_rec(RecQ4K & R) { pin(R.qs); pin(R.hd); }
__device__ __forceinline__ void pin_rec(RecQ5K & R) { pin(R.qs); pin(R.hd); pin(R.qh); }
#endif
__device__ __forceinline__ void pin_rec(RecQ6K & R) { pin(R.ql); pin(R.qh); pin(R.sc); pin(R.d); }

// Weight records are fetched with BUFFER loads: the matrix is one 128-bit resource descriptor in scalar registers, the record a scalar
// byte offset (soffset), the lane's share a constant 32-bit vector offset + an immediate — no vector instruction computes an address
// (a global_load of base + per-lane offset took one 64-bit vector add per request: 2 of the 62 vector instructions of a Q4_K record).
// Weights are read exactly once per token: non-temporal (MI355X_MICROARCH.md, row nt-weights).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#ifndef BAMD_BUFFER_LOADS
#define BAMD_BUFFER_LOADS 1
#endif
#define BAMD_LOAD_NT 2                  /* aux bits of the raw buffer loads: nt */
template <typename T> __device__ __forceinline__ T ldnt(const uint8_t * rec, uint32_t off) { return __builtin_nontemporal_load((const T *) (rec + off)); }
template <> __device__ __forceinline__ uint4 ldnt<uint4>(const uint8_t * rec, uint32_t off) {
    const u32x4_t v = __builtin_nontemporal_load((const u32x4_t *) (rec + off)); return make_uint4(v.x, v.y, v.z, v.w);
}
template <> __device__ __forceinline__ uint2 ldnt<uint2>(const uint8_t * rec, uint32_t off) {
    const u32x2_t v = __builtin_nontemporal_load((const u32x2_t *) (rec + off)); return make_uint2(v.x, v.y);
}
#if BAMD_BUFFER_LOADS
typedef __amdgpu_buffer_rsrc_t bamd_rsrc;
__device__ __forceinline__ bamd_rsrc weight_rsrc(const void * base) {       // base must be wave-uniform; the window is 2 GiB - 1 (offsets are checked by the launcher's shapes)
    return __builtin_amdgcn_make_buffer_rsrc((void *) uniform_ptr((const uint8_t *) base), 0, 0x7fffffff, 0x00020000);
}
// the same base with ZERO records: every load through it is out of range — it returns 0 and fetches nothing.  The requests a streaming loop issues past the end of
// its work (they stay unconditional so that the compiler's wait counts stay counted) go through this descriptor, chosen by scalar selects: no branch, no bytes
__device__ __forceinline__ bamd_rsrc null_rsrc(const void * base) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) uniform_ptr((const uint8_t *) base), 0, 0, 0x00020000);
}
__device__ __forceinline__ uint4 bl128(bamd_rsrc r, uint32_t voff, int soff) { const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int) voff, soff, BAMD_LOAD_NT); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 bl64(bamd_rsrc r, uint32_t voff, int soff) { const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (int) voff, soff, BAMD_LOAD_NT); return make_uint2(v.x, v.y); }
__device__ __forceinline__ uint32_t bl32(bamd_rsrc r, uint32_t voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b32(r, (int) voff, soff, BAMD_LOAD_NT); }
__device__ __forceinline__ uint32_t bl16(bamd_rsrc r, uint32_t voff, int soff) { return (uint32_t) __builtin_amdgcn_raw_buffer_load_b16(r, (int) voff, soff, BAMD_LOAD_NT); }
#else
struct bamd_rsrc { const uint8_t * base; };
__device__ __forceinline__ bamd_rsrc weight_rsrc(const void * base) { bamd_rsrc r; r.base = uniform_ptr((const uint8_t *) base); return r; }
__device__ __forceinline__ bamd_rsrc null_rsrc(const void * base) { return weight_rsrc(base); }
__device__ __forceinline__ uint4 bl128(bamd_rsrc r, uint32_t voff, int soff) { return ldnt<uint4>(r.base + soff, voff); }
__device__ __forceinline__ uint2 bl64(bamd_rsrc r, uint32_t voff, int soff) { return ldnt<uint2>(r.base + soff, voff); }
__device__ __forceinline__ uint32_t bl32(bamd_rsrc r, uint32_t voff, int soff) { return ldnt<uint32_t>(r.base + soff, voff); }
__device__ __forceinline__ uint32_t bl16(bamd_rsrc r, uint32_t voff, int soff) { return (uint32_t) ldnt<unsigned short>(r.base + soff, voff); }
#endif
// soff: byte offset of the record inside the matrix (wave-uniform)
__device__ __forceinline__ void load_rec(RecQ4K & R, bamd_rsrc rs, int soff, int lane) {
    const uint32_t l = (uint32_t) lane;
    R.qs = bl128(rs, l * 16u, soff);
    R.hd = bl128(rs, 1024u + (l >> 3) * 16u, soff);
#if BAMD_XSCALES
    R.mn47 = bl32(rs, 1152u + (l >> 3) * 4u, soff);
#else
    R.mn47 = 0u;
#endif
}
__device__ __forceinline__ void load_rec(RecQ5K & R, bamd_rsrc rs, int soff, int lane) {
    const uint32_t l = (uint32_t) lane;
    R.qs = bl128(rs, l * 16u, soff);
    R.qh = bl32(rs, 1024u + l * 4u, soff);
    R.hd = bl128(rs, 1280u + (l >> 3) * 16u, soff);
#if BAMD_XSCALES
    R.mn47 = bl32(rs, 1408u + (l >> 3) * 4u, soff);
#else
    R.mn47 = 0u;
#endif
}
__device__ __forceinline__ void load_rec(RecQ6K & R, bamd_rsrc rs, int soff, int lane) {
    const uint32_t l = (uint32_t) lane;
    R.ql = bl128(rs, l * 16u, soff);
    R.qh = bl64(rs, 1024u + l * 8u, soff);
    R.sc = bl64(rs, 1536u + (l >> 3) * 16u + ((l >> 2) & 1u) * 8u, soff);
    R.d  = bl16(rs, 1664u + (l >> 3) * 2u, soff);
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
#if BAMD_XSCALES
    const uint32_t sc03 = R.hd.y, sc47 = R.hd.z, mn03 = R.hd.w, mn47 = R.mn47;      // unpacked at load time (repack_kernel)
#else
    uint32_t sc03, sc47, mn03, mn47; unpack_k4(R.hd, sc03, sc47, mn03, mn47);
#endif
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
#if BAMD_XSCALES
    const uint32_t sc03 = R.hd.y, sc47 = R.hd.z, mn03 = R.hd.w, mn47 = R.mn47;
#else
    uint32_t sc03, sc47, mn03, mn47; unpack_k4(R.hd, sc03, sc47, mn03, mn47);
#endif
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

// ---- block_terms in two halves: what depends on the weight record only (PreTerms, computed while a co-launched workgroup waits for its
//      activations: bamd_colaunch.hip) and the rest.  Same operations on the same values as block_terms above, in the same order per result.
template <int TYPE> struct PreTerms;
template <> struct PreTerms<BAMD_Q4_K> {
    uint32_t wq[8], sc03, sc47; int ma, mb; float dh, dminh;
    __device__ __forceinline__ void prep(const RecQ4K & R, int lane) {
        const int l = lane & 3;
        dh = h2f(R.hd.x & 0xffffu); dminh = h2f(R.hd.x >> 16);
        uint32_t mn03, mn47; unpack_k4(R.hd, sc03, sc47, mn03, mn47);
        wq[0] = R.qs.x & 0x0f0f0f0fu; wq[1] = (R.qs.x >> 4) & 0x0f0f0f0fu; wq[2] = R.qs.y & 0x0f0f0f0fu; wq[3] = (R.qs.y >> 4) & 0x0f0f0f0fu;
        wq[4] = R.qs.z & 0x0f0f0f0fu; wq[5] = (R.qs.z >> 4) & 0x0f0f0f0fu; wq[6] = R.qs.w & 0x0f0f0f0fu; wq[7] = (R.qs.w >> 4) & 0x0f0f0f0fu;
        const uint32_t mw = (l < 2) ? mn03 : mn47;
        const int sh = (l & 1) * 16;
        ma = (int) ((mw >> sh) & 0xffu); mb = (int) ((mw >> (sh + 8)) & 0xffu);
    }
    __device__ __forceinline__ Terms finish(int ci, int lane, const uint32_t * q8, const int * S, const float * yd) const {
        const int e = lane & 7, l = e & 3;
        const float ydv = yd[ci];
        Terms T;
        T.d = ydv * dh; T.dmin = (-ydv) * dminh;
        const uint4 a0 = *(const uint4 *) (q8 + ci * 64 + e * 8), a1 = *(const uint4 *) (q8 + ci * 64 + e * 8 + 4);
        const uint32_t aq[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
        T.fs = (float) dotscale8<false>(wq, aq, sc03, sc47);
        const int2 sp = *(const int2 *) (S + ci * 8 + 2 * l);
        T.pm = (float) (mul24(ma, sp.x) + mul24(mb, sp.y));
        return T;
    }
};
template <> struct PreTerms<BAMD_Q5_K> {
    uint32_t wq[8], sc03, sc47; int me; float dh, dminh;
    __device__ __forceinline__ void prep(const RecQ5K & R, int lane) {
        const int e = lane & 7;
        dh = h2f(R.hd.x & 0xffffu); dminh = h2f(R.hd.x >> 16);
        uint32_t mn03, mn47; unpack_k4(R.hd, sc03, sc47, mn03, mn47);
        const uint32_t qh = R.qh;
#define Q5(w, shift, c) ((((w) >> (shift)) & 0x0f0f0f0fu) | (((qh >> (c)) & 0x01010101u) << 4))
        wq[0] = Q5(R.qs.x, 0, 0); wq[1] = Q5(R.qs.x, 4, 1); wq[2] = Q5(R.qs.y, 0, 2); wq[3] = Q5(R.qs.y, 4, 3);
        wq[4] = Q5(R.qs.z, 0, 4); wq[5] = Q5(R.qs.z, 4, 5); wq[6] = Q5(R.qs.w, 0, 6); wq[7] = Q5(R.qs.w, 4, 7);
#undef Q5
        const uint32_t mw = (e < 4) ? mn03 : mn47;
        me = (int) ((mw >> (8 * (e & 3))) & 0xffu);
    }
    __device__ __forceinline__ Terms finish(int ci, int lane, const uint32_t * q8, const int * S, const float * yd) const {
        const int e = lane & 7;
        const float ydv = yd[ci];
        Terms T;
        T.d = ydv * dh; T.dmin = (-ydv) * dminh;
        const uint4 a0 = *(const uint4 *) (q8 + ci * 64 + e * 8), a1 = *(const uint4 *) (q8 + ci * 64 + e * 8 + 4);
        const uint32_t aq[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
        T.fs = (float) dotscale8<false>(wq, aq, sc03, sc47);
        T.pm = (float) group8_sum(mul24(me, S[ci * 8 + e]));
        return T;
    }
};
template <> struct PreTerms<BAMD_Q6_K> {
    uint32_t wq[8], s0, s1; float dh;
    __device__ __forceinline__ void prep(const RecQ6K & R, int lane) {
        (void) lane;
        dh = h2f(R.d); s0 = R.sc.x; s1 = R.sc.y;
#define Q6(lo, hb) ((((lo) | ((hb) << 4)) + 0x60606060u) ^ 0x80808080u)
        const uint32_t A0 = R.ql.x, B0 = R.ql.y, h0 = R.qh.x, A1 = R.ql.z, B1 = R.ql.w, h1 = R.qh.y;
        wq[0] = Q6(A0 & 0x0f0f0f0fu, h0 & 0x03030303u); wq[1] = Q6(B0 & 0x0f0f0f0fu, (h0 >> 2) & 0x03030303u);
        wq[2] = Q6((A0 >> 4) & 0x0f0f0f0fu, (h0 >> 4) & 0x03030303u); wq[3] = Q6((B0 >> 4) & 0x0f0f0f0fu, (h0 >> 6) & 0x03030303u);
        wq[4] = Q6(A1 & 0x0f0f0f0fu, h1 & 0x03030303u); wq[5] = Q6(B1 & 0x0f0f0f0fu, (h1 >> 2) & 0x03030303u);
        wq[6] = Q6((A1 >> 4) & 0x0f0f0f0fu, (h1 >> 4) & 0x03030303u); wq[7] = Q6((B1 >> 4) & 0x0f0f0f0fu, (h1 >> 6) & 0x03030303u);
#undef Q6
    }
    __device__ __forceinline__ Terms finish(int ci, int lane, const uint32_t * q8, const int * S, const float * yd) const {
        (void) S;
        const int e = lane & 7;
        Terms T;
        T.d = yd[ci] * dh; T.dmin = 0.f; T.pm = 0.f;
        const uint4 a0 = *(const uint4 *) (q8 + ci * 64 + e * 8), a1 = *(const uint4 *) (q8 + ci * 64 + e * 8 + 4);
        const uint32_t aq[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
        T.fs = (float) dotscale8<true>(wq, aq, s0, s1);
        return T;
    }
};

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

struct ProArgs { const float * x, * nw; float eps; int K; uint32_t * q8; int * S; float * yd; double * red; unsigned long long * tl; };
#define BAMD_PRO_ISSUE(ap, pa) do { (ap).tl = (pa).tl; (ap).issue((pa).x, (pa).nw, (pa).K, wave_id()); } while (0)
#define BAMD_PRO_FINISH(ap, pa) (ap).finish((pa).x, (pa).nw, (pa).eps, (pa).K, (pa).q8, (pa).S, (pa).yd, (pa).red)
#define BAMD_PRO_FINISH_SMALLK(ap, pa) (ap).template finish<true>((pa).x, (pa).nw, (pa).eps, (pa).K, (pa).q8, (pa).S, (pa).yd, (pa).red)
#define BAMD_PRO_FINISH_SMALLK_MID(ap, pa, mid) (ap).template finish<true>((pa).x, (pa).nw, (pa).eps, (pa).K, (pa).q8, (pa).S, (pa).yd, (pa).red, mid)
// the same with NB batch slots (the launcher guarantees K <= 256 * NB * waves)
#define BAMD_PRO_ISSUE_NB(ap, pa, NB_) do { (ap).tl = (pa).tl; (ap).template issue<NB_>((pa).x, (pa).nw, (pa).K, wave_id()); } while (0)
#define BAMD_PRO_FINISH_NB_MID(ap, pa, mid, NB_) (ap).template finish<true, decltype(mid), NB_>((pa).x, (pa).nw, (pa).eps, (pa).K, (pa).q8, (pa).S, (pa).yd, (pa).red, mid)

__device__ __forceinline__ void get_scale_min_k4(int j, const uint8_t * q, int & d, int & m) {
    if (j < 4) { d = q[j] & 63; m = q[j + 4] & 63; }
    else { d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4); }
}

// get_rows of one token (ggml.c:13186-13228 -> dequantize_row_*): row `tok` of the embedding matrix (GGUF layout) -> x[E]
__device__ __forceinline__ void embed_row(const uint8_t * embd, int embd_type, int E, int tok, float * x) {
    // get_rows: ggml.c:13186-13228 -> dequantize_row_*
    if (embd_type == BAMD_F32) {
        const float * src = (const float *) embd + (size_t) tok * E;
        for (int i = threadIdx.x; i < E; i += blockDim.x) ik_st(x + i, src[i]);
    } else if (embd_type == BAMD_F16) {
        const unsigned short * src = (const unsigned short *) embd + (size_t) tok * E;
        for (int i = threadIdx.x; i < E; i += blockDim.x) ik_st(x + i, h2f(src[i]));
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
            ik_st(x + i, y);
        }
    }
}

// ===========================================================================================================
// Attention (single token): RoPE + KV store + scores + softmax + P.V         (reference: llm_build_kv, llama.cpp:8318)
// ===========================================================================================================
// KV cache, "chain-major" physical order (logically the reference's K [n_ctx][Hkv*hd] f16 and V^T [Hkv*hd][n_ctx] f16,
// llama.cpp:7845-7875; bamd_op_attention converts at the boundary):
//   K : inside each head row, element n = 8l + e (lane e, chain step l) is stored at index (l >> 3)*64 + e*8 + (l & 7): the 16-byte group g = l >> 3 of
//       the eight lanes is ONE 128-byte line, so a wave's load instruction for group g covers whole lines (round 6; rounds 1-5 kept a lane's L = hd/8
//       halves contiguous — e*L + l — and every load instruction touched HALF of two lines per position: harmless while the vector L1 caught the second
//       half, 1.4 us per attention launch once the loads had to bypass it — sc1, "Inter-kernel data" above; profiles/r06_levers.txt)
//   V^T: inside each row, position p = 64B + 8l + e is stored at index 64B + 8e + l
// The reference's attention mat-muls (tinyBLAS, sgemm.cpp:405-431) keep 8 SIMD lanes e, each a sequential f32 chain over
// the steps l.  With this order the wave lane that stands for SIMD lane e finds the operands of consecutive steps
// in 16-byte runs: 16-byte loads straight from HBM/L2, no LDS staging, no gather.  The LDS copies of q / k (qt, q16t, k16t) use the same order.
__device__ __forceinline__ int kperm(int n, int L) { (void) L; const int l = n >> 3; return ((l >> 3) << 6) + ((n & 7) << 3) + (l & 7); }
#define BAMD_KGRP 64                 /* halves (or floats) between consecutive 16-byte groups of one lane's chain */
__device__ __forceinline__ int vperm(int p) { return (p & ~63) + ((p & 7) << 3) + ((p & 63) >> 3); }

// eight steps of ggml_vec_dot_f16's four interleaved accumulators on f16 pairs: v_fma_mix_f32 extends both halves and fuses the
// multiply-add in one instruction (= fmaf((float) k, (float) q, acc) exactly); hipcc emits two conversions + half a v_pk_fma_f32 per
// step instead.  Element u is the low (u even) / high half of word u >> 1; steps u and u + 4 share an accumulator (distance 4).
__device__ __forceinline__ void fma_mix8(float (&a4)[4], const uint32_t (&k)[4], const uint32_t (&q)[4]) {
    asm("v_fma_mix_f32 %0, %4, %8, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
        "v_fma_mix_f32 %1, %4, %8, %1 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n\t"
        "v_fma_mix_f32 %2, %5, %9, %2 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
        "v_fma_mix_f32 %3, %5, %9, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n\t"
        "v_fma_mix_f32 %0, %6, %10, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
        "v_fma_mix_f32 %1, %6, %10, %1 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n\t"
        "v_fma_mix_f32 %2, %7, %11, %2 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
        "v_fma_mix_f32 %3, %7, %11, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]"
        : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3])
        : "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]));
}
// N (4 or 8) steps of ONE sequential chain acc = fmaf((float) v_u, p_u, acc) over the halves of w (element u = low / high half of word u >> 1): the
// tinyBLAS P.V step.  Inline asm because hipcc puts an s_nop behind every dependent v_fma_mix_f32 whose f16 operand is selected by op_sel_hi (it
// takes that modifier bit for a write to the high half of the destination); a dependent vector instruction issues ~12 clocks after its producer
// on this chip anyway, and chains of 4096 such steps give the same bits with and without the wait state (tools/chain_probe.hip).
template <int N>
__device__ __forceinline__ float fma_mix_chain(float acc, const uint32_t (&w)[4], const float (&p)[8]) {
    static_assert(N == 4 || N == 8, "half a block or a block");
    if (N == 4)
        asm("v_fma_mix_f32 %0, %1, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %1, %4, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %2, %5, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %2, %6, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "+v"(acc) : "v"(w[0]), "v"(w[1]), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]));
    else                                                       // one block: a single wait for its operands in front of it
        asm("v_fma_mix_f32 %0, %1, %5, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %1, %6, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %2, %7, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %2, %8, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %3, %9, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %3, %10, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %4, %11, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %4, %12, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "+v"(acc) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]));
    return acc;
}
// dot of up to 32 steps for lane e: kv = this lane's L halves of the K row (L <= 32), qf / qh = this lane's floats / halves of group 0 (base + e * 8),
// group g another BAMD_KGRP elements on (kperm)
template <bool PREFILL>
__device__ __forceinline__ float kq_chain(const uint4 (&kv)[4], int L, const float * qf, const unsigned short * qh) {
    if (!PREFILL) {
        float acc = 0.f;                                           // tinyBLAS F16 x F32, KN = 8 (sgemm.cpp:405-431)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g * 8 < L) {
                const uint32_t w[4] = { kv[g].x, kv[g].y, kv[g].z, kv[g].w };
                const float4 qa = *(const float4 *) (qf + g * BAMD_KGRP), qb = *(const float4 *) (qf + g * BAMD_KGRP + 4);
                const float qv[8] = { qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w };
                acc = fma_mix_chain<8>(acc, w, qv);              // acc = fmaf((float) k_u, q_u, acc), u = 0..7 (hipcc: a v_cvt_f32_f16 per step beside the fma)
            }
        }
        return acc;
    } else {
        float a4[4] = { 0.f, 0.f, 0.f, 0.f };                      // ggml_vec_dot_f16: 4 accumulators x 8 lanes (ggml.c:2038)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g * 8 < L) {
                const uint32_t w[4] = { kv[g].x, kv[g].y, kv[g].z, kv[g].w };
                const uint4 qq = *(const uint4 *) (qh + g * BAMD_KGRP);
                const uint32_t qw[4] = { qq.x, qq.y, qq.z, qq.w };
                fma_mix8(a4, w, qw);                               // a4[u & 3] = fmaf((float) k_u, (float) q_u, a4[u & 3]), u = 0..7
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


// dynamic LDS of a mat-vec workgroup: quantised activations + reduction scratch (host side of carve_lds)
static inline size_t act_lds_bytes(int K) {
    const int nb = K >> 8;
    size_t b = (size_t) nb * (256 + 32 + 4);
    b = (b + 15) & ~(size_t) 15;
    return b + 16 * sizeof(double) + 16 * sizeof(unsigned long long);
}


// one token's Q8_K activations in global memory: LDS layout (matmul_batch_kernel) and f16 MFMA layout (matmul_mfma_*), bamd_prefill.hip
#define BAMD_TT 8                       /* tokens per workgroup tile of matmul_batch_kernel */
#define BAMD_BLOB_BYTES(nb) (BAMD_ACT_RED_OFF(nb))
#ifndef BAMD_B16_REC
#define BAMD_B16_REC 608                 /* 512 B of dot operands + 32 B of min-term operands + 16 B of i16 block sums + pad; 152 dwords = 24 mod 64: the
                                            ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) see 16 distinct banks-of-4 (560 B: 2-way conflicts, SQ_LDS_BANK_CONFLICT 17.9M -> 9.1M per launch) */
#endif
#define BAMD_B16_Q (BAMD_B16_REC / 16)
#define BAMD_BLOB16_BYTES(nb) ((((size_t) (nb) * (BAMD_B16_REC + 4)) + 15) & ~(size_t) 15)   /* per-token stride: records + d_y floats, 16-byte multiple */
