/* booster_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See booster_oracle.h.
 *
 * Restates, operation for operation, the arithmetic of the reference's AVX2+FMA+F16C CPU build ("A2").
 * Vector code of the reference is written here as scalar loops over the 8 SIMD lanes; a lane of a 256-bit
 * register is an index e in [0,8).  All float expressions are single IEEE operations (this file is compiled
 * with -ffp-contract=off, like the reference's -std=c11 C files); fused multiply-adds of the reference's
 * intrinsics are explicit fmaf().  Integer work is exact, so only its values (not its order) are restated.
 * Must NOT be compiled with -ffast-math.
 */
#include "booster_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------------
 * fp16 <-> fp32: IEEE binary16, round-to-nearest-even — what F16C _cvtss_sh/_cvtsh_ss do
 * (ggml-impl.h GGML_COMPUTE_FP32_TO_FP16 / GGML_COMPUTE_FP16_TO_FP32 under __F16C__).
 * ------------------------------------------------------------------------------------------------------ */
float bo_fp16_to_fp32(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu, u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; ++sh; } m &= 0x3ffu; u = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}
uint16_t bo_fp32_to_fp16(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000u; int32_t e = (int32_t)((u >> 23) & 0xff); uint32_t m = u & 0x7fffffu;
    if (e == 255) return (uint16_t)(s | 0x7c00u | (m ? (0x200u | (m >> 13)) : 0));
    int32_t he = e - 127 + 15;
    if (he >= 31) return (uint16_t)(s | 0x7c00u);
    if (he <= 0) {
        if (he < -10) return (uint16_t) s;
        m |= 0x800000u;
        int sh = 14 - he;                       /* 14..24 */
        uint32_t hm = m >> sh, rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (hm & 1))) ++hm;
        return (uint16_t)(s | hm);
    }
    uint32_t hm = m >> 13, rem = m & 0x1fffu;
    uint32_t r = ((uint32_t) he << 10) | hm;
    if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) ++r;   /* carry into exponent is the right answer */
    return (uint16_t)(s | r);
}

size_t bo_row_size(int type, int64_t n) {
    switch (type) {
        case BO_TYPE_F32:  return (size_t) n * 4;
        case BO_TYPE_F16:  return (size_t) n * 2;
        case BO_TYPE_Q4_K: return (size_t)(n / BO_QK_K) * sizeof(bo_block_q4_K);
        case BO_TYPE_Q5_K: return (size_t)(n / BO_QK_K) * sizeof(bo_block_q5_K);
        case BO_TYPE_Q6_K: return (size_t)(n / BO_QK_K) * sizeof(bo_block_q6_K);
        case BO_TYPE_Q8_K: return (size_t)(n / BO_QK_K) * sizeof(bo_block_q8_K);
    }
    return 0;
}

/* ggml-quants.c:1632-1637 nearest_int: magic-number round-half-even */
static inline int nearest_int(float fval) {
    float val = fval + 12582912.f;
    int i; memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

/* ggml-quants.c:3593-3630 quantize_row_q8_K_ref (quantize_row_q8_K :3643 forwards to it) */
void bo_quantize_row_q8_K(const float * x, bo_block_q8_K * y, int64_t k) {
    const int64_t nb = k / BO_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        float max = 0, amax = 0;
        for (int j = 0; j < BO_QK_K; ++j) {
            float ax = fabsf(x[j]);
            if (ax > amax) { amax = ax; max = x[j]; }
        }
        if (!amax) {
            /* the reference leaves bsums untouched here (:3608-3612); its callers' wdata is not zeroed, but a
             * zero d multiplies every use of bsums by 0 only in the scale path, so we define them as 0 */
            y[i].d = 0; memset(y[i].qs, 0, BO_QK_K); memset(y[i].bsums, 0, sizeof y[i].bsums);
            x += BO_QK_K; continue;
        }
        const float iscale = -127.f / max;
        for (int j = 0; j < BO_QK_K; ++j) {
            int v = nearest_int(iscale * x[j]);
            y[i].qs[j] = (int8_t)(v < 127 ? v : 127);
        }
        for (int j = 0; j < BO_QK_K / 16; ++j) {
            int sum = 0;
            for (int ii = 0; ii < 16; ++ii) sum += y[i].qs[j * 16 + ii];
            y[i].bsums[j] = (int16_t) sum;
        }
        y[i].d = 1 / iscale;
        x += BO_QK_K;
    }
}

/* ggml-quants.c:1891-1899 get_scale_min_k4 */
static inline void get_scale_min_k4(int j, const uint8_t * q, uint8_t * d, uint8_t * m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4); }
}

/* ggml-quants.c:2548-2570 / 2756-2783 / 2970-2999 */
static void dequantize_row_q4_K(const bo_block_q4_K * x, float * y, int64_t k) {
    for (int64_t i = 0; i < k / BO_QK_K; i++) {
        const uint8_t * q = x[i].qs;
        const float d = bo_fp16_to_fp32(x[i].d), min = bo_fp16_to_fp32(x[i].dmin);
        int is = 0; uint8_t sc, m;
        for (int j = 0; j < BO_QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m); const float d1 = d * sc, m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m); const float d2 = d * sc, m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l] >> 4) - m2;
            q += 32; is += 2;
        }
    }
}
static void dequantize_row_q5_K(const bo_block_q5_K * x, float * y, int64_t k) {
    for (int64_t i = 0; i < k / BO_QK_K; i++) {
        const uint8_t * ql = x[i].qs, * qh = x[i].qh;
        const float d = bo_fp16_to_fp32(x[i].d), min = bo_fp16_to_fp32(x[i].dmin);
        int is = 0; uint8_t sc, m, u1 = 1, u2 = 2;
        for (int j = 0; j < BO_QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m); const float d1 = d * sc, m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m); const float d2 = d * sc, m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * ((ql[l] & 0xF) + (qh[l] & u1 ? 16 : 0)) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * ((ql[l] >> 4) + (qh[l] & u2 ? 16 : 0)) - m2;
            ql += 32; is += 2; u1 <<= 2; u2 <<= 2;
        }
    }
}
static void dequantize_row_q6_K(const bo_block_q6_K * x, float * y, int64_t k) {
    for (int64_t i = 0; i < k / BO_QK_K; i++) {
        const float d = bo_fp16_to_fp32(x[i].d);
        const uint8_t * ql = x[i].ql, * qh = x[i].qh; const int8_t * sc = x[i].scales;
        for (int n = 0; n < BO_QK_K; n += 128) {
            for (int l = 0; l < 32; ++l) {
                int is = l / 16;
                const int8_t q1 = (int8_t)((ql[l +  0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int8_t q2 = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int8_t q3 = (int8_t)((ql[l +  0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int8_t q4 = (int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l +  0] = d * sc[is + 0] * q1;
                y[l + 32] = d * sc[is + 2] * q2;
                y[l + 64] = d * sc[is + 4] * q3;
                y[l + 96] = d * sc[is + 6] * q4;
            }
            y += 128; ql += 64; qh += 32; sc += 8;
        }
    }
}
void bo_dequantize_row(int type, const void * x, float * y, int64_t k) {
    switch (type) {
        case BO_TYPE_F32:  memcpy(y, x, (size_t) k * 4); break;
        case BO_TYPE_F16:  for (int64_t i = 0; i < k; ++i) y[i] = bo_fp16_to_fp32(((const uint16_t *) x)[i]); break;
        case BO_TYPE_Q4_K: dequantize_row_q4_K((const bo_block_q4_K *) x, y, k); break;
        case BO_TYPE_Q5_K: dequantize_row_q5_K((const bo_block_q5_K *) x, y, k); break;
        case BO_TYPE_Q6_K: dequantize_row_q6_K((const bo_block_q6_K *) x, y, k); break;
        default: assert(0);
    }
}

/* hsum_float_8, ggml-quants.c:47-53: (hi128+lo128) -> (+movehl) -> (+movehdup) */
static inline float hsum_float_8(const float a[8]) {
    float r0 = a[4] + a[0], r1 = a[5] + a[1], r2 = a[6] + a[2], r3 = a[7] + a[3];
    r0 = r0 + r2; r1 = r1 + r3;
    return r0 + r1;
}

/* the 6-bit scale/min unpack of ggml-quants.c:6928-6933 (utmp shuffle) = get_scale_min_k4 for j = 0..7 */
static inline void unpack_k4(const uint8_t * scales, uint8_t sc[8], uint8_t mn[8]) {
    for (int j = 0; j < 8; ++j) get_scale_min_k4(j, scales, &sc[j], &mn[j]);
}

/* ggml-quants.c:6914-6978 (__AVX2__ branch of ggml_vec_dot_q4_K_q8_K).
 * lane e of `sumi` = sum over the 8 sub-blocks c of scale_c * sum_{t<4} q4[c][4e+t]*q8[c][4e+t]
 * (maddubs pairs bytes, madd_epi16 pairs words: 4 consecutive elements per 32-bit lane). */
float bo_vec_dot_q4_K_q8_K(int n, const bo_block_q4_K * x, const bo_block_q8_K * y) {
    const int nb = n / BO_QK_K;
    float acc[8] = {0}, acc_m[4] = {0};
    for (int i = 0; i < nb; ++i) {
        const float d = y[i].d * bo_fp16_to_fp32(x[i].d);
        const float dmin = -y[i].d * bo_fp16_to_fp32(x[i].dmin);
        uint8_t sc[8], mn[8]; unpack_k4(x[i].scales, sc, mn);
        int S[8]; for (int j = 0; j < 8; ++j) S[j] = y[i].bsums[2 * j] + y[i].bsums[2 * j + 1];      /* _mm_hadd_epi16 */
        for (int l = 0; l < 4; ++l) {                                                                  /* _mm_madd_epi16(mins, q8s) */
            const int prod = mn[2 * l] * S[2 * l] + mn[2 * l + 1] * S[2 * l + 1];
            acc_m[l] = fmaf(dmin, (float) prod, acc_m[l]);
        }
        int sumi[8] = {0};
        const uint8_t * q4 = x[i].qs; const int8_t * q8 = y[i].qs;
        for (int j = 0; j < BO_QK_K / 64; ++j) {
            for (int e = 0; e < 8; ++e) {
                int pl = 0, ph = 0;
                for (int t = 0; t < 4; ++t) {
                    const uint8_t q = q4[32 * j + 4 * e + t];
                    pl += (q & 0xF) * q8[64 * j + 4 * e + t];
                    ph += (q >> 4)  * q8[64 * j + 32 + 4 * e + t];
                }
                sumi[e] += sc[2 * j] * pl + sc[2 * j + 1] * ph;
            }
        }
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(d, (float) sumi[e], acc[e]);
    }
    const float m0 = acc_m[0] + acc_m[2], m1 = acc_m[1] + acc_m[3];      /* movehl, then movehdup */
    return hsum_float_8(acc) + (m0 + m1);
}

/* ggml-quants.c:7487-7564 (__AVX2__ branch of ggml_vec_dot_q5_K_q8_K) */
float bo_vec_dot_q5_K_q8_K(int n, const bo_block_q5_K * x, const bo_block_q8_K * y) {
    const int nb = n / BO_QK_K;
    float acc[8] = {0};
    float summs = 0.f;
    for (int i = 0; i < nb; ++i) {
        const float d = y[i].d * bo_fp16_to_fp32(x[i].d);
        const float dmin = -y[i].d * bo_fp16_to_fp32(x[i].dmin);
        uint8_t sc[8], mn[8]; unpack_k4(x[i].scales, sc, mn);
        int hs = 0;
        for (int j = 0; j < 8; ++j) hs += mn[j] * (y[i].bsums[2 * j] + y[i].bsums[2 * j + 1]);
        summs += dmin * hs;                                                /* :7518, int -> float, mul, add */
        int sumi[8] = {0};
        const uint8_t * q5 = x[i].qs, * qh = x[i].qh; const int8_t * q8 = y[i].qs;
        for (int j = 0; j < BO_QK_K / 64; ++j) {
            for (int e = 0; e < 8; ++e) {
                int p0 = 0, p1 = 0;
                for (int t = 0; t < 4; ++t) {
                    const int l = 4 * e + t;
                    const uint8_t q = q5[32 * j + l];
                    const int v0 = (q & 0xF) + (((qh[l] >> (2 * j))     & 1) << 4);
                    const int v1 = (q >> 4)  + (((qh[l] >> (2 * j + 1)) & 1) << 4);
                    p0 += v0 * q8[64 * j + l];
                    p1 += v1 * q8[64 * j + 32 + l];
                }
                sumi[e] += sc[2 * j] * p0 + sc[2 * j + 1] * p1;
            }
        }
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(d, (float) sumi[e], acc[e]);
    }
    return hsum_float_8(acc) + summs;
}

/* ggml-quants.c:8145-8222 (__AVX2__ branch of ggml_vec_dot_q6_K_q8_K).
 * chunk c (32 elements) of the super-block: c = 4*half + {0,1,2,3} <- (ql lo[0..31], ql lo[32..63], ql hi[0..31], ql hi[32..63]);
 * 16-element scale index = 2c + (e >= 4). */
float bo_vec_dot_q6_K_q8_K(int n, const bo_block_q6_K * x, const bo_block_q8_K * y) {
    const int nb = n / BO_QK_K;
    float acc[8] = {0};
    for (int i = 0; i < nb; ++i) {
        const float d = y[i].d * bo_fp16_to_fp32(x[i].d);
        int sumi[8] = {0};
        for (int half = 0; half < 2; ++half) {
            const uint8_t * ql = x[i].ql + 64 * half, * qh = x[i].qh + 32 * half;
            const int8_t * q8 = y[i].qs + 128 * half, * sc = x[i].scales + 8 * half;
            for (int e = 0; e < 8; ++e) {
                int p[4] = {0, 0, 0, 0};
                for (int t = 0; t < 4; ++t) {
                    const int l = 4 * e + t;
                    const int v0 = ((ql[l]      & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                    const int v1 = ((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                    const int v2 = ((ql[l]      >> 4)  | (((qh[l] >> 4) & 3) << 4)) - 32;
                    const int v3 = ((ql[l + 32] >> 4)  | (((qh[l] >> 6) & 3) << 4)) - 32;
                    p[0] += v0 * q8[l]; p[1] += v1 * q8[32 + l]; p[2] += v2 * q8[64 + l]; p[3] += v3 * q8[96 + l];
                }
                const int hi = e >= 4;
                sumi[e] += sc[0 + hi] * p[0] + sc[2 + hi] * p[1] + sc[4 + hi] * p[2] + sc[6 + hi] * p[3];
            }
        }
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(d, (float) sumi[e], acc[e]);
    }
    return hsum_float_8(acc);
}

static float vec_dot_q(int type, int n, const void * w, const bo_block_q8_K * y) {
    switch (type) {
        case BO_TYPE_Q4_K: return bo_vec_dot_q4_K_q8_K(n, (const bo_block_q4_K *) w, y);
        case BO_TYPE_Q5_K: return bo_vec_dot_q5_K_q8_K(n, (const bo_block_q5_K *) w, y);
        case BO_TYPE_Q6_K: return bo_vec_dot_q6_K_q8_K(n, (const bo_block_q6_K *) w, y);
    }
    assert(0); return 0;
}

/* ggml.c:12277-12492: src1 rows -> Q8_K once (:12345-12372), then one vec_dot per (row, col) (:12186-12275).
 * The result does not depend on the thread count or chunking. */
void bo_mul_mat_q(int type, const void * W, int64_t nrows, int64_t K, const float * x, int64_t T, float * y, int nthreads) {
    const size_t rs = bo_row_size(type, K);
    const int64_t nb = K / BO_QK_K;
    bo_block_q8_K * q = (bo_block_q8_K *) malloc((size_t) T * nb * sizeof(bo_block_q8_K));
    for (int64_t t = 0; t < T; ++t) bo_quantize_row_q8_K(x + t * K, q + t * nb, K);
    (void) nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t r = 0; r < nrows; ++r)
        for (int64_t t = 0; t < T; ++t)
            y[t * nrows + r] = vec_dot_q(type, (int) K, (const char *) W + r * rs, q + t * nb);
    free(q);
}

/* ggml.c:11850-11896 ggml_compute_forward_rms_norm_f32 (ggml_float = double, sequential sum) */
void bo_rms_norm(const float * x, float * y, int64_t n, float eps) {
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) sum += (double)(x[i] * x[i]);
    const float mean = (float)(sum / (double) n);
    const float scale = 1.0f / sqrtf(mean + eps);
    for (int64_t i = 0; i < n; i++) y[i] = x[i] * scale;               /* ggml_vec_scale_f32 */
}

/* ggml.c:2490-2522 ggml_v_expf (__AVX2__ && __FMA__), one lane */
float bo_v_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = fmaf(-n, 0x1.7f7d1cp-20f, fmaf(-n, 0x1.62e4p-1f, x));       /* fnmadd(a,b,c) = -(a*b)+c */
    uint32_t zb; memcpy(&zb, &z, 4);
    const uint32_t e = zb << 23;
    uint32_t one; { const float o = 1.0f; memcpy(&one, &o, 4); }
    uint32_t kb = e + one; float k; memcpy(&k, &kb, 4);
    const int c = fabsf(n) > 126.0f;                                             /* _CMP_GT_OQ: false on NaN */
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u,
                         0x1.ffffecp-1f * b);
    if (!c) return fmaf(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    uint32_t s1b = g + 0x7f000000u, s2b = e - g; float s1, s2; memcpy(&s1, &s1b, 4); memcpy(&s2, &s2b, 4);
    const int dd = fabsf(n) > 192.0f;
    if (dd) return s1 * s1;
    return fmaf(s2, j, s2) * s1;
}
/* ggml.c:2524-2531 ggml_v_silu */
float bo_v_silu(float x) {
    const float neg_x = 0.0f - x;
    const float one_plus = 1.0f + bo_v_expf(neg_x);
    return x / one_plus;
}

/* ggml.c:13682-13778 ggml_compute_forward_soft_max_f32 + ggml_vec_soft_max_f32 (:2619-2671, AVX2 branch).
 * n must be a multiple of 8 here (n_kv is padded to 32, llama.cpp:14693-14701). mask may be NULL. */
void bo_soft_max(const float * s, const float * mask, float scale, float * p, int n) {
    float * wp = (float *) malloc((size_t) n * 4);
    for (int i = 0; i < n; ++i) wp[i] = s[i] * scale;                    /* ggml_vec_scale_f32 */
    if (mask) for (int i = 0; i < n; ++i) wp[i] += 1.0f * mask[i];      /* slope = 1 */
    float max = -INFINITY;
    for (int i = 0; i < n; ++i) max = wp[i] > max ? wp[i] : max;        /* ggml_vec_max_f32 */
    double sum = 0;
    int i = 0;
    for (; i + 7 < n; i += 8) {
        float v[8];
        for (int e = 0; e < 8; ++e) { v[e] = bo_v_expf(wp[i + e] - max); p[i + e] = v[e]; }
        float a0 = v[4] + v[0], a1 = v[5] + v[1], a2 = v[6] + v[2], a3 = v[7] + v[3];
        a0 = a0 + a2; a1 = a1 + a3;
        sum += (double)(a0 + a1);
    }
    for (; i < n; ++i) { float val = expf(wp[i] - max); sum += (double) val; p[i] = val; }
    sum = 1.0 / sum;
    const float fs = (float) sum;
    for (int k = 0; k < n; ++k) p[k] *= fs;                              /* ggml_vec_scale_f32(nc, dp, sum) */
    free(wp);
}

/* ggml.c:13994-14041 rope_yarn, ggml_rope_yarn_corr_dims, ggml_rope_cache_init */
static float rope_yarn_ramp(const float low, const float high, const int i0) {
    const float y = (i0 / 2 - low) / fmaxf(0.001f, high - low);
    return 1 - fminf(1, fmaxf(0, y));
}
static float rope_yarn_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}
void bo_rope_cache(float * cache, int32_t pos, int n_dims, float freq_base, float freq_scale, const float * freq_factors,
                   float ext_factor, float attn_factor, int n_ctx_orig, float beta_fast, float beta_slow) {
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    float corr_dims[2];
    {
        float start = floorf(rope_yarn_corr_dim(n_dims, n_ctx_orig, beta_fast, freq_base));
        float end   = ceilf(rope_yarn_corr_dim(n_dims, n_ctx_orig, beta_slow, freq_base));
        corr_dims[0] = start > 0 ? start : 0;
        corr_dims[1] = end < n_dims - 1 ? end : n_dims - 1;
    }
    float theta = (float) pos;
    for (int i0 = 0; i0 < n_dims; i0 += 2) {
        const float ff = freq_factors ? freq_factors[i0 / 2] : 1.0f;
        const float theta_extrap = theta / ff;
        float theta_interp = freq_scale * theta_extrap;
        float th = theta_interp, mscale = attn_factor;
        if (ext_factor != 0.0f) {
            float ramp_mix = rope_yarn_ramp(corr_dims[0], corr_dims[1], i0) * ext_factor;
            th = theta_interp * (1 - ramp_mix) + theta_extrap * ramp_mix;
            mscale *= 1.0f + 0.1f * logf(1.0f / freq_scale);
        }
        cache[i0 + 0] = cosf(th) * mscale;
        cache[i0 + 1] = sinf(th) * mscale;
        cache[i0 + 1] *= 1.0f;                                           /* sin_sign (forward) */
        theta *= theta_scale;
    }
}
/* ggml.c:14130-14143 (NORM mode: adjacent pairs) */
void bo_rope_apply(float * x, const float * cache, int n_dims) {
    for (int i0 = 0; i0 < n_dims; i0 += 2) {
        const float c = cache[i0], s = cache[i0 + 1];
        const float x0 = x[i0], x1 = x[i0 + 1];
        x[i0]     = x0 * c - x1 * s;
        x[i0 + 1] = x0 * s + x1 * c;
    }
}

/* sgemm.cpp:405-431 tinyBLAS<8,__m256,__m256,ggml_fp16_t,float,float>::gemm — one C element:
 * Cv = madd(load(A+l), load(B+l), Cv) for l += 8, then hsum (sgemm.cpp:164-184). */
float bo_dot_f16_f32_tinyblas(const uint16_t * a, const float * b, int64_t k) {
    float cv[8] = {0};
    for (int64_t l = 0; l < k; l += 8)
        for (int e = 0; e < 8; ++e) cv[e] = fmaf(bo_fp16_to_fp32(a[l + e]), b[l + e], cv[e]);
    float r0 = cv[4] + cv[0], r1 = cv[5] + cv[1], r2 = cv[6] + cv[2], r3 = cv[7] + cv[3];
    r0 = r0 + r2; r1 = r1 + r3;
    return r0 + r1;
}

/* ggml.c:2038-2079 ggml_vec_dot_f16 with the AVX F16C macros (:1268-1345): 4 accumulators x 8 lanes */
float bo_vec_dot_f16(int n, const uint16_t * x, const uint16_t * y) {
    float sum[4][8]; memset(sum, 0, sizeof sum);
    const int np = n & ~31;
    for (int i = 0; i < np; i += 32)
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 8; ++e)
                sum[j][e] = fmaf(bo_fp16_to_fp32(x[i + 8 * j + e]), bo_fp16_to_fp32(y[i + 8 * j + e]), sum[j][e]);
    float t[8];
    for (int e = 0; e < 8; ++e) { float a = sum[0][e] + sum[2][e], b = sum[1][e] + sum[3][e]; t[e] = a + b; }
    float t0[4]; for (int e = 0; e < 4; ++e) t0[e] = t[e] + t[e + 4];
    double sumf = (double)((t0[0] + t0[1]) + (t0[2] + t0[3]));           /* two _mm_hadd_ps */
    for (int i = np; i < n; ++i) sumf += (double)(bo_fp16_to_fp32(x[i]) * bo_fp16_to_fp32(y[i]));
    return (float) sumf;
}

/* ======================================================================================================
 * whole model
 * ====================================================================================================== */
struct bo_ctx {
    const bo_model * m; int n_ctx, nthreads;
    uint16_t ** k, ** v;
    float * logits;
    float * rope;           /* [n_ctx][hd] lazily built? no: built per call */
    bo_tap_fn tap; void * tap_ud;
};

bo_ctx * bo_ctx_new(const bo_model * m, int n_ctx, int nthreads) {
    bo_ctx * c = (bo_ctx *) calloc(1, sizeof *c);
    c->m = m; c->n_ctx = n_ctx; c->nthreads = nthreads > 0 ? nthreads : 1;
    c->k = (uint16_t **) calloc(m->L, sizeof(void *)); c->v = (uint16_t **) calloc(m->L, sizeof(void *));
    const size_t kvn = (size_t) n_ctx * m->Hkv * m->hd;
    for (int il = 0; il < m->L; ++il) { c->k[il] = (uint16_t *) calloc(kvn, 2); c->v[il] = (uint16_t *) calloc(kvn, 2); }   /* llama.cpp:2989-3020 zero-init */
    c->logits = (float *) calloc(m->V, 4);
    return c;
}
void bo_ctx_free(bo_ctx * c) {
    for (int il = 0; il < c->m->L; ++il) { free(c->k[il]); free(c->v[il]); }
    free(c->k); free(c->v); free(c->logits); free(c);
}
void bo_ctx_set_tap(bo_ctx * c, bo_tap_fn fn, void * ud) { c->tap = fn; c->tap_ud = ud; }
void bo_kv_clear(bo_ctx * c) { (void) c; /* llama_kv_cache_clear only resets cell metadata (llama.cpp:3230-3245); data stays */ }
const float * bo_get_logits(const bo_ctx * c) { return c->logits; }
const uint16_t * bo_kv_k(const bo_ctx * c, int il) { return c->k[il]; }
const uint16_t * bo_kv_v(const bo_ctx * c, int il) { return c->v[il]; }

#define TAP(name, il, ptr, n) do { if (c->tap) c->tap(c->tap_ud, name, il, ptr, (int64_t)(n)); } while (0)

int bo_decode(bo_ctx * c, const int32_t * tokens, int T, int n_past) {
    const bo_model * m = c->m;
    const int E = m->E, H = m->H, Hkv = m->Hkv, hd = m->hd, F = m->F, V = m->V, n_ctx = c->n_ctx, nth = c->nthreads;
    const int Ekv = Hkv * hd, gq = H / Hkv;
    if (n_past + T > n_ctx) return 1;
    /* llama.cpp:14693-14701: n_kv = min(size, max(32, GGML_PAD(cell_max, 32))) */
    int n_kv = ((n_past + T) + 31) / 32 * 32; if (n_kv < 32) n_kv = 32; if (n_kv > n_ctx) n_kv = n_ctx;
    const float kq_scale = 1.0f / sqrtf((float) hd);                     /* llama.cpp:8829 */

    float * x    = (float *) malloc((size_t) T * E * 4);                 /* residual stream inpL */
    float * cur  = (float *) malloc((size_t) T * E * 4);
    float * q    = (float *) malloc((size_t) T * E * 4);
    float * kk   = (float *) malloc((size_t) T * Ekv * 4);
    float * vv   = (float *) malloc((size_t) T * Ekv * 4);
    float * att  = (float *) malloc((size_t) T * E * 4);
    float * ffi  = (float *) malloc((size_t) T * E * 4);
    float * g    = (float *) malloc((size_t) T * F * 4);
    float * u    = (float *) malloc((size_t) T * F * 4);
    float * kq   = (float *) malloc((size_t) n_kv * 4);
    float * pr   = (float *) malloc((size_t) n_kv * 4);
    float * mask = (float *) malloc((size_t) n_kv * 4);
    float * rc   = (float *) malloc((size_t) T * hd * 4);
    uint16_t * q16 = (uint16_t *) malloc((size_t) hd * 2);

    for (int t = 0; t < T; ++t) {                                        /* llm_build_inp_embd :7802, get_rows ggml.c:13186 */
        bo_dequantize_row(m->t_embd, (const char *) m->tok_embd + (size_t) tokens[t] * bo_row_size(m->t_embd, E), x + (size_t) t * E, E);
        bo_rope_cache(rc + (size_t) t * hd, n_past + t, hd, m->rope_theta, m->rope_freq_scale, m->rope_freqs, 0.0f, 1.0f,
                      m->n_ctx_orig, 32.0f, 1.0f);
    }
    TAP("inp_embd", -1, x, (size_t) T * E);

    int Tl = T;                     /* tokens still alive; the last layer keeps only the last token after attention */
    for (int il = 0; il < m->L; ++il) {
        const bo_layer * ly = &m->layers[il];
        for (int t = 0; t < T; ++t) {                                    /* llm_build_norm :7928 = rms_norm then mul */
            bo_rms_norm(x + (size_t) t * E, cur + (size_t) t * E, E, m->eps);
            for (int i = 0; i < E; ++i) cur[(size_t) t * E + i] = cur[(size_t) t * E + i] * ly->attn_norm[i];
        }
        TAP("attn_norm", il, cur, (size_t) T * E);
        bo_mul_mat_q(ly->tq, ly->wq, E,   E, cur, T, q,  nth);
        bo_mul_mat_q(ly->tk, ly->wk, Ekv, E, cur, T, kk, nth);
        bo_mul_mat_q(ly->tv, ly->wv, Ekv, E, cur, T, vv, nth);
        TAP("Vcur", il, vv, (size_t) T * Ekv);
        for (int t = 0; t < T; ++t) {                                    /* ggml_rope_ext :8837-8849 */
            for (int h = 0; h < H; ++h)   bo_rope_apply(q  + (size_t) t * E   + h * hd, rc + (size_t) t * hd, hd);
            for (int h = 0; h < Hkv; ++h) bo_rope_apply(kk + (size_t) t * Ekv + h * hd, rc + (size_t) t * hd, hd);
        }
        TAP("Qcur", il, q, (size_t) T * E);
        TAP("Kcur", il, kk, (size_t) T * Ekv);
        for (int t = 0; t < T; ++t) {                                    /* llm_build_kv_store :7830-7875 */
            const int cell = n_past + t;
            for (int i = 0; i < Ekv; ++i) {
                c->k[il][(size_t) cell * Ekv + i] = bo_fp32_to_fp16(kk[(size_t) t * Ekv + i]);
                c->v[il][(size_t) i * n_ctx + cell] = bo_fp32_to_fp16(vv[(size_t) t * Ekv + i]);
            }
        }
        /* llm_build_kqv :8188-8316 */
        for (int t = 0; t < T; ++t) {
            const int pos = n_past + t;
            for (int i = 0; i < n_kv; ++i) mask[i] = i <= pos ? 0.0f : -INFINITY;   /* llama_set_inputs :14152-14200 */
            for (int h = 0; h < H; ++h) {
                const int hk = h / gq;
                const float * qh = q + (size_t) t * E + h * hd;
                if (T == 1) {
                    /* llamafile_sgemm F16 x F32 (q stays f32 when T == 1) */
                    for (int i = 0; i < n_kv; ++i)
                        kq[i] = bo_dot_f16_f32_tinyblas(c->k[il] + (size_t) i * Ekv + hk * hd, qh, hd);
                } else {
                    /* q rounded to f16 (ggml.c:12345-12372), ggml_vec_dot_f16 */
                    for (int i = 0; i < hd; ++i) q16[i] = bo_fp32_to_fp16(qh[i]);
                    for (int i = 0; i < n_kv; ++i)
                        kq[i] = bo_vec_dot_f16(hd, c->k[il] + (size_t) i * Ekv + hk * hd, q16);
                }
                if (t == T - 1 && h == 0) TAP("kq_h0_last", il, kq, n_kv);
                bo_soft_max(kq, mask, kq_scale, pr, n_kv);
                if (t == T - 1 && h == 0) TAP("kq_soft_max_h0_last", il, pr, n_kv);
                /* kqv = mul_mat(v^T, p): llamafile_sgemm F16 x F32 for every T (p is contiguous) */
                for (int d = 0; d < hd; ++d)
                    att[(size_t) t * E + h * hd + d] = bo_dot_f16_f32_tinyblas(c->v[il] + (size_t)(hk * hd + d) * n_ctx, pr, n_kv);
            }
        }
        TAP("kqv_merged_cont", il, att, (size_t) T * E);
        bo_mul_mat_q(ly->to, ly->wo, E, E, att, T, cur, nth);
        TAP("kqv_out", il, cur, (size_t) T * E);
        const float * res = x;
        if (il == m->L - 1 && T > 1) {                                   /* inp_out_ids :8856-8862: keep the last token only */
            memmove(cur, cur + (size_t)(T - 1) * E, (size_t) E * 4);
            memmove(x, x + (size_t)(T - 1) * E, (size_t) E * 4);
            Tl = 1;
        }
        for (size_t i = 0; i < (size_t) Tl * E; ++i) ffi[i] = cur[i] + res[i];       /* ffn_inp :8864 */
        TAP("ffn_inp", il, ffi, (size_t) Tl * E);
        for (int t = 0; t < Tl; ++t) {
            bo_rms_norm(ffi + (size_t) t * E, cur + (size_t) t * E, E, m->eps);
            for (int i = 0; i < E; ++i) cur[(size_t) t * E + i] = cur[(size_t) t * E + i] * ly->ffn_norm[i];
        }
        TAP("ffn_norm", il, cur, (size_t) Tl * E);
        /* llm_build_ffn :7960-8085 (LLM_FFN_SILU, LLM_FFN_PAR): up first, then gate */
        bo_mul_mat_q(ly->tu, ly->wu, F, E, cur, Tl, u, nth);
        bo_mul_mat_q(ly->tg, ly->wg, F, E, cur, Tl, g, nth);
        TAP("ffn_gate", il, g, (size_t) Tl * F);
        TAP("ffn_up", il, u, (size_t) Tl * F);
        for (size_t i = 0; i < (size_t) Tl * F; ++i) g[i] = bo_v_silu(g[i]);         /* F % 8 == 0: vector path only */
        for (size_t i = 0; i < (size_t) Tl * F; ++i) g[i] = g[i] * u[i];             /* ggml_mul(silu(gate), up) */
        TAP("ffn_gate_par", il, g, (size_t) Tl * F);
        bo_mul_mat_q(ly->td, ly->wd, E, F, g, Tl, cur, nth);
        TAP("ffn_out", il, cur, (size_t) Tl * E);
        for (size_t i = 0; i < (size_t) Tl * E; ++i) x[i] = cur[i] + ffi[i];         /* :8902 */
        TAP("l_out", il, x, (size_t) Tl * E);
    }
    {   /* :8911-8919 ; only the last token has an output */
        const float * xl = x + (size_t)(Tl - 1) * E;
        bo_rms_norm(xl, cur, E, m->eps);
        for (int i = 0; i < E; ++i) cur[i] = cur[i] * m->out_norm[i];
        TAP("result_norm", -1, cur, E);
        bo_mul_mat_q(m->t_out, m->output, V, E, cur, 1, c->logits, nth);
        TAP("result_output", -1, c->logits, V);
    }
    free(x); free(cur); free(q); free(kk); free(vv); free(att); free(ffi); free(g); free(u); free(kq); free(pr); free(mask); free(rc); free(q16);
    return 0;
}
