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


template <int TYPE, typename REC, int D, int EPI, int TT>
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
        float gate_val[TT];
#pragma unroll
        for (int part = 0; part < NPARTS; ++part) {
            const bool last = !(PAIR && part == 0) && r + 1 >= count;
            const bool after_b = (PAIR && part == 0) || (last && part == 1);
            const int after_off = (PAIR && part == 0) ? rowoff : (last ? rowoff + (nb - 1) * RECB : rowoff + rg_step);
            RowAcc A[TT];
#pragma unroll
            for (int u = 0; u < TT; ++u) { A[u].acc = 0.f; A[u].accm = 0.f; }
            for (int c = 0; c < chunks; ++c) {
                const bool inrow = c + 1 < chunks;
                const bamd_rsrc nrs = (inrow ? part == 1 : after_b) ? rsB : rsA;
                const int nxt = inrow ? rowoff + (c + 1) * (D * RECB) : after_off;
                const int step = (inrow || !last) ? RECB : 0;
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    pin_rec(ring[s]);
#pragma unroll
                    for (int u = 0; u < TT; ++u) {               // tokens beyond nt read stale-but-valid LDS and are never stored
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
            for (int u = 0; u < TT; ++u) {
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

// (Rounds 2-5 kept a first generation of matrix-core mat-mul kernels here — every wave expanding its own 16 rows — and a second one in bamd_prefill2.hip;
// round 6 measured what bounds them (profiles/r06_prefill_ceiling.txt) and kept ONE: bamd_prefill2.hip.  This file keeps the integer-dot kernel — the second
// implementation the matrix-core kernels are tested against, and what a matrix without a side table runs on — and the plumbing.)

// grid (token tiles, row slots): consecutive workgroups share the weights (L2) and differ in the token tile.  TT tokens per tile: BAMD_TT = 8 while their Q8_K
// activations fit the LDS (K <= 17920), 4 beyond (K <= 35840: the 70B ffn_down at K = 28672 — round 6; before, a 70B-width model without side tables had no batched
// kernel at all and evaluated prompts token by token)
template <int EPI, int TT>
__global__ void __launch_bounds__(512) matmul_batch_kernel(bamd_mm_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nb = a.K >> 8;
    const size_t bb = BAMD_BLOB_BYTES(nb);
    const int t0 = blockIdx.x * TT;
    const int nt = a.T - t0 < TT ? a.T - t0 : TT;
    {   // this tile's activations -> LDS (rows past T: repeat the last token; results discarded)
        const int n16 = (int) (bb / 16);
        for (int i = threadIdx.x; i < n16 * TT; i += blockDim.x) {
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
                if (t == BAMD_Q4_K)      batch_segment<BAMD_Q4_K, RecQ4K, 4, EPI, TT>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
                else if (t == BAMD_Q5_K) batch_segment<BAMD_Q5_K, RecQ5K, 4, EPI, TT>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
                else                     batch_segment<BAMD_Q6_K, RecQ6K, 4, EPI, TT>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
            } else {
                if (t == BAMD_Q4_K)      batch_segment<BAMD_Q4_K, RecQ4K, 1, EPI, TT>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
                else if (t == BAMD_Q5_K) batch_segment<BAMD_Q5_K, RecQ5K, 1, EPI, TT>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
                else                     batch_segment<BAMD_Q6_K, RecQ6K, 1, EPI, TT>(wA, wB, nb, g0 - off, count, stride, a.seg[s].out, a.res, a.ldo, t0, nt, smem, bb, nv);
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
    const int tt = (size_t) BAMD_TT * BAMD_BLOB_BYTES(a.K >> 8) <= 160 * 1024 ? BAMD_TT : 4;     // tokens per tile: 8 while their activations fit the LDS, else 4
    const size_t lds = (size_t) tt * BAMD_BLOB_BYTES(a.K >> 8);
    if (lds > 160 * 1024) return 1;                           // K > 35840: would need a K-split of the activation tile
    const int tiles = (a.T + tt - 1) / tt;
    // row slots: enough workgroups to fill the chip a few times over, at least one row-group per wave
    int gy = (4 * (n_cu > 0 ? n_cu : 256) + tiles - 1) / tiles;
    if (gy * 8 > nrg) gy = (nrg + 7) / 8;
    if (gy < 1) gy = 1;
    dim3 grid(tiles, gy);
#define BAMD_MB(EPI_) do { if (tt == BAMD_TT) hipLaunchKernelGGL((matmul_batch_kernel<EPI_, BAMD_TT>), grid, dim3(512), lds, s, a); else hipLaunchKernelGGL((matmul_batch_kernel<EPI_, 4>), grid, dim3(512), lds, s, a); } while (0)
    switch (epi) {
        case BAMD_EPI_STORE:    BAMD_MB(BAMD_EPI_STORE); break;
        case BAMD_EPI_ADD:      BAMD_MB(BAMD_EPI_ADD); break;
        case BAMD_EPI_SILU_MUL: BAMD_MB(BAMD_EPI_SILU_MUL); break;
        default: return 1;
    }
#undef BAMD_MB
    return 0;
}
void bamd_launch_embed_batch(const int32_t * tokens, int T, const void * embd, int embd_type, int E, int V, float * x, hipStream_t s) {
    hipLaunchKernelGGL(embed_batch_kernel, dim3(T), dim3(256), 0, s, tokens, (const uint8_t *) embd, embd_type, E, V, x);
}
