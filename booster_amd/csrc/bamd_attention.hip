// bamd_attention.hip — attention kernels: fused single-launch (decode of short sequences), three-launch long-sequence path, batched
// prefill attention with all query heads of a KV head per workgroup, batched KV store.  Helpers and layout: bamd_device.h.
#include "bamd_device.h"

// ---- long contexts: three launches (scores | softmax | P.V), positions / rows spread over many workgroups ------------------
// grid (Hkv, tiles of 64 positions), block 512 = 8 waves x (8 positions x 8 lanes); the GQ query heads of a KV head share K
#define BAMD_QK_NT 4                     /* tiles of 64 positions per workgroup whose K rows are in flight together */
// SH: cells no longer follow positions (a context shift happened: a.cellpos).  The token's K / V go to cell st->cell, a cell is attended when it
// holds a position <= pos (llama_set_inputs' mask, llama.cpp:14152-14200) — in CELL order, as the reference's soft_max and P.V run over the cache
template <int GQ, int LG, bool SH, int NT = BAMD_QK_NT>       // LG = head_dim / 64 = 16-byte groups of a K row per lane; NT = tiles of 64 positions per workgroup in flight together
// (the launcher picks 2 when the context has no more than two tiles per workgroup: no request for a tile that cannot exist)
__global__ void __launch_bounds__(512) attn_qk_kernel(bamd_attn_args a) {
    // Round 6: the launch is a chain of latencies, not bytes — 16 MB of K at 8000 positions are 2.7 us at the HBM rate and the kernel took 7.9.
    // A wave's loads return in order, so the q / k / RoPE requests that rounds 2-5 issued BEHIND the K prefetch (the prefetch clamp needed the position first)
    // came back behind 64 KB of K rows per CU, and the prologue's barrier stood at 4.7 us.  Now: (1) the small requests go out first — this token's q / k / v pairs
    // and the cos / sin row of the current position, which step_begin_kernel leaves at a FIXED address (a.rope_cur: no dependent load of the position in front of
    // them); (2) the K rows of the first NT tiles right behind, unclamped (rows past the sequence are masked by a select; the cache is finite everywhere);
    // (3) the scores leave through LDS as 16-byte write-through stores instead of scattered 4-byte ones.  All cross-launch data by the rules of bamd_device.h
    // ("Inter-kernel data"): the step can be replayed from the own AQL queue.
    constexpr int hd = 64 * LG, L = 8 * LG, hp = hd / 2;
    constexpr int NPAIR = (GQ + 1) * hp, IT = (NPAIR + 511) / 512;      // adjacent pairs of the GQ query heads, then of the k head; pairs per thread
    __shared__ __attribute__((aligned(16))) float qt[GQ * 256];
    __shared__ __attribute__((aligned(16))) unsigned short q16t[GQ * 256];
    __shared__ __attribute__((aligned(16))) unsigned short k16t[256];
    __shared__ __attribute__((aligned(16))) float scs[NT][GQ][64];   // scores of this workgroup's tiles in the V^T position order, on their way out
    const bamd_step_state * st = a.st;
    const int Hkv = a.Hkv, Ekv = Hkv * hd, n_ctx = a.n_ctx;
    const int hk = blockIdx.x, by = (int) blockIdx.y, gy = (int) gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), e = lane & 7, r = lane >> 3;
    // ---- 1. the small requests ----
    float2 xin[IT], cs[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * 512, idc = idx < NPAIR ? idx : 0, hh = idc / hp, p = idc - hh * hp;
        const float * src = hh < GQ ? a.q + (size_t) (hk * GQ + hh) * hd : a.k + (size_t) hk * hd;
        xin[it] = ik_ld2f(src + 2 * p);
        cs[it] = ik_ld2f(a.rope_cur + 2 * p);                    // (the launcher refuses a null rope_cur: no conditional request, no join in front of the prefetch)
    }
    const float vst = ik_ld(a.v + hk * hd + (tid < hd ? tid : 0));
    const int pos = ik_ld(&st->pos), n_kv = ik_ld(&st->n_kv), cell = SH ? ik_ld(&st->cell) : pos;
    // ---- 2. the K rows of this workgroup's first NT tiles: one memory latency for all of them, overlapped with the prologue ----
    const bamd_ik_rsrc rk = ik_rsrc(a.kc);
    uint4 kpre[NT][LG];
#define BAMD_QK_PREFETCH(t0_) do { \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) { \
            int i_ = ((t0_) + j * gy) * 64 + wave * 8 + r; i_ = i_ < n_ctx ? i_ : 0; \
            _Pragma("unroll") for (int g = 0; g < LG; ++g) kpre[j][g] = ik_ld128(rk, (uint32_t) (i_ * Ekv + hk * hd + g * BAMD_KGRP + e * 8) * 2u); \
        } } while (0)
#ifndef BAMD_QK_KNOCK
#define BAMD_QK_KNOCK 0              /* timing-only experiment builds (results wrong): 1 = no K requests, 2 = no chains, 4 = no score stores, 8 = no RoPE prologue loads */
#endif
    if (!(BAMD_QK_KNOCK & 1)) BAMD_QK_PREFETCH(by);
    else { _Pragma("unroll") for (int j = 0; j < NT; ++j) _Pragma("unroll") for (int g = 0; g < LG; ++g) kpre[j][g] = make_uint4(tid, j, g, 1); }
    // ---- 3. RoPE (NORM mode, adjacent pairs; ggml.c:14130-14143 — rope_heads' arithmetic) into the chain-major LDS copies ----
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * 512;
        if (idx < NPAIR) {
            const int hh = idx / hp, p = idx - hh * hp;
            const float t0 = xin[it].x * cs[it].x, t1 = xin[it].y * cs[it].y, t2 = xin[it].x * cs[it].y, t3 = xin[it].y * cs[it].x;
            const float r0 = t0 - t1, r1 = t2 + t3;
            const int i0 = kperm(2 * p, L), i1 = kperm(2 * p + 1, L);
            if (hh < GQ) { qt[hh * hd + i0] = r0; qt[hh * hd + i1] = r1; q16t[hh * hd + i0] = f2h(r0); q16t[hh * hd + i1] = f2h(r1); }
            else { k16t[i0] = f2h(r0); k16t[i1] = f2h(r1); }
        }
    }
    __syncthreads();
    // KV store by the block that owns the tile of `pos` — llm_build_kv_store, llama.cpp:7830-7875
    if (by == ((cell >> 6) % gy) && tid < hd) {
        ik_st(a.kc + (size_t) cell * Ekv + hk * hd + tid, k16t[tid]);
        ik_st(a.vc + (size_t) (hk * hd + tid) * n_ctx + vperm(cell), f2h(vst));
    }
    const bamd_ik_rsrc rs = ik_rsrc(a.scores);
    const int tiles = (n_kv + 63) >> 6;
    for (int tile0 = by; tile0 < tiles; tile0 += NT * gy) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int i = (tile0 + j * gy) * 64 + wave * 8 + r;         // position
            float sc[GQ];
#pragma unroll
            for (int g = 0; g < GQ; ++g) sc[g] = -INFINITY;     // masked (KQ_mask, llama.cpp:14152-14200)
            bool live = i <= pos;                                // (i <= pos < n_kv)
            if (SH) live = i < n_kv && (i == cell || (uint32_t) ik_ld(a.cellpos + (i < n_kv ? i : 0)) <= (uint32_t) pos);
            if (live) {
                uint4 kreg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    kreg[g] = make_uint4(0, 0, 0, 0);
                    if (g < LG) { const uint4 own = *(const uint4 *) (k16t + g * BAMD_KGRP + e * 8); kreg[g] = i == cell ? own : kpre[j][g < LG ? g : 0]; }
                }
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    if (BAMD_QK_KNOCK & 2) { sc[g] = __uint_as_float(kreg[0].x ^ kreg[1 < LG ? 1 : 0].w) + qt[g * hd + e * 8]; continue; }
                    const float v = a.prefill_mode ? kq_chain<true>(kreg, L, nullptr, q16t + g * hd + e * 8) : kq_chain<false>(kreg, L, qt + g * hd + e * 8, nullptr);
                    sc[g] = a.prefill_mode ? hsum8_vecdot(v) : hsum8_tinyblas(v);
                }
            }
            if (e == 0) {
#pragma unroll
                for (int g = 0; g < GQ; ++g) scs[j][g][r * 8 + wave] = sc[g];      // position 8 w + r of a tile sits at 8 r + w (vperm)
            }
        }
        __syncthreads();
        // the tiles' scores in the V^T position order (the softmax passes read 16 bytes at a time), 16 bytes per thread, write-through
        for (int q4 = tid; q4 < NT * GQ * 16; q4 += 512) {
            const int j = q4 / (GQ * 16), g = (q4 >> 4) % GQ, c4 = q4 & 15, tile = tile0 + j * gy;
            if (tile < tiles && (!(BAMD_QK_KNOCK & 4) || scs[j][g][c4 * 4] == 12345.f)) ik_st128f(rs, (uint32_t) ((hk * GQ + g) * n_ctx + tile * 64 + c4 * 4) * 4u, *(const float4 *) &scs[j][g][c4 * 4]);
        }
        if (tile0 + NT * gy < tiles) { __syncthreads(); BAMD_QK_PREFETCH(tile0 + NT * gy); }   // (n_ctx > 64 x NT x gridDim.y only)
    }
#undef BAMD_QK_PREFETCH
}

// seq_expsum8 over values stored in the V^T position order (position i + j, i % 8 == 0, sits at (i & ~63) + 8 j + ((i & 63) >> 3))
__device__ __forceinline__ double seq_expsum8_vt(const float * v, int n) {
    double sq = 0.0;
    for (int i = 0; i < n; i += 8) {
        const float * q = v + (i & ~63) + ((i & 63) >> 3);
        const float a0 = q[0] + q[32], a1 = q[8] + q[40], a2 = q[16] + q[48], a3 = q[24] + q[56];
        const float b0 = a0 + a2, b1 = a1 + a3;
        sq += (double) (b0 + b1);
    }
    return sq;
}
// softmax over n_kv scores of one head (stored in the V^T position order by attn_qk_kernel): grid (H), block 1024.  ggml.c:13682-13778 + :2619-2671 (AVX2 branch).
// Probabilities are written back in the V^T position order (vperm) so the P.V lanes read them contiguously.
#define BAMD_SM_R 8                      /* values per thread kept in registers: one pass over global memory up to n_kv = 8192 */
__global__ void __launch_bounds__(1024) attn_softmax_kernel(bamd_attn_args a) {
    __shared__ float redf[16];
    __shared__ double redd[16];
    const bamd_step_state * st = a.st;
    const int n_kv = st->n_kv, n_ctx = a.n_ctx;
    const int h = blockIdx.x;
    float * s = a.scores + (size_t) h * n_ctx;
    float * pr = a.probs + (size_t) h * n_ctx;
    const float scale = a.kq_scale;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), nw = blockDim.x >> 6;
    const bool cached = n_kv <= BAMD_SM_R * (int) blockDim.x;
    float v[BAMD_SM_R];
    float mx = -INFINITY;
    if (cached) {
#pragma unroll
        for (int k = 0; k < BAMD_SM_R; ++k) { const int i = tid + k * blockDim.x; v[k] = i < n_kv ? s[vperm(i)] * scale : -INFINITY; mx = v[k] > mx ? v[k] : mx; }
    } else for (int i = tid; i < n_kv; i += blockDim.x) { const float w = s[vperm(i)] * scale; mx = w > mx ? w : mx; }
    for (int o = 32; o; o >>= 1) { const float om = __shfl_xor(mx, o); mx = om > mx ? om : mx; }
    if (lane == 0) redf[wave] = mx;
    __syncthreads();
    mx = redf[0]; for (int w = 1; w < nw; ++w) mx = redf[w] > mx ? redf[w] : mx;
    double sum = 0.0;
    if (cached) {
#pragma unroll
        for (int k = 0; k < BAMD_SM_R; ++k) {
            const int i = tid + k * blockDim.x;
            if (i < n_kv) {                                      // n_kv % 32 == 0: 8-lane groups are all-active or all-idle
                const float val = v_expf(v[k] - mx);
                v[k] = val;
                const float c = hsum8_tinyblas(val);             // the reference's 8-wide partial sum (same tree shape)
                if ((lane & 7) == 0) sum += (double) c;
            }
        }
    } else for (int i = tid; i < n_kv; i += blockDim.x) {
        const float w = s[vperm(i)] * scale;
        const float val = v_expf(w - mx);
        s[vperm(i)] = val;                                              // same index this thread just read: no hazard
        const float c = hsum8_tinyblas(val);
        if ((lane & 7) == 0) sum += (double) c;
    }
    sum = wave_sum_f64(sum);
    if (lane == 0) redd[wave] = sum;
    __syncthreads();
    double tot = 0.0; for (int w = 0; w < nw; ++w) tot += redd[w];
    double rs = 1.0 / tot;
    float fs = (float) rs;
    if (!f32_rounding_safe(rs, BAMD_F64_GUARD_ULPS(n_kv / 8))) {          // workgroup-uniform, rare: the reference's sequential order (bamd_device.h)
        if (cached) {
#pragma unroll
            for (int k = 0; k < BAMD_SM_R; ++k) { const int i = tid + k * blockDim.x; if (i < n_kv) s[vperm(i)] = v[k]; }
        }
        __syncthreads();
        if (tid == 0) redd[0] = seq_expsum8_vt(s, n_kv);
        __syncthreads();
        rs = 1.0 / redd[0]; fs = (float) rs;
    }
    if (cached) {
#pragma unroll
        for (int k = 0; k < BAMD_SM_R; ++k) { const int i = tid + k * blockDim.x; if (i < n_kv) pr[vperm(i)] = v[k] * fs; }
    } else for (int i = tid; i < n_kv; i += blockDim.x) pr[vperm(i)] = s[vperm(i)] * fs;
}
// P.V: grid (H, hd/8), block 64: lane = d_local*8 + e carries the tinyBLAS chain Cv[e] of output (h, d).  sgemm.cpp:405-431 with
// A = V^T rows (f16), B = p (f32).  The chain over positions is sequential per lane, but the loads are not: BAMD_PV_U blocks of 64
// positions are requested together (16 KiB of V^T per wave in flight).  Workgroup = half of the gq query heads of one KV head (grid z = 2:
// Hkv x hd/8 alone would occupy only 128 of the 256 CUs), one wave each: they stream the same V^T rows in step and share the CU's vector L1.
#define BAMD_PV_U 8
__global__ void __launch_bounds__(512) attn_pv_kernel(bamd_attn_args a, int gq) {
    const bamd_step_state * st = a.st;
    const int n_kv = st->n_kv, n_ctx = a.n_ctx, hd = a.hd;
    const int hk = blockIdx.x, h = hk * gq + (int) blockIdx.z * (int) (blockDim.x >> 6) + wave_id();   // blockIdx.z: which part of the KV head's query heads
    const int lane = threadIdx.x & 63, e = lane & 7;
    const int d = blockIdx.y * 8 + (lane >> 3);
    const unsigned short * vrow = a.vc + (size_t) (hk * hd + d) * n_ctx + e * 8;
    const float * p = a.probs + (size_t) h * n_ctx + e * 8;
    float acc = 0.f;
    // n_kv is a multiple of 32 (llama.cpp:14693-14701): nfull blocks of 64 positions (8 chain steps per lane) and possibly half a block
    // (4 steps).  Sets of BAMD_PV_U full blocks run without a single condition — one v_fma_mix_f32 per chain step, three 16-byte
    // requests per block — with the next set's requests in flight (two register sets); what does not fill a set is chained from
    // the last prefetched set under wave-uniform branches.  (The first form of this kernel tested the block and the step count at
    // every chain step: seven instructions per step, 15.7 us per layer at 8000 positions for 3.3 us of chain latency.)
    const int nfull = n_kv >> 6, half = (n_kv >> 5) & 1, last = nfull - 1 + half;
    const int nsets = nfull / BAMD_PV_U, rem = nfull - nsets * BAMD_PV_U;
    uint4 vA[BAMD_PV_U], vB[BAMD_PV_U]; float4 paA[BAMD_PV_U], pbA[BAMD_PV_U], paB[BAMD_PV_U], pbB[BAMD_PV_U];
#define BAMD_PV_LOAD(V_, PA_, PB_, set_) do { \
        _Pragma("unroll") for (int u = 0; u < BAMD_PV_U; ++u) { \
            int blk = (set_) * BAMD_PV_U + u; blk = blk < last ? blk : last;       /* unconditional requests (the last block again past the end) */ \
            const int b = blk * 64; \
            V_[u] = *(const uint4 *) (vrow + b); PA_[u] = *(const float4 *) (p + b); PB_[u] = *(const float4 *) (p + b + 4); \
        } } while (0)
#define BAMD_PV_STEPS(V_, PA_, PB_, u_, n_) do { \
        const uint32_t w[4] = { V_[u_].x, V_[u_].y, V_[u_].z, V_[u_].w }; \
        const float pv[8] = { PA_[u_].x, PA_[u_].y, PA_[u_].z, PA_[u_].w, PB_[u_].x, PB_[u_].y, PB_[u_].z, PB_[u_].w }; \
        acc = fma_mix_chain<n_>(acc, w, pv); } while (0)
#define BAMD_PV_CHAIN(V_, PA_, PB_) do { _Pragma("unroll") for (int u = 0; u < BAMD_PV_U; ++u) BAMD_PV_STEPS(V_, PA_, PB_, u, 8); } while (0)
#define BAMD_PV_TAIL(V_, PA_, PB_) do { \
        _Pragma("unroll") for (int u = 0; u < BAMD_PV_U; ++u) { \
            if (u < rem) BAMD_PV_STEPS(V_, PA_, PB_, u, 8); \
            else if (u == rem && half) BAMD_PV_STEPS(V_, PA_, PB_, u, 4); \
        } } while (0)
    BAMD_PV_LOAD(vA, paA, pbA, 0);
    int s = 0;
    for (; s + 2 <= nsets; s += 2) {
        BAMD_PV_LOAD(vB, paB, pbB, s + 1);
        BAMD_PV_CHAIN(vA, paA, pbA);
        BAMD_PV_LOAD(vA, paA, pbA, s + 2);
        BAMD_PV_CHAIN(vB, paB, pbB);
    }
    if (s < nsets) {                                             // an odd number of full sets: one more, then the tail sits in set B
        BAMD_PV_LOAD(vB, paB, pbB, s + 1);
        BAMD_PV_CHAIN(vA, paA, pbA);
        BAMD_PV_TAIL(vB, paB, pbB);
    } else BAMD_PV_TAIL(vA, paA, pbA);
#undef BAMD_PV_LOAD
#undef BAMD_PV_STEPS
#undef BAMD_PV_CHAIN
#undef BAMD_PV_TAIL
    const float v = hsum8_tinyblas(acc);
    if (e == 0) a.out[(size_t) h * hd + d] = v;
}

// softmax + P.V in ONE launch (long contexts, after attn_qk_kernel): grid (Hkv, hd/8, Z), block 1024; a workgroup serves NH query heads of one
// KV head (NH * Z = gq) and 8 rows of V^T.  All sixteen waves turn the NH score rows into probabilities IN LDS (scaled score -> exponential ->
// probability, in the V^T position order the scores arrive in), every workgroup for itself: hd/8 workgroups repeat the same rows, which is cheaper
// than a launch of its own (attn_softmax_kernel: one workgroup per head, 9.2 us per layer, 224 CUs idle).  A wave of this chip issues one vector
// instruction per ~4.2 ns (tools/chain_probe.hip: 10 clocks, dependent or not; a SIMD reaches its rate only with four or more waves), so the softmax
// passes are written for few instructions per element — 16-byte LDS / global accesses in LDS order, thread = (block of 64 positions, SIMD lane j,
// half) so that the reference's 8-wide partial sums are DPP sums over lane bits 0..2 — and run on 16 waves.
// Then wave g < NH carries the tinyBLAS chains of head g exactly as attn_pv_kernel does — lane = (d, e), sequential over positions: 5.2 ns per
// dependent step, 1000 steps at 8000 positions, the floor of this path — with nothing but the chain on its vector pipe: probabilities from LDS
// (two ds_read_b128 per block of 64 positions, three blocks ahead), V^T from a register ring of BAMD_SPV_R blocks (requested R blocks = ~1 us
// ahead of their turn: attn_pv_kernel's two sets of 8 were 64 steps ahead and every set waited out most of a memory latency) through buffer
// loads whose block offset is an immediate (no address arithmetic; the descriptor's bound makes requests past the cache return 0).
// The ring is first filled before the softmax passes.  Same values, same order as the two kernels it replaces (tests).
#define BAMD_SPV_R 20
#define BAMD_SPV_SLACK 1024                 /* bytes of LDS past the rows: the chain wave's look-ahead reads up to three blocks past the last one */
#define BAMD_SPV_LDS_MAX (152 * 1024)
template <int NH>
__global__ void __launch_bounds__(1024) attn_spv_kernel(bamd_attn_args a, int gq, uint32_t vc_bytes) {
    extern __shared__ __attribute__((aligned(16))) float spv_p[];          // [NH][n_ctx] (+ slack)
    __shared__ float redf[NH][16];
    __shared__ double redd[NH][16];
    const bamd_step_state * st = a.st;
    const int n_ctx = a.n_ctx, hd = a.hd;
    const int hk = blockIdx.x, h0 = hk * gq + (int) blockIdx.z * NH;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), e = lane & 7;
    const int d = blockIdx.y * 8 + (lane >> 3);
    const bamd_rsrc vr = __builtin_amdgcn_make_buffer_rsrc((void *) uniform_ptr((const uint8_t *) a.vc), 0, (int) vc_bytes, 0x00020000);
    const uint32_t voff = (uint32_t) (((size_t) (hk * hd + d) * n_ctx + e * 8) * 2);
    uint4 ring[BAMD_SPV_R];
#define BAMD_SPV_VLOAD(blk_) ({ const u32x4_t t_ = __builtin_amdgcn_raw_buffer_load_b128(vr, (int) voff, (blk_) * 128, BAMD_IK_AUX); make_uint4(t_.x, t_.y, t_.z, t_.w); })   /* V^T rows: inter-kernel data (bamd_device.h) */
    if (wave < NH) {
#pragma unroll
        for (int u = 0; u < BAMD_SPV_R; ++u) ring[u] = BAMD_SPV_VLOAD(u);
    }
    // ---- softmax of the NH rows: ggml.c:13682-13778 + :2619-2671 (AVX2 branch), as attn_softmax_kernel.  Thread = (block B0 + 64 k, lane j of the
    //      reference's 8-wide vector, half lh): the four floats at [64 B + 8 j + 4 lh ..] of a row = positions 64 B + 8 (4 lh + c) + j, c = 0..3 ----
    const int B0 = tid >> 4, foff = (tid & 7) * 8 + ((tid >> 3) & 1) * 4, lh = (tid >> 3) & 1;
#define BAMD_SPV_VALID(B_) ((B_) < nfull || ((B_) == nfull && half && lh == 0))
    const float scale = a.kq_scale;
    float mx[NH];
#pragma unroll
    for (int g = 0; g < NH; ++g) mx[g] = -INFINITY;
    const int n_kv = ik_ld(&st->n_kv);
    const bamd_ik_rsrc rsc = ik_rsrc(a.scores);                  // the score rows: the previous launch's output
    // n_kv is a multiple of 32 (llama.cpp:14693-14701): nfull blocks of 64 positions and possibly half a block (attn_pv_kernel)
    const int nfull = n_kv >> 6, half = (n_kv >> 5) & 1, last = nfull - 1 + half;
    for (int B = B0; B <= last; B += 128) {                      // two blocks per thread and head requested together (one latency, not four)
        float4 w[NH][2];
#pragma unroll
        for (int g = 0; g < NH; ++g)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int Bk = B + 64 * k;
                w[g][k] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                if (BAMD_SPV_VALID(Bk)) w[g][k] = ik_ld128f(rsc, (uint32_t) ((h0 + g) * n_ctx + Bk * 64 + foff) * 4u);
            }
#pragma unroll
        for (int g = 0; g < NH; ++g)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int Bk = B + 64 * k;
                if (BAMD_SPV_VALID(Bk)) {
                    float4 v = w[g][k];
                    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
                    float m = mx[g];
                    m = v.x > m ? v.x : m; m = v.y > m ? v.y : m; m = v.z > m ? v.z : m; m = v.w > m ? v.w : m;
                    mx[g] = m;
                    *(float4 *) (spv_p + g * n_ctx + Bk * 64 + foff) = v;
                }
            }
    }
#pragma unroll
    for (int g = 0; g < NH; ++g) {
        float m = mx[g];
        for (int o = 32; o; o >>= 1) { const float om = __shfl_xor(m, o); m = om > m ? om : m; }
        if (lane == 0) redf[g][wave] = m;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NH; ++g) {
        float m = redf[g][0];
#pragma unroll
        for (int w = 1; w < 16; ++w) m = redf[g][w] > m ? redf[g][w] : m;
        double sm = 0.0;
        for (int B = B0; B <= last; B += 64) {                   // (8-lane groups j = 0..7 are all-valid or all-idle)
            if (BAMD_SPV_VALID(B)) {
                float4 * pp = (float4 *) (spv_p + g * n_ctx + B * 64 + foff);
                float4 v = *pp;
                v.x = v_expf(v.x - m); v.y = v_expf(v.y - m); v.z = v_expf(v.z - m); v.w = v_expf(v.w - m);
                *pp = v;
                // the reference's 8-wide partial sums (same tree shape), one per c: positions 64 B + 8 (4 lh + c) + 0..7
                const float c0 = hsum8_tinyblas(v.x), c1 = hsum8_tinyblas(v.y), c2 = hsum8_tinyblas(v.z), c3 = hsum8_tinyblas(v.w);
                if ((lane & 7) == 0) { sm += (double) c0; sm += (double) c1; sm += (double) c2; sm += (double) c3; }
            }
        }
        sm = wave_sum_f64(sm);
        if (lane == 0) redd[g][wave] = sm;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NH; ++g) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += redd[g][w];
        double rs = 1.0 / tot;
        if (!f32_rounding_safe(rs, BAMD_F64_GUARD_ULPS(n_kv / 8))) {          // workgroup-uniform, rare: the reference's sequential order (bamd_device.h)
            __syncthreads();                                     // (the exponentials of the other threads; redd read by all)
            if (tid == 0) redd[g][0] = seq_expsum8_vt(spv_p + g * n_ctx, n_kv);
            __syncthreads();
            rs = 1.0 / redd[g][0];
            __syncthreads();
        }
        const float fs = (float) rs;
        float * pr = a.probs + (size_t) (h0 + g) * n_ctx;
        for (int B = B0; B <= last; B += 64) {
            if (BAMD_SPV_VALID(B)) {
                float4 * pp = (float4 *) (spv_p + g * n_ctx + B * 64 + foff);
                float4 v = *pp;
                v.x *= fs; v.y *= fs; v.z *= fs; v.w *= fs;
                *pp = v;
                if (blockIdx.y == 0 && a.probs) *(float4 *) (pr + B * 64 + foff) = v;      // the probability rows, once per head, when asked for (bamd_op_attention hands them to the tests; a decode step passes null)
            }
        }
    }
    __syncthreads();
#undef BAMD_SPV_VALID
    if (wave >= NH) return;
    // ---- P.V: lane = d_local*8 + e carries the chain Cv[e] of output (h0 + wave, d) — sgemm.cpp:405-431, attn_pv_kernel ----
    const float * pw = spv_p + wave * n_ctx + e * 8;
    float acc = 0.f;
    // probabilities of a block: two ds_read_b128, issued three blocks ahead of their chain steps (four register slots; BAMD_SPV_R % 4 == 0 keeps
    // slot = block & 3 across turns).  A scheduling barrier after every block: left alone, the scheduler hoists LDS reads and V^T requests across the
    // whole unrolled turn until the register file overflows.
    float4 pa[4], pb[4];
#define BAMD_SPV_PREAD(slot_, blk_) do { pa[slot_] = *(const float4 *) (pw + (blk_) * 64); pb[slot_] = *(const float4 *) (pw + (blk_) * 64 + 4); } while (0)
#define BAMD_SPV_STEPS(V_, u_, n_) do { \
        const uint32_t w[4] = { V_[u_].x, V_[u_].y, V_[u_].z, V_[u_].w }; \
        const float pv[8] = { pa[(u_) & 3].x, pa[(u_) & 3].y, pa[(u_) & 3].z, pa[(u_) & 3].w, pb[(u_) & 3].x, pb[(u_) & 3].y, pb[(u_) & 3].z, pb[(u_) & 3].w }; \
        acc = fma_mix_chain<n_>(acc, w, pv); } while (0)
    BAMD_SPV_PREAD(0, 0); BAMD_SPV_PREAD(1, 1); BAMD_SPV_PREAD(2, 2);
    int base = 0;
    for (; base + BAMD_SPV_R <= nfull; base += BAMD_SPV_R) {     // whole turns of the ring: full blocks only
#pragma unroll
        for (int u = 0; u < BAMD_SPV_R; ++u) {
            BAMD_SPV_PREAD((u + 3) & 3, base + u + 3);
            BAMD_SPV_STEPS(ring, u, 8);
            ring[u] = BAMD_SPV_VLOAD(base + BAMD_SPV_R + u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int u = 0; u < BAMD_SPV_R; ++u) {                       // what is left of the last turn (wave-uniform branches)
        const int blk = base + u;
        if (blk + 3 <= last) BAMD_SPV_PREAD((u + 3) & 3, blk + 3);
        if (blk < nfull) BAMD_SPV_STEPS(ring, u, 8);
        else if (blk == nfull && half) BAMD_SPV_STEPS(ring, u, 4);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef BAMD_SPV_PREAD
#undef BAMD_SPV_VLOAD
#undef BAMD_SPV_STEPS
    const float v = hsum8_tinyblas(acc);
    if (e == 0) ik_st(a.out + (size_t) (h0 + wave) * hd + d, v);
}

#include "bamd_attn_fused.h"
template <int LG>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) attn_fused_kernel(bamd_attn_args a, int gq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_dyn[];
    attn_fused_body<LG, false>(a, gq, (int) blockIdx.x, (int) blockIdx.y, attn_dyn, nullptr, 0u);
}

// ---- batched prefill attention: one workgroup per (KV head, token) computes ALL GQH query heads that share the KV head -------
// Same arithmetic per (token, head) as attn_fused_kernel in its T > 1 mode (q rounded to f16, ggml_vec_dot_f16 order for the scores,
// softmax with the reference's 8-wide partial sums, tinyBLAS chains for P.V), but every K row and V^T chunk is loaded once for the
// GQH heads, and the token's own K/V are already in the cache (kv_store_batch_kernel).  Dynamic LDS: GQH x ld floats.
template <int GQH>
__global__ void __launch_bounds__(512) attn_batch_kernel(bamd_attn_args a, int gq) {
    __shared__ __attribute__((aligned(16))) unsigned short q16t[GQH][256];
    __shared__ float redf[GQH][8];
    __shared__ double redd[GQH][8];
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_dyn[];
    const int ld = a.lds_ld ? a.lds_ld : a.n_ctx;                            // LDS row length: bounds the padded sequence length of the micro-batch
    float * sc = (float *) attn_dyn;                                         // [GQH][ld] scores, then exp values, then — IN PLACE — the
    float * pt = sc;                                                         // probabilities in V^T position order (vperm stays inside a 64-block)
    const bamd_step_state * st = a.st;
    const int tokb = blockIdx.y;
    const int pos = st->pos + tokb;
    int n_kv = (pos + 1 + 31) / 32 * 32; n_kv = n_kv < st->n_ctx ? n_kv : st->n_ctx;
    const int hd = a.hd, Hkv = a.Hkv, Ekv = Hkv * hd, n_ctx = a.n_ctx, L = hd >> 3;
    const int h0 = blockIdx.x * GQH, hk = h0 / gq;           // GQH divides gq: the heads of a workgroup share one KV head
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
        // unconditional requests (a masked position reads row `pos`, which this micro-batch stored), every chain runs, the mask is a select:
        // a conditional load is a branch with a full wait at its join
        const unsigned short * krow = a.kc + (size_t) (valid ? i : pos) * Ekv + hk * hd + e * 8;
        uint4 kl[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) kl[g] = g * 8 < L ? *(const uint4 *) (krow + g * BAMD_KGRP) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int hh = 0; hh < GQH; ++hh) {
            float v = hsum8_vecdot(kq_chain<true>(kl, L, nullptr, &q16t[hh][0] + e * 8));
            v = valid ? v : -INFINITY;                             // masked (KQ_mask, llama.cpp:14152-14200)
            if (e == 0 && i < n_kv) sc[(size_t) hh * ld + i] = v;
        }
    }
    __syncthreads();
    // ---- softmax per head (ggml.c:13682-13778 + :2619-2671) ----
    const float scale = a.kq_scale;
#pragma unroll
    for (int hh = 0; hh < GQH; ++hh) {
        const float * s_ = sc + (size_t) hh * ld;
        float mx = -INFINITY;
        for (int i = tid; i < n_kv; i += blockDim.x) { const float w = s_[i] * scale; mx = w > mx ? w : mx; }
        uint32_t u = __float_as_uint(mx); u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        u = wave_max_u32(u);
        if (lane == 0) redf[hh][wave] = __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
    }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < GQH; ++hh) {
        float * s_ = sc + (size_t) hh * ld;
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
        const float * s_ = sc + (size_t) hh * ld; float * p_ = pt + (size_t) hh * ld;
        double tot = 0.0;
        for (int w = 0; w < 8; ++w) tot += redd[hh][w];
        double rs = 1.0 / tot;
        float fs = (float) rs;
        if (!f32_rounding_safe(rs, BAMD_F64_GUARD_ULPS(n_kv / 8))) {      // workgroup-uniform, rare: the reference's sequential order (bamd_device.h)
            __syncthreads();
            if (tid == 0) redd[hh][0] = seq_expsum8(s_, n_kv);
            __syncthreads();
            rs = 1.0 / redd[hh][0]; fs = (float) rs;
        }
        // one wave per 64-block: all 64 values are read before the permuted ones are written, so the block is permuted in place;
        // the idle half of a half-filled last block becomes zeros (exact no-ops in the chains)
        for (int i = tid; i < ((n_kv + 63) & ~63); i += blockDim.x) { const float val = i < n_kv ? s_[i] * fs : 0.f; p_[vperm(i)] = val; }
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
                    const float * p_ = pt + (size_t) hh * ld + b0 + e * 8;
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
// ---- K-shift: after llama_kv_cache_seq_add the cached K rows of the moved cells are re-rotated by their position delta ------------
// (llama_kv_cache_update_internal -> build_k_shift, llama.cpp:15245-15277, :8482-8512: ggml_rope_ext_inplace over the whole f16 K cache,
// i.e. ggml_compute_forward_rope_f16, ggml.c:14169-14290 — x0, x1 from f16, x0*cos - x1*sin and x0*sin + x1*cos as separate f32
// multiplies and one add each, back to f16; cells with delta 0 go through the same arithmetic with cos 1 / sin 0).
// grid (cells), block Hkv*hd/2: one adjacent pair per thread, at its chain-major place (kperm).
__global__ void __launch_bounds__(1024) k_shift_kernel(unsigned short * kc, int Hkv, int hd, const int32_t * tab_of_cell, const float * tab) {
    const int cell = blockIdx.x, hp = hd >> 1, L = hd >> 3;
    const float * row = tab + (size_t) tab_of_cell[cell] * hd;
    for (int t = threadIdx.x; t < Hkv * hp; t += blockDim.x) {
        const int hk = t / hp, p = t - hk * hp;
        unsigned short * kr = kc + (size_t) cell * Hkv * hd + hk * hd;
        const int i0 = kperm(2 * p, L), i1 = kperm(2 * p + 1, L);
        const float x0 = h2f(kr[i0]), x1 = h2f(kr[i1]);
        const float c = row[2 * p], sn = row[2 * p + 1];
        const float t0 = x0 * c, t1 = x1 * sn, t2 = x0 * sn, t3 = x1 * c;
        kr[i0] = f2h(t0 - t1); kr[i1] = f2h(t2 + t3);
    }
}
void bamd_launch_k_shift(unsigned short * kc, int n_cells, int Hkv, int hd, const int32_t * tab_of_cell, const float * tab, hipStream_t s) {
    int nt = Hkv * hd / 2; if (nt > 1024) nt = 1024; nt = (nt + 63) & ~63;
    BAMD_LAUNCH(k_shift_kernel, dim3(n_cells), dim3(nt), 0, s, kc, Hkv, hd, tab_of_cell, tab);
}

static void launch_attn_fused(const bamd_attn_args & a, int gq, dim3 grid, size_t lds, hipStream_t s) {
    switch (a.hd >> 6) {                                       // head_dim 64 / 128 / 192 / 256 (checked by the callers)
        case 1: BAMD_LAUNCH((attn_fused_kernel<1>), grid, dim3(512), lds, s, a, gq); break;
        case 2: BAMD_LAUNCH((attn_fused_kernel<2>), grid, dim3(512), lds, s, a, gq); break;
        case 3: BAMD_LAUNCH((attn_fused_kernel<3>), grid, dim3(512), lds, s, a, gq); break;
        default: BAMD_LAUNCH((attn_fused_kernel<4>), grid, dim3(512), lds, s, a, gq); break;
    }
}

// attention of a micro-batch of T tokens (a.batch = 1, a.ld_qkv / a.ld_out set): KV store for all tokens, then (head, token) workgroups
int bamd_launch_attention_batch(const bamd_attn_args & a, int gq, int T, hipStream_t s) {
    const int ld = a.lds_ld ? a.lds_ld : a.n_ctx;
    if (a.hd > 256 || (a.hd & 63) || (ld & 63) || (size_t) ld * 8 > BAMD_ATTN_LDS_MAX || !a.batch) return 1;
    if (gq < 1 || gq > 8) return 1;
    BAMD_LAUNCH(kv_store_batch_kernel, dim3(a.Hkv, T), dim3(256), 0, s, a);
    if (bamd_launch_attention_batch_mfma(a, gq, T, s) == 0) return 0;        // head_dim 128 (beyond 512 positions with a.batch_scratch): the matrix-core kernel (bamd_attention_mfma.hip)
    // as many query heads of a KV head per workgroup as have their score rows fit the LDS (ld floats each: the probabilities replace
    // the scores in place); a single head per workgroup runs on attn_fused_kernel (separate rows: 2 x ld floats)
    int gqh = (gq == 2 || gq == 4 || gq == 8) ? gq : 1;          // other ratios (3: Llama-3.2-3B): one query head per workgroup
    while (gqh > 1 && (size_t) gqh * ld * 4 > BAMD_ATTN_LDS_MAX) gqh >>= 1;
    const size_t lds_g = (size_t) gqh * ld * 4;
    const dim3 grid(a.Hkv * gq / (gqh > 1 ? gqh : 1), T);
    if (gqh == 8)      BAMD_LAUNCH((attn_batch_kernel<8>), grid, dim3(512), lds_g, s, a, gq);
    else if (gqh == 4) BAMD_LAUNCH((attn_batch_kernel<4>), grid, dim3(512), lds_g, s, a, gq);
    else if (gqh == 2) BAMD_LAUNCH((attn_batch_kernel<2>), grid, dim3(512), lds_g, s, a, gq);
    else launch_attn_fused(a, gq, dim3(a.Hkv * gq, T), (size_t) ld * 8, s);
    return 0;
}

static const bool g_attn_spv = [] { const char * e = getenv("BAMD_ATTN_SPV"); return !(e && e[0] == '0'); }();
// softmax + P.V in one launch when the probability rows of a workgroup fit in LDS; false = the caller launches attn_softmax_kernel + attn_pv_kernel
template <int G> static bool spv_ok(const bamd_attn_args & a) {
    constexpr int Z = (G & 1) ? 1 : 2, NH = G / Z;              // even ratios: two workgroups per KV head (all 256 CUs at Hkv x hd/8 = 128)
    if (NH != 1 && NH != 2 && NH != 4) return false;
    const size_t lds = (size_t) NH * a.n_ctx * 4 + BAMD_SPV_SLACK, vcb = (size_t) a.Hkv * a.hd * a.n_ctx * 2;
    return g_attn_spv && lds <= BAMD_SPV_LDS_MAX && vcb <= 0x7fffffffu;
}
template <int G> static bool spv_launch(const bamd_attn_args & a, hipStream_t s) {
    constexpr int Z = (G & 1) ? 1 : 2, NH = G / Z;
    if (!spv_ok<G>(a)) return false;
    const size_t lds = (size_t) NH * a.n_ctx * 4 + BAMD_SPV_SLACK, vcb = (size_t) a.Hkv * a.hd * a.n_ctx * 2;
    BAMD_LAUNCH((attn_spv_kernel<(NH == 1 || NH == 2 || NH == 4) ? NH : 1>), dim3(a.Hkv, a.hd / 8, Z), dim3(1024), lds, s, a, G, (uint32_t) vcb);
    return true;
}
// 1 when the long-sequence path of this shape launches only kernels that keep the inter-kernel rules of bamd_device.h (scores | softmax + P.V in LDS): the step may be
// replayed from the own AQL queue; the fallback pair attn_softmax_kernel + attn_pv_kernel (score rows beyond the LDS: n_ctx > ~19 K at four heads per KV head) may not
int bamd_attention_split_is_ik_clean(const bamd_attn_args & a, int gq) {
    switch (gq) {
#define CASE(G) case G: return spv_ok<G>(a) ? 1 : 0;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    return 0;
}
int bamd_launch_attention(const bamd_attn_args & a, int gq, int max_tiles, hipStream_t s) {
    if (a.hd > 256 || (a.hd & 63)) return 1;           // chain-major K rows are read in 16-byte (8-step) groups
    if (gq < 1 || gq > 8) return 1;
    const int ld = a.lds_ld ? a.lds_ld : a.n_ctx;
    if (a.cellpos && max_tiles >= 0) return 1;                  // shifted cells: the three-launch path only (the caller passes -tiles)
    if (max_tiles >= 0 && !(ld & 63) && (size_t) ld * 8 <= BAMD_ATTN_LDS_MAX) {
        // the caller knows the sequence is short enough for one workgroup per query head and that ld bounds its padded length
        launch_attn_fused(a, gq, dim3(a.Hkv * gq), (size_t) ld * 8, s);
        return 0;
    }
    if (!a.rope_cur) return 1;                                  // the score kernel takes the cos / sin row of the position from a fixed address (step_begin_kernel)
    int ty = max_tiles < 0 ? -max_tiles : max_tiles;
    if (ty < 1) ty = 1;
    dim3 g1(a.Hkv, ty), g3(a.Hkv, a.hd / 8);
    const bool nt2 = (a.n_ctx / 64 + ty - 1) / ty <= 2;         // at most two tiles per score workgroup in this context: the two-slot instance
    switch (gq) {
#define CASE(G) case G: \
        if (a.cellpos) switch (a.hd >> 6) { \
            case 1: BAMD_LAUNCH((attn_qk_kernel<G, 1, true>), g1, dim3(512), 0, s, a); break; \
            case 2: BAMD_LAUNCH((attn_qk_kernel<G, 2, true>), g1, dim3(512), 0, s, a); break; \
            case 3: BAMD_LAUNCH((attn_qk_kernel<G, 3, true>), g1, dim3(512), 0, s, a); break; \
            default: BAMD_LAUNCH((attn_qk_kernel<G, 4, true>), g1, dim3(512), 0, s, a); break; \
        } else if (nt2) switch (a.hd >> 6) { \
            case 1: BAMD_LAUNCH((attn_qk_kernel<G, 1, false, 2>), g1, dim3(512), 0, s, a); break; \
            case 2: BAMD_LAUNCH((attn_qk_kernel<G, 2, false, 2>), g1, dim3(512), 0, s, a); break; \
            case 3: BAMD_LAUNCH((attn_qk_kernel<G, 3, false, 2>), g1, dim3(512), 0, s, a); break; \
            default: BAMD_LAUNCH((attn_qk_kernel<G, 4, false, 2>), g1, dim3(512), 0, s, a); break; \
        } else switch (a.hd >> 6) { \
            case 1: BAMD_LAUNCH((attn_qk_kernel<G, 1, false>), g1, dim3(512), 0, s, a); break; \
            case 2: BAMD_LAUNCH((attn_qk_kernel<G, 2, false>), g1, dim3(512), 0, s, a); break; \
            case 3: BAMD_LAUNCH((attn_qk_kernel<G, 3, false>), g1, dim3(512), 0, s, a); break; \
            default: BAMD_LAUNCH((attn_qk_kernel<G, 4, false>), g1, dim3(512), 0, s, a); break; \
        } \
        if (spv_launch<G>(a, s)) break; \
        BAMD_LAUNCH(attn_softmax_kernel, dim3(a.Hkv * G), dim3(1024), 0, s, a); \
        if ((G & 1) == 0) BAMD_LAUNCH(attn_pv_kernel, dim3(a.Hkv, a.hd / 8, 2), dim3(64 * (G / 2)), 0, s, a, gq); /* even ratios: two workgroups per KV head (all 256 CUs at Hkv x hd/8 = 128) */ \
        else BAMD_LAUNCH(attn_pv_kernel, g3, dim3(64 * G), 0, s, a, gq); \
        break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        default: return 1;
    }
    return 0;
}
