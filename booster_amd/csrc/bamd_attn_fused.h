// bamd_attn_fused.h — the single-launch attention of one (query head, token): body shared by attn_fused_kernel (bamd_attention.hip) and by the
// co-launched attention + wo kernel (bamd_colaunch.hip).  Numerics contract and reference citations: bamd_device.h.
#pragma once
#include "bamd_device.h"

// ---- ONE launch per layer, one workgroup per QUERY head (and per token of a prefill micro-batch) ---------------------------
// scores and probabilities live in dynamic LDS (2 x ld floats, ld = the caller's bound on the padded sequence length, independent
// of n_ctx).  Single-token decode uses this kernel for short sequences (beyond a few hundred positions one workgroup per head no
// longer has the bandwidth: three-kernel path); batched prefill, with T x H workgroups, while the rows fit the LDS.
// LG = head_dim / 64 (chain steps per lane / 8): compile-time, so that every request below is unconditional and the chains unroll.
// Latency-bound (the K / V bytes of a few hundred positions are nothing): what matters is the ORDER of the requests and that none
// of them sits behind a branch — a conditional load costs a full s_waitcnt at its join.
#define BAMD_ATTN_LDS_MAX (144 * 1024)    /* score + probability rows of one workgroup */
// COLAUNCH: the output is not stored as floats but as 8-byte GRANULES {value bits, tag} with one write-through (sc1) store each — the data is its own
// flag (MI355X_MICROARCH.md, R2): the wo workgroups of the same launch re-read their granules until every tag is this launch's tag, with no drain,
// barrier or flag hop on this side.  done_flags then points at the granule array [H * hd].
// ENV: where the body runs.  AttnEnvWG = a whole 512-thread workgroup (attn_fused_kernel, attn_wo_kernel): thread / wave ids from the hardware,
// __syncthreads, this token's q / k / v read from the f32 vectors of the argument block.  (Round 4's weight-stream engine ran the same body on eight of its
// consumer waves through another ENV; the engine was removed in round 6 — profiles/r04_engine_vs_launches.txt — the parameter stays for a role that
// takes its inputs from somewhere else.)
struct AttnEnvWG {
    int tid, wave, nthr;
    __device__ __forceinline__ AttnEnvWG() : tid((int) threadIdx.x), wave(wave_id()), nthr((int) blockDim.x) { }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ float2 qk2(const bamd_attn_args & a, int role, int h, int hk, int hd, int rp) const {
        const float * rsrc = role == 1 ? a.k + (size_t) hk * hd : a.q + (size_t) h * hd;
        return BAMD_IK_QKV ? ik_ld2f(rsrc + 2 * rp) : *(const float2 *) (rsrc + 2 * rp);                        // q | k | v: the QKV launch's output (inter-kernel data, bamd_device.h)
    }
    __device__ __forceinline__ float vel(const bamd_attn_args & a, int hk, int hd, int i) const { return ik_ld_if<BAMD_IK_QKV != 0>(a.v + hk * hd + i); }
};
template <int LG, bool COLAUNCH, typename ENV = AttnEnvWG>
__device__ __forceinline__ void attn_fused_body(bamd_attn_args a, int gq, const int h, const int tokb_in, unsigned char * attn_dyn, uint32_t * done_flags, uint32_t tag, const ENV env = ENV()) {
    __shared__ __attribute__((aligned(16))) float qt[256];
    __shared__ __attribute__((aligned(16))) unsigned short q16t[256];
    __shared__ __attribute__((aligned(16))) unsigned short k16t[256];
    float * sc = (float *) attn_dyn;                                         // [ld] scores, then exp values (natural order)
    float * pt = sc + (a.lds_ld ? a.lds_ld : a.n_ctx);                       // [ld] probabilities in V^T position order (ld bounds the padded sequence length)
    __shared__ float redf[8];
    __shared__ double redd[8];
    const bamd_step_state * st = a.st;
    // batched prefill (a.batch): tokb_in = token of the micro-batch (the kernel's blockIdx.y); its K/V rows and those of the earlier tokens of the batch
    // were stored by kv_store_batch_kernel, and masked positions are exact no-ops, so each token uses its own padded length
    const int tokb = a.batch ? tokb_in : 0;
    const int pos = ik_ld_if<BAMD_IK_ST != 0>(&st->pos) + tokb;                  // the step state is inter-kernel data: never through the scalar cache
    int n_kv = ik_ld_if<BAMD_IK_ST != 0>(&st->n_kv);
    if (a.batch) { const int nc = ik_ld(&st->n_ctx); n_kv = (pos + 1 + 31) / 32 * 32; n_kv = n_kv < nc ? n_kv : nc; }
    a.q += (size_t) tokb * a.ld_qkv; a.k += (size_t) tokb * a.ld_qkv; a.v += (size_t) tokb * a.ld_qkv; a.out += (size_t) tokb * a.ld_out;
    constexpr int hd = LG * 64, L = LG * 8, hp = hd / 2;
    const int Hkv = a.Hkv, Ekv = Hkv * hd, n_ctx = a.n_ctx;
    const int hk = h / gq;
    const int tid = env.tid, lane = tid & 63, wave = env.wave, e = lane & 7;
    const float * rope = a.rope + (size_t) pos * hd;
    TL_STAMP(a.tl, 0);
    // 1. the small, latency-critical requests go out FIRST: this token's q / k pair and its cos / sin for the thread's RoPE role, and
    //    the v elements the KV store and the P.V splice need.  Loads return in order: they land ~1 us before the K / V^T chunks
    //    requested behind them, and RoPE runs while those stream in.
    const int role = tid / hp, rp = tid - role * hp;              // role 0: q pair rp, role 1: k pair rp, others: (a redundant copy of role 0)
    const float2 xin = env.qk2(a, role, h, hk, hd, rp);
    const float2 cs = *(const float2 *) (rope + 2 * rp);
    const float vst = env.vel(a, hk, hd, tid < hd ? tid : 0);     // KV store: element tid of this token's v
    const int r_pos = wave * 8 + (lane >> 3);                     // position inside a 64-tile (scores) / d inside a 64-block (P.V)
    float vcf[LG];
#pragma unroll
    for (int dd = 0; dd < LG; ++dd) vcf[dd] = env.vel(a, hk, hd, r_pos + 64 * dd);
    // 2. this lane's K chunks of the first 4 x 64 positions and its V^T chunks of the first 4 blocks (rows d = r_pos + 64 dd), all
    //    unconditional: a tile past the end of the cache is clamped to the last one, positions >= pos hold zeros or stale finite
    //    values whose scores are masked below and whose probabilities are exactly 0
    // the KV cache is inter-kernel data too (earlier steps' launches wrote the rows): sc1 buffer loads, 32-bit byte offsets into this layer's K / V^T
    const bamd_ik_rsrc rk = ik_rsrc(a.kc), rv = ik_rsrc(a.vc);
    uint4 kreg[4][LG];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = t * 64 + r_pos < n_ctx ? t * 64 + r_pos : n_ctx - 64 + r_pos;
#pragma unroll
        for (int g = 0; g < LG; ++g) kreg[t][g] = ik_ld128_if<BAMD_IK_KLD != 0>(rk, (uint32_t) (i * Ekv + hk * hd + g * BAMD_KGRP + e * 8) * 2u);
    }
    uint4 vreg[4][LG < 2 ? LG : 2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tb = t * 64 < n_ctx ? t * 64 : n_ctx - 64;
#pragma unroll
        for (int dd = 0; dd < (LG < 2 ? LG : 2); ++dd) vreg[t][dd] = ik_ld128_if<BAMD_IK_VLD != 0>(rv, (uint32_t) ((hk * hd + r_pos + 64 * dd) * n_ctx + tb + e * 8) * 2u);
    }
    {   // RoPE (NORM mode, adjacent pairs; ggml.c:14130-14143 — rope_heads' arithmetic) into the chain-major LDS copies
        const float t0 = xin.x * cs.x, t1 = xin.y * cs.y, t2 = xin.x * cs.y, t3 = xin.y * cs.x;
        const float r0 = t0 - t1, r1 = t2 + t3;
        const int i0 = kperm(2 * rp, L), i1 = kperm(2 * rp + 1, L);
        if (role == 0) { qt[i0] = r0; qt[i1] = r1; q16t[i0] = f2h(r0); q16t[i1] = f2h(r1); }
        else if (role == 1) { k16t[i0] = f2h(r0); k16t[i1] = f2h(r1); }
    }
    env.sync();
    // KV store by the first query head of each KV head — llm_build_kv_store, llama.cpp:7830-7875
    if (h == hk * gq && !a.batch && tid < hd) {
        ik_st_if<BAMD_IK_KVST != 0>(a.kc + (size_t) pos * Ekv + hk * hd + tid, k16t[tid]);
        ik_st_if<BAMD_IK_KVST != 0>(a.vc + (size_t) (hk * hd + tid) * n_ctx + vperm(pos), f2h(vst));
    }
    TL_STAMP(a.tl, 1);
    // ---- scores: every chain runs unconditionally (independent chains interleave), the mask is a select at the end ----
    uint4 kself[4];                                                // this token's K row is not visible in the cache yet
#pragma unroll
    for (int g = 0; g < 4; ++g) kself[g] = g < LG ? *(const uint4 *) (k16t + g * BAMD_KGRP + e * 8) : make_uint4(0, 0, 0, 0);
#define BAMD_SCORE_TILE(t0_, KL_) do { \
        const int i = (t0_) + r_pos; \
        uint4 kk[4]; \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) { \
            const uint4 kc_ = g < LG ? KL_[g < LG ? g : 0] : make_uint4(0, 0, 0, 0); \
            kk[g].x = i == pos ? kself[g].x : kc_.x; kk[g].y = i == pos ? kself[g].y : kc_.y; kk[g].z = i == pos ? kself[g].z : kc_.z; kk[g].w = i == pos ? kself[g].w : kc_.w; \
        } \
        float v = a.prefill_mode ? hsum8_vecdot(kq_chain<true>(kk, L, nullptr, q16t + e * 8)) : hsum8_tinyblas(kq_chain<false>(kk, L, qt + e * 8, nullptr)); \
        v = (i < n_kv && i <= pos) ? v : -INFINITY;               /* masked (KQ_mask, llama.cpp:14152-14200) */ \
        if (e == 0 && i < n_kv) sc[i] = v; \
    } while (0)
#pragma unroll
    for (int t = 0; t < 4; ++t) { if (t * 64 < n_kv) BAMD_SCORE_TILE(t * 64, kreg[t]); }
    for (int t0 = 256; t0 < n_kv; t0 += 64) {
        const int i2 = t0 + r_pos;
        uint4 kl[LG];
#pragma unroll
        for (int g = 0; g < LG; ++g) kl[g] = ik_ld128_if<BAMD_IK_KLD != 0>(rk, (uint32_t) (i2 * Ekv + hk * hd + g * BAMD_KGRP + e * 8) * 2u);    // n_kv <= n_ctx: in bounds
        BAMD_SCORE_TILE(t0, kl);
    }
#undef BAMD_SCORE_TILE
    env.sync();
    TL_STAMP(a.tl, 2);
    // ---- softmax (ggml.c:13682-13778 + :2619-2671): wp = s*scale (+mask), max, exp, 8-chunk f32 sums, double total ----
    const float scale = a.kq_scale;
    float mx = -INFINITY;
    for (int i = tid; i < n_kv; i += env.nthr) { const float w = sc[i] * scale; mx = w > mx ? w : mx; }
    {   // -inf..inf floats: order-preserving key for an unsigned max
        uint32_t u = __float_as_uint(mx); u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        u = wave_max_u32(u);
        if (lane == 0) redf[wave] = __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
    }
    env.sync();
    mx = redf[0];
    for (int w = 1; w < 8; ++w) mx = redf[w] > mx ? redf[w] : mx;
    double sum = 0.0;
    for (int i = tid; i < n_kv; i += env.nthr) {                 // n_kv % 32 == 0: 8-lane groups are all-active or all-idle
        const float w = sc[i] * scale;
        const float val = v_expf(w - mx);
        sc[i] = val;
        const float c = hsum8_tinyblas(val);
        if (e == 0) sum += (double) c;
    }
    sum = wave_sum_f64(sum);
    if (lane == 0) redd[wave] = sum;
    env.sync();
    double tot = 0.0;
    for (int w = 0; w < 8; ++w) tot += redd[w];
    double rs = 1.0 / tot;
    float fs = (float) rs;
    if (!f32_rounding_safe(rs, BAMD_F64_GUARD_ULPS(n_kv / 8))) {          // workgroup-uniform, rare: the reference's sequential order (bamd_device.h)
        env.sync();
        if (tid == 0) redd[0] = seq_expsum8(sc, n_kv);
        env.sync();
        rs = 1.0 / redd[0]; fs = (float) rs;
    }
    for (int i = tid; i < n_kv; i += env.nthr) pt[vperm(i)] = sc[i] * fs;
    // half-filled last block (n_kv % 64 == 32): p = 0 for the missing positions, so the chain steps there are exact no-ops
    for (int i = n_kv + tid; i < ((n_kv + 63) & ~63); i += env.nthr) pt[vperm(i)] = 0.f;
    env.sync();
    TL_STAMP(a.tl, 3);
    // ---- P.V: lane (d, e) carries the tinyBLAS chain Cv[e] of output d (A = V^T row, B = p), up to 4 rows d per lane ----
    unsigned short vcur[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int dd = 0; dd < LG; ++dd) vcur[dd] = f2h(vcf[dd]);       // requested at entry
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
        acc4[dd_] = fma_mix_chain<8>(acc4[dd_], w, pv); \
    } while (0)
#pragma unroll
    for (int t = 0; t < 4; ++t) {                                  // blocks whose V chunks were requested at kernel entry
        if (t * 64 < n_kv) {
#pragma unroll
            for (int dd = 0; dd < (LG < 2 ? LG : 2); ++dd) BAMD_PV_BLOCK(t * 64, dd, vreg[t][dd]);
#pragma unroll
            for (int dd = 2; dd < LG; ++dd) {                      // hd > 128
                const uint4 vv = ik_ld128_if<BAMD_IK_VLD != 0>(rv, (uint32_t) ((hk * hd + r_pos + 64 * dd) * n_ctx + t * 64 + e * 8) * 2u);
                BAMD_PV_BLOCK(t * 64, dd, vv);
            }
        }
    }
    for (int b0 = 256; b0 < n_kv; b0 += 64) {
#pragma unroll
        for (int dd = 0; dd < LG; ++dd) {
            const uint4 vv = ik_ld128_if<BAMD_IK_VLD != 0>(rv, (uint32_t) ((hk * hd + r_pos + 64 * dd) * n_ctx + b0 + e * 8) * 2u);
            BAMD_PV_BLOCK(b0, dd, vv);
        }
    }
#undef BAMD_PV_BLOCK
#pragma unroll
    for (int dd = 0; dd < LG; ++dd) {
        const float v = hsum8_tinyblas(acc4[dd]);
        if (e == 0) {
            if (COLAUNCH) __hip_atomic_store((unsigned long long *) done_flags + (size_t) h * hd + r_pos + 64 * dd, ((unsigned long long) tag << 32) | __float_as_uint(v),
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ik_st(a.out + (size_t) h * hd + r_pos + 64 * dd, v);
        }
    }
    TL_STAMP(a.tl, 7);
}
