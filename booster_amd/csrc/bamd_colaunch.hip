// bamd_colaunch.hip — attention and the wo projection of a decode step in ONE launch.
//
// The single-launch attention (one workgroup per query head, latency-bound: ~4.5 us for a few hundred cached positions) leaves 224 of the
// 256 CUs idle, and the wo launch behind it spends most of its ~5 us waiting: kernel boundary, wave launch, weight requests, HBM latency.
// Two launches cannot overlap on this runtime — the command processor starts a dispatch only when its predecessor in the queue has
// finished, barrier bit or not, and a second queue costs ~7 us per cross-queue edge (tools/aql_probe.hip, profiles/r03_aql_probe.txt) —
// but the workgroups of ONE launch run side by side.  So this kernel has two roles, by workgroup index:
//   blockIdx <  H : attention of query head blockIdx (attn_fused_body: the same code as attn_fused_kernel); its output is stored as 8-byte
//                   granules {value bits, tag of this launch}, one write-through (sc1) store each: the data is its own flag;
//   blockIdx >= H : wo, split-K as matvec_split_fast_kernel does it (8 waves share a row-group, terms parked in LDS, one wave replays the
//                   reference's sequential f32 chain), over the other n_cu - H CUs: the weight records of ALL its row-groups are requested
//                   early (a little after entry, so that the burst does not sit in front of the attention's latency-critical first loads) and
//                   land while the attention runs; every wave then re-reads the granules of its own K-slice (sc1 loads) until all carry the
//                   tag, quantises them, parks its terms.
// Placement-independent: the grid is one workgroup per CU, so all roles are resident together whatever the dispatch order; every spin is
// bounded and reports through `err` (the host checks it).  The tag = (host call serial, device step, layer) is unique among consecutive
// uses of the granules, which are never reset.  Same arithmetic, same order, same bits as the two separate launches (tests).
#include "bamd_matvec_core.h"
#include "bamd_attn_fused.h"

// two granules of another workgroup's write-through output: L1 bypassed (sc1), as the producer stored them; each 8-byte half is one store
__device__ __forceinline__ uint4 ld_coh128(bamd_rsrc r, uint32_t byte_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int) byte_off, 0, 16);      // aux 16 = sc1
    return make_uint4(v.x, v.y, v.z, v.w);
}

#define BAMD_COLAUNCH_SPINS (1u << 22)      /* ~1 s of polling: a launch whose attention role never publishes gives up instead of hanging the device */

// wo role: workgroup j of G takes row-groups j, j + G, ... (M of them), NBW records per wave and row-group
template <int TYPE, int NBW, int M>
__device__ __forceinline__ void wo_role(const bamd_mv_args & a, const ProArgs & pa, const int j, const int G, float * part0, const unsigned long long * gran,
                                        const bamd_step_state * st, const int il, const int ring_delay_in, uint32_t * err) {
    const int ring_delay = ring_delay_in & 0xff; const bool poll_sleep = (ring_delay_in >> 8) & 1;
    typedef typename RecOf<TYPE>::type REC;
    constexpr int RECB = TYPE == BAMD_Q4_K ? BAMD_RECB_Q4K : TYPE == BAMD_Q5_K ? BAMD_RECB_Q5K : 1680;
    const int nb = pa.K >> 8;
    const int lane = threadIdx.x & 63, wave = wave_id(), r8 = lane >> 3;
    const int i0 = wave * NBW;
    const bamd_rsrc rs = weight_rsrc(a.seg[0].w);
    const int rgb = nb * RECB;
    for (int d = 0; d < ring_delay; ++d) __builtin_amdgcn_s_sleep(8);   // ~0.2 us each: the attention role's first requests go first
    REC ring[M * NBW];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int jj = 0; jj < NBW; ++jj) load_rec(ring[m * NBW + jj], rs, (j + m * G) * rgb + (i0 + jj) * RECB, lane);
    const int nv = a.seg[0].nvalid > 0 ? a.seg[0].nvalid : a.seg[0].nrows;
    const int crow = (j + (wave < M ? wave : 0) * G) * 8 + r8;          // the row whose chain this wave replays
    float resv = 0.f;
    if (crow < nv) resv = ik_ld(a.res + crow);                            // the residual: written two launches ago
    // what the terms need from the weights alone — nibbles, scales, mins, d — while the attention role works (the records have landed long before its granules)
    PreTerms<TYPE> pre[M * NBW];
#pragma unroll
    for (int r = 0; r < M * NBW; ++r) { pin_rec(ring[r]); pre[r].prep(ring[r], lane); }
    TL_STAMP(pa.tl, 1);
    // ---- this wave's slice of the attention output: blocks i0 .. i0 + NBW - 1 = 4 granules per lane and block, re-read until every tag is
    //      this launch's; then Q8_K into LDS (no workgroup barrier: the wave consumes only what it quantised itself) ----
    ActPro<false> ap; ap.tl = pa.tl; ap.okmask = (1 << NBW) - 1;
    {
        const uint32_t tag = ((uint32_t) ik_ld_if<BAMD_IK_ST != 0>(&st->serial) << 20) | (((uint32_t) ik_ld_if<BAMD_IK_ST != 0>(&st->step) & 0xfffu) << 8) | (uint32_t) il;
        const bamd_rsrc gr = weight_rsrc(gran);
        unsigned spins = 0;
        for (;;) {
            asm volatile("" ::: "memory");                               // the granules change under us: every pass re-reads them
            bool ok = true;
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                const uint32_t off = (uint32_t) ((i0 + b) * 256 + lane * 4) * 8u;
                const uint4 g0 = ld_coh128(gr, off), g1 = ld_coh128(gr, off + 16u);
                ap.v[b] = make_float4(__uint_as_float(g0.x), __uint_as_float(g0.z), __uint_as_float(g1.x), __uint_as_float(g1.z));
                ok = ok && g0.y == tag && g0.w == tag && g1.y == tag && g1.w == tag;
            }
            if (__all(ok)) break;
            if (poll_sleep) __builtin_amdgcn_s_sleep(1);
            if (++spins > BAMD_COLAUNCH_SPINS) { if (lane == 0) atomicAdd(err, 1u); break; }
        }
    }
    TL_STAMP(pa.tl, 2);
    ap.template quantize_batch<NBW>(1.0f, pa.K, i0, pa.q8, pa.S, pa.yd, 1, i0 + NBW);
    const uint32_t * q8 = pa.q8; const int * S = pa.S; const float * yd = pa.yd;
    const size_t rg_floats = BAMD_TERM_FLOATS(nb);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        float4 * P = (float4 *) (part0 + (size_t) m * rg_floats);
#pragma unroll
        for (int jj = 0; jj < NBW; ++jj) {
            const Terms T = pre[m * NBW + jj].finish(i0 + jj, lane, q8, S, yd);
            P[(i0 + jj) * 64 + lane] = make_float4(T.d, T.fs, T.dmin, T.pm);
        }
    }
    TL_STAMP(pa.tl, 3);
    __syncthreads();
    TL_STAMP(pa.tl, 4);
    if (wave < M) {                                                      // the reference's chain, in order, for lane (r, e) (split_stream)
        const float4 * P = (const float4 *) (part0 + (size_t) wave * rg_floats);
        RowAcc A = { 0.f, 0.f };
        for (int i = 0; i < nb; i += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = P[(i + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) chain_step<TYPE>(A, t[u].x, t[u].y, t[u].z, t[u].w);
        }
        const float val = finish_row<TYPE>(A);
        if ((lane & 7) == 0 && crow < nv) ik_st(a.seg[0].out + crow, val + resv);
        TL_STAMP(pa.tl, 5);
    }
}

// wo role at the Llama-3-70B width (K = 8192: four records per wave and row-group, five or six row-groups per workgroup): too many records to hold them
// all in registers as wo_role does, so this role is the batched split-K loop of the stand-alone launch (split_stream, two row-groups per batch,
// double-buffered terms) with the granule wait in the place of its activation requests: the records of the FIRST batch are requested at entry and
// land while the attention role runs, the later batches stream behind the hand-over.
template <int TYPE>
__device__ __forceinline__ void wo_role_batched(const bamd_mv_args & a, const ProArgs & pa, const int j, const int G, const int count, float * part0, const unsigned long long * gran,
                                                const bamd_step_state * st, const int il, const int ring_delay_in, uint32_t * err) {
    typedef typename RecOf<TYPE>::type REC;
    constexpr int NBW = 4;
    const int ring_delay = ring_delay_in & 0xff; const bool poll_sleep = (ring_delay_in >> 8) & 1;
    const int nb = pa.K >> 8, lane = threadIdx.x & 63, i0 = wave_id() * NBW;
    for (int d = 0; d < ring_delay; ++d) __builtin_amdgcn_s_sleep(8);
    ActPro<false> ap, ap2; ap.tl = pa.tl; ap.okmask = (1 << NBW) - 1;
    auto wait_for_attention = [&]() {
        const uint32_t tag = ((uint32_t) ik_ld_if<BAMD_IK_ST != 0>(&st->serial) << 20) | (((uint32_t) ik_ld_if<BAMD_IK_ST != 0>(&st->step) & 0xfffu) << 8) | (uint32_t) il;
        const bamd_rsrc gr = weight_rsrc(gran);
        unsigned spins = 0;
        for (;;) {
            asm volatile("" ::: "memory");
            bool ok = true;
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                const uint32_t off = (uint32_t) ((i0 + b) * 256 + lane * 4) * 8u;
                const uint4 g0 = ld_coh128(gr, off), g1 = ld_coh128(gr, off + 16u);
                ap.v[b] = make_float4(__uint_as_float(g0.x), __uint_as_float(g0.z), __uint_as_float(g1.x), __uint_as_float(g1.z));
                ok = ok && g0.y == tag && g0.w == tag && g1.y == tag && g1.w == tag;
            }
            if (__all(ok)) break;
            if (poll_sleep) __builtin_amdgcn_s_sleep(1);
            if (++spins > BAMD_COLAUNCH_SPINS) { if (lane == 0) atomicAdd(err, 1u); break; }
        }
        TL_STAMP(pa.tl, 2);
    };
    int batchctr = 0;
    const int nv = a.seg[0].nvalid > 0 ? a.seg[0].nvalid : a.seg[0].nrows;
    split_stream<TYPE, REC, NBW, 2, 2, BAMD_EPI_ADD, BAMD_PRO_PLAIN, true, false, false>((const uint8_t *) a.seg[0].w, nb, j, count, G, a.seg[0].out, a.res, pa, ap, ap2, false, true,
                                                                                         part0, batchctr, nv, wait_for_attention);
}

template <int LG, int TYPE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) attn_wo_kernel(bamd_attn_args at, int gq, bamd_mv_args wo, unsigned long long * gran, int il, int extra,
                                                                                                    int ring_delay, uint32_t * err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = at.Hkv * gq;
    if ((int) blockIdx.x < H) {
        const bamd_step_state * st = at.st;
        const uint32_t tag = ((uint32_t) ik_ld_if<BAMD_IK_ST != 0>(&st->serial) << 20) | (((uint32_t) ik_ld_if<BAMD_IK_ST != 0>(&st->step) & 0xfffu) << 8) | (uint32_t) il;
        attn_fused_body<LG, true>(at, gq, (int) blockIdx.x, 0, smem, (uint32_t *) gran, tag);
        return;
    }
    TL_STAMP(wo.tl, 0);
    const int j = (int) blockIdx.x - H, G = (int) gridDim.x - H;
    const ProArgs pa = carve_lds(wo, smem);
    float * part0 = (float *) (smem + BAMD_ACT_RED_OFF(wo.K >> 8) + 16 * sizeof(double));
    if ((wo.K >> 8) == 32) {                                                                          // 70B width: batched (count = 5 or 6 row-groups)
        const int nrg = wo.seg[0].nrows >> 3;
        wo_role_batched<TYPE>(wo, pa, j, G, nrg / G + (j < nrg % G ? 1 : 0), part0, gran, at.st, il, ring_delay, err);
    }
    else if (j < extra) wo_role<TYPE, 2, 3>(wo, pa, j, G, part0, gran, at.st, il, ring_delay, err);  // the first `extra` workgroups take a third row-group
    else                wo_role<TYPE, 2, 2>(wo, pa, j, G, part0, gran, at.st, il, ring_delay, err);
    TL_STAMP(wo.tl, 7);
}

static const bool g_colaunch = [] { const char * e = getenv("BAMD_COLAUNCH"); return !(e && e[0] == '0'); }();
static const int g_ring_delay = [] { const char * e = getenv("BAMD_COLAUNCH_DELAY"); return e ? atoi(e) : 12 + 256; }();   // low byte: x ~0.2 us; + 256: s_sleep between polls (A/B on the MI355X, round 3 end: 707 / 708 / 710 / 713-717 / 712 / 705 / 694 tok/s at 6 / 8 / 10 / 12 / 14 / 16 / 20, all + sleep)      // x ~0.2 us before the wo role requests its weights

// 0 = launched; 1 = this shape has no co-launch kernel (the caller issues the two ordinary launches)
int bamd_launch_attn_wo(const bamd_attn_args & t, int gq, const bamd_mv_args & wo, int n_cu, unsigned long long * gran, int il, uint32_t * err, hipStream_t s) {
    if (!g_colaunch || !gran || !err) return 1;
    const int H = t.Hkv * gq, ld = t.lds_ld ? t.lds_ld : t.n_ctx;
    if (t.batch || t.cellpos || t.hd > 256 || (t.hd & 63) || gq < 1 || gq > 8 || (ld & 63) || (size_t) ld * 8 > BAMD_ATTN_LDS_MAX || il < 0 || il > 255) return 1;
    const int nb = wo.K >> 8, type = wo.seg[0].type;
    if (wo.nseg != 1 || (nb != 16 && nb != 32) || (wo.mode & 31) != 0 || !wo.res) return 1;  // K = 4096: two records per wave and row-group; K = 8192: four, in batches
    const int G = n_cu - H, nrg = wo.seg[0].nrows >> 3;
    if (G < 8) return 1;
    if (nb == 16 && nrg / G != 2) return 1;                                                   // two or three row-groups per wo workgroup
    if (nb == 32) {
        // the 70B width: MEASURED NEUTRAL (round 4: a 10-layer stage 1.267 ms per token co-launched against 1.260 with the two launches; the whole model 110.9
        // against 115.2 tok/s in one bench.py pairing) — the wo role's 5 - 6 row-groups x 32 records are 190 KB per workgroup, of which only the first
        // batch can be in flight while the attention role works, and the hand-over costs what the boundary did.  BAMD_COLAUNCH70=1 selects it.
        static const bool on70 = [] { const char * e = getenv("BAMD_COLAUNCH70"); return e && e[0] == '1'; }();
        if (!on70 || nrg / G < 2) return 1;
    }
    const int extra = nrg - 2 * G;
    const size_t lds_wo = act_lds_bytes(wo.K) + 16 + (nb == 32 ? (size_t) 2 * 2 * nb * 256 * 4 : (size_t) 3 * nb * 256 * 4), lds_at = (size_t) ld * 8;
    const size_t lds = lds_wo > lds_at ? lds_wo : lds_at;
    const dim3 grid(n_cu), block(512);
#define BAMD_CL(LG_, T_) BAMD_LAUNCH((attn_wo_kernel<LG_, T_>), grid, block, lds, s, t, gq, wo, gran, il, extra, g_ring_delay, err)
#define BAMD_CL_T(LG_) do { if (type == BAMD_Q4_K) BAMD_CL(LG_, BAMD_Q4_K); else if (type == BAMD_Q5_K) BAMD_CL(LG_, BAMD_Q5_K); else if (type == BAMD_Q6_K) BAMD_CL(LG_, BAMD_Q6_K); else return 1; } while (0)
    switch (t.hd >> 6) {
        case 1: BAMD_CL_T(1); break;
        case 2: BAMD_CL_T(2); break;
        case 3: BAMD_CL_T(3); break;
        default: BAMD_CL_T(4); break;
    }
#undef BAMD_CL_T
#undef BAMD_CL
    return 0;
}
