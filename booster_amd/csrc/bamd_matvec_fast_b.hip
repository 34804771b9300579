// bamd_matvec_fast_b.hip — fast mode-B (split-K) mat-vec kernels and their host-side dispatch.  See bamd_matvec_fast_a.hip for the
// rationale of the fast kernels.
#include "bamd_matvec_core.h"

// NW: waves per workgroup (8; 14 for K = 14336 = 14 x 4 super-blocks: the ffn_down launch is not paced by its bytes but by what ONE wave walks
// between entry and its last store — Q8_K of its blocks, the terms of its records, the chain — at one vector instruction per ~10 clocks)
// COMPACT (NW = 16, NBW = 7: K = 28672 = 16 x 7 super-blocks, the 70B ffn_down): two compact term buffers (bamd_matvec_core.h)
template <int TYPE, int NBW, int M, int PRO, int EPI, bool ONEB, int NW = 8, bool COMPACT = false>
__global__ void __launch_bounds__(64 * NW) matvec_split_fast_kernel(BAMD_LEAD_PARAMS, bamd_mv_args a) {
    BAMD_LEAD_TAKE(a);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    ActPro<PRO == BAMD_PRO_NORM> ap, ap2;
    if (PRO == BAMD_PRO_PLAIN) {                             // own K-slice only (see split_stream)
        const int i0 = wave_id() * NBW;
        ap.template issue<BAMD_NB1(NBW)>(pa.x, pa.nw, pa.K, i0, 1, i0 + NBW);
        if (NBW > BAMD_ACT_BATCH) ap2.template issue<BAMD_NB2(NBW)>(pa.x, pa.nw, pa.K, i0 + BAMD_ACT_BATCH, 1, i0 + NBW);
    } else BAMD_PRO_ISSUE_NB(ap, pa, BAMD_NB1(NBW));         // shared prologue: wave w takes blocks w, w + 8, ...: NBW of them
    float * part0 = (float *) (smem + BAMD_ACT_RED_OFF(nb) + 16 * sizeof(double));
    int rgctr = 0;
    const int count = a.cnt_q + ((int) blockIdx.x < a.cnt_r ? 1 : 0);
    const int nv = a.seg[0].nvalid > 0 ? a.seg[0].nvalid : a.seg[0].nrows;
    typedef typename RecOf<TYPE>::type REC;
    constexpr int NBUF = COMPACT ? 2 : NBW * M > 8 ? 1 : 2;  // term buffers: 2 x M x K/256 KiB must fit the LDS
    split_stream<TYPE, REC, NBW, M, NBUF, EPI, PRO, true, ONEB, false, SplitNoPre, COMPACT>((const uint8_t *) a.seg[0].w, nb, (int) blockIdx.x, count, (int) gridDim.x, a.seg[0].out, a.res, pa,
                                                       ap, ap2, false, true, part0, rgctr, nv, SplitNoPre());      // the launcher's grid gives every workgroup >= 1 row-group
    TL_STAMP(a.tl, 7);
}


// ---- fused QKV with a differently typed attn_v (wq|wk Q4_K + wv Q6_K / Q5_K): split-K over BOTH segments in one pass -------------
// Every workgroup takes MA row-groups of segment 0 (type TA) and ONE more row-group that is of type TA (still segment 0) or of
// type TB (segment 1) depending on the workgroup: all MA + 1 rings are requested at entry, all terms are parked in one batch, waves
// 0..MA replay one chain each.  The type of the last row-group is a template parameter and the kernel branches on it ONCE, right
// after the activation requests, so that each side is straight-line code with counted waits.  (Through mode A this launch ran on 3
// of 8 waves per CU, 16 records each: 10 us against 6.4 us for the all-Q4_K layers.)
// COMPACT (the 70B widths: five row-groups of 32 super-blocks per workgroup): a parked record takes 576 bytes — {fs, pm} per lane, {d, dmin} per row —
// instead of a float4 per lane, so that the five term buffers fit the LDS (92 KB); the first MA - 2 rings go out at entry, the others behind the
// prologue's first barrier
template <int TA, int TB, int NBW, int MA, bool COMPACT = false>
__device__ __forceinline__ void split_mixed_body(const bamd_mv_args & a, const ProArgs & pa, ActPro<true> & ap, float * part0, int g_last) {
    typedef typename RecOf<TA>::type RECA;
    typedef typename RecOf<TB>::type RECB_T;
    constexpr int RA = TA == BAMD_Q4_K ? BAMD_RECB_Q4K : TA == BAMD_Q5_K ? BAMD_RECB_Q5K : 1680;
    constexpr int RB = TB == BAMD_Q4_K ? BAMD_RECB_Q4K : TB == BAMD_Q5_K ? BAMD_RECB_Q5K : 1680;
    constexpr bool SAME = TA == TB;                          // the last row-group still belongs to segment 0
    const int nb = pa.K >> 8;
    const int lane = threadIdx.x & 63, wave = wave_id(), r8 = lane >> 3;
    const int i0 = wave * NBW, grid = (int) gridDim.x, b = (int) blockIdx.x;
    const int nrg0 = a.seg[0].nrows >> 3;
    const bamd_rsrc rs0 = weight_rsrc(a.seg[0].w), rs1 = weight_rsrc(SAME ? a.seg[0].w : a.seg[1].w);
    const int rgbA = nb * RA, rgbB = nb * RB;
    RECA ring[MA * NBW]; RECB_T ringL[NBW];
    constexpr int MEARLY = COMPACT ? (MA > 2 ? MA - 2 : MA) : MA;      // rings requested at entry
#pragma unroll
    for (int m = 0; m < MEARLY; ++m)
#pragma unroll
        for (int j = 0; j < NBW; ++j) load_rec(ring[m * NBW + j], rs0, (b + m * grid) * rgbA + (i0 + j) * RA, lane);
    const int lastoff = SAME ? g_last * rgbA : (g_last - nrg0) * rgbB;
    TL_STAMP(pa.tl, 1);
    // the last row-group's records are requested behind the prologue's first barrier (stream_segment: the texture path of a CU is busy
    // accepting the first rings for ~1 us, and the barrier would wait for the slowest wave's issue stage)
    auto last_ring = [&]() {
#pragma unroll
        for (int m = MEARLY; m < MA; ++m)
#pragma unroll
            for (int j = 0; j < NBW; ++j) load_rec(ring[m * NBW + j], rs0, (b + m * grid) * rgbA + (i0 + j) * RA, lane);
#pragma unroll
        for (int j = 0; j < NBW; ++j) load_rec(ringL[j], rs1, lastoff + (i0 + j) * RB, lane);
    };
    BAMD_PRO_FINISH_NB_MID(ap, pa, last_ring, BAMD_NB1(NBW));
    TL_STAMP(pa.tl, 2);
    const uint32_t * q8 = pa.q8; const int * S = pa.S; const float * yd = pa.yd;
    const size_t rg_floats = COMPACT ? (size_t) nb * 144 : BAMD_TERM_FLOATS(nb);         // floats per parked row-group (compact: 576 bytes per record)
    auto park = [&](float * base, int ci, const Terms & T) {
        if (COMPACT) {
            float * rec = base + (size_t) ci * 144;
            *(float2 *) (rec + lane * 2) = make_float2(T.fs, T.pm);
            if ((lane & 7) == 0) *(float2 *) (rec + 128 + r8 * 2) = make_float2(T.d, T.dmin);
        } else ((float4 *) base)[ci * 64 + lane] = make_float4(T.d, T.fs, T.dmin, T.pm);
    };
#pragma unroll
    for (int m = 0; m < MA; ++m) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            pin_rec(ring[m * NBW + j]);
            const Terms T = block_terms(ring[m * NBW + j], i0 + j, lane, q8, S, yd);
            park(part0 + (size_t) m * rg_floats, i0 + j, T);
        }
    }
    {
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            pin_rec(ringL[j]);
            const Terms T = block_terms(ringL[j], i0 + j, lane, q8, S, yd);
            park(part0 + (size_t) MA * rg_floats, i0 + j, T);
        }
    }
    TL_STAMP(pa.tl, 3);
    __syncthreads();
    TL_STAMP(pa.tl, 4);
    if (wave <= MA) {                                        // one chain per wave, in the reference's order (split_stream)
        const float * Pf = part0 + (size_t) wave * rg_floats;
        const float4 * P = (const float4 *) Pf;
        const bool lastw = wave == MA;
        RowAcc A = { 0.f, 0.f };
        float val;
        auto fetch8 = [&](int i, float4 (&t)[8]) {           // terms of super-blocks i .. i + 7 as {d, fs, dmin, pm}
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (COMPACT) {
                    const float * rec = Pf + (size_t) (i + u) * 144;
                    const float2 fp = *(const float2 *) (rec + lane * 2), dd = *(const float2 *) (rec + 128 + r8 * 2);
                    t[u] = make_float4(dd.x, fp.x, dd.y, fp.y);
                } else t[u] = P[(i + u) * 64 + lane];
            }
        };
        if (SAME || !lastw) {
            for (int i = 0; i < nb; i += 8) {
                float4 t[8];
                fetch8(i, t);
#pragma unroll
                for (int u = 0; u < 8; ++u) chain_step<TA>(A, t[u].x, t[u].y, t[u].z, t[u].w);
            }
            val = finish_row<TA>(A);
        } else {
            for (int i = 0; i < nb; i += 8) {
                float4 t[8];
                fetch8(i, t);
#pragma unroll
                for (int u = 0; u < 8; ++u) chain_step<TB>(A, t[u].x, t[u].y, t[u].z, t[u].w);
            }
            val = finish_row<TB>(A);
        }
        const bool in1 = !SAME && lastw;
        const int row = in1 ? (g_last - nrg0) * 8 + r8 : ((lastw ? g_last : b + wave * grid) * 8 + r8);
        const bamd_mv_seg & sg = a.seg[in1 ? 1 : 0];
        const int nv = sg.nvalid > 0 ? sg.nvalid : sg.nrows;
        if ((lane & 7) == 0 && row < nv) ik_st(sg.out + row, val);
        TL_STAMP(pa.tl, 5);
    }
}
template <int TA, int TB, int NBW, int MA, bool COMPACT = false, int NW = 8>
__global__ void __launch_bounds__(64 * NW) matvec_split_mixed_kernel(BAMD_LEAD_PARAMS, bamd_mv_args a) {
    BAMD_LEAD_TAKE(a);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    ActPro<true> ap;
    BAMD_PRO_ISSUE_NB(ap, pa, BAMD_NB1(NBW));
    float * part0 = (float *) (smem + BAMD_ACT_RED_OFF(nb) + 16 * sizeof(double));
    const int g_last = MA * (int) gridDim.x + (int) blockIdx.x;       // index of this workgroup's last row-group in the concatenated segments
    if (g_last < (a.seg[0].nrows >> 3)) split_mixed_body<TA, TA, NBW, MA, COMPACT>(a, pa, ap, part0, g_last);
    else                                split_mixed_body<TA, TB, NBW, MA, COMPACT>(a, pa, ap, part0, g_last);
    TL_STAMP(a.tl, 7);
}
template <int TA, int TB, int NBW, int MA, bool COMPACT = false, int NW = 8>
static void launch_mixed_inst(const bamd_mv_args & a, int grid, hipStream_t s) {
    const int nb = a.K >> 8;
    const size_t lds = act_lds_bytes(a.K) + 16 + (size_t) (MA + 1) * nb * (COMPACT ? 576 : 1024);
    BAMD_LAUNCH((matvec_split_mixed_kernel<TA, TB, NBW, MA, COMPACT, NW>), dim3(grid), dim3(64 * NW), lds, s, BAMD_LEAD_ARGS(a), a);
}
// fused QKV launch with two differently typed segments; false: shape not covered (mode A takes it)
bool bamd_launch_fast_mixed(const bamd_mv_args & a, int pro, int epi, int grid, hipStream_t s) {
    static const bool on = [] { const char * e = getenv("BAMD_MIXED_SPLIT"); return !(e && e[0] == '0'); }();
    if (!on || pro != BAMD_PRO_NORM || epi != BAMD_EPI_STORE || a.nseg != 2) return false;
    const int nb = a.K >> 8, t0 = a.seg[0].type, t1 = a.seg[1].type;
    const int nrg0 = a.seg[0].nrows >> 3, nrg1 = a.seg[1].nrows >> 3;
    if ((nb & 7) != 0 || nb > 8 * BAMD_ACT_BATCH || t0 == t1 || grid < 1 || (nrg0 + nrg1) % grid) return false;
    const int cnt = (nrg0 + nrg1) / grid, nbw = nb >> 3;
    if (cnt == 5 && nrg0 >= 4 * grid && nbw == 4) {                             // the Llama-3-70B shape: K = 8192, five row-groups per workgroup (four of wq | wk, the fifth wk or wv)
        // MEASURED AND NOT USED BY DEFAULT (round 4, MI355X, a 10-layer stage at the 70B widths, ms per token): one wave per row-group (mode A) 1.260, this
        // split-K instance on eight waves (20 records per wave in registers) 1.275, on sixteen waves (10 records, 128 VGPRs) 1.293 — the prologue, the
        // terms of a wave's records and a 32-step chain per workgroup cost more than five of eight waves streaming 32 records each.  BAMD_QKV70_WAVES=8 / 16
        // selects it for the record (tests/test_gpu_fullsize_ref.py::test_config4_70b_stage runs bit-exact through either).
        static const int nw70 = [] { const char * e = getenv("BAMD_QKV70_WAVES"); return e ? atoi(e) : 0; }();
        if (nw70 != 8 && nw70 != 16) return false;
#define BAMD_MX70(TA_, TB_) if (t0 == TA_ && t1 == TB_) { if (nw70 == 16) launch_mixed_inst<TA_, TB_, 2, 4, true, 16>(a, grid, s); else launch_mixed_inst<TA_, TB_, 4, 4, true>(a, grid, s); return true; }
        BAMD_MX70(BAMD_Q4_K, BAMD_Q6_K) BAMD_MX70(BAMD_Q4_K, BAMD_Q5_K)
#undef BAMD_MX70
        return false;
    }
    if (cnt != 3 || nrg0 < 2 * grid || nbw != 2) return false;                  // the Llama-3-8B / Mistral-7B shape: K = 4096, three row-groups per workgroup
#define BAMD_MX(TA_, TB_) if (t0 == TA_ && t1 == TB_) { launch_mixed_inst<TA_, TB_, 2, 2>(a, grid, s); return true; }
    BAMD_MX(BAMD_Q4_K, BAMD_Q6_K) BAMD_MX(BAMD_Q4_K, BAMD_Q5_K) BAMD_MX(BAMD_Q5_K, BAMD_Q6_K)
#undef BAMD_MX
    return false;
}

template <int PRO, int EPI, int T, int NBW, int M, bool ONEB = false, int NW = 8, bool COMPACT = false>
static void launch_fast_b_inst(const bamd_mv_args & a, int grid, hipStream_t s) {
    const int nb = a.K >> 8;
    const size_t lds = act_lds_bytes(a.K) + 16 + (COMPACT ? (size_t) 2 * M * nb * 576 : (size_t) (NBW * M > 8 ? 1 : 2) * M * nb * 256 * 4);
    BAMD_LAUNCH((matvec_split_fast_kernel<T, NBW, M, PRO, EPI, ONEB, NW, COMPACT>), dim3(grid), dim3(64 * NW), lds, s, BAMD_LEAD_ARGS(a), a);
}
static const bool g_down14 = [] { const char * e = getenv("BAMD_DOWN14"); return !(e && e[0] == '0'); }();
static const bool g_down112 = [] { const char * e = getenv("BAMD_DOWN112"); return !(e && e[0] == '0'); }();
// K = 28672 (112 super-blocks: the Llama-3-70B ffn_down) as split-K over SIXTEEN waves x seven records, one row-group per batch, two COMPACT term buffers.
// With one wave per row-group (mode A) this launch is bound by its prologue — every workgroup quantises all 112 blocks behind shared barriers: 22 us of the
// 30 (tools/mvbench.py 70b, `prologue-only K28672`) — and only four waves per CU have a row-group; here a wave quantises the seven blocks of its own K-slice.
bool bamd_launch_fast_b112_supported(int K, int pro, int epi, int nseg, int type) {
    return g_down112 && (K >> 8) == 112 && pro == BAMD_PRO_PLAIN && (epi == BAMD_EPI_STORE || epi == BAMD_EPI_ADD) && nseg == 1 &&
           (type == BAMD_Q4_K || type == BAMD_Q5_K || type == BAMD_Q6_K);
}
template <int PRO, int EPI>
static bool launch_fast_b_types(const bamd_mv_args & a, int t, int nbw, int grid, hipStream_t s) {
    // row-groups per batch: the largest M of the kernel family that every workgroup can fill (cnt_q = the smallest count)
    // (K = 14336 with both row-groups of a workgroup in flight — M = 2, 14 records per wave — measured no faster, again: the wave that
    // issues 130 KB of requests up front sits in the issue stage until most of them have landed, and its prologue starts that much later)
    if constexpr (PRO == BAMD_PRO_PLAIN) if ((a.K >> 8) == 112) {                               // bamd_launch_fast_b112_supported
        if (a.cnt_q < 1) return false;
#define BAMD_B112(T_) if (t == T_) { launch_fast_b_inst<PRO, EPI, T_, 7, 1, false, 16, true>(a, grid, s); return true; }
        BAMD_B112(BAMD_Q4_K) BAMD_B112(BAMD_Q5_K) BAMD_B112(BAMD_Q6_K)
#undef BAMD_B112
        return false;
    }
    if (g_down14 && PRO == BAMD_PRO_PLAIN && (a.K >> 8) == 56 && a.cnt_q >= 1) {                // K = 14336 on fourteen waves, four records each (one row-group per batch: 128 VGPRs)
#define BAMD_B14(T_) if (t == T_) { launch_fast_b_inst<PRO, EPI, T_, 4, 1, false, 14>(a, grid, s); return true; }
        BAMD_B14(BAMD_Q4_K) BAMD_B14(BAMD_Q5_K) BAMD_B14(BAMD_Q6_K)
#undef BAMD_B14
    }
    const int mmax = nbw == 1 ? 8 : nbw == 2 ? 4 : nbw == 4 ? 2 : 1;
    int m = 1; while (m * 2 <= mmax && m * 2 <= a.cnt_q) m *= 2;
    // K = 8192 with exactly FOUR row-groups per workgroup behind a plain prologue and a residual add (the 70B wo): all four in one batch — sixteen records per wave
    // requested at entry, one barrier, four chains side by side, one 128-KB term buffer — instead of two batches of two
    static const bool wo4 = [] { const char * e = getenv("BAMD_WO4"); return !(e && e[0] == '0'); }();
    if (wo4 && PRO == BAMD_PRO_PLAIN && EPI == BAMD_EPI_ADD && nbw == 4 && a.cnt_q == 4 && a.cnt_r == 0) m = 4;
    // exactly M row-groups in every workgroup: the single-batch instances (residual-add launches: wo, ffn_down)
    const bool oneb = a.cnt_r == 0 && a.cnt_q == m;
#define BAMD_B_ONE(T_, NBW_, M_) if (PRO == BAMD_PRO_PLAIN && EPI == BAMD_EPI_ADD && oneb && t == T_ && nbw == NBW_ && m == M_) { launch_fast_b_inst<PRO, EPI, T_, NBW_, M_, true>(a, grid, s); return true; }
#define BAMD_B_ONES(NBW_, M_) BAMD_B_ONE(BAMD_Q4_K, NBW_, M_) BAMD_B_ONE(BAMD_Q5_K, NBW_, M_) BAMD_B_ONE(BAMD_Q6_K, NBW_, M_)
    BAMD_B_ONES(2, 2) BAMD_B_ONES(2, 4) BAMD_B_ONES(4, 2) BAMD_B_ONES(4, 4) BAMD_B_ONES(1, 8)
#undef BAMD_B_ONES
#undef BAMD_B_ONE
#define BAMD_B_CASE(T_, NBW_, M_) if (t == T_ && nbw == NBW_ && m == M_) { launch_fast_b_inst<PRO, EPI, T_, NBW_, M_>(a, grid, s); return true; }
#define BAMD_B_TYPES(NBW_, M_) BAMD_B_CASE(BAMD_Q4_K, NBW_, M_) BAMD_B_CASE(BAMD_Q5_K, NBW_, M_) BAMD_B_CASE(BAMD_Q6_K, NBW_, M_)
    BAMD_B_TYPES(1, 8) BAMD_B_TYPES(1, 4) BAMD_B_TYPES(1, 2) BAMD_B_TYPES(1, 1)
    BAMD_B_TYPES(2, 4) BAMD_B_TYPES(2, 2) BAMD_B_TYPES(2, 1)
    BAMD_B_TYPES(4, 2) BAMD_B_TYPES(4, 1)
    if (PRO == BAMD_PRO_PLAIN) { BAMD_B_TYPES(7, 1) }
#undef BAMD_B_TYPES
#undef BAMD_B_CASE
    return false;
}
static const bool g_qkv3 = [] { const char * e = getenv("BAMD_QKV3"); return !(e && e[0] == '0'); }();
bool bamd_launch_fast_b(bamd_mv_args a, int pro, int epi, int grid, hipStream_t s) {
    const int nb = a.K >> 8;
    if ((nb & 7) != 0 || a.nseg != 1) return false;
    const int nrg = a.seg[0].nrows >> 3;
    a.cnt_q = nrg / grid; a.cnt_r = nrg % grid;
    const int t = a.seg[0].type, nbw = nb >> 3;
    // exactly THREE row-groups per workgroup at K = 4096 behind an RMSNorm prologue (the fused QKV launch of a layer whose wq | wk | wv are of one type, 8B / Mistral
    // widths): all three in ONE batch — six records per wave requested at entry, one barrier, three chains side by side — on the body of the mixed-type kernel with
    // both types equal, instead of two batches of two and one row-groups
    if (g_qkv3 && pro == BAMD_PRO_NORM && epi == BAMD_EPI_STORE && nbw == 2 && a.cnt_q == 3 && a.cnt_r == 0 && (a.mode & 15) == 0) {
#define BAMD_Q3(T_) if (t == T_) { launch_mixed_inst<T_, T_, 2, 2>(a, grid, s); return true; }
        BAMD_Q3(BAMD_Q4_K) BAMD_Q3(BAMD_Q5_K) BAMD_Q3(BAMD_Q6_K)
#undef BAMD_Q3
    }
    if (pro == BAMD_PRO_NORM) { if (epi == BAMD_EPI_STORE && nb <= 8 * BAMD_ACT_BATCH) return launch_fast_b_types<BAMD_PRO_NORM, BAMD_EPI_STORE>(a, t, nbw, grid, s); return false; }
    if (epi == BAMD_EPI_STORE) return launch_fast_b_types<BAMD_PRO_PLAIN, BAMD_EPI_STORE>(a, t, nbw, grid, s);
    if (epi == BAMD_EPI_ADD)   return launch_fast_b_types<BAMD_PRO_PLAIN, BAMD_EPI_ADD>(a, t, nbw, grid, s);
    return false;
}

// BAMD_MV_GENERIC=1: every launch on the generic kernels (A/B comparison, tests of the fallback)
