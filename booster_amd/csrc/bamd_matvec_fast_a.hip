// bamd_matvec_fast_a.hip — fast mode-A mat-vec kernels (one wave per row-group) and their host-side dispatch.
#include "bamd_matvec_core.h"
// ===========================================================================================================
// FAST KERNELS — one kernel per (weight type, launch shape): straight-line code from the first instruction to the streaming loop.
// The generic kernels above pick the weight type, the ring depth and the segment at run time inside ONE kernel; hipcc then has to
// merge register states at every join, which (a) put a full s_waitcnt behind each conditional activation load and held the weight
// ring back until the activations had arrived, (b) spilled scalar registers to vector lanes, and (c) made every launch walk through
// a 200 KB code object.  Here the dispatch happens on the host (bamd_launch_matvec): activation requests at entry, ring requests
// right behind them, counted waits all the way.  Same device functions (block_terms / chain_step / finish_row), same bits.
// Shapes outside the table (K/256 not a multiple of 8, three differently typed segments, ...) keep using the generic kernels.
// ===========================================================================================================

// mode A.  TYPE1 == 0: one segment (or the gate/up pair: seg[0] and seg[1] of TYPE0, EPI_SILU_MUL), any number of row-groups per wave
// (a.cnt_q / a.cnt_r = row-groups / wave slots, quotient and remainder).  TYPE1 != 0: two segments of different types with at most one
// row-group per wave (fused QKV with a Q6_K / Q5_K attn_v): the wave's row-group picks the branch, each branch is straight-line.
// NBP: activation batch slots per wave (K <= 256 * NBP * 8): 2 for K <= 4096, else 4
template <int TYPE0, int TYPE1, int PRO, int EPI, int NBP>
__global__ void __launch_bounds__(512) matvec_fast_kernel(BAMD_LEAD_PARAMS, bamd_mv_args a) {
    BAMD_LEAD_TAKE(a);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    ActPro<PRO == BAMD_PRO_NORM> ap;
    BAMD_PRO_ISSUE_NB(ap, pa, NBP);                          // activation requests: the first memory instructions of the kernel
    const int wave = wave_id(), nwaves = blockDim.x >> 6;
    const int slot = blockIdx.x + gridDim.x * wave;          // consecutive row-groups land on different CUs
    const int stride = gridDim.x * nwaves;
    unsigned long long best = 0ull;
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
    typedef typename RecOf<TYPE0>::type REC0;
    const int nrg0 = a.seg[0].nrows >> 3;
    const int nv0 = a.seg[0].nvalid > 0 ? a.seg[0].nvalid : a.seg[0].nrows;
    if (TYPE1 == 0) {
        const int count = a.cnt_q + (slot < a.cnt_r ? 1 : 0);
        const uint8_t * wA = (const uint8_t *) a.seg[0].w;
        const uint8_t * wB = PAIR ? (const uint8_t *) a.seg[1].w : wA;
        stream_segment<TYPE0, REC0, 8, EPI, PRO, true, NBP>(wA, wB, nb, slot, count, stride, a.seg[0].out, a.res, pa, ap, false, true, best, nv0);   // count == 0: prologue only
    } else {
        typedef typename RecOf<TYPE1 == 0 ? TYPE0 : TYPE1>::type REC1;
        constexpr int T1 = TYPE1 == 0 ? TYPE0 : TYPE1;
        const int nrg1 = a.seg[1].nrows >> 3;
        const int nv1 = a.seg[1].nvalid > 0 ? a.seg[1].nvalid : a.seg[1].nrows;
        if (slot >= nrg0 && slot < nrg0 + nrg1) {
            const uint8_t * w1 = (const uint8_t *) a.seg[1].w;
            stream_segment<T1, REC1, 8, EPI, PRO, true, NBP>(w1, w1, nb, slot - nrg0, 1, stride, a.seg[1].out, a.res, pa, ap, false, true, best, nv1);
        } else {                                             // segment 0, or no work (count 0: prologue only)
            const uint8_t * w0 = (const uint8_t *) a.seg[0].w;
            stream_segment<TYPE0, REC0, 8, EPI, PRO, true, NBP>(w0, w0, nb, slot, slot < nrg0 ? 1 : 0, stride, a.seg[0].out, a.res, pa, ap, false, true, best, nv0);
        }
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
    TL_STAMP(a.tl, 7);
}

// gate/up launch with seven row-group pairs per workgroup (stream_pair_short): waves 0-3 stream a whole pair each, waves 4-6 three quarters of
// theirs, wave 7 the last quarters of those three pairs.  K = 4096 (16 super-blocks), n_ff = 56 x grid.
template <int TYPE, int NBP>
__global__ void __launch_bounds__(512) matvec_gateup7_kernel(BAMD_LEAD_PARAMS, bamd_mv_args a) {
    BAMD_LEAD_TAKE(a);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    ActPro<true> ap;
    BAMD_PRO_ISSUE_NB(ap, pa, NBP);
    const int wave = wave_id(), grid = (int) gridDim.x, b = (int) blockIdx.x;
    float4 * park = (float4 *) (smem + BAMD_ACT_RED_OFF(nb) + 16 * sizeof(double) + 16 * sizeof(unsigned long long));
    int * flags = (int *) (smem + BAMD_ACT_RED_OFF(nb) + 16 * sizeof(double) + 16 * sizeof(unsigned long long) + BAMD_GU7_PARK_BYTES(16));
    if (threadIdx.x < 4) flags[threadIdx.x] = 0;             // ordered before their first use by the prologue's workgroup barriers
    typedef typename RecOf<TYPE>::type REC;
    const uint8_t * wG = (const uint8_t *) a.seg[0].w, * wU = (const uint8_t *) a.seg[1].w;
    const int nv = a.seg[0].nvalid > 0 ? a.seg[0].nvalid : a.seg[0].nrows;
    unsigned long long best = 0ull;
    if (wave < 4) stream_segment<TYPE, REC, 8, BAMD_EPI_SILU_MUL, BAMD_PRO_NORM, true, NBP>(wG, wU, nb, b + grid * wave, 1, grid * 8, a.seg[0].out, a.res, pa, ap, false, true, best, nv);
    else if (wave < 7) stream_pair_short<TYPE, REC, NBP, false>(wG, wU, b + grid * wave, grid, wave - 4, a.seg[0].out, pa, ap, park, flags, nv);
    else stream_pair_short<TYPE, REC, NBP, true>(wG, wU, b + grid * 4, grid, 0, a.seg[0].out, pa, ap, park, flags, nv);
    TL_STAMP(a.tl, 7);
}
static const bool g_gateup7 = [] { const char * e = getenv("BAMD_GATEUP7"); return !(e && e[0] == '0'); }();

// gate/up launch with FOURTEEN row-group pairs per workgroup (n_ff = 112 x grid, K = 8192: Llama-3-70B on 256 CUs).  One pair or two per wave (the generic
// dealing: waves 0-5 two, waves 6-7 one) put four pairs on SIMDs 0 and 1 and three on SIMDs 2 and 3, and the launch ended with waves 4 and 5 alone
// (timeline, round 6: waves 0-3 exit at 37.5 us, waves 6-7 at 30, waves 4-5 at 47).  Here every SIMD streams SEVEN rows: waves 0-3 two pairs each
// (pairs w, w + 4), waves 4-7 one pair (8 + w - 4) and then HALF of a pair — waves 4 / 6 the gate row of pairs 12 / 13, waves 5 / 7 the up row, the gate
// value crossing to the up row's wave through LDS for silu(gate) * up.  Whole rows only: every chain is one wave's sequential chain, as everywhere.
template <int TYPE, int NBP>
__global__ void __launch_bounds__(512) matvec_gateup14_kernel(BAMD_LEAD_PARAMS, bamd_mv_args a) {
    BAMD_LEAD_TAKE(a);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    ActPro<true> ap;
    BAMD_PRO_ISSUE_NB(ap, pa, NBP);
    const int wave = wave_id(), grid = (int) gridDim.x, b = (int) blockIdx.x;
    float * slots = (float *) (smem + BAMD_ACT_RED_OFF(nb) + 16 * sizeof(double) + 16 * sizeof(unsigned long long));     // [2 half pairs][8 rows]
    int * flags = (int *) (slots + 16);
    if (threadIdx.x < 2) flags[threadIdx.x] = 0;             // ordered before their first use by the prologue's workgroup barriers
    typedef typename RecOf<TYPE>::type REC;
    const uint8_t * wG = (const uint8_t *) a.seg[0].w, * wU = (const uint8_t *) a.seg[1].w;
    const int nv = a.seg[0].nvalid > 0 ? a.seg[0].nvalid : a.seg[0].nrows;
    unsigned long long best = 0ull;
    if (wave < 4) stream_segment<TYPE, REC, 8, BAMD_EPI_SILU_MUL, BAMD_PRO_NORM, true, NBP>(wG, wU, nb, b + grid * wave, 2, grid * 4, a.seg[0].out, a.res, pa, ap, false, true, best, nv);
    else {
        stream_segment<TYPE, REC, 8, BAMD_EPI_SILU_MUL, BAMD_PRO_NORM, true, NBP>(wG, wU, nb, b + grid * (4 + wave), 1, grid, a.seg[0].out, a.res, pa, ap, false, true, best, nv);
        const int h = (wave - 4) >> 1;                       // half pair 0 (pair 12): waves 4, 5; half pair 1 (pair 13): waves 6, 7
        const int rg = b + grid * (12 + h);
        if (((wave - 4) & 1) == 0) stream_segment<TYPE, REC, 8, BAMD_EPI_HALF_GATE, BAMD_PRO_NORM, true, NBP>(wG, wG, nb, rg, 1, grid, a.seg[0].out, a.res, pa, ap, false, false, best, nv, slots + 8 * h, flags + h);
        else                       stream_segment<TYPE, REC, 8, BAMD_EPI_HALF_UP, BAMD_PRO_NORM, true, NBP>(wU, wU, nb, rg, 1, grid, a.seg[0].out, a.res, pa, ap, false, false, best, nv, slots + 8 * h, flags + h);
    }
    TL_STAMP(a.tl, 7);
}
static const bool g_gateup14 = [] { const char * e = getenv("BAMD_GATEUP14"); return !(e && e[0] == '0'); }();

// mode B (split-K), one segment of one type, NBW = K / 2048 records per wave and row-group, M row-groups per batch

// ---- host-side dispatch of the fast kernels; false = no instance for this shape (the caller takes the generic kernel) ----
template <int PRO, int EPI, int T0, int T1>
static void launch_fast_a_inst(const bamd_mv_args & a, int grid, hipStream_t s) {
    if ((a.K >> 8) <= 16) BAMD_LAUNCH((matvec_fast_kernel<T0, T1, PRO, EPI, 2>), dim3(grid), dim3(512), act_lds_bytes(a.K), s, BAMD_LEAD_ARGS(a), a);
    else                  BAMD_LAUNCH((matvec_fast_kernel<T0, T1, PRO, EPI, 4>), dim3(grid), dim3(512), act_lds_bytes(a.K), s, BAMD_LEAD_ARGS(a), a);
}
template <int PRO, int EPI>
static bool launch_fast_a_types(const bamd_mv_args & a, int t0, int t1, int grid, hipStream_t s) {
    constexpr bool MIX = PRO == BAMD_PRO_NORM && EPI == BAMD_EPI_STORE;      // two differently typed segments: the fused QKV launch only
#define BAMD_A_CASE(T0_, T1_) if (t0 == T0_ && t1 == T1_) { launch_fast_a_inst<PRO, EPI, T0_, T1_>(a, grid, s); return true; }
    BAMD_A_CASE(BAMD_Q4_K, 0) BAMD_A_CASE(BAMD_Q5_K, 0) BAMD_A_CASE(BAMD_Q6_K, 0)
    if (MIX) {
        BAMD_A_CASE(BAMD_Q4_K, BAMD_Q5_K) BAMD_A_CASE(BAMD_Q4_K, BAMD_Q6_K) BAMD_A_CASE(BAMD_Q5_K, BAMD_Q4_K)
        BAMD_A_CASE(BAMD_Q5_K, BAMD_Q6_K) BAMD_A_CASE(BAMD_Q6_K, BAMD_Q4_K) BAMD_A_CASE(BAMD_Q6_K, BAMD_Q5_K)
    }
#undef BAMD_A_CASE
    return false;
}
bool bamd_launch_fast_a(bamd_mv_args a, int pro, int epi, int grid, hipStream_t s) {
    const int nb = a.K >> 8;
    if ((nb & 7) != 0 || nb < 8 || nb > 8 * BAMD_ACT_BATCH) return false;      // SMALLK prologue: K <= 8192
    const int slots = grid * 8;
    int t0 = a.seg[0].type, t1 = 0;
    const int nrg0 = a.seg[0].nrows >> 3;
    if (epi == BAMD_EPI_SILU_MUL) { if (a.nseg != 2 || a.seg[1].type != t0 || a.seg[1].nrows != a.seg[0].nrows) return false; }
    else if (a.nseg == 2) {
        t1 = a.seg[1].type;
        if (t1 == t0 || nrg0 + (a.seg[1].nrows >> 3) > slots) return false;
    } else if (a.nseg != 1) return false;
    a.cnt_q = nrg0 / slots; a.cnt_r = nrg0 % slots;
    if (g_gateup7 && pro == BAMD_PRO_NORM && epi == BAMD_EPI_SILU_MUL && nb == 16 && nrg0 == 7 * grid && (a.mode & 15) == 0) {
        const size_t lds = act_lds_bytes(a.K) + BAMD_GU7_PARK_BYTES(16) + 16;
#define BAMD_G7(T_) if (t0 == T_) { BAMD_LAUNCH((matvec_gateup7_kernel<T_, 2>), dim3(grid), dim3(512), lds, s, BAMD_LEAD_ARGS(a), a); return true; }
        BAMD_G7(BAMD_Q4_K) BAMD_G7(BAMD_Q5_K) BAMD_G7(BAMD_Q6_K)
#undef BAMD_G7
    }
    if (g_gateup14 && pro == BAMD_PRO_NORM && epi == BAMD_EPI_SILU_MUL && nb == 32 && nrg0 == 14 * grid && (a.mode & 15) == 0) {
        const size_t lds = act_lds_bytes(a.K) + 16 * 4 + 2 * 4 + 8;
#define BAMD_G14(T_) if (t0 == T_) { BAMD_LAUNCH((matvec_gateup14_kernel<T_, 4>), dim3(grid), dim3(512), lds, s, BAMD_LEAD_ARGS(a), a); return true; }
        BAMD_G14(BAMD_Q4_K) BAMD_G14(BAMD_Q5_K) BAMD_G14(BAMD_Q6_K)
#undef BAMD_G14
    }
    if (pro == BAMD_PRO_NORM) {
        if (epi == BAMD_EPI_STORE)    return launch_fast_a_types<BAMD_PRO_NORM, BAMD_EPI_STORE>(a, t0, t1, grid, s);
        if (epi == BAMD_EPI_SILU_MUL) return launch_fast_a_types<BAMD_PRO_NORM, BAMD_EPI_SILU_MUL>(a, t0, 0, grid, s);
        if (epi == BAMD_EPI_ARGMAX)   return launch_fast_a_types<BAMD_PRO_NORM, BAMD_EPI_ARGMAX>(a, t0, t1, grid, s);
        return false;
    }
    if (t1 != 0) return false;
    if (epi == BAMD_EPI_STORE) return launch_fast_a_types<BAMD_PRO_PLAIN, BAMD_EPI_STORE>(a, t0, 0, grid, s);
    if (epi == BAMD_EPI_ADD)   return launch_fast_a_types<BAMD_PRO_PLAIN, BAMD_EPI_ADD>(a, t0, 0, grid, s);
    return false;
}
