// bamd_matvec_fast_b.hip — fast mode-B (split-K) mat-vec kernels and their host-side dispatch.  See bamd_matvec_fast_a.hip for the
// rationale of the fast kernels.
#include "bamd_matvec_core.h"

template <int TYPE, int NBW, int M, int PRO, int EPI, bool ONEB>
__global__ void __launch_bounds__(512) matvec_split_fast_kernel(bamd_mv_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    ActPro<PRO == BAMD_PRO_NORM> ap, ap2;
    if (PRO == BAMD_PRO_PLAIN) {                             // own K-slice only (see split_stream)
        const int i0 = wave_id() * NBW;
        ap.issue(pa.x, pa.nw, pa.K, i0, 1, i0 + NBW);
        if (NBW > BAMD_ACT_BATCH) ap2.issue(pa.x, pa.nw, pa.K, i0 + BAMD_ACT_BATCH, 1, i0 + NBW);
    } else BAMD_PRO_ISSUE(ap, pa);
    float * part0 = (float *) (smem + BAMD_ACT_RED_OFF(nb) + 16 * sizeof(double));
    int rgctr = 0;
    const int count = a.cnt_q + ((int) blockIdx.x < a.cnt_r ? 1 : 0);
    const int nv = a.seg[0].nvalid > 0 ? a.seg[0].nvalid : a.seg[0].nrows;
    typedef typename RecOf<TYPE>::type REC;
    constexpr int NBUF = NBW * M > 8 ? 1 : 2;                // term buffers: 2 x M x K/256 KiB must fit the LDS
    split_stream<TYPE, REC, NBW, M, NBUF, EPI, PRO, true, ONEB>((const uint8_t *) a.seg[0].w, nb, (int) blockIdx.x, count, (int) gridDim.x, a.seg[0].out, a.res, pa,
                                                       ap, ap2, false, true, part0, rgctr, nv);      // the launcher's grid gives every workgroup >= 1 row-group
    TL_STAMP(a.tl, 7);
}


template <int PRO, int EPI, int T, int NBW, int M, bool ONEB = false>
static void launch_fast_b_inst(const bamd_mv_args & a, int grid, hipStream_t s) {
    const int nb = a.K >> 8;
    const size_t lds = act_lds_bytes(a.K) + 16 + (size_t) (NBW * M > 8 ? 1 : 2) * M * nb * 256 * 4;
    hipLaunchKernelGGL((matvec_split_fast_kernel<T, NBW, M, PRO, EPI, ONEB>), dim3(grid), dim3(512), lds, s, a);
}
template <int PRO, int EPI>
static bool launch_fast_b_types(const bamd_mv_args & a, int t, int nbw, int grid, hipStream_t s) {
    // row-groups per batch: the largest M of the kernel family that every workgroup can fill (cnt_q = the smallest count)
    // (K = 14336 with both row-groups of a workgroup in flight — M = 2, 14 records per wave — measured no faster, again: the wave that
    // issues 130 KB of requests up front sits in the issue stage until most of them have landed, and its prologue starts that much later)
    const int mmax = nbw == 1 ? 8 : nbw == 2 ? 4 : nbw == 4 ? 2 : 1;
    int m = 1; while (m * 2 <= mmax && m * 2 <= a.cnt_q) m *= 2;
    // exactly M row-groups in every workgroup: the single-batch instances (residual-add launches: wo, ffn_down)
    const bool oneb = a.cnt_r == 0 && a.cnt_q == m;
#define BAMD_B_ONE(T_, NBW_, M_) if (PRO == BAMD_PRO_PLAIN && EPI == BAMD_EPI_ADD && oneb && t == T_ && nbw == NBW_ && m == M_) { launch_fast_b_inst<PRO, EPI, T_, NBW_, M_, true>(a, grid, s); return true; }
#define BAMD_B_ONES(NBW_, M_) BAMD_B_ONE(BAMD_Q4_K, NBW_, M_) BAMD_B_ONE(BAMD_Q5_K, NBW_, M_) BAMD_B_ONE(BAMD_Q6_K, NBW_, M_)
    BAMD_B_ONES(2, 2) BAMD_B_ONES(2, 4) BAMD_B_ONES(4, 2) BAMD_B_ONES(1, 8)
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
bool bamd_launch_fast_b(bamd_mv_args a, int pro, int epi, int grid, hipStream_t s) {
    const int nb = a.K >> 8;
    if ((nb & 7) != 0 || a.nseg != 1) return false;
    const int nrg = a.seg[0].nrows >> 3;
    a.cnt_q = nrg / grid; a.cnt_r = nrg % grid;
    const int t = a.seg[0].type, nbw = nb >> 3;
    if (pro == BAMD_PRO_NORM) { if (epi == BAMD_EPI_STORE && nb <= 8 * BAMD_ACT_BATCH) return launch_fast_b_types<BAMD_PRO_NORM, BAMD_EPI_STORE>(a, t, nbw, grid, s); return false; }
    if (epi == BAMD_EPI_STORE) return launch_fast_b_types<BAMD_PRO_PLAIN, BAMD_EPI_STORE>(a, t, nbw, grid, s);
    if (epi == BAMD_EPI_ADD)   return launch_fast_b_types<BAMD_PRO_PLAIN, BAMD_EPI_ADD>(a, t, nbw, grid, s);
    return false;
}

// BAMD_MV_GENERIC=1: every launch on the generic kernels (A/B comparison, tests of the fallback)
