// bamd_matvec_core.h — device code shared by the mat-vec translation units (bamd_matvec.hip: generic kernels and launchers;
// bamd_matvec_fast_a.hip / bamd_matvec_fast_b.hip: the host-dispatched fast kernels): the mode-A streaming loop and the mode-B
// split-K loop.  Numerics contract and reference citations: bamd_device.h.
#pragma once
#include "bamd_device.h"

template <int TYPE> struct RecOf;
template <> struct RecOf<BAMD_Q4_K> { typedef RecQ4K type; };
template <> struct RecOf<BAMD_Q5_K> { typedef RecQ5K type; };
template <> struct RecOf<BAMD_Q6_K> { typedef RecQ6K type; };

// The fields a launch needs for its FIRST requests travel as leading scalar kernel parameters: with -amdgpu-kernarg-preload-count the
// hardware places them in SGPRs at wave launch (gfx950 kernarg preload), so the activation and ring requests do not wait for an s_load of
// the argument block; the struct behind them carries everything else.
// activation batch slots of a wave that takes NBW_ blocks of the vector: the first ActPro holds up to BAMD_ACT_BATCH, a second one the rest
#define BAMD_NB1(NBW_) ((NBW_) < BAMD_ACT_BATCH ? (NBW_) : BAMD_ACT_BATCH)
#define BAMD_NB2(NBW_) ((NBW_) > BAMD_ACT_BATCH ? (NBW_) - BAMD_ACT_BATCH : 1)
#define BAMD_LEAD_PARAMS const float * x_, const float * nw_, const void * w0_, const void * w1_, int K_, float eps_
#define BAMD_LEAD_TAKE(a_) do { (a_).x = x_; (a_).normw = nw_; (a_).seg[0].w = w0_; (a_).seg[1].w = w1_; (a_).K = K_; (a_).eps = eps_; } while (0)
#define BAMD_LEAD_ARGS(a_) (a_).x, (a_).normw, (a_).seg[0].w, (a_).seg[1].w, (a_).K, (a_).eps
__device__ __forceinline__ ProArgs carve_lds(const bamd_mv_args & a, unsigned char * smem) {
    const int nb = a.K >> 8;
    ProArgs pa;
    pa.x = a.x; pa.nw = a.normw; pa.eps = a.eps; pa.K = a.K;
    pa.q8 = (uint32_t *) smem; pa.S = (int *) (pa.q8 + nb * 64); pa.yd = (float *) (pa.S + nb * 8);
    pa.tl = a.tl;
    pa.red = (double *) (smem + BAMD_ACT_RED_OFF(nb));     // byte offsets, never a pointer->integer->pointer round trip: that loses
                                                           // the LDS address space and turns every access into a FLAT instruction
    return pa;
}


// device-internal epilogues of stream_segment (matvec_gateup14_kernel): HALF a gate/up pair per wave — the gate row's value goes to the partner wave
// through LDS (`res` = the slot of 8 floats, `out` reinterpreted: see the kernel), the up row's wave waits for it and stores silu(gate) * up
#define BAMD_EPI_HALF_GATE 16
#define BAMD_EPI_HALF_UP 17
// ---- MODE A: one wave per row-group --------------------------------------------------------------------------
// The wave walks row-groups rg = first, first+stride, ... (count of them).  A register ring of D records is kept
// in flight by a LOADER cursor that runs D records ahead of the consumer and crosses row-group boundaries by
// pure (branch-free, scalar) arithmetic, so the prefetch never drains and the compiler can keep counted
// s_waitcnt vmcnt(N) waits.  The ring is filled BEFORE the activation prologue (weights do not depend on it), so
// the first HBM round trip overlaps the RMSNorm/Q8_K work.  With PAIR each row-group is streamed twice back to
// back — gate (wA) then up (wB) — and the epilogue fuses silu(gate)*up.
template <int TYPE, typename REC, int D, int EPI, int PRO, bool SMALLK = false, int NBP = BAMD_ACT_BATCH>
__device__ __forceinline__ void stream_segment(const uint8_t * __restrict__ wA, const uint8_t * __restrict__ wB, int nb,
                                               int first, int count, int stride, float * __restrict__ out,
                                               const float * __restrict__ res, const ProArgs & pa, ActPro<PRO == BAMD_PRO_NORM> & ap,
                                               bool issue_here, bool do_pro, unsigned long long & best, int nvalid, float * half_slot = nullptr, int * half_flag = nullptr) {
    constexpr int RECB = TYPE == BAMD_Q4_K ? BAMD_RECB_Q4K : TYPE == BAMD_Q5_K ? BAMD_RECB_Q5K : 1680;     // bamd_record_bytes
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
    constexpr int NPARTS = PAIR ? 2 : 1;
    const int lane = threadIdx.x & 63;
    // records are addressed as (matrix descriptor, 32-bit byte offset): see load_rec.  A matrix stays below 2 GiB (launcher).
    const bamd_rsrc rsA = weight_rsrc(wA), rsB = PAIR ? weight_rsrc(wB) : rsA, rsN = null_rsrc(wA);
    const int rgb = nb * RECB;                           // D divides nb (chosen by the dispatcher below)
    const int rg_step = stride * rgb;
    const int chunks = nb / D;
    if (issue_here) BAMD_PRO_ISSUE(ap, pa);              // activation loads go out FIRST (see ActPro::issue); the fast kernels issue them at entry
    REC ring[D];
    // The loader runs exactly one CHUNK (D records = the whole ring) ahead of the consumer: slot s is refilled, right after it
    // is consumed, with record s of the chunk that follows in this wave's sequence (next chunk of the row, else the other half
    // of a gate/up pair, else the next row-group).  One wave-uniform base address per chunk: the per-record cost of the cursor
    // is a constant offset, and the loads stay unconditional so the compiler keeps counted s_waitcnt vmcnt(N) waits.
    // (a wave without work — count == 0, fast kernels only — requests record 0 of the matrix D times: L1 hits, and its code path stays
    // the one of the busy waves: one copy of the prologue, no join in front of the counted waits)
    const int offA = count > 0 ? first * rgb : 0;
    const int fill_step = count > 0 ? RECB : 0;
#ifndef BAMD_RING_SPLIT
#define BAMD_RING_SPLIT 1            /* fast mode-A kernels: the second half of the ring is requested behind the prologue's first barrier */
#endif
    constexpr bool RSPLIT = BAMD_RING_SPLIT && SMALLK && PRO == BAMD_PRO_NORM;
#pragma unroll
    for (int s = 0; s < (RSPLIT ? D / 2 : D); ++s) load_rec(ring[s], rsA, offA + s * fill_step, lane);
    TL_STAMP(pa.tl, 1);
    if (do_pro) {
        if (RSPLIT) {
            auto second_half = [&]() {
#pragma unroll
                for (int s = D / 2; s < D; ++s) load_rec(ring[s], rsA, offA + s * fill_step, lane);
            };
            BAMD_PRO_FINISH_NB_MID(ap, pa, second_half, NBP);
        } else if (SMALLK) { auto nm = []() { }; BAMD_PRO_FINISH_NB_MID(ap, pa, nm, NBP); }
        else BAMD_PRO_FINISH(ap, pa);
    } else if (RSPLIT) {
#pragma unroll
        for (int s = D / 2; s < D; ++s) load_rec(ring[s], rsA, offA + s * fill_step, lane);
    }
    TL_STAMP(pa.tl, 2);
    const uint32_t * q8 = pa.q8; const int * S = pa.S; const float * yd = pa.yd;
    for (int r = 0; r < count; ++r) {
        const int rg = first + r * stride;
        const int row = rg * 8 + (lane >> 3);
        const int rowoff = rg * rgb;
        float gate_val = 0.f;
#pragma unroll
        for (int part = 0; part < NPARTS; ++part) {
            // after the last chunk of this row-part: the other half of the pair, the next row-group, or — at the very end of the
            // wave's stream — D requests through the zero-record descriptor (null_rsrc: they return 0 and fetch nothing; rounds 1-5
            // re-requested the wave's last record, one record of redundant traffic; the requests stay unconditional so that the
            // waits stay counted)
            const bool last = !(PAIR && part == 0) && r + 1 >= count;
            const bool after_b = (PAIR && part == 0) || (last && part == 1);
            const int after_off = (PAIR && part == 0) ? rowoff : (last ? rowoff + (nb - 1) * RECB : rowoff + rg_step);
            // residual fetched at the START of the row: by the epilogue it is the oldest outstanding load
            float resv = 0.f;
            if (EPI == BAMD_EPI_ADD && row < nvalid) resv = ik_ld(res + row);
            RowAcc A = { 0.f, 0.f };
            for (int c = 0; c < chunks; ++c) {
                const bool inrow = c + 1 < chunks;
                const bool tail = !inrow && last;                // the requests behind this wave's last chunk: never consumed — the zero-record descriptor
                const bamd_rsrc nrs = tail ? rsN : (inrow ? part == 1 : after_b) ? rsB : rsA;
                const int nxt = inrow ? rowoff + (c + 1) * (D * RECB) : after_off;
                const int step = (inrow || !last) ? RECB : 0;
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    pin_rec(ring[s]);
                    const Terms T = block_terms(ring[s], c * D + s, lane, q8, S, yd);
                    chain_step<TYPE>(A, T.d, T.fs, T.dmin, T.pm);
                    load_rec(ring[s], nrs, nxt + s * step, lane);
                    if ((s & (BAMD_SCHED_GROUP - 1)) == BAMD_SCHED_GROUP - 1)
                        __builtin_amdgcn_sched_barrier(0);   // keep hipcc from clustering the refills at the loop tail
                }
                if (r == 0 && part == 0 && c == 0) TL_STAMP(pa.tl, 3);
            }
            if (r + 1 == count && part == NPARTS - 1) TL_STAMP(pa.tl, 4);
            const float val = finish_row<TYPE>(A);
            if (PAIR) {
                if (part == 0) gate_val = val;
                else if ((lane & 7) == 0 && row < nvalid) ik_st(out + row, v_silu(gate_val) * val);
            } else if (EPI == BAMD_EPI_HALF_GATE) {              // the gate row of a pair whose up row another wave streams: hand the value over through LDS
                if ((lane & 7) == 0) half_slot[lane >> 3] = val;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(half_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (EPI == BAMD_EPI_HALF_UP) {
                while (__hip_atomic_load(half_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const float g = half_slot[lane >> 3];
                if ((lane & 7) == 0 && row < nvalid) ik_st(out + row, v_silu(g) * val);
            } else if ((lane & 7) == 0 && row < nvalid) {
                float o = val;
                if (EPI == BAMD_EPI_ADD) o = val + resv;
                ik_st(out + row, o);
                if (EPI == BAMD_EPI_ARGMAX) { const unsigned long long k = argmax_key(o, row); best = k > best ? k : best; }
            }
        }
    }
}


// ---- MODE A, gate/up launch with exactly SEVEN row-group pairs per 8-wave workgroup (n_ff = 7 x 8 x 8 x CUs: Llama-3-8B / Mistral-7B, 14336 rows on
// 256 CUs).  One pair per wave leaves wave 7 idle: SIMD 3 streams 32 records where SIMDs 0-2 stream 64, and the launch is paced by the younger
// waves 4-6 (timeline, round 2: waves 0-3 exit at 11.4 us, waves 4-6 at 13.8, wave 7 at 3.4).  Here waves 4-6 stop three quarters into their
// gate row and their up row (super-blocks 0..CUT-1, CUT = 3 nb / 4), and wave 7 computes the TERMS {d, fs, dmin, pm} of the last quarter of those six rows
// and parks them in LDS (the split-K kernels' mechanism); waves 4-6 then replay them IN ORDER behind their own chain steps: every SIMD streams
// 56 records, and each lane's f32 chain is still the reference's sequential chain over super-blocks 0..nb-1 (ggml-quants.c:6937-6941, :6970).
// HELPER = wave 7; otherwise wave 4 + j of the workgroup (its pair: row-group rg0).  LDS: park[3 pairs][gate | up][nb / 4][64 lanes] float4, flags[3].
#define BAMD_GU7_PARK_BYTES(nb) ((size_t) 3 * 2 * ((nb) / 4) * 64 * 16)
template <int TYPE, typename REC, int NBP, bool HELPER>
__device__ __forceinline__ void stream_pair_short(const uint8_t * __restrict__ wG, const uint8_t * __restrict__ wU, int rg0, int rg_stride, int j,
                                                  float * __restrict__ out, const ProArgs & pa, ActPro<true> & ap, float4 * park, int * flags, int nvalid) {
    constexpr int RECB = TYPE == BAMD_Q4_K ? BAMD_RECB_Q4K : TYPE == BAMD_Q5_K ? BAMD_RECB_Q5K : 1680;
    constexpr int NB = 16, Q = NB / 4, CUT = NB - Q, D = 8, NREC = HELPER ? 3 * 2 * Q : 2 * CUT;      // 24 records either way
    static_assert(NREC % D == 0, "whole ring chunks");
    const int lane = threadIdx.x & 63;
    const bamd_rsrc rsG = weight_rsrc(wG), rsU = weight_rsrc(wU);
    const int rgb = NB * RECB;
    // record i of this wave's sequence (compile-time i): which matrix, which row-group, which super-block
#define BAMD_GU7_UP(i_)  (HELPER ? (((i_) % (2 * Q)) / Q) == 1 : ((i_) / CUT) == 1)
#define BAMD_GU7_SB(i_)  (HELPER ? CUT + ((i_) % Q) : (i_) % CUT)
#define BAMD_GU7_OFF(i_) ((HELPER ? rg0 + ((i_) / (2 * Q)) * rg_stride : rg0) * rgb + BAMD_GU7_SB(i_) * RECB)
    REC ring[D];
#pragma unroll
    for (int s = 0; s < D / 2; ++s) load_rec(ring[s], BAMD_GU7_UP(s) ? rsU : rsG, BAMD_GU7_OFF(s), lane);
    TL_STAMP(pa.tl, 1);
    auto second_half = [&]() {
#pragma unroll
        for (int s = D / 2; s < D; ++s) load_rec(ring[s], BAMD_GU7_UP(s) ? rsU : rsG, BAMD_GU7_OFF(s), lane);
    };
    BAMD_PRO_FINISH_NB_MID(ap, pa, second_half, NBP);
    TL_STAMP(pa.tl, 2);
    const uint32_t * q8 = pa.q8; const int * S = pa.S; const float * yd = pa.yd;
    RowAcc Ag = { 0.f, 0.f }, Au = { 0.f, 0.f };
#pragma unroll
    for (int c = 0; c < NREC / D; ++c) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int i = c * D + s;
            pin_rec(ring[s]);
            const Terms T = block_terms(ring[s], BAMD_GU7_SB(i), lane, q8, S, yd);
            if (HELPER) park[((i / (2 * Q)) * 2 + (BAMD_GU7_UP(i) ? 1 : 0)) * Q * 64 + (i % Q) * 64 + lane] = make_float4(T.d, T.fs, T.dmin, T.pm);
            else if (BAMD_GU7_UP(i)) chain_step<TYPE>(Au, T.d, T.fs, T.dmin, T.pm);
            else chain_step<TYPE>(Ag, T.d, T.fs, T.dmin, T.pm);
            if (i + D < NREC) load_rec(ring[s], BAMD_GU7_UP(i + D) ? rsU : rsG, BAMD_GU7_OFF(i + D), lane);
            if (HELPER && (i % (2 * Q)) == 2 * Q - 1) {            // the six quarter-rows of one pair are parked: tell its wave
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(flags + i / (2 * Q), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if ((s & (BAMD_SCHED_GROUP - 1)) == BAMD_SCHED_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (c == 0) TL_STAMP(pa.tl, 3);
    }
    TL_STAMP(pa.tl, 4);
    if (!HELPER) {
        while (__hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const float4 * Pg = park + (j * 2 + 0) * Q * 64 + lane, * Pu = park + (j * 2 + 1) * Q * 64 + lane;
        float4 tg[Q], tu[Q];
#pragma unroll
        for (int u = 0; u < Q; ++u) { tg[u] = Pg[u * 64]; tu[u] = Pu[u * 64]; }
#pragma unroll
        for (int u = 0; u < Q; ++u) chain_step<TYPE>(Ag, tg[u].x, tg[u].y, tg[u].z, tg[u].w);
#pragma unroll
        for (int u = 0; u < Q; ++u) chain_step<TYPE>(Au, tu[u].x, tu[u].y, tu[u].z, tu[u].w);
        const float gate_val = finish_row<TYPE>(Ag), up_val = finish_row<TYPE>(Au);
        const int row = rg0 * 8 + (lane >> 3);
        if ((lane & 7) == 0 && row < nvalid) ik_st(out + row, v_silu(gate_val) * up_val);
    }
#undef BAMD_GU7_UP
#undef BAMD_GU7_SB
#undef BAMD_GU7_OFF
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

// ONEB (fast kernels): every workgroup has exactly M row-groups — one batch, no refills, no loop: the waits for the ring stay counted
// (record by record) instead of one full wait at the loop head
// UNEVEN: K / 256 is not a multiple of 8 (Llama-2's 11008 = 43, 13824 = 54, 5120 = 20 super-blocks): the first nb % 8 waves take NBW records of
// a row-group, the others NBW - 1; a wave's ring slot past its share re-requests its last record and its term is not parked (wave-uniform).
// PRE: called once between the first ring requests and the activation prologue (default: nothing).  The co-launched attention || wo kernel waits
// there for its activations (granules of the attention role, bamd_colaunch.hip) and fills ap.v itself: the weights of the first batch are in
// flight while it waits.
// COMPACT (K = 28672, the 70B ffn_down: 112 super-blocks): a parked record takes 576 bytes — {fs, pm} per lane, {d, dmin} once per row — instead of a float4 per
// lane, so that TWO term buffers of a whole row-group fit the LDS beside the activations (2 x 63 KB + 32 KB) and the chain of row-group n overlaps the
// streaming of row-group n + 1 (with one buffer every batch ends in barrier + 112-step chain + barrier: measured slower than one wave per row-group, round 5)
struct SplitNoPre { __device__ __forceinline__ void operator()() const { } };
template <int TYPE, typename REC, int NBW, int M, int NBUF, int EPI, int PRO, bool SMALLK = false, bool ONEB = false, bool UNEVEN = false, typename PRE = SplitNoPre, bool COMPACT = false>
__device__ __forceinline__ void split_stream(const uint8_t * __restrict__ w, int nb, int first, int count, int stride,
                                             float * __restrict__ out, const float * __restrict__ res, const ProArgs & pa,
                                             ActPro<PRO == BAMD_PRO_NORM> & ap, ActPro<PRO == BAMD_PRO_NORM> & ap2, bool issue_here, bool do_pro,
                                             float * part0, int & batchctr, int nvalid, PRE pre = PRE()) {
    constexpr int RECB = TYPE == BAMD_Q4_K ? BAMD_RECB_Q4K : TYPE == BAMD_Q5_K ? BAMD_RECB_Q5K : 1680;     // bamd_record_bytes
    constexpr int D = NBW * M;                               // ring depth = one batch (M row-groups) of this wave's records
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int r8 = lane >> 3, l4 = lane & 3;
    const bamd_rsrc rs = weight_rsrc(w), rsn = null_rsrc(w);
    const int rgb = nb * RECB;
    const int rg_step = stride * rgb;
    const int n_w = UNEVEN ? (NBW - 1) + (wave < (nb & 7) ? 1 : 0) : NBW;                  // this wave's records per row-group
    const int i0 = UNEVEN ? wave * (NBW - 1) + (wave < (nb & 7) ? wave : (nb & 7)) : wave * NBW;   // its first super-block inside a row
    const size_t rg_floats = COMPACT ? (size_t) nb * 144 : BAMD_TERM_FLOATS(nb);
    // PLAIN prologue: wave w consumes only the activations of its own K-slice (blocks i0 .. i0+NBW-1), so it quantises exactly
    // those — no workgroup barrier, and a wave starts on its records as soon as ITS blocks are done.  (NORM needs the sum of
    // squares of the whole vector: shared prologue as in mode A.)
    constexpr bool OWN = PRO == BAMD_PRO_PLAIN;
    if (issue_here) {                                        // (the fast kernels issue these at entry)
        if (OWN) { ap.template issue<BAMD_NB1(NBW)>(pa.x, pa.nw, pa.K, i0, 1, i0 + n_w); if (NBW > BAMD_ACT_BATCH) ap2.template issue<BAMD_NB2(NBW)>(pa.x, pa.nw, pa.K, i0 + BAMD_ACT_BATCH, 1, i0 + n_w); }
        else BAMD_PRO_ISSUE(ap, pa);                         // activation loads go out FIRST
    }
    // ring slot (m, j) holds record i0+j of row-group r0+m; after it is consumed it is refilled with the same record of row-group
    // r0+M+m, i.e. a constant M*rg_step further on: the loader needs one wave-uniform base per batch and nothing per record
    int bbase = first * rgb + i0 * RECB;
    REC ring[D];
#ifndef BAMD_RING_SPLIT_B
#define BAMD_RING_SPLIT_B 1          /* fast split-K kernels: the second half of the ring is requested behind (inside) the activation prologue */
#endif
#ifndef BAMD_RING_SPLIT_MIN
#define BAMD_RING_SPLIT_MIN 6        /* ... for rings of at least this many records per wave */
#endif
    constexpr bool RSPLIT = BAMD_RING_SPLIT_B && SMALLK && !UNEVEN && D >= BAMD_RING_SPLIT_MIN;
    constexpr int DH = RSPLIT ? D / 2 : D;                   // slots requested before the prologue
    auto ring_fill = [&](const int s0, const int s1) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            // generic kernels: no redundant requests when the stream is short.  Fast kernels (SMALLK): the launcher picks M <= the row-groups of
            // every workgroup, so the requests are unconditional — a branch around them costs a full s_waitcnt at the join, i.e. the
            // activation prologue would wait for the whole ring to land before it starts
            if (SMALLK || m < count) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const int sl = m * NBW + j;
                    if (sl >= s0 && sl < s1) load_rec(ring[sl], rs, bbase + m * rg_step + (UNEVEN && j >= n_w ? n_w - 1 : j) * RECB, lane);
                }
            }
        }
    };
    ring_fill(0, DH);
    TL_STAMP(pa.tl, 1);
    pre();
    if (do_pro) {
        if (OWN) {
            static_assert(NBW <= 2 * BAMD_ACT_BATCH, "own-slice prologue handles two batches");
            ap.template quantize_batch<BAMD_NB1(NBW)>(1.0f, pa.K, i0, pa.q8, pa.S, pa.yd, 1, i0 + n_w);
            if (RSPLIT) ring_fill(DH, D);
            if (NBW > BAMD_ACT_BATCH) ap2.template quantize_batch<BAMD_NB2(NBW)>(1.0f, pa.K, i0 + BAMD_ACT_BATCH, pa.q8, pa.S, pa.yd, 1, i0 + n_w);
        } else if (SMALLK && !UNEVEN) {                      // shared prologue of a fast kernel: NBW blocks per wave (issued with BAMD_PRO_ISSUE_NB at entry)
            auto second_half = [&]() { if (RSPLIT) ring_fill(DH, D); };
            BAMD_PRO_FINISH_NB_MID(ap, pa, second_half, BAMD_NB1(NBW));
        }
        else if (SMALLK) BAMD_PRO_FINISH_SMALLK(ap, pa);
        else BAMD_PRO_FINISH(ap, pa);
    } else if (RSPLIT) ring_fill(DH, D);
    TL_STAMP(pa.tl, 2);
    const uint32_t * q8 = pa.q8; const int * S = pa.S; const float * yd = pa.yd;
    for (int r0 = 0; r0 < (ONEB ? 1 : count); r0 += M) {
        const int nbatch = ONEB ? M : (count - r0 < M ? count - r0 : M);  // workgroup-uniform
        float * B0 = part0 + (NBUF == 2 ? (size_t) (batchctr & 1) * M * rg_floats : (size_t) 0);
        // the wave that will run the chain of row-group r0+wave fetches its residual now (old by chain time)
        const int crow = (first + (r0 + (wave < nbatch ? wave : 0)) * stride) * 8 + r8;
        float resv = 0.f;
        if (EPI == BAMD_EPI_ADD && crow < nvalid) resv = ik_ld(res + crow);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (m < nbatch) {
                float * Pf = B0 + (size_t) m * rg_floats;
                float4 * P = (float4 *) Pf;
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const int s = m * NBW + j;
                    const int ci = i0 + j;
                    pin_rec(ring[s]);
                    if (!UNEVEN || j < n_w) {
                        const Terms T = block_terms(ring[s], ci, lane, q8, S, yd);
                        if (COMPACT) {
                            float * rec = Pf + (size_t) ci * 144;
                            *(float2 *) (rec + lane * 2) = make_float2(T.fs, T.pm);
                            if ((lane & 7) == 0) *(float2 *) (rec + 128 + r8 * 2) = make_float2(T.d, T.dmin);
                        } else P[ci * 64 + lane] = make_float4(T.d, T.fs, T.dmin, T.pm);   // every lane owns the terms of its chain: one 16-byte store
                    }
#ifndef BAMD_SPLIT_UNCOND_REFILL
#define BAMD_SPLIT_UNCOND_REFILL 1
#endif
                    if (!ONEB) {
                        const int jj = UNEVEN && j >= n_w ? n_w - 1 : j;
                        if (BAMD_SPLIT_UNCOND_REFILL && SMALLK) {
                            // fast kernels: the refill is UNCONDITIONAL — behind the last batch it goes through the zero-record descriptor (null_rsrc: returns 0,
                            // fetches nothing; a re-request of a real record showed as + 6 % FETCH_SIZE on the 8B ffn_down: nt lines do not stay in the L2).  A branch around the request made the compiler merge the wait counts of
                            // both paths at every join: the waits of a batch ran down from vmcnt(13) to vmcnt(0), i.e. the last record of every batch waited for
                            // the six refills issued just before it (a drained ring per batch); now every record waits for exactly its own two loads
                            const bool more = r0 + M + m < count;
                            load_rec(ring[s], more ? rs : rsn, bbase + (more ? (M + m) * rg_step + jj * RECB : m * rg_step), lane);
                        } else if (r0 + M + m < count) load_rec(ring[s], rs, bbase + (M + m) * rg_step + jj * RECB, lane);
                    }
                    if ((s & (BAMD_SCHED_GROUP - 1)) == BAMD_SCHED_GROUP - 1 || s == D - 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (r0 == 0) TL_STAMP(pa.tl, 3);
#if BAMD_CEILING & 2
        // TIMING-ONLY ceiling build: no barrier, no chain replay — every wave stores from its own first parked term (garbage results)
        if (wave < nbatch) { const float4 t = ((const float4 *) (B0 + (size_t) wave * rg_floats))[i0 * 64 + lane]; if ((lane & 7) == 0 && crow < nvalid) ik_st(out + crow, t.x + t.y + resv); }
        batchctr += 1; bbase += M * rg_step;
        continue;
#endif
        __syncthreads();
        if (r0 == 0) TL_STAMP(pa.tl, 4);
        if (wave < nbatch) {
            // the reference's chains, in order, for lane (r, e)   (ggml-quants.c:6937-6941, :6970, :7518, :8219)
            const float * Pc = B0 + (size_t) wave * rg_floats;
            const float4 * P = (const float4 *) Pc;
            auto term = [&](int ib) -> float4 {              // {d, fs, dmin, pm} of super-block ib for this lane
                if (COMPACT) {
                    const float * rec = Pc + (size_t) ib * 144;
                    const float2 fp = *(const float2 *) (rec + lane * 2), dd = *(const float2 *) (rec + 128 + r8 * 2);
                    return make_float4(dd.x, fp.x, dd.y, fp.y);
                }
                return P[ib * 64 + lane];
            };
            RowAcc A = { 0.f, 0.f };
            int i = 0;
            constexpr int CB = COMPACT ? 4 : 8;              // chain steps per LDS read group (COMPACT runs at 128 VGPRs beside a ring of seven records: 2 x 4 terms in flight)
            if (nb >= CB) {                                  // the LDS reads of CB blocks issued together, the next CB in flight behind them
                float4 t[CB], tn[CB];
#pragma unroll
                for (int u = 0; u < CB; ++u) t[u] = term(u);
                for (; i + 2 * CB <= nb; i += CB) {
#pragma unroll
                    for (int u = 0; u < CB; ++u) tn[u] = term(i + CB + u);
#pragma unroll
                    for (int u = 0; u < CB; ++u) chain_step<TYPE>(A, t[u].x, t[u].y, t[u].z, t[u].w);
#pragma unroll
                    for (int u = 0; u < CB; ++u) t[u] = tn[u];
                }
#pragma unroll
                for (int u = 0; u < CB; ++u) chain_step<TYPE>(A, t[u].x, t[u].y, t[u].z, t[u].w);
                i += CB;
            }
            if (UNEVEN) for (; i < nb; ++i) { const float4 t = term(i); chain_step<TYPE>(A, t.x, t.y, t.z, t.w); }
            const float val = finish_row<TYPE>(A);
            if ((lane & 7) == 0 && crow < nvalid) ik_st(out + crow, EPI == BAMD_EPI_ADD ? val + resv : val);
            if (r0 == 0) TL_STAMP(pa.tl, 5);
        }
        batchctr += 1;
        bbase += M * rg_step;
        if (NBUF == 1 && r0 + M < count) __syncthreads();    // single term buffer: the chains must be done before the next batch writes
    }
}


// host-side dispatch of the fast kernels (bamd_matvec_fast_a.hip / _b.hip); false = no instance for this shape: generic kernel
bool bamd_launch_fast_a(bamd_mv_args a, int pro, int epi, int grid, hipStream_t s);
bool bamd_launch_fast_b(bamd_mv_args a, int pro, int epi, int grid, hipStream_t s);
bool bamd_launch_fast_mixed(const bamd_mv_args & a, int pro, int epi, int grid, hipStream_t s);
bool bamd_launch_fast_b112_supported(int K, int pro, int epi, int nseg, int type);
