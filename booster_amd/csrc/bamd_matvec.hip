// bamd_matvec.hip — single-token mat-vec kernels (mode A: one wave per row-group; mode B: split-K), load-time repack, step begin.
// Numerics contract and reference citations: bamd_device.h.
#include "bamd_device.h"

// ===========================================================================================================
// Load-time repack: GGUF row-major blocks -> wave-stream records (bamd_formats.h).  One thread per (row, block).
// ===========================================================================================================
__global__ void repack_kernel(const uint8_t * __restrict__ raw, uint8_t * __restrict__ dst, int type, int nrows, int nb) {
    const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t) nrows * nb) return;
    const int row = (int) (idx / nb), i = (int) (idx % nb);
    const int rg = row >> 3, r = row & 7;
    const int bb = type == BAMD_Q4_K ? 144 : type == BAMD_Q5_K ? 176 : 210;
    const uint8_t * src = raw + ((int64_t) row * nb + i) * bb;
    uint8_t * rec = dst + ((int64_t) rg * nb + i) * (8 * bb);
    if (type == BAMD_Q4_K || type == BAMD_Q5_K) {
        const uint8_t * qs = src + (type == BAMD_Q4_K ? 16 : 48);
        for (int e = 0; e < 8; ++e)
            for (int j = 0; j < 4; ++j)
                for (int t = 0; t < 4; ++t) rec[(r * 8 + e) * 16 + 4 * j + t] = qs[32 * j + 4 * e + t];
        int hdr_off = 1024;
        if (type == BAMD_Q5_K) {
            for (int e = 0; e < 8; ++e)
                for (int t = 0; t < 4; ++t) rec[1024 + (r * 8 + e) * 4 + t] = src[16 + 4 * e + t];
            hdr_off = 1280;
        }
        for (int t = 0; t < 16; ++t) rec[hdr_off + r * 16 + t] = src[t];
    } else {
        const uint8_t * ql = src, * qh = src + 128, * sc = src + 192;
        for (int e = 0; e < 8; ++e) {
            for (int j = 0; j < 4; ++j)
                for (int t = 0; t < 4; ++t) rec[(r * 8 + e) * 16 + 4 * j + t] = ql[32 * j + 4 * e + t];
            for (int m = 0; m < 2; ++m)
                for (int t = 0; t < 4; ++t) rec[1024 + (r * 8 + e) * 8 + 4 * m + t] = qh[32 * m + 4 * e + t];
        }
        for (int hi = 0; hi < 2; ++hi)
            for (int c = 0; c < 8; ++c) rec[1536 + r * 16 + hi * 8 + c] = sc[2 * c + hi];
        rec[1664 + r * 2] = src[208]; rec[1664 + r * 2 + 1] = src[209];
    }
}

// test entry: standard block_q8_K bytes out of the prologue (for parity tests against quantize_row_q8_K)
__global__ void __launch_bounds__(512) quantize_q8k_test_kernel(const float * x, const float * nw, float eps, int K, int norm, uint8_t * out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nb = K >> 8;
    uint32_t * q8 = (uint32_t *) smem; int * S = (int *) (q8 + nb * 64); float * yd = (float *) (S + nb * 8);
    double * red = (double *) (smem + BAMD_ACT_RED_OFF(nb));
    if (norm) { ActPro<true> ap; ap.issue(x, nw, K, wave_id()); ap.finish(x, nw, eps, K, q8, S, yd, red); }
    else { ActPro<false> ap; ap.issue(x, nw, K, wave_id()); ap.finish(x, nw, eps, K, q8, S, yd, red); }
    for (int i = threadIdx.x; i < nb * 64; i += blockDim.x) {
        const int blk = i >> 6, e = (i >> 3) & 7, c = i & 7;
        const uint32_t w = q8[i];
        uint8_t * o = out + (size_t) blk * 292;
        for (int t = 0; t < 4; ++t) o[4 + 32 * c + 4 * e + t] = (uint8_t) (w >> (8 * t));
    }
    for (int i = threadIdx.x; i < nb; i += blockDim.x) *(float *) (out + (size_t) i * 292) = yd[i];
    __syncthreads();
    for (int i = threadIdx.x; i < nb * 16; i += blockDim.x) {          // bsums from the stored int8
        const int blk = i >> 4, j = i & 15;
        const int8_t * q = (const int8_t *) (out + (size_t) blk * 292 + 4);
        int s = 0; for (int t = 0; t < 16; ++t) s += q[j * 16 + t];
        *(int16_t *) (out + (size_t) blk * 292 + 260 + 2 * j) = (int16_t) (yd[blk] == 0.f ? 0 : s);
    }
}

// ---- MODE A: one wave per row-group --------------------------------------------------------------------------
// The wave walks row-groups rg = first, first+stride, ... (count of them).  A register ring of D records is kept
// in flight by a LOADER cursor that runs D records ahead of the consumer and crosses row-group boundaries by
// pure (branch-free, scalar) arithmetic, so the prefetch never drains and the compiler can keep counted
// s_waitcnt vmcnt(N) waits.  The ring is filled BEFORE the activation prologue (weights do not depend on it), so
// the first HBM round trip overlaps the RMSNorm/Q8_K work.  With PAIR each row-group is streamed twice back to
// back — gate (wA) then up (wB) — and the epilogue fuses silu(gate)*up.
template <int TYPE, typename REC, int D, int EPI, int PRO, bool SMALLK = false>
__device__ __forceinline__ void stream_segment(const uint8_t * __restrict__ wA, const uint8_t * __restrict__ wB, int nb,
                                               int first, int count, int stride, float * __restrict__ out,
                                               const float * __restrict__ res, const ProArgs & pa, ActPro<PRO == BAMD_PRO_NORM> & ap,
                                               bool issue_here, bool do_pro, unsigned long long & best, int nvalid) {
    constexpr int RECB = TYPE == BAMD_Q4_K ? 1152 : TYPE == BAMD_Q5_K ? 1408 : 1680;
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
    constexpr int NPARTS = PAIR ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const long rgb = (long) nb * RECB;                   // D divides nb (chosen by the dispatcher below)
    const long rg_step = (long) stride * rgb;
    const int chunks = nb / D;
    if (issue_here) BAMD_PRO_ISSUE(ap, pa);              // activation loads go out FIRST (see ActPro::issue); the fast kernels issue them at entry
    REC ring[D];
    // The loader runs exactly one CHUNK (D records = the whole ring) ahead of the consumer: slot s is refilled, right after it
    // is consumed, with record s of the chunk that follows in this wave's sequence (next chunk of the row, else the other half
    // of a gate/up pair, else the next row-group).  One wave-uniform base address per chunk: the per-record cost of the cursor
    // is a constant offset, and the loads stay unconditional so the compiler keeps counted s_waitcnt vmcnt(N) waits.
    // (a wave without work — count == 0, fast kernels only — requests record 0 of the matrix D times: L1 hits, and its code path stays
    // the one of the busy waves: one copy of the prologue, no join in front of the counted waits)
    const uint8_t * rowA = wA + (count > 0 ? (long) first * rgb : 0l);
    const int fill_step = count > 0 ? RECB : 0;
#pragma unroll
    for (int s = 0; s < D; ++s) load_rec(ring[s], rowA + s * fill_step, lane);
    TL_STAMP(pa.tl, 1);
    if (do_pro) { if (SMALLK) BAMD_PRO_FINISH_SMALLK(ap, pa); else BAMD_PRO_FINISH(ap, pa); }
    TL_STAMP(pa.tl, 2);
    const uint32_t * q8 = pa.q8; const int * S = pa.S; const float * yd = pa.yd;
    for (int r = 0; r < count; ++r) {
        const int rg = first + r * stride;
        const int row = rg * 8 + (lane >> 3);
        const long rowoff = (long) rg * rgb;
        float gate_val = 0.f;
#pragma unroll
        for (int part = 0; part < NPARTS; ++part) {
            const uint8_t * pbase = (part ? wB : wA) + rowoff;
            // after the last chunk of this row-part: the other half of the pair, the next row-group, or — at the very end of the
            // wave's stream — its own last record again, D times (step 0: one record of redundant traffic, never consumed; the
            // requests stay unconditional so that the waits stay counted)
            const bool last = !(PAIR && part == 0) && r + 1 >= count;
            const uint8_t * after = (PAIR && part == 0) ? wB + rowoff : (last ? pbase + (long) (nb - 1) * RECB : wA + rowoff + rg_step);
            // residual fetched at the START of the row: by the epilogue it is the oldest outstanding load
            float resv = 0.f;
            if (EPI == BAMD_EPI_ADD && row < nvalid) resv = res[row];
            RowAcc A = { 0.f, 0.f };
            for (int c = 0; c < chunks; ++c) {
                const bool inrow = c + 1 < chunks;
                const uint8_t * nxt = inrow ? pbase + (long) (c + 1) * (D * RECB) : after;
                const int step = (inrow || !last) ? RECB : 0;
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    pin_rec(ring[s]);
                    const Terms T = block_terms(ring[s], c * D + s, lane, q8, S, yd);
                    chain_step<TYPE>(A, T.d, T.fs, T.dmin, T.pm);
                    load_rec(ring[s], nxt + s * step, lane);
                    if ((s & (BAMD_SCHED_GROUP - 1)) == BAMD_SCHED_GROUP - 1)
                        __builtin_amdgcn_sched_barrier(0);   // keep hipcc from clustering the refills at the loop tail
                }
                if (r == 0 && part == 0 && c == 0) TL_STAMP(pa.tl, 3);
            }
            if (r + 1 == count && part == NPARTS - 1) TL_STAMP(pa.tl, 4);
            const float val = finish_row<TYPE>(A);
            if (PAIR) {
                if (part == 0) gate_val = val;
                else if ((lane & 7) == 0 && row < nvalid) out[row] = v_silu(gate_val) * val;
            } else if ((lane & 7) == 0 && row < nvalid) {
                float o = val;
                if (EPI == BAMD_EPI_ADD) o = val + resv;
                out[row] = o;
                if (EPI == BAMD_EPI_ARGMAX) { const unsigned long long k = argmax_key(o, row); best = k > best ? k : best; }
            }
        }
    }
}

template <int TYPE, typename REC, int EPI, int PRO>
__device__ __forceinline__ void stream_dispatch_depth(const uint8_t * wA, const uint8_t * wB, int nb, int first, int count, int stride,
                                                      float * out, const float * res, const ProArgs & pa, bool do_pro,
                                                      unsigned long long & best, int nvalid) {
    ActPro<PRO == BAMD_PRO_NORM> ap;
    if ((nb & 7) == 0)      stream_segment<TYPE, REC, 8, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, ap, do_pro, do_pro, best, nvalid);
    else if ((nb & 3) == 0) stream_segment<TYPE, REC, 4, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, ap, do_pro, do_pro, best, nvalid);
    else if ((nb & 1) == 0) stream_segment<TYPE, REC, 2, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, ap, do_pro, do_pro, best, nvalid);
    else                    stream_segment<TYPE, REC, 1, EPI, PRO>(wA, wB, nb, first, count, stride, out, res, pa, ap, do_pro, do_pro, best, nvalid);
}

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

template <int PRO, int EPI>
__global__ void __launch_bounds__(512) matvec_kernel(bamd_mv_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    const int wave = wave_id(), nwaves = blockDim.x >> 6;
    const int slot = blockIdx.x + gridDim.x * wave;          // consecutive row-groups land on different CUs
    const int stride = gridDim.x * nwaves;
    unsigned long long best = 0ull;
    constexpr bool PAIR = EPI == BAMD_EPI_SILU_MUL;
    bool pro_done = false;
    int off = 0;
    const int nseg = PAIR ? 1 : a.nseg;
    for (int s = 0; s < nseg; ++s) {
        const int nrg = a.seg[s].nrows >> 3;
        // my row-groups inside the concatenated index space [off, off+nrg): g = slot + k*stride
        const int k0 = off <= slot ? 0 : (off - slot + stride - 1) / stride;
        const int g0 = slot + k0 * stride;
        const int count = g0 < off + nrg ? (off + nrg - 1 - g0) / stride + 1 : 0;
        if (count > 0) {
            const int t = a.seg[s].type;
            const uint8_t * wA = (const uint8_t *) a.seg[s].w;
            const uint8_t * wB = PAIR ? (const uint8_t *) a.seg[1].w : wA;
            float * out = a.seg[s].out;
            const float * res = a.res;
            const int nv = a.seg[s].nvalid > 0 ? a.seg[s].nvalid : a.seg[s].nrows;
            if (t == BAMD_Q4_K)      stream_dispatch_depth<BAMD_Q4_K, RecQ4K, EPI, PRO>(wA, wB, nb, g0 - off, count, stride, out, res, pa, !pro_done, best, nv);
            else if (t == BAMD_Q5_K) stream_dispatch_depth<BAMD_Q5_K, RecQ5K, EPI, PRO>(wA, wB, nb, g0 - off, count, stride, out, res, pa, !pro_done, best, nv);
            else                     stream_dispatch_depth<BAMD_Q6_K, RecQ6K, EPI, PRO>(wA, wB, nb, g0 - off, count, stride, out, res, pa, !pro_done, best, nv);
            pro_done = true;
        }
        off += nrg;
    }
    if (!pro_done) { ActPro<PRO == BAMD_PRO_NORM> ap; BAMD_PRO_ISSUE(ap, pa); BAMD_PRO_FINISH(ap, pa); }   // idle waves still owe the block its barriers
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

// ---- MODE B: split-K, one 8-wave workgroup per row-group ------------------------------------------------------
// For matrices with few row-groups (wq/wk/wv/wo, ffn_down: 512..768 of them) one wave per row-group leaves the chip
// short of bytes in flight.  Here the 8 waves of a workgroup share a row-group: wave w streams super-blocks
// [w*nb/8, (w+1)*nb/8) and writes the per-block TERMS (d, fs, dmin, pm — exact integers already converted) to LDS;
// after a workgroup barrier ONE wave replays the reference's sequential f32 chain over all nb blocks in order.
// Same arithmetic, same order, 8x the parallelism.  Term buffers are double-buffered so the chain of row-group n
// overlaps the streaming of row-group n+1; the prefetch ring spans row-group boundaries (M row-groups per body).
// LDS term buffers: 2 (double buffer) x M (row-groups per batch) x nb x 64 lanes x float4 {d, fs, dmin, pm}
#define BAMD_TERM_FLOATS(nb) ((size_t) (nb) * 256)      /* one float4 {d, fs, dmin, pm} per lane per super-block */

template <int TYPE, typename REC, int NBW, int M, int NBUF, int EPI, int PRO, bool SMALLK = false>
__device__ __forceinline__ void split_stream(const uint8_t * __restrict__ w, int nb, int first, int count, int stride,
                                             float * __restrict__ out, const float * __restrict__ res, const ProArgs & pa,
                                             ActPro<PRO == BAMD_PRO_NORM> & ap, ActPro<PRO == BAMD_PRO_NORM> & ap2, bool issue_here, bool do_pro,
                                             float * part0, int & batchctr, int nvalid) {
    constexpr int RECB = TYPE == BAMD_Q4_K ? 1152 : TYPE == BAMD_Q5_K ? 1408 : 1680;
    constexpr int D = NBW * M;                               // ring depth = one batch (M row-groups) of this wave's records
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int r8 = lane >> 3, l4 = lane & 3;
    const long rgb = (long) nb * RECB;
    const long rg_step = (long) stride * rgb;
    const int i0 = wave * NBW;                               // this wave's first super-block inside a row
    const size_t rg_floats = BAMD_TERM_FLOATS(nb);
    // PLAIN prologue: wave w consumes only the activations of its own K-slice (blocks i0 .. i0+NBW-1), so it quantises exactly
    // those — no workgroup barrier, and a wave starts on its records as soon as ITS blocks are done.  (NORM needs the sum of
    // squares of the whole vector: shared prologue as in mode A.)
    constexpr bool OWN = PRO == BAMD_PRO_PLAIN;
    if (issue_here) {                                        // (the fast kernels issue these at entry)
        if (OWN) { ap.issue(pa.x, pa.nw, pa.K, i0, 1, i0 + NBW); if (NBW > BAMD_ACT_BATCH) ap2.issue(pa.x, pa.nw, pa.K, i0 + BAMD_ACT_BATCH, 1, i0 + NBW); }
        else BAMD_PRO_ISSUE(ap, pa);                         // activation loads go out FIRST
    }
    // ring slot (m, j) holds record i0+j of row-group r0+m; after it is consumed it is refilled with the same record of row-group
    // r0+M+m, i.e. a constant M*rg_step further on: the loader needs one wave-uniform base per batch and nothing per record
    const uint8_t * bbase = w + (long) first * rgb + (long) i0 * RECB;
    REC ring[D];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        if (m < count) {                                     // no redundant requests when the stream is short
#pragma unroll
            for (int j = 0; j < NBW; ++j) load_rec(ring[m * NBW + j], bbase + (long) m * rg_step + j * RECB, lane);
        }
    }
    TL_STAMP(pa.tl, 1);
    if (do_pro) {
        if (OWN) {
            static_assert(NBW <= 2 * BAMD_ACT_BATCH, "own-slice prologue handles two batches");
            ap.quantize_batch(1.0f, pa.K, i0, pa.q8, pa.S, pa.yd, 1, i0 + NBW);
            if (NBW > BAMD_ACT_BATCH) ap2.quantize_batch(1.0f, pa.K, i0 + BAMD_ACT_BATCH, pa.q8, pa.S, pa.yd, 1, i0 + NBW);
        } else if (SMALLK) BAMD_PRO_FINISH_SMALLK(ap, pa);
        else BAMD_PRO_FINISH(ap, pa);
    }
    TL_STAMP(pa.tl, 2);
    const uint32_t * q8 = pa.q8; const int * S = pa.S; const float * yd = pa.yd;
    for (int r0 = 0; r0 < count; r0 += M) {
        const int nbatch = count - r0 < M ? count - r0 : M;  // workgroup-uniform
        float * B0 = part0 + (NBUF == 2 ? (size_t) (batchctr & 1) * M * rg_floats : (size_t) 0);
        // the wave that will run the chain of row-group r0+wave fetches its residual now (old by chain time)
        const int crow = (first + (r0 + (wave < nbatch ? wave : 0)) * stride) * 8 + r8;
        float resv = 0.f;
        if (EPI == BAMD_EPI_ADD && crow < nvalid) resv = res[crow];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (m < nbatch) {
                float4 * P = (float4 *) (B0 + (size_t) m * rg_floats);
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const int s = m * NBW + j;
                    const int ci = i0 + j;
                    pin_rec(ring[s]);
                    const Terms T = block_terms(ring[s], ci, lane, q8, S, yd);
                    P[ci * 64 + lane] = make_float4(T.d, T.fs, T.dmin, T.pm);   // every lane owns the terms of its chain: one 16-byte store
                    if (r0 + M + m < count) load_rec(ring[s], bbase + (long) (M + m) * rg_step + j * RECB, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (r0 == 0) TL_STAMP(pa.tl, 3);
        __syncthreads();
        if (r0 == 0) TL_STAMP(pa.tl, 4);
        if (wave < nbatch) {
            // the reference's chains, in order, for lane (r, e)   (ggml-quants.c:6937-6941, :6970, :7518, :8219)
            const float4 * P = (const float4 *) (B0 + (size_t) wave * rg_floats);
            RowAcc A = { 0.f, 0.f };
            for (int i = 0; i < nb; i += 8) {                // nb % 8 == 0 here; the 16-byte LDS reads of 8 blocks issued together
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = P[(i + u) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) chain_step<TYPE>(A, t[u].x, t[u].y, t[u].z, t[u].w);
            }
            const float val = finish_row<TYPE>(A);
            if ((lane & 7) == 0 && crow < nvalid) out[crow] = EPI == BAMD_EPI_ADD ? val + resv : val;
            if (r0 == 0) TL_STAMP(pa.tl, 5);
        }
        batchctr += 1;
        bbase += (long) M * rg_step;
        if (NBUF == 1 && r0 + M < count) __syncthreads();    // single term buffer: the chains must be done before the next batch writes
    }
}

template <int TYPE, typename REC, int EPI, int PRO>
__device__ __forceinline__ void split_dispatch(const uint8_t * w, int nb, int first, int count, int stride, float * out, const float * res,
                                               const ProArgs & pa, bool do_pro, float * part0, int & rgctr, int nvalid) {
    const int nbw = nb >> 3;
    ActPro<PRO == BAMD_PRO_NORM> ap, ap2;
    // (records per wave per row-group, row-groups per batch, term buffers): the batch is the prefetch depth.  K = 14336 with M = 2
    // (all of ffn_down's work per workgroup in flight from the first instruction, single-buffered) measured no better for Q4_K and
    // 14 % worse for Q6_K than M = 1: the kernel is instruction-issue bound, not latency bound.
    if (nbw == 2)       split_stream<TYPE, REC, 2, 4, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
    else if (nbw == 7)  split_stream<TYPE, REC, 7, 1, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
    else if (nbw == 4)  split_stream<TYPE, REC, 4, 2, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
    else if (nbw == 1)  split_stream<TYPE, REC, 1, 8, 2, EPI, PRO>(w, nb, first, count, stride, out, res, pa, ap, ap2, do_pro, do_pro, part0, rgctr, nvalid);
    else __builtin_trap();                               // the launcher only picks this kernel for the shapes above
}

// host must check bamd_split_supported(nb) before choosing this kernel
template <int PRO, int EPI>
__global__ void __launch_bounds__(512) matvec_split_kernel(bamd_mv_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    float * part0 = (float *) (smem + BAMD_ACT_RED_OFF(nb) + 16 * sizeof(double));
    int rgctr = 0;
    bool pro_done = false;
    int off = 0;
    const int slot = blockIdx.x, stride = gridDim.x;         // row-groups are dealt to WORKGROUPS here
    for (int s = 0; s < a.nseg; ++s) {
        const int nrg = a.seg[s].nrows >> 3;
        const int k0 = off <= slot ? 0 : (off - slot + stride - 1) / stride;
        const int g0 = slot + k0 * stride;
        const int count = g0 < off + nrg ? (off + nrg - 1 - g0) / stride + 1 : 0;
        if (count > 0) {
            const int t = a.seg[s].type;
            const uint8_t * w = (const uint8_t *) a.seg[s].w;
            const int nv = a.seg[s].nvalid > 0 ? a.seg[s].nvalid : a.seg[s].nrows;
            if (t == BAMD_Q4_K)      split_dispatch<BAMD_Q4_K, RecQ4K, EPI, PRO>(w, nb, g0 - off, count, stride, a.seg[s].out, a.res, pa, !pro_done, part0, rgctr, nv);
            else if (t == BAMD_Q5_K) split_dispatch<BAMD_Q5_K, RecQ5K, EPI, PRO>(w, nb, g0 - off, count, stride, a.seg[s].out, a.res, pa, !pro_done, part0, rgctr, nv);
            else                     split_dispatch<BAMD_Q6_K, RecQ6K, EPI, PRO>(w, nb, g0 - off, count, stride, a.seg[s].out, a.res, pa, !pro_done, part0, rgctr, nv);
            pro_done = true;
        }
        off += nrg;
    }
    if (!pro_done) { ActPro<PRO == BAMD_PRO_NORM> ap; BAMD_PRO_ISSUE(ap, pa); BAMD_PRO_FINISH(ap, pa); }
    TL_STAMP(a.tl, 7);
}


// ===========================================================================================================
// FAST KERNELS — one kernel per (weight type, launch shape): straight-line code from the first instruction to the streaming loop.
// The generic kernels above pick the weight type, the ring depth and the segment at run time inside ONE kernel; hipcc then has to
// merge register states at every join, which (a) put a full s_waitcnt behind each conditional activation load and held the weight
// ring back until the activations had arrived, (b) spilled scalar registers to vector lanes, and (c) made every launch walk through
// a 200 KB code object.  Here the dispatch happens on the host (bamd_launch_matvec): activation requests at entry, ring requests
// right behind them, counted waits all the way.  Same device functions (block_terms / chain_step / finish_row), same bits.
// Shapes outside the table (K/256 not a multiple of 8, three differently typed segments, ...) keep using the generic kernels.
// ===========================================================================================================
template <int TYPE> struct RecOf;
template <> struct RecOf<BAMD_Q4_K> { typedef RecQ4K type; };
template <> struct RecOf<BAMD_Q5_K> { typedef RecQ5K type; };
template <> struct RecOf<BAMD_Q6_K> { typedef RecQ6K type; };

// mode A.  TYPE1 == 0: one segment (or the gate/up pair: seg[0] and seg[1] of TYPE0, EPI_SILU_MUL), any number of row-groups per wave
// (a.cnt_q / a.cnt_r = row-groups / wave slots, quotient and remainder).  TYPE1 != 0: two segments of different types with at most one
// row-group per wave (fused QKV with a Q6_K / Q5_K attn_v): the wave's row-group picks the branch, each branch is straight-line.
template <int TYPE0, int TYPE1, int PRO, int EPI>
__global__ void __launch_bounds__(512) matvec_fast_kernel(bamd_mv_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TL_STAMP(a.tl, 0);
    const int nb = a.K >> 8;
    const ProArgs pa = carve_lds(a, smem);
    ActPro<PRO == BAMD_PRO_NORM> ap;
    BAMD_PRO_ISSUE(ap, pa);                                  // activation requests: the first memory instructions of the kernel
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
        stream_segment<TYPE0, REC0, 8, EPI, PRO, true>(wA, wB, nb, slot, count, stride, a.seg[0].out, a.res, pa, ap, false, true, best, nv0);   // count == 0: prologue only
    } else {
        typedef typename RecOf<TYPE1 == 0 ? TYPE0 : TYPE1>::type REC1;
        constexpr int T1 = TYPE1 == 0 ? TYPE0 : TYPE1;
        const int nrg1 = a.seg[1].nrows >> 3;
        const int nv1 = a.seg[1].nvalid > 0 ? a.seg[1].nvalid : a.seg[1].nrows;
        if (slot >= nrg0 && slot < nrg0 + nrg1) {
            const uint8_t * w1 = (const uint8_t *) a.seg[1].w;
            stream_segment<T1, REC1, 8, EPI, PRO, true>(w1, w1, nb, slot - nrg0, 1, stride, a.seg[1].out, a.res, pa, ap, false, true, best, nv1);
        } else {                                             // segment 0, or no work (count 0: prologue only)
            const uint8_t * w0 = (const uint8_t *) a.seg[0].w;
            stream_segment<TYPE0, REC0, 8, EPI, PRO, true>(w0, w0, nb, slot, slot < nrg0 ? 1 : 0, stride, a.seg[0].out, a.res, pa, ap, false, true, best, nv0);
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

// mode B (split-K), one segment of one type, NBW = K / 2048 records per wave and row-group, M row-groups per batch
template <int TYPE, int NBW, int M, int PRO, int EPI>
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
    split_stream<TYPE, REC, NBW, M, 2, EPI, PRO, true>((const uint8_t *) a.seg[0].w, nb, (int) blockIdx.x, count, (int) gridDim.x, a.seg[0].out, a.res, pa,
                                                       ap, ap2, false, true, part0, rgctr, nv);      // the launcher's grid gives every workgroup >= 1 row-group
    TL_STAMP(a.tl, 7);
}

// ===========================================================================================================
// Step begin: pick the token of this step (forced prompt token, or the arg-max of the previous step's logits),
// advance the position, and dequantise its embedding row into the residual stream.
// ===========================================================================================================
__global__ void __launch_bounds__(1024) step_begin_kernel(bamd_step_state * st, const int32_t * forced, int n_forced,
                                                         int32_t * out_tokens, const uint8_t * embd, int embd_type, int E, int V,
                                                         float * x, int do_embed) {
    __shared__ int tok_s;
    if (threadIdx.x == 0) {
        int step = st->step;
        int tok;
        const unsigned long long key = st->best_key;         // arg-max of the previous lm_head, 0 = none ran
        if (key != 0ull) {
            tok = (int) (0xffffffffu - (uint32_t) (key & 0xffffffffull));
            out_tokens[st->n_out] = tok; st->n_out += 1;
        } else tok = 0;
        if (step < n_forced) tok = forced[step];
        if (tok < 0 || tok >= V) tok = 0;
        st->token = tok;
        if (do_embed) {
            st->pos = st->pos_base + step;
            int n_kv = (st->pos + 1 + 31) / 32 * 32;
            if (n_kv > st->n_ctx) n_kv = st->n_ctx;
            st->n_kv = n_kv;
            st->step = step + 1;
        }
        if (do_embed) st->best_key = 0ull;                   // a flush-only call leaves the key for the next generate call
        tok_s = tok;
    }
    __syncthreads();
    if (!do_embed) return;
    embed_row(embd, embd_type, E, tok_s, x);
}


// ===========================================================================================================
// launchers
// ===========================================================================================================
int bamd_timing_enabled(void) {
#ifdef BAMD_TIMING
    return 1;
#else
    return 0;
#endif
}
void bamd_launch_repack(const void * raw, void * dst, int type, int nrows, int K, hipStream_t s) {
    const int nb = K >> 8;
    const int64_t n = (int64_t) nrows * nb;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, (const uint8_t *) raw, (uint8_t *) dst, type, nrows, nb);
}

void bamd_launch_quantize_q8k_test(const float * x, const float * nw, float eps, int K, int norm, void * out, hipStream_t s) {
    hipLaunchKernelGGL(quantize_q8k_test_kernel, dim3(1), dim3(512), act_lds_bytes(K), s, x, nw, eps, K, norm, (uint8_t *) out);
}

template <int PRO>
static void launch_mv_epi(const bamd_mv_args & a, int epi, int grid, hipStream_t s) {
    const size_t lds = act_lds_bytes(a.K);
    switch (epi) {
        case BAMD_EPI_STORE:    hipLaunchKernelGGL((matvec_kernel<PRO, BAMD_EPI_STORE>),    dim3(grid), dim3(512), lds, s, a); break;
        case BAMD_EPI_ADD:      hipLaunchKernelGGL((matvec_kernel<PRO, BAMD_EPI_ADD>),      dim3(grid), dim3(512), lds, s, a); break;
        case BAMD_EPI_SILU_MUL: hipLaunchKernelGGL((matvec_kernel<PRO, BAMD_EPI_SILU_MUL>), dim3(grid), dim3(512), lds, s, a); break;
        case BAMD_EPI_ARGMAX:   hipLaunchKernelGGL((matvec_kernel<PRO, BAMD_EPI_ARGMAX>),   dim3(grid), dim3(512), lds, s, a); break;
    }
}
template <int PRO>
static void launch_mv_split(const bamd_mv_args & a, int epi, int grid, hipStream_t s) {
    const int nb = a.K >> 8;
    const int nbw = nb >> 3;
    const int M = nbw == 2 ? 4 : nbw == 7 ? 1 : nbw == 4 ? 2 : 8, NBUF = 2;                  // must match split_dispatch
    const size_t lds = act_lds_bytes(a.K) + 16 + (size_t) NBUF * M * nb * 256 * 4;   // 112..128 KiB of term buffers
    if (epi == BAMD_EPI_ADD) hipLaunchKernelGGL((matvec_split_kernel<PRO, BAMD_EPI_ADD>),   dim3(grid), dim3(512), lds, s, a);
    else                     hipLaunchKernelGGL((matvec_split_kernel<PRO, BAMD_EPI_STORE>), dim3(grid), dim3(512), lds, s, a);
}

static bool split_supported(int nb) { const int nbw = nb >> 3; return (nb & 7) == 0 && (nbw == 1 || nbw == 2 || nbw == 4 || nbw == 7); }

// ---- host-side dispatch of the fast kernels; false = no instance for this shape (the caller takes the generic kernel) ----
template <int PRO, int EPI, int T0, int T1>
static void launch_fast_a_inst(const bamd_mv_args & a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((matvec_fast_kernel<T0, T1, PRO, EPI>), dim3(grid), dim3(512), act_lds_bytes(a.K), s, a);
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
static bool launch_fast_a(bamd_mv_args a, int pro, int epi, int grid, hipStream_t s) {
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
template <int PRO, int EPI, int T, int NBW, int M>
static void launch_fast_b_inst(const bamd_mv_args & a, int grid, hipStream_t s) {
    const int nb = a.K >> 8;
    const size_t lds = act_lds_bytes(a.K) + 16 + (size_t) 2 * M * nb * 256 * 4;
    hipLaunchKernelGGL((matvec_split_fast_kernel<T, NBW, M, PRO, EPI>), dim3(grid), dim3(512), lds, s, a);
}
template <int PRO, int EPI>
static bool launch_fast_b_types(const bamd_mv_args & a, int t, int nbw, int grid, hipStream_t s) {
#define BAMD_B_CASE(T_, NBW_, M_) if (t == T_ && nbw == NBW_) { launch_fast_b_inst<PRO, EPI, T_, NBW_, M_>(a, grid, s); return true; }
#define BAMD_B_TYPES(NBW_, M_) BAMD_B_CASE(BAMD_Q4_K, NBW_, M_) BAMD_B_CASE(BAMD_Q5_K, NBW_, M_) BAMD_B_CASE(BAMD_Q6_K, NBW_, M_)
    BAMD_B_TYPES(1, 8) BAMD_B_TYPES(2, 4) BAMD_B_TYPES(4, 2)
    if (PRO == BAMD_PRO_PLAIN) { BAMD_B_TYPES(7, 1) }
#undef BAMD_B_TYPES
#undef BAMD_B_CASE
    return false;
}
static bool launch_fast_b(bamd_mv_args a, int pro, int epi, int grid, hipStream_t s) {
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
static const bool g_mv_generic = [] { const char * e = getenv("BAMD_MV_GENERIC"); return e && e[0] == '1'; }();

void bamd_launch_matvec(const bamd_mv_args & a, int pro, int epi, int n_cu, hipStream_t s) {
    int nrg = 0;
    if (epi == BAMD_EPI_SILU_MUL) nrg = a.seg[0].nrows >> 3;
    else for (int i = 0; i < a.nseg; ++i) nrg += a.seg[i].nrows >> 3;
    const int cus = n_cu > 0 ? n_cu : 256;
    // few row-groups: split K over the 8 waves of a workgroup (mode B); otherwise one wave per row-group (mode A)
    const bool can_split = (epi == BAMD_EPI_STORE || epi == BAMD_EPI_ADD) && split_supported(a.K >> 8);
    // differently typed segments (wq|wk Q4_K + wv Q6_K): a split-K workgroup would stream them one after the other, each with its
    // own ring fill; with one wave per row-group every wave has a single row-group of a single type
    const bool mixed = a.nseg > 1 && epi == BAMD_EPI_STORE && nrg <= 8 * cus;
    const int fmode = a.mode & 15;
    const bool split = fmode == 2 ? can_split : fmode == 1 ? false : (can_split && nrg < 8 * cus && !mixed);
    int grid = cus;                                          // one 8-wave workgroup per CU
    if (grid > nrg) grid = nrg;
    if (grid < 1) grid = 1;
    const bool generic = g_mv_generic || a.mode >= 16;       // mode bit 4: force the generic kernels (tests)
    if (split) {
        if (!generic && launch_fast_b(a, pro, epi, grid, s)) return;
        if (pro == BAMD_PRO_NORM) launch_mv_split<BAMD_PRO_NORM>(a, epi, grid, s);
        else                      launch_mv_split<BAMD_PRO_PLAIN>(a, epi, grid, s);
        return;
    }
    if (!generic && launch_fast_a(a, pro, epi, grid, s)) return;
    if (pro == BAMD_PRO_NORM) launch_mv_epi<BAMD_PRO_NORM>(a, epi, grid, s);
    else                      launch_mv_epi<BAMD_PRO_PLAIN>(a, epi, grid, s);
}

void bamd_launch_step_begin(bamd_step_state * st, const int32_t * forced, int n_forced, int32_t * out_tokens, const void * embd,
                            int embd_type, int E, int V, float * x, int do_embed, hipStream_t s) {
    hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(1024), 0, s, st, forced, n_forced, out_tokens, (const uint8_t *) embd, embd_type, E, V, x, do_embed);
}

