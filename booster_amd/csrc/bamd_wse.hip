// bamd_wse.hip — the weight-stream engine kernel: one persistent launch runs a whole decode step (or any sub-range of its ops).
// Design, wave roles and the reference functions it replaces: bamd_wse.h.  Numerics: the per-record terms are block_terms() and the chains
// chain_step() / finish_row() of bamd_device.h — the code the launch kernels run — so every output is the launch path's, bit for bit.
//
// Intra-CU hand-overs are LDS words (the LDS of a CU is one in-order unit: a wave's data writes are visible before its later flag write):
//   filled       = number of slots whose DMA has landed, in program order             (loader -> consumers; slot g lives at ring position g % ns)
//   freec[slot] += 1 per record whose LDS reads have been issued                     (consumers -> loader; monotonic; expect[slot] is the loader's own tally)
//   chunk[c]    += 1 per record parked in term chunk c (8 records)                  (consumers -> chainers; the chainer that owns the chunk resets it)
//   rel[c]       = how many times term chunk c has been read and handed back        (chainers -> consumers: record g may be parked when rel >= g / tr)
//   pieces[k]    = pieces chainer k has finished                                    (chainers -> consumers: the attention scratch aliases the term ring)
//   cbar / cbar8 = counting barriers among the consumer waves (all of them / the eight that run an attention head)
// Inter-CU hand-overs are 8-byte {value, tag} granules, one sc1 store each, re-read with sc1 loads until every tag matches
// (cdna_hip_programming.md, Guideline 16 R2); tag = (host serial, device step, consumer layer) is unique among consecutive uses of a word.
#include "bamd_matvec_core.h"
#include "bamd_attn_fused.h"
#include "bamd_wse.h"

#define WSE_SPINS_LDS (1u << 21)       /* bounded waits: ~0.2 s of LDS polling / ~1 s of granule polling, then give up (err) and run on */
#define WSE_SPINS_GLB (1u << 20)

enum { W_FREE = 16, W_EXPECT = 32, W_CBAR = 49, W_CBAR8 = 50, W_ABORT = 51, W_GATHERING = 52, W_FILLED = 53, W_PIECES = 56, W_CHUNK = 64, W_REL = 80,
       W_RED_BYTES = 384, W_STASH_BYTES = 512 };       /* words of the control block; red: 16 doubles; stash: BAMD_WSE_STASH floats */

__device__ __forceinline__ uint32_t lds_ld(const uint32_t * w) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void lds_st(uint32_t * w, uint32_t v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wse_fail(const bamd_wse_args & a, uint32_t * misc, uint32_t code) {
    if ((threadIdx.x & 63) == 0) {
        lds_st(misc + W_ABORT, 1u);
        if (atomicAdd(a.err, 1u) == 0u) { a.err[1] = code; a.err[2] = blockIdx.x; a.err[3] = threadIdx.x >> 6; }
    }
}
// wait until (int) (*w - want) >= 0 (monotonic counters) or, EQ, until *w == want
template <bool EQ>
__device__ __forceinline__ void lds_wait(const bamd_wse_args & a, uint32_t * misc, const uint32_t * w, uint32_t want, uint32_t code) {
    for (unsigned spins = 0;; ++spins) {
        const uint32_t v = lds_ld(w);
        if (EQ ? v == want : (int) (v - want) >= 0) break;
        if (spins > WSE_SPINS_LDS || lds_ld(misc + W_ABORT)) { if (spins > WSE_SPINS_LDS) wse_fail(a, misc, code); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");          // nothing that follows may be read ahead of the flag (the LDS itself serves a wave in order)
}
// a pointer that arrives through memory (a field of the argument block picked by a run-time index, an op's 64-bit address) is a GENERIC pointer to
// hipcc: its loads become FLAT instructions, which count on both wait counters.  Everything here is global memory: say so.
template <typename T> __device__ __forceinline__ T * as_global(const void * p) { return (T *) (__attribute__((address_space(1))) T *) (uintptr_t) p; }
template <typename T> __device__ __forceinline__ T * as_global(uint64_t p) { return (T *) (__attribute__((address_space(1))) T *) (uintptr_t) p; }
// The roles of a wave are inlined into ONE loop over its program; without this LLVM hoists every lane-dependent address expression of every role out
// of that loop and keeps them all live through the attention body (204 VGPRs instead of ~110).  An opaque copy of the lane / thread id at the entry
// of each role keeps its arithmetic inside the role.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// monotonic counter: wait until (int) (*w - want) >= 0; returns the value seen (callers cache it: the next few waits need no LDS round trip)
__device__ __forceinline__ uint32_t lds_wait_ge(const bamd_wse_args & a, uint32_t * misc, const uint32_t * w, uint32_t want, uint32_t code) {
    uint32_t v;
    for (unsigned spins = 0;; ++spins) {
        v = lds_ld(w);
        if ((int) (v - want) >= 0) break;
        if (spins > WSE_SPINS_LDS || lds_ld(misc + W_ABORT)) { if (spins > WSE_SPINS_LDS) wse_fail(a, misc, code); v = want; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return v;
}
struct CBar { uint32_t target = 0; };
__device__ __forceinline__ void cbar(const bamd_wse_args & a, uint32_t * misc, int word, CBar & b, int n, uint32_t code) {
    asm volatile("" ::: "memory");           // the wave's earlier LDS writes reach the LDS before its arrival (in-order unit): no hardware wait needed
    if ((threadIdx.x & 63) == 0) atomicAdd(misc + word, 1u);
    b.target += (uint32_t) n;
    lds_wait<false>(a, misc, misc + word, b.target, code);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void tl_stamp(const bamd_wse_args & a, const bamd_wse_op & op, int ev) {
    if (a.tl && op.tlslot != 255 && (threadIdx.x & 63) == 0) a.tl[((size_t) blockIdx.x * a.tl_ops + op.tlslot) * 8 + ev] = wall_clock64();
}
__device__ __forceinline__ uint32_t wse_tagbase(const bamd_step_state * st) { return ((uint32_t) st->serial << 20) | (((uint32_t) st->step & 0xfffu) << 8); }

// ---- LOADER ------------------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void * wse_lds_vp;
// one 1 KiB piece of a fill: lane l moves the 16 bytes at gsrc (its own address) to LDS dst + 16 l.  Inline asm: invisible to hipcc's wait counting
// (the loader counts vmcnt itself); M0 is written in the statement that reads it (cdna_hip_programming.md 5.7)
__device__ __forceinline__ void dma1k(const uint8_t * gsrc, uint32_t dst_) {
    const uint32_t dst = (uint32_t) __builtin_amdgcn_readfirstlane((int) dst_);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void wse_loader(const bamd_wse_args & a, const bamd_wse_op * ops, unsigned char * smem, uint32_t * misc) {
    const int lane = threadIdx.x & 63;
    const uint32_t ring0 = (uint32_t) (size_t) (wse_lds_vp) smem;
    const uint32_t ns = (uint32_t) a.ns;
    uint32_t gs = 0, slot = 0;            // next slot to issue (global number, ring position)
    uint32_t pub = 0;                     // slots published so far
    auto publish_to = [&](uint32_t upto) {        // everything below `upto` has landed
        if (pub < upto) { pub = upto; lds_st(misc + W_FILLED, pub); }
    };
    for (int io = 0;; ++io) {
        const bamd_wse_op op = ops[io];
        if (op.kind == BAMD_WSE_END) break;
        if (op.kind != BAMD_WSE_MATVEC) continue;
        const uint32_t recb = (uint32_t) bamd_record_bytes((int) op.type), nrec = op.ntask * op.nb;
        const uint8_t * src = as_global<const uint8_t>(op.src);
        for (uint32_t done = 0; done < nrec; done += op.rps, ++gs) {
            const uint32_t nr = nrec - done < op.rps ? nrec - done : op.rps;
            const uint32_t want = lds_ld(misc + W_EXPECT + slot);
            if (gs >= ns && lds_ld(misc + W_FREE + slot) != want) {          // ring full: nothing to issue, so let everything land and publish it
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                publish_to(gs);
                lds_wait<true>(a, misc, misc + W_FREE + slot, want, 0x100u);
            }
            lds_st(misc + W_EXPECT + slot, want + nr);
            const uint32_t nld = (nr * recb + 1023u) >> 10, dst = ring0 + slot * BAMD_WSE_SLOT;
            const uint8_t * p = src + (size_t) done * recb + lane * 16;
#pragma unroll
            for (uint32_t i = 0; i < 16; ++i) {       // always 16 requests (vmcnt arithmetic stays constant); past the data: the last KiB again (a cache hit)
                const uint32_t k = i < nld ? i : nld - 1u;
                dma1k(p + (size_t) k * 1024u, dst + k * 1024u);
            }
            if ((a.thin & 1) && lds_ld(misc + W_GATHERING)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); publish_to(gs + 1u); }
            else { asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); if (gs >= 3u) publish_to(gs - 2u); }       // three fills (48 KiB per CU, 12 MiB on the chip) may be in flight
            slot = slot + 1u == ns ? 0u : slot + 1u;
        }
        if (!(a.thin & 2)) tl_stamp(a, op, BAMD_WSE_TL_LOADED);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish_to(gs);
}

// ---- CONSUMERS ---------------------------------------------------------------------------------------------------------------------------------
// records out of an LDS slot: the record layout of bamd_formats.h, addressed as load_rec() does in global memory
__device__ __forceinline__ void lds_rec(RecQ4K & R, const unsigned char * p, int lane) {
    R.qs = *(const uint4 *) (p + lane * 16); R.hd = *(const uint4 *) (p + 1024 + (lane >> 3) * 16); R.mn47 = 0u;
}
__device__ __forceinline__ void lds_rec(RecQ5K & R, const unsigned char * p, int lane) {
    R.qs = *(const uint4 *) (p + lane * 16); R.qh = *(const uint32_t *) (p + 1024 + lane * 4); R.hd = *(const uint4 *) (p + 1280 + (lane >> 3) * 16); R.mn47 = 0u;
}
__device__ __forceinline__ void lds_rec(RecQ6K & R, const unsigned char * p, int lane) {
    R.ql = *(const uint4 *) (p + lane * 16); R.qh = *(const uint2 *) (p + 1024 + lane * 8);
    R.sc = *(const uint2 *) (p + 1536 + (lane >> 3) * 16 + ((lane >> 2) & 1) * 8); R.d = (uint32_t) *(const unsigned short *) (p + 1664 + (lane >> 3) * 2);
}

struct ActPtrs { uint32_t * q8; int * S; float * yd; };
__device__ __forceinline__ ActPtrs act_ptrs(unsigned char * smem, const bamd_wse_args & a, int buf, int nb) {
    ActPtrs p; p.q8 = (uint32_t *) (smem + a.off_act[buf]); p.S = (int *) (p.q8 + nb * 64); p.yd = (float *) (p.S + nb * 8); return p;
}

// two granules (16 bytes) of a vector another CU publishes: sc1, past this CU's L1
__device__ __forceinline__ uint4 gran2(bamd_rsrc r, uint32_t byte_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int) byte_off, 0, 16);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ bamd_rsrc vec_rsrc(const void * base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) uniform_ptr((const uint8_t *) base), 0, (int) bytes, 0x00020000);
}

// the activation vector of an op -> Q8_K (optionally RMSNorm * weight first) in LDS buffer op.actbuf, by all NC consumer waves:
// consumer cw takes the blocks cw, cw + NC, ... in batches of BAMD_ACT_BATCH.  Arithmetic: ActPro::quantize_batch and the sum of ActPro::finish.
template <bool NORM, int NB, int NBAT>
__device__ __forceinline__ void wse_gather(const bamd_wse_args & a, const bamd_wse_op & op, unsigned char * smem, uint32_t * misc, int cw, CBar & cb, uint32_t tagbase) {
    static_assert(!NORM || NBAT == 1, "the RMSNorm prologue keeps a wave's whole share in one batch");
    const int lane = opaque((int) threadIdx.x & 63), nc = a.nc, nb = (int) op.nb, K = nb * 256;
    const bamd_wse_vec vin = a.vec[op.in_vec];
    const ActPtrs ap_ = act_ptrs(smem, a, op.actbuf, nb);
    double * red = (double *) ((unsigned char *) misc + W_RED_BYTES);
    const uint32_t tag = tagbase | op.in_tag;
    const bamd_rsrc gr = vec_rsrc(vin.p, vin.n * (vin.gran ? 8u : 4u));
    const float * nw = as_global<const float>(op.normw);
    if (cw == 0) { tl_stamp(a, op, BAMD_WSE_TL_GATHER0); if ((a.thin & 1) && lane == 0) lds_st(misc + W_GATHERING, 1u); }
    float scale = 1.0f;
    // this wave's blocks: cw + (bt * NB + b) * nc — ALL of them requested in one sweep (a second batch behind the first quantisation would cost a
    // second memory round trip), quantised batch by batch
  for (int first = cw; first < nb || first == cw; first += NB * NBAT * nc) {        // (one sweep unless a wave's share exceeds NB * NBAT blocks: 70B ffn_down)
    ActPro<NORM> ap[NBAT];
    int blk[NBAT][NB];
#pragma unroll
    for (int bt = 0; bt < NBAT; ++bt) {
        ap[bt].okmask = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) { const int i = first + (bt * NB + b) * nc; const bool ok = i < nb; ap[bt].okmask |= ok ? 1 << b : 0; blk[bt][b] = ok ? i : nb - 1; }
    }
    if (NORM) {
#pragma unroll
        for (int b = 0; b < NB; ++b) ap[0].w[b] = *(const float4 *) (nw + blk[0][b] * 256 + lane * 4);
    }
    if (vin.gran) {
        for (unsigned spins = 0;; ++spins) {
            asm volatile("" ::: "memory");
            bool ok = true;
#pragma unroll
            for (int bt = 0; bt < NBAT; ++bt)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const uint32_t off = (uint32_t) (blk[bt][b] * 256 + lane * 4) * 8u;
                    const uint4 g0 = gran2(gr, off), g1 = gran2(gr, off + 16u);
                    ap[bt].v[b] = make_float4(__uint_as_float(g0.x), __uint_as_float(g0.z), __uint_as_float(g1.x), __uint_as_float(g1.z));
                    ok = ok && g0.y == tag && g0.w == tag && g1.y == tag && g1.w == tag;
                }
            if (__all(ok)) break;
            if (spins > WSE_SPINS_GLB || lds_ld(misc + W_ABORT) || ((spins & 255u) == 255u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (spins > WSE_SPINS_GLB) wse_fail(a, misc, 0x200u | op.in_vec);
                break;
            }
            // not there yet: do not sweep the whole share again and again (polling-cost: 2560 waves re-reading 2 - 8 KB each slow every producer down);
            // watch ONE granule per lane — the last one of each lane's quarter of the first block — and sweep again when those carry the tag
            for (unsigned sp2 = 0;; ++sp2) {
                const unsigned long long x = __hip_atomic_load(as_global<const unsigned long long>(vin.p) + blk[0][0] * 256 + lane * 4 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((uint32_t) (x >> 32) == tag) || sp2 > WSE_SPINS_GLB || lds_ld(misc + W_ABORT)) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
    } else {
#pragma unroll
        for (int bt = 0; bt < NBAT; ++bt)
#pragma unroll
            for (int b = 0; b < NB; ++b) ap[bt].v[b] = *(const float4 *) (as_global<const float>(vin.p) + blk[bt][b] * 256 + lane * 4);
    }
    if (cw == 0 && first == cw) tl_stamp(a, op, BAMD_WSE_TL_VALID);
    if (NORM) {      // sum of squares in double, ggml.c:11874-11877 (tree here; f32_rounding_safe decides whether the order can matter)
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 v = ap[0].v[b];
            if (ap[0].okmask >> b & 1) { s += (double) (v.x * v.x); s += (double) (v.y * v.y); s += (double) (v.z * v.z); s += (double) (v.w * v.w); }
        }
        s = wave_sum_f64(s);
        if (lane == 0) red[cw] = s;
        cbar(a, misc, W_CBAR, cb, nc, 0x300u);
        double tot = 0.0;
        for (int w2 = 0; w2 < nc; ++w2) tot += red[w2];
        double md = (K & (K - 1)) == 0 ? tot * (1.0 / (double) K) : tot / (double) K;
        float mean = (float) md;
        if (!f32_rounding_safe(md, BAMD_F64_GUARD_ULPS(K))) {            // uniform over the consumers (same tot); rare: the reference's order, one lane
            cbar(a, misc, W_CBAR, cb, nc, 0x301u);                       // everybody has read red[]
            if (cw == 0 && lane == 0) {
                double sq = 0.0;
                for (int i = 0; i < K; ++i) {
                    float xv;
                    if (vin.gran) xv = __uint_as_float((uint32_t) __hip_atomic_load(as_global<const unsigned long long>(vin.p) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    else xv = as_global<const float>(vin.p)[i];
                    sq += (double) (xv * xv);
                }
                red[15] = sq;
            }
            cbar(a, misc, W_CBAR, cb, nc, 0x302u);
            md = red[15] / (double) K; mean = (float) md;
        }
        scale = 1.0f / sqrtf(mean + a.eps);
    }
#pragma unroll
    for (int bt = 0; bt < NBAT; ++bt) ap[bt].template quantize_batch<NB>(scale, K, first + bt * NB * nc, ap_.q8, ap_.S, ap_.yd, nc, nb);
    if (NORM) break;
  }
    cbar(a, misc, W_CBAR, cb, nc, 0x303u);
    if (cw == 0) { tl_stamp(a, op, BAMD_WSE_TL_ACTREADY); if ((a.thin & 1) && lane == 0) lds_st(misc + W_GATHERING, 0u); }
}
// blocks per wave and batch: the share of a wave (ceil(nb / nc)) in as few equal batches of <= 4 as possible
__device__ __forceinline__ void wse_gather_any(const bamd_wse_args & a, const bamd_wse_op & op, unsigned char * smem, uint32_t * misc, int cw, CBar & cb, uint32_t tagbase) {
    const int share = ((int) op.nb + a.nc - 1) / a.nc, nbat = (share + 3) / 4, per = (share + nbat - 1) / nbat;
#define WSE_G(N_, NB_, NBAT_) wse_gather<N_, NB_, NBAT_>(a, op, smem, misc, cw, cb, tagbase)
    if (op.act & BAMD_WSE_ACT_NORM) {
        switch (per) { case 1: WSE_G(true, 1, 1); break; case 2: WSE_G(true, 2, 1); break; case 3: WSE_G(true, 3, 1); break; default: WSE_G(true, 4, 1); break; }
    } else if (nbat <= 1) {
        switch (per) { case 1: WSE_G(false, 1, 1); break; case 2: WSE_G(false, 2, 1); break; case 3: WSE_G(false, 3, 1); break; default: WSE_G(false, 4, 1); break; }
    } else {
        switch (per) { case 3: WSE_G(false, 3, 2); break; default: WSE_G(false, 4, 2); break; }       // shares beyond 8 blocks: a second sweep
    }
#undef WSE_G
}

// One piece on one consumer wave: the records j = cw, cw + NC, ... : slot -> registers -> terms -> term ring.
//  * the wave's FIRST record is read out of the ring BEFORE the activations are gathered (the weights are there long before: that is the point of the
//    engine), and inside the loop the NEXT record's LDS reads are issued before the terms of the current one are computed;
//  * the loader's progress counter is cached, so that in the steady state a record costs no flag round trip; the release counter of the term chunk is
//    requested ahead of the terms and looked at behind them;
//  * hand-backs need no hardware wait (the LDS serves a wave's instructions in order: a ds_add behind the reads of a record executes after them).
// Term chunk layout (8 records, 4608 B): [pair p][lane] float4 {fs, pm of record 2p | fs, pm of record 2p + 1}, then [pair p][row] float4 {d, dmin | d, dmin}:
// the chainer reads a chunk with 8 conflict-free 16-byte reads per lane.
template <int TYPE>
__device__ __forceinline__ void wse_piece(const bamd_wse_args & a, const bamd_wse_op & op, unsigned char * smem, uint32_t * misc, int cw, uint32_t & filled, CBar & cb, uint32_t tagbase) {
    typedef typename RecOf<TYPE>::type REC;
    constexpr int RECB = TYPE == BAMD_Q4_K ? BAMD_RECB_Q4K : TYPE == BAMD_Q5_K ? BAMD_RECB_Q5K : 1680;
    const int lane = opaque((int) threadIdx.x & 63);
    const uint32_t nc = (uint32_t) a.nc, nb = op.nb, nrec = op.ntask * nb, ns = (uint32_t) a.ns, tr = (uint32_t) a.tr, ckm = (tr >> 3) - 1u, rps = op.rps;
    const uint32_t trs = (uint32_t) (31 - __builtin_clz(tr));
    const ActPtrs ap = act_ptrs(smem, a, op.actbuf, (int) nb);
    unsigned char * terms = smem + a.off_terms;
    uint32_t j = (uint32_t) cw;
    const bool any = j < nrec;
    const bool dbg = (a.thin & 2) && cw == 0 && a.tl;      // experiment builds of the timeline: ticks consumer 0 spends waiting for the loader / the chainer
    unsigned long long w_fill = 0, w_rel = 0;
    // (slot of the piece, index in the slot, super-block, ring position) of record j, advanced incrementally: no divisions in the loop
    uint32_t s = 0, r = 0, sb = 0, slot = 0;
    REC Rn;
    if (any) {
        s = j / rps; r = j - s * rps; sb = j % nb; slot = (op.gs0 + s) % ns;
        if ((int) (filled - (op.gs0 + s + 1u)) < 0) filled = lds_wait_ge(a, misc, misc + W_FILLED, op.gs0 + s + 1u, 0x400u);
        lds_rec(Rn, smem + slot * BAMD_WSE_SLOT + r * RECB, lane);
        if (lane == 0) atomicAdd(misc + W_FREE + slot, 1u);
    }
    if (op.act & BAMD_WSE_ACT_GATHER) wse_gather_any(a, op, smem, misc, cw, cb, tagbase);
    if (any) {
        for (;;) {
            REC R = Rn;
            const uint32_t sb_cur = sb, grec = op.grec0 + j, ck = (grec >> 3) & ckm, gen = grec >> trs;
            const uint32_t relv = __hip_atomic_load(misc + W_REL + ck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // looked at behind the terms
            j += nc;
            const bool more = j < nrec;
            if (more) {
                r += nc; while (r >= rps) { r -= rps; ++s; slot = slot + 1u == ns ? 0u : slot + 1u; }
                sb += nc; while (sb >= nb) sb -= nb;
                if ((int) (filled - (op.gs0 + s + 1u)) < 0) {
                    const unsigned long long t0 = dbg ? wall_clock64() : 0ull;
                    filled = lds_wait_ge(a, misc, misc + W_FILLED, op.gs0 + s + 1u, 0x400u);
                    if (dbg) w_fill += wall_clock64() - t0;
                }
                lds_rec(Rn, smem + slot * BAMD_WSE_SLOT + r * RECB, lane);
                if (lane == 0) atomicAdd(misc + W_FREE + slot, 1u);
            }
            const Terms T = block_terms(R, (int) sb_cur, lane, ap.q8, ap.S, ap.yd);
            if ((int) ((uint32_t) __builtin_amdgcn_readfirstlane((int) relv) - gen) < 0) {       // the chunk's previous tenants have not been chained yet
                const unsigned long long t0 = dbg ? wall_clock64() : 0ull;
                lds_wait_ge(a, misc, misc + W_REL + ck, gen, 0x401u);
                if (dbg) w_rel += wall_clock64() - t0;
            }
            const uint32_t u = grec & 7u;
            unsigned char * t = terms + ck * (8 * BAMD_WSE_TERM_BYTES) + (u >> 1) * 1024u + (u & 1u) * 8u;
            *(float2 *) (t + lane * 16) = make_float2(T.fs, T.pm);
            if ((lane & 7) == 0) *(float2 *) (terms + ck * (8 * BAMD_WSE_TERM_BYTES) + 4096u + (u >> 1) * 128u + (u & 1u) * 8u + (lane >> 3) * 16) = make_float2(T.d, T.dmin);
            asm volatile("" ::: "memory");
            if (lane == 0) atomicAdd(misc + W_CHUNK + ck, 1u);
            if (grec == op.grec0 && cw == 0) tl_stamp(a, op, BAMD_WSE_TL_FIRSTREC);
            if (!more) break;
        }
    }
    if (cw == 0) tl_stamp(a, op, BAMD_WSE_TL_LASTREC);
    if (dbg && op.tlslot != 255 && lane == 0) { unsigned long long * row = a.tl + ((size_t) blockIdx.x * a.tl_ops + op.tlslot) * 8; row[BAMD_WSE_TL_VALID] = w_fill; row[BAMD_WSE_TL_LOADED] = w_rel; }
}

// attention of query head h on consumer waves 0..7 (attn_fused_body): q / k / v of this token come from the QKV granules through an LDS stage
struct AttnEnvWSE {
    int tid, wave, nthr;
    const bamd_wse_args * a; uint32_t * misc; CBar * cb; const float * stage;      // stage: [q hd | k hd | v hd] f32 in LDS
    __device__ __forceinline__ void sync() const { cbar(*a, misc, W_CBAR8, *cb, 8, 0x500u); }
    __device__ __forceinline__ float2 qk2(const bamd_attn_args &, int role, int, int, int hd, int rp) const { return *(const float2 *) (stage + (role == 1 ? hd : 0) + 2 * rp); }
    __device__ __forceinline__ float vel(const bamd_attn_args &, int, int hd, int i) const { return stage[2 * hd + i]; }
};
template <int LG>
__device__ __forceinline__ void wse_attention(const bamd_wse_args & a, const bamd_wse_op & op, unsigned char * smem, uint32_t * misc, int cw, CBar & cb8, uint32_t tagbase) {
    constexpr int hd = LG * 64;
    const int tid = opaque((int) threadIdx.x - 64 * (1 + a.nch)), lane = tid & 63, h = (int) blockIdx.x, hk = h / a.gq, Hq = a.H, Ekv = (Hq / a.gq) * hd;
    unsigned char * scratch = smem + a.off_attn;
    const int ld = a.at.lds_ld;
    float * stage = (float *) (scratch + (size_t) ld * 8);
    if (cw == 0) tl_stamp(a, op, BAMD_WSE_TL_GATHER0);
    {   // q head h | k head hk | v head hk of the QKV vector: 3 hd granules, thread t takes t and t + 512
        const bamd_wse_vec vin = a.vec[op.in_vec];
        const uint32_t tag = tagbase | op.in_tag;
        const unsigned long long * g = as_global<const unsigned long long>(vin.p);
        int idx[2]; bool use[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + u * 512; use[u] = t < 3 * hd;
            const int tt = use[u] ? t : 0, part = tt / hd, d = tt - part * hd;
            idx[u] = part == 0 ? h * hd + d : part == 1 ? Hq * hd + hk * hd + d : Hq * hd + Ekv + hk * hd + d;
        }
        uint32_t val[2] = { 0u, 0u };
        for (unsigned spins = 0;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned long long x = __hip_atomic_load(g + idx[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                val[u] = (uint32_t) x; ok = ok && (!use[u] || (uint32_t) (x >> 32) == tag);
            }
            if (__all(ok)) break;
            if (spins > WSE_SPINS_GLB || lds_ld(misc + W_ABORT) || (spins & 255u) == 255u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                if (spins > WSE_SPINS_GLB) wse_fail(a, misc, 0x210u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) if (use[u]) stage[tid + u * 512] = __uint_as_float(val[u]);
    }
    cbar(a, misc, W_CBAR8, cb8, 8, 0x501u);
    if (cw == 0) tl_stamp(a, op, BAMD_WSE_TL_ACTREADY);
    bamd_attn_args at = a.at;
    at.kc = as_global<unsigned short>(a.kc[op.layer]); at.vc = as_global<unsigned short>(a.vc[op.layer]);
    AttnEnvWSE env; env.tid = tid; env.wave = cw; env.nthr = 512; env.a = &a; env.misc = misc; env.cb = &cb8; env.stage = stage;
    attn_fused_body<LG, true, AttnEnvWSE>(at, a.gq, h, 0, scratch, as_global<uint32_t>(a.vec[op.out_vec].p), tagbase | op.out_tag, env);
    if (cw == 0) tl_stamp(a, op, BAMD_WSE_TL_PUBLISHED);
}

template <int LG>
__device__ __forceinline__ void wse_consumer(const bamd_wse_args & a, const bamd_wse_op * ops, unsigned char * smem, uint32_t * misc, int cw) {
    CBar cb, cb8;
    uint32_t filled = 0;                   // cached copy of the loader's progress counter
    const uint32_t tagbase = wse_tagbase(a.st);
    for (int io = 0;; ++io) {
        const bamd_wse_op op = ops[io];
        if (op.kind == BAMD_WSE_END) break;
        if (op.kind == BAMD_WSE_ATTN) {
            if ((int) blockIdx.x < a.H && cw < 8) {
                for (int k = 0; k < a.nch; ++k) lds_wait_ge(a, misc, misc + W_PIECES + k, op.rps, 0x402u);   // the attention scratch aliases the term ring: every chainer must be through the pieces before this op (planner: rps = their number)
                if (LG > 0) wse_attention<(LG > 0 ? LG : 1)>(a, op, smem, misc, cw, cb8, tagbase);
            }
            continue;
        }
        if (op.type == BAMD_Q4_K) wse_piece<BAMD_Q4_K>(a, op, smem, misc, cw, filled, cb, tagbase);
        else if (op.type == BAMD_Q5_K) wse_piece<BAMD_Q5_K>(a, op, smem, misc, cw, filled, cb, tagbase);
        else wse_piece<BAMD_Q6_K>(a, op, smem, misc, cw, filled, cb, tagbase);
    }
}

// ---- CHAINER -----------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float vec_value(const bamd_wse_args & a, const bamd_wse_vec & v, uint32_t row, uint32_t tag, uint32_t * misc, uint32_t code) {
    if (!v.gran) return as_global<const float>(v.p)[row];
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long x = __hip_atomic_load(as_global<const unsigned long long>(v.p) + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t) (x >> 32) == tag) return __uint_as_float((uint32_t) x);
        if (spins > WSE_SPINS_GLB || lds_ld(misc + W_ABORT)) { if (spins > WSE_SPINS_GLB) wse_fail(a, misc, code); return 0.f; }
        __builtin_amdgcn_s_sleep(2);
    }
}
__device__ __forceinline__ void vec_store(const bamd_wse_vec & v, uint32_t row, float val, uint32_t tag) {
    if (v.gran) __hip_atomic_store(as_global<unsigned long long>(v.p) + row, ((unsigned long long) tag << 32) | __float_as_uint(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else as_global<float>(v.p)[row] = val;
}
// chainer k of NCH takes the row-groups t = k, k + NCH, ... of every piece (the chains of different row-groups are independent; a chunk of 8 records
// belongs to one row-group, so exactly one chainer reads and releases it)
template <int TYPE>
__device__ __forceinline__ void wse_chain_piece(const bamd_wse_args & a, const bamd_wse_op & op, unsigned char * smem, uint32_t * misc, uint32_t tagbase, unsigned long long & best, int k) {
    const int lane = opaque((int) threadIdx.x & 63), r8 = lane >> 3;
    const uint32_t nb = op.nb, tr = (uint32_t) a.tr, ckm = (tr >> 3) - 1u, nch = (uint32_t) a.nch;            // tr: a power of two (planner)
    const uint32_t trs = (uint32_t) (31 - __builtin_clz(tr));
    unsigned char * terms = smem + a.off_terms;
    float * stash = (float *) ((unsigned char *) misc + W_STASH_BYTES);
    const bamd_wse_vec vout = a.vec[op.out_vec];
    if ((uint32_t) k < op.ntask) {
        // residual rows of this chainer's tasks, fetched now: old by the time the first chain ends
        float resv[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        if (op.epi == BAMD_WSE_EPI_ADD) {
            const bamd_wse_vec vr = a.vec[op.res_vec];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t row = op.row0 + (uint32_t) t * 8u + (uint32_t) r8;
                if ((uint32_t) t < op.ntask && (uint32_t) t % nch == (uint32_t) k && row < op.nvalid && (lane & 7) == 0) resv[t] = vec_value(a, vr, row, tagbase | op.res_tag, misc, 0x600u);
            }
        }
        // the terms of the next chunk are requested before the chain of the current one runs (a chunk = 8 records = 8 16-byte LDS reads per lane);
        // a chunk is handed back to the consumers right behind its reads (in-order LDS: the two stores execute after them)
        float4 fpn[4], ddn[4];
        uint32_t pk_cnt = 0, pk_rel = 0;                      // the flags of the chunk after next, requested one chain early: no flag round trip per chunk in the steady state
        auto peek = [&](uint32_t g) {
            const uint32_t ck = (g >> 3) & ckm;
            pk_cnt = __hip_atomic_load(misc + W_CHUNK + ck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pk_rel = __hip_atomic_load(misc + W_REL + ck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto fetch = [&](uint32_t g, bool peeked) {
            const uint32_t ck = (g >> 3) & ckm;
            const bool ready = peeked && (uint32_t) __builtin_amdgcn_readfirstlane((int) pk_cnt) == 8u && (uint32_t) __builtin_amdgcn_readfirstlane((int) pk_rel) == g >> trs;
            if (!ready) {
                lds_wait<true>(a, misc, misc + W_REL + ck, g >> trs, 0x602u);     // (two chainers: the other one may still own the slot's previous tenant — a full count must be THIS chunk's)
                lds_wait<true>(a, misc, misc + W_CHUNK + ck, 8u, 0x601u);
            }
            asm volatile("" ::: "memory");
            const unsigned char * p = terms + ck * (8 * BAMD_WSE_TERM_BYTES);
#pragma unroll
            for (int q = 0; q < 4; ++q) { fpn[q] = *(const float4 *) (p + q * 1024 + lane * 16); ddn[q] = *(const float4 *) (p + 4096 + q * 128 + r8 * 16); }
            asm volatile("" ::: "memory");
            lds_st(misc + W_CHUNK + ck, 0u);
            asm volatile("" ::: "memory");
            lds_st(misc + W_REL + ck, (g >> trs) + 1u);
        };
        // this chainer's chunks in order: (t, c) -> next chunk, or none
        auto next_of = [&](uint32_t t, uint32_t c, uint32_t & g) -> bool {
            if (c + 8u < nb) { g = op.grec0 + t * nb + c + 8u; return true; }
            if (t + nch < op.ntask) { g = op.grec0 + (t + nch) * nb; return true; }
            return false;
        };
        fetch(op.grec0 + (uint32_t) k * nb, false);
        bool peeked = false;
        for (uint32_t t = (uint32_t) k; t < op.ntask; t += nch) {
            RowAcc A = { 0.f, 0.f };
            for (uint32_t c = 0; c < nb; c += 8) {
                float4 fp[4], dd[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { fp[q] = fpn[q]; dd[q] = ddn[q]; }
                uint32_t g1 = 0, g2 = 0;
                const bool has1 = next_of(t, c, g1);
                if (has1) {
                    fetch(g1, peeked);
                    const uint32_t t1 = c + 8u < nb ? t : t + nch, c1 = c + 8u < nb ? c + 8u : 0u;
                    peeked = next_of(t1, c1, g2);
                    if (peeked) peek(g2);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { chain_step<TYPE>(A, dd[q].x, fp[q].x, dd[q].y, fp[q].y); chain_step<TYPE>(A, dd[q].z, fp[q].z, dd[q].w, fp[q].w); }
                if (t == 0 && c == 0) tl_stamp(a, op, BAMD_WSE_TL_CHAIN0);
            }
            const float val = finish_row<TYPE>(A);
            const uint32_t row = op.row0 + t * 8u + (uint32_t) r8;
            if ((lane & 7) == 0 && row < op.nvalid) {
                if (op.epi == BAMD_WSE_EPI_GATE) stash[t * 8u + (uint32_t) r8] = val;
                else {
                    float o = val;
                    if (op.epi == BAMD_WSE_EPI_ADD) { float rv = 0.f;
#pragma unroll
                        for (int q = 0; q < 8; ++q) rv = (uint32_t) q == t ? resv[q] : rv;
                        o = val + rv; }
                    else if (op.epi == BAMD_WSE_EPI_UP) o = v_silu(stash[t * 8u + (uint32_t) r8]) * val;
                    vec_store(vout, row, o, tagbase | op.out_tag);
                    if (op.epi == BAMD_WSE_EPI_ARGMAX) { const unsigned long long key = argmax_key(o, (int) row); best = key > best ? key : best; }
                }
            }
        }
    }
    tl_stamp(a, op, BAMD_WSE_TL_PUBLISHED);
}
__device__ __forceinline__ void wse_chainer(const bamd_wse_args & a, const bamd_wse_op * ops, unsigned char * smem, uint32_t * misc, int k) {
    const uint32_t tagbase = wse_tagbase(a.st);
    unsigned long long best = 0ull; bool any_best = false;
    uint32_t pieces = 0;
    for (int io = 0;; ++io) {
        const bamd_wse_op op = ops[io];
        if (op.kind == BAMD_WSE_END) break;
        if (op.kind != BAMD_WSE_MATVEC) continue;
        if (op.type == BAMD_Q4_K) wse_chain_piece<BAMD_Q4_K>(a, op, smem, misc, tagbase, best, k);
        else if (op.type == BAMD_Q5_K) wse_chain_piece<BAMD_Q5_K>(a, op, smem, misc, tagbase, best, k);
        else wse_chain_piece<BAMD_Q6_K>(a, op, smem, misc, tagbase, best, k);
        any_best = any_best || op.epi == BAMD_WSE_EPI_ARGMAX;
        asm volatile("" ::: "memory");
        lds_st(misc + W_PIECES + k, ++pieces);
    }
    if (any_best && a.best_key) {
        // wave maximum of the keys (64-bit), then one atomic per chainer
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = ((unsigned long long) (uint32_t) __shfl_xor((int) (uint32_t) (best >> 32), off) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) best, off);
            best = o > best ? o : best;
        }
        if ((threadIdx.x & 63) == 0 && best) atomicMax(a.best_key, best);
    }
}

// LG = head_dim / 64 of the attention role (0: a program without attention ops): one instance per head size, so that the register budget of a
// launch (16 waves per CU: 128 VGPRs) is that of ITS attention body, not of the largest
// MAXT = threads per workgroup the instance is compiled for: 768 (loader + chainers + consumers = 12 waves: 3 per SIMD, 168 VGPRs — what the
// attention body of head_dim 128 needs without spilling) or 1024 (16 waves, 128 VGPRs)
template <int LG, int MAXT>
__global__ void __launch_bounds__(MAXT) wse_kernel(const bamd_wse_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t * misc = (uint32_t *) (smem + a.off_misc);
    for (int i = (int) threadIdx.x; i < BAMD_WSE_MISC_BYTES / 4; i += (int) blockDim.x) misc[i] = 0u;
    __syncthreads();
    const bamd_wse_op * ops = a.ops + (size_t) blockIdx.x * a.ops_per_cu;
    const int wave = wave_id();
    // the loader and the chainers issue few instructions, every one of them on the critical path of ten consumer waves: they go first on their SIMDs
    if (wave == 0) { __builtin_amdgcn_s_setprio(3); wse_loader(a, ops, smem, misc); }
    else if (wave <= a.nch) { __builtin_amdgcn_s_setprio(2); wse_chainer(a, ops, smem, misc, wave - 1); }
    else if (wave - 1 - a.nch < a.nc) wse_consumer<LG>(a, ops, smem, misc, wave - 1 - a.nch);
}

// hardware facts the loader relies on, probed once on the device (bamd_wse_selftest): an LDS-DMA destination above 64 KiB (M0 carries the full
// LDS byte address), and what the instruction's offset field moves (global address only, or the LDS address as well).
// src: u32 word i holds i.  out[0]: words correct of 256 at LDS 100 KiB; out[1..4]: for offset:1024 with M0 = 8 KiB and source byte 4096: the first
// word found at LDS 8 KiB, at 9 KiB (0xffffffff = untouched) — word 1280 there means the offset moved both addresses.
__global__ void wse_selftest_kernel(const uint8_t * src, uint32_t * out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    uint32_t * w = (uint32_t *) smem;
    for (int i = lane; i < 160 * 256; i += 64) w[i] = 0xffffffffu;
    __syncthreads();
    const uint32_t base = (uint32_t) (size_t) (wse_lds_vp) smem;
    dma1k(src + lane * 16, base + 100u * 1024u);
    {
        const uint8_t * p = src + 4096 + lane * 16; const uint32_t dst = (uint32_t) __builtin_amdgcn_readfirstlane((int) (base + 8192u));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:1024 nt" :: "v"(p), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int ok = 0;
    for (int i = lane; i < 256; i += 64) ok += w[100 * 256 + i] == (uint32_t) i ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) ok += __shfl_xor(ok, off);
    if (lane == 0) { out[0] = (uint32_t) ok; out[1] = w[8192 / 4]; out[2] = w[9216 / 4]; out[3] = w[8192 / 4 + 255]; out[4] = w[9216 / 4 + 255]; out[5] = base; }
}
int bamd_wse_selftest_launch(const uint8_t * src, uint32_t * out, hipStream_t s) {
    if (hipFuncSetAttribute((const void *) wse_selftest_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { (void) hipGetLastError(); return 1; }
    hipLaunchKernelGGL(wse_selftest_kernel, dim3(1), dim3(64), 160 * 1024, s, src, out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// the engine kernels use more dynamic LDS than the default limit: raise it once per instance (160 KiB less the static arrays of attn_fused_body:
// BAMD_WSE_LDS_LIMIT of the planner).  Not allowed while a stream is being captured, so the
// host calls this when it plans (head_dim = 0: programs without attention ops)
static bool g_wse_attr[2][5] = { { false, false, false, false, false }, { false, false, false, false, false } };
int bamd_wse_setup(int head_dim, size_t * static_lds) {
    const int lg = head_dim >> 6;
    if (lg < 0 || lg > 4 || (head_dim & 63)) return 1;
    if (static_lds) {                                      // the instance's static LDS (the arrays of attn_fused_body): the planner's budget must leave room for it
        hipFuncAttributes fa;
        const void * fn = lg == 0 ? (const void *) wse_kernel<0, 768> : lg == 1 ? (const void *) wse_kernel<1, 768> : lg == 2 ? (const void *) wse_kernel<2, 768> :
                          lg == 3 ? (const void *) wse_kernel<3, 768> : (const void *) wse_kernel<4, 768>;
        if (hipFuncGetAttributes(&fa, fn) != hipSuccess) { (void) hipGetLastError(); return 1; }
        *static_lds = fa.sharedSizeBytes;
    }
#define WSE_ATTR(LG_) do { \
        if (!g_wse_attr[0][LG_] && hipFuncSetAttribute((const void *) wse_kernel<LG_, 768>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2560) != hipSuccess) { (void) hipGetLastError(); return 1; } \
        if (!g_wse_attr[1][LG_] && hipFuncSetAttribute((const void *) wse_kernel<LG_, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2560) != hipSuccess) { (void) hipGetLastError(); return 1; } \
        g_wse_attr[0][LG_] = g_wse_attr[1][LG_] = true; } while (0)
    switch (lg) { case 0: WSE_ATTR(0); break; case 1: WSE_ATTR(1); break; case 2: WSE_ATTR(2); break; case 3: WSE_ATTR(3); break; default: WSE_ATTR(4); break; }
#undef WSE_ATTR
    return 0;
}
int bamd_launch_wse(const bamd_wse_args & a, int n_cu, size_t lds_bytes, hipStream_t s) {
    if (a.ns < 3 || a.ns > BAMD_WSE_MAX_SLOTS || a.tr < 8 || a.tr > BAMD_WSE_MAX_TERMS || (a.tr & (a.tr - 1)) || a.nc < 8 || a.nch < 1 || a.nch > 2 || 1 + a.nch + a.nc > 16) return 1;
    const int lg = a.H > 0 ? a.at.hd >> 6 : 0;
    if (lg < 0 || lg > 4 || (a.H > 0 && (a.at.hd & 63))) return 1;
    const int big = 1 + a.nch + a.nc > 12 ? 1 : 0;
    if (!g_wse_attr[big][lg]) return 1;                 // bamd_wse_setup(head_dim) must have run (outside any stream capture)
    const dim3 grid(n_cu), block(64 * (1 + a.nch + a.nc));
#define WSE_GO(LG_, MT_) hipLaunchKernelGGL((wse_kernel<LG_, MT_>), grid, block, lds_bytes, s, a)
#define WSE_GO_LG(MT_) do { switch (lg) { case 0: WSE_GO(0, MT_); break; case 1: WSE_GO(1, MT_); break; case 2: WSE_GO(2, MT_); break; case 3: WSE_GO(3, MT_); break; default: WSE_GO(4, MT_); break; } } while (0)
    if (big) WSE_GO_LG(1024); else WSE_GO_LG(768);
#undef WSE_GO_LG
#undef WSE_GO
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
