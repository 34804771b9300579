// bamd_wse_plan.cpp — plans the per-CU programs of the weight-stream engine (bamd_wse.h).  Pure host code: no HIP calls, so the plan is
// unit-tested without a GPU (tests/test_wse_plan.py through bamd_wse_plan_describe).
//
// The graph being planned is build_llama's (cpp/src/llama.cpp:8781-8925): per layer QKV <- rms_norm(x), attention, wo + residual,
// gate / up <- rms_norm(x2), down + residual; then output_norm + lm_head.  Every matrix is cut into contiguous runs of row-groups, one run per
// CU (so a CU's records of a piece are ONE contiguous byte range of the wave-stream copy: the loader's fills are 16 KiB sequential reads), and
// a run is split where it crosses a matrix boundary (wq | wk | wv may differ in type).
#include "bamd_wse.h"
#include "bamd_formats.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace {
struct CuProg { std::vector<bamd_wse_op> ops; uint32_t gs = 0, grec = 0; int gathers = 0; };

static size_t act_bytes(int nb) { return (((size_t) nb * (256 + 32 + 4)) + 15) & ~(size_t) 15; }      // q8 | S | yd of BAMD_ACT_RED_OFF

struct Planner {
    int n_cu, nc;
    std::vector<CuProg> cu;
    size_t act_need[2] = { 0, 0 };
    int tl_ops = 0;
    char why[160] = { 0 };
    bool fail(const char * m) { snprintf(why, sizeof why, "%s", m); return false; }

    // one op over a list of matrices whose rows are concatenated in the output vector (QKV: wq | wk | wv); in_*: the activation
    bool matvec(const bamd_wse_mat * const * mats, int nmat, int act, int in_vec, int in_tag, uint64_t normw, int epi, int out_vec, int out_tag,
                int res_vec, int res_tag, int layer, int tlslot, bool continues_prev_act) {
        int total_rg = 0;
        for (int i = 0; i < nmat; ++i) {
            const bamd_wse_mat & m = *mats[i];
            if (!bamd_is_kquant(m.type)) return fail("a matrix is not Q4_K / Q5_K / Q6_K");
            if (m.K % 2048) return fail("K / 256 is not a multiple of 8 (the chainer takes the terms in chunks of 8 records)");
            if (m.K != mats[0]->K) return fail("matrices of one op differ in K");
            if (m.nrows_pad % 8) return fail("stream rows not a multiple of 8");
            total_rg += m.nrows_pad / 8;
        }
        const int nb = mats[0]->K / 256;
        if ((act & BAMD_WSE_ACT_NORM) && (nb + nc - 1) / nc > 4) return fail("RMSNorm prologue: more than 4 blocks per consumer wave");
        for (int c = 0; c < n_cu; ++c) {
            const long lo = (long) total_rg * c / n_cu, hi = (long) total_rg * (c + 1) / n_cu;
            CuProg & P = cu[c];
            bool first_piece = !continues_prev_act;
            long base_rg = 0; uint32_t row_off = 0;
            for (int i = 0; i < nmat; ++i) {
                const bamd_wse_mat & m = *mats[i];
                const long nrg = m.nrows_pad / 8;
                const long a = std::max(lo, base_rg), b = std::min(hi, base_rg + nrg);
                if (a < b) {
                    bamd_wse_op op; memset(&op, 0, sizeof op);
                    const int recb = bamd_record_bytes(m.type);
                    op.kind = BAMD_WSE_MATVEC; op.type = (uint32_t) m.type; op.nb = (uint32_t) nb; op.ntask = (uint32_t) (b - a);
                    op.src = m.stream + (uint64_t) (a - base_rg) * nb * recb;
                    op.normw = normw;
                    op.row0 = row_off + (uint32_t) (a - base_rg) * 8; op.nvalid = row_off + (uint32_t) m.nrows;
                    const uint32_t nrec = op.ntask * op.nb, rmax = (uint32_t) (BAMD_WSE_SLOT / recb), nslots = (nrec + rmax - 1) / rmax;
                    op.rps = (nrec + nslots - 1) / nslots;
                    op.gs0 = P.gs; op.grec0 = P.grec;
                    P.gs += (nrec + op.rps - 1) / op.rps; P.grec += nrec;
                    if (first_piece) {
                        op.act = (uint8_t) (BAMD_WSE_ACT_GATHER | act); op.actbuf = (uint8_t) (P.gathers & 1); P.gathers++;
                        act_need[op.actbuf] = std::max(act_need[op.actbuf], act_bytes(nb));
                    } else {
                        op.act = BAMD_WSE_ACT_REUSE; op.actbuf = (uint8_t) ((P.gathers - 1) & 1);
                        if (P.gathers == 0) return fail("internal: reuse without a gather");
                    }
                    first_piece = false;
                    op.in_vec = (uint8_t) in_vec; op.in_tag = (uint8_t) in_tag; op.epi = (uint8_t) epi; op.out_vec = (uint8_t) out_vec; op.out_tag = (uint8_t) out_tag;
                    op.res_vec = (uint8_t) res_vec; op.res_tag = (uint8_t) res_tag; op.layer = (uint8_t) layer; op.tlslot = (uint8_t) tlslot;
                    if ((epi == BAMD_WSE_EPI_GATE || epi == BAMD_WSE_EPI_UP) && op.ntask * 8 > BAMD_WSE_STASH) return fail("gate / up: more than 16 row-groups per CU");
                    if (epi == BAMD_WSE_EPI_ADD && op.ntask > 8) return fail("residual epilogue: more than 8 row-groups per CU");
                    P.ops.push_back(op);
                }
                base_rg += nrg; row_off += (uint32_t) m.nrows;
            }
        }
        return true;
    }
    void attn(int layer, int tlslot) {
        for (int c = 0; c < n_cu; ++c) {
            bamd_wse_op op; memset(&op, 0, sizeof op);
            op.kind = BAMD_WSE_ATTN; op.layer = (uint8_t) layer; op.in_vec = BAMD_WSE_V_QKV; op.in_tag = (uint8_t) layer; op.out_vec = BAMD_WSE_V_ATT; op.out_tag = (uint8_t) layer;
            op.grec0 = cu[c].grec; op.gs0 = cu[c].gs; op.tlslot = (uint8_t) tlslot;
            uint32_t pieces = 0; for (const bamd_wse_op & o : cu[c].ops) pieces += o.kind == BAMD_WSE_MATVEC ? 1u : 0u;
            op.rps = pieces;                                          // the chainers must be through these before the attention scratch (aliasing the term ring) is written
            cu[c].ops.push_back(op);
        }
    }
    int tr_force = 0;                                      // experiments (BAMD_WSE_TR): term-ring records, a power of two (0: 32, or more when the ring has 8 slots)
    int finish(bamd_wse_plan * plan, size_t attn_lds, int lds_limit) {
        size_t mx = 0;
        for (auto & P : cu) mx = std::max(mx, P.ops.size());
        plan->n_cu = n_cu; plan->nc = nc; plan->ops_per_cu = (int) mx + 1; plan->tl_ops = tl_ops;
        // LDS: ring | act 0 | act 1 | term ring (the attention scratch aliases it: the consumers wait for the chainer before an ATTN op) | control words
        const size_t fixed = act_need[0] + act_need[1] + BAMD_WSE_MISC_BYTES;
        int tr = tr_force >= 8 && tr_force <= BAMD_WSE_MAX_TERMS && !(tr_force & (tr_force - 1)) ? tr_force : 32;
        size_t terms = std::max((size_t) tr * BAMD_WSE_TERM_BYTES, (attn_lds + 15) & ~(size_t) 15);
        long ring = (long) lds_limit - (long) fixed - (long) terms;
        int ns = (int) (ring / BAMD_WSE_SLOT);
        if (ns > 8 && !tr_force) {            // room to spare: a deeper term ring (a power of two: the kernel masks record numbers)
            ns = 8;
            const size_t room = ((size_t) lds_limit - fixed - (size_t) ns * BAMD_WSE_SLOT) / BAMD_WSE_TERM_BYTES;
            tr = room >= 128 ? 128 : room >= 64 ? 64 : 32;
            terms = std::max((size_t) tr * BAMD_WSE_TERM_BYTES, (attn_lds + 15) & ~(size_t) 15);
        }
        if (ns > 8) ns = 8;
        if (ns < 3) { snprintf(plan->why, sizeof plan->why, "LDS: %zu B of activations + %zu B of terms leave %d ring slots", fixed, terms, ns); return 1; }
        plan->ns = ns; plan->tr = tr;
        plan->off_act[0] = (uint32_t) ((size_t) ns * BAMD_WSE_SLOT); plan->off_act[1] = plan->off_act[0] + (uint32_t) act_need[0];
        plan->off_terms = plan->off_act[1] + (uint32_t) act_need[1]; plan->off_attn = plan->off_terms;
        plan->off_misc = plan->off_terms + (uint32_t) terms;
        plan->lds_bytes = plan->off_misc + BAMD_WSE_MISC_BYTES;
        plan->ops = (bamd_wse_op *) calloc((size_t) n_cu * plan->ops_per_cu, sizeof(bamd_wse_op));      // zero = BAMD_WSE_END
        if (!plan->ops) { snprintf(plan->why, sizeof plan->why, "out of memory"); return 1; }
        for (int c = 0; c < n_cu; ++c) memcpy(plan->ops + (size_t) c * plan->ops_per_cu, cu[c].ops.data(), cu[c].ops.size() * sizeof(bamd_wse_op));
        return 0;
    }
};
}   // namespace

int bamd_wse_plan_build(bamd_wse_plan * plan, const bamd_wse_layer * L, int l0, int l1, int n_cu, int E, int H, int Hkv, int hd, int F,
                        const bamd_wse_mat * head, uint64_t head_norm, int V, size_t attn_lds, int nc, int lds_limit) {
    memset(plan, 0, sizeof *plan);
    (void) E; (void) F; (void) V; (void) Hkv; (void) hd;
    Planner pl; pl.n_cu = n_cu; pl.nc = nc; pl.cu.resize((size_t) n_cu);
    { const char * e = getenv("BAMD_WSE_TR"); pl.tr_force = e ? atoi(e) : 0; }
    bool ok = true;
    if (H > n_cu) { snprintf(plan->why, sizeof plan->why, "more query heads than CUs"); return 1; }
    if (l1 - l0 > 250) { snprintf(plan->why, sizeof plan->why, "more than 250 layers in one program (8-bit tags)"); return 1; }
    int tls = 0;
    for (int l = l0; l < l1 && ok; ++l) {
        const bamd_wse_layer & y = L[l];
        const int t = l - l0;                                          // tag byte: layer index inside this program
        const int xin = t == 0 ? BAMD_WSE_V_XIN : BAMD_WSE_V_X;
        const bamd_wse_mat * qkv[3] = { &y.wq, &y.wk, &y.wv };
        ok = ok && pl.matvec(qkv, 3, BAMD_WSE_ACT_NORM, xin, t, y.attn_norm, BAMD_WSE_EPI_STORE, BAMD_WSE_V_QKV, t, 0, 0, t, tls++, false);
        if (ok) pl.attn(t, tls++);
        const bamd_wse_mat * wo[1] = { &y.wo };
        ok = ok && pl.matvec(wo, 1, 0, BAMD_WSE_V_ATT, t, 0, BAMD_WSE_EPI_ADD, BAMD_WSE_V_X2, t, xin, t, t, tls++, false);
        const bamd_wse_mat * wg[1] = { &y.wg }, * wu[1] = { &y.wu };
        if (ok && (y.wg.nrows_pad != y.wu.nrows_pad || y.wg.K != y.wu.K)) { ok = pl.fail("gate / up shapes differ"); }
        ok = ok && pl.matvec(wg, 1, BAMD_WSE_ACT_NORM, BAMD_WSE_V_X2, t, y.ffn_norm, BAMD_WSE_EPI_GATE, 0, 0, 0, 0, t, tls++, false);
        ok = ok && pl.matvec(wu, 1, 0, BAMD_WSE_V_X2, t, 0, BAMD_WSE_EPI_UP, BAMD_WSE_V_HID, t, 0, 0, t, tls++, true);
        const bamd_wse_mat * wd[1] = { &y.wd };
        const bool last = l + 1 == l1;
        ok = ok && pl.matvec(wd, 1, 0, BAMD_WSE_V_HID, t, 0, BAMD_WSE_EPI_ADD, (last && !head) ? BAMD_WSE_V_XOUT : BAMD_WSE_V_X, t + 1, BAMD_WSE_V_X2, t, t, tls++, false);
    }
    if (ok && head) {
        const bamd_wse_mat * hm[1] = { head };
        const int t = l1 - l0;
        ok = pl.matvec(hm, 1, BAMD_WSE_ACT_NORM, t == 0 ? BAMD_WSE_V_XIN : BAMD_WSE_V_X, t, head_norm, BAMD_WSE_EPI_ARGMAX, BAMD_WSE_V_LOGITS, 0, 0, 0, t, tls++, false);
    }
    if (!ok) { memcpy(plan->why, pl.why, sizeof plan->why); return 1; }
    pl.tl_ops = tls;
    return pl.finish(plan, attn_lds, lds_limit);
}

int bamd_wse_plan_single(bamd_wse_plan * plan, const bamd_wse_mat * wA, const bamd_wse_mat * wB, uint64_t normw, int epi, int n_cu, int nc, int lds_limit) {
    memset(plan, 0, sizeof *plan);
    Planner pl; pl.n_cu = n_cu; pl.nc = nc; pl.cu.resize((size_t) n_cu);
    const bamd_wse_mat * a[1] = { wA }, * b[1] = { wB };
    const int act = normw ? BAMD_WSE_ACT_NORM : 0;
    bool ok;
    if (wB) {
        ok = pl.matvec(a, 1, act, BAMD_WSE_V_XIN, 0, normw, BAMD_WSE_EPI_GATE, 0, 0, 0, 0, 0, 0, false);
        ok = ok && pl.matvec(b, 1, 0, BAMD_WSE_V_XIN, 0, 0, BAMD_WSE_EPI_UP, BAMD_WSE_V_XOUT, 0, 0, 0, 0, 1, true);
        pl.tl_ops = 2;
    } else {
        ok = pl.matvec(a, 1, act, BAMD_WSE_V_XIN, 0, normw, epi, epi == BAMD_WSE_EPI_ARGMAX ? BAMD_WSE_V_LOGITS : BAMD_WSE_V_XOUT, 0, BAMD_WSE_V_X2, 0, 0, 0, false);
        pl.tl_ops = 1;
    }
    if (!ok) { memcpy(plan->why, pl.why, sizeof plan->why); return 1; }
    return pl.finish(plan, 0, lds_limit);
}

void bamd_wse_plan_free(bamd_wse_plan * plan) { free(plan->ops); plan->ops = nullptr; }
