"""Constructs a softmax row on which the reciprocal of the denominator sits next to a rounding boundary of its f32 — inside the band in which
f32_rounding_safe (bamd_device.h) must send the sum down the reference's sequential order — for tests/test_f64_order.py.

The row has 64 scores s_i (f16-representable, so that an attention test can produce them exactly as dot products k_i . q with q = e_0), head
dimension 64 (the scale 1/8 is an exact scaling), s_0 = 0 the maximum.  61 scores are fixed at random, three are searched: positions 47, 55, 63,
the last elements of the 8-wide groups 5, 6, 7, whose f32 partial sums ((v0+v4)+(v2+v6))+((v1+v5)+(v3+v7)) (ggml.c:2635-2644) are tabulated by
candidate; the sequential double sum of the eight partial sums is then enumerated over all triples.  The exponentials come from the oracle's ggml_v_expf restatement (oracle/, test infrastructure).
usage: python tools/f64_softmax_search.py   -> tests/golden/f64_softmax_kat.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 64
GUARD_ULPS = 2 * (N // 8) + 8                   # BAMD_F64_GUARD_ULPS(n_kv / 8)


def expf_table(po, vals):
    L = po.lib()
    return np.array([L.bo_v_expf(float(v)) for v in vals], np.float32)


def group_sum(v):
    """the reference's 8-wide partial sum, f32"""
    v = v.astype(np.float32)
    a0, a1, a2, a3 = v[0] + v[4], v[1] + v[5], v[2] + v[6], v[3] + v[7]
    return np.float32(np.float32(a0 + a2) + np.float32(a1 + a3))


def dist(rs):
    """|low 29 mantissa bits - 2^28| of doubles: how far the value is from an f32 rounding boundary, in units of the double's ulp"""
    lo = rs.view(np.uint64) & np.uint64(0x1fffffff)
    return np.abs(lo.astype(np.int64) - (1 << 28))


def denominators(e):
    """the row's denominator in the reference's order and in a few others (exps e[64] f32): dict name -> double"""
    c = np.array([group_sum(e[g * 8:g * 8 + 8]) for g in range(N // 8)], np.float64)
    seq = 0.0
    for x in c: seq = seq + x
    rev = 0.0
    for x in c[::-1]: rev = rev + x
    tree = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]))
    tree2 = ((c[0] + c[4]) + (c[2] + c[6])) + ((c[1] + c[5]) + (c[3] + c[7]))
    return dict(seq=seq, rev=rev, tree=tree, tree2=tree2)


def main():
    from oracle import pyoracle as po
    rng = np.random.default_rng(20260927)
    # candidate scores: every f16 in [-64, -1]
    h = np.arange(0x3c00, 0x5401, dtype=np.uint16)                      # 1.0 .. 64.0
    cand = -h.view(np.float16).astype(np.float32)
    e_c = expf_table(po, cand * np.float32(0.125))                      # exp(s / 8 - 0)
    fixed = -(rng.random(N) * 60 + 1).astype(np.float16).astype(np.float32)
    fixed[0] = 0.0
    # groups 1 and 2 far below the rest (exp ~ 1e-18 .. 1e-13): their partial sums fall off the end of the running double, so the ORDER of the
    # eight additions shows in its last bits — the search keeps a row on which the orders round to different f32 reciprocals
    fixed[8:24] = -(rng.random(16) * 80 + 240).astype(np.float16).astype(np.float32)
    fixed[40:48] = -(rng.random(8) * 100 + 100).astype(np.float16).astype(np.float32)     # group 5 too: its searched element moves the denominator in fine steps
    e_f = expf_table(po, fixed * np.float32(0.125))
    # searched positions 47, 55, 63 = the last element of groups 5, 6, 7: the sequential denominator is ((base + c5(x)) + c6(y)) + c7(z) in double
    base = 0.0
    for g in range(5): base = base + float(group_sum(e_f[g * 8:g * 8 + 8]))
    def c_of(g):                                                          # the group's partial sum by candidate for its element 7
        v = e_f[g * 8:g * 8 + 8]
        return (((v[0] + v[4]) + (v[2] + v[6])).astype(np.float32) + (np.float32(v[1] + v[5]) + (v[3] + e_c)).astype(np.float32)).astype(np.float32).astype(np.float64)
    hx = np.arange(0x5400, 0x5c01, dtype=np.uint16)                      # group 5's candidates: every f16 in [-256, -64]
    candx = -hx.view(np.float16).astype(np.float32)
    e_cx = expf_table(po, candx * np.float32(0.125))
    v5 = e_f[40:48]
    c5 = (((v5[0] + v5[4]) + (v5[2] + v5[6])).astype(np.float32) + (np.float32(v5[1] + v5[5]) + (v5[3] + e_cx)).astype(np.float32)).astype(np.float32).astype(np.float64)
    c6, c7 = c_of(6), c_of(7)
    T1 = base + c5
    best = None
    for zi in rng.permutation(cand.size)[:512]:
        for lo in range(0, candx.size, 512):
            rs = 1.0 / ((T1[lo:lo + 512, None] + c6[None, :]) + c7[zi])
            d = dist(rs)
            k = int(np.argmin(d))
            if d.flat[k] <= 3:
                xi, yi = lo + k // cand.size, k % cand.size
                s = fixed.copy(); s[47] = candx[xi]; s[55] = cand[yi]; s[63] = cand[zi]
                e = expf_table(po, s * np.float32(0.125))
                den = denominators(e)
                ds = {n: int(dist(np.array([1.0 / t]))[0]) for n, t in den.items()}
                f32s = {n: np.float32(1.0 / t) for n, t in den.items()}
                nd = len({int(x.view(np.uint32)) for x in f32s.values()})
                print("hit", candx[xi], cand[yi], cand[zi], ds, {n: hex(int(x.view(np.uint32))) for n, x in f32s.items()}, flush=True)
                if max(ds.values()) <= GUARD_ULPS - 8 and (best is None or nd > best[1]):
                    best = (s, nd, den)
        if best is not None and best[1] > 1:
            break
    assert best is not None, "nothing found"
    s, nd, den = best
    out = os.path.join(ROOT, "tests", "golden", "f64_softmax_kat.npz")
    np.savez(out, scores=s, inv_seq=np.float32(1.0 / den["seq"]), inv_tree=np.float32(1.0 / den["tree"]), distinct=np.int32(nd))
    print("saved", out, "distinct f32 reciprocals over the orders:", nd)


if __name__ == "__main__":
    main()
