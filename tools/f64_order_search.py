#!/usr/bin/env python3
"""The one documented numerics deviation: the RMSNorm sum of squares is accumulated in double in a fixed TREE
order on the GPU (ActPro::finish, bamd_device.h) and SEQUENTIALLY in the reference (ggml_compute_forward_rms_norm_f32,
cpp/ggml/src/ggml.c:11874-11879); both round the mean to f32 right after.  This tool (CPU only, numpy) restates both orders for
n = 4096 and
  (a) counts, over random activation vectors, how often the f32 mean differs (expected ~1e-8 per reduction);
  (b) CONSTRUCTS an adversarial vector: two elements are tuned until the sum sits a few double-ulps below a rounding boundary of
      the f32 mean, then a tiny last element walks it across one double-ulp at a time; where the two orders cross the boundary
      at different steps the two means differ by one f32 ulp.
    python tools/f64_order_search.py [n_random] [out.npz]
"""
import sys

import numpy as np

K = 4096


def terms(x):
    x = np.asarray(x, np.float32)
    return (x * x).astype(np.float32).astype(np.float64)           # (ggml_float)(x[i] * x[i]): the product is rounded to f32 first


def sum_seq(t):
    return float(np.cumsum(t, dtype=np.float64)[-1])               # cumsum adds strictly left to right


def sum_tree(t):
    """ActPro<true>::finish for K = 4096, 8 waves: wave w owns blocks w and w + 8; lane l the elements 4l..4l+3 of each, added in
    order; wave_sum_f64 = xor 1, xor 2, half-mirror, mirror inside rows of 16 lanes, then ((r0 + r1) + r2) + r3; the eight wave
    sums are added in order starting from 0."""
    b = t.reshape(16, 64, 4)                                       # [block][lane][element]
    tot = 0.0
    for w in range(8):
        lane = np.zeros(64, np.float64)
        for blk in (w, w + 8):
            for c in range(4):
                lane = lane + b[blk, :, c]
        s = lane
        s = s + s[np.arange(64) ^ 1]
        s = s + s[np.arange(64) ^ 2]
        i = np.arange(64); s = s + s[(i & ~7) | (7 - (i & 7))]
        s = s + s[(i & ~15) | (15 - (i & 15))]
        tot = tot + (((s[15] + s[31]) + s[47]) + s[63])
    return float(tot)


def mean32(s):
    return np.float32(s / K)


def rounding_safe(v, ulps):
    """bamd_device.h f32_rounding_safe, restated: a double rounds to f32 by its low 29 mantissa bits D (boundary D = 2^28); a relative
    reordering error of (2 n + 8) 2^-53 moves D by at most 2 n + 8, so the f32 cannot depend on the order when |D - 2^28| > 2 n + 8"""
    if v == 0.0:
        return True
    bits = int(np.float64(v).view(np.uint64))
    ex = (bits >> 52) & 0x7ff
    dist = (bits & 0x1fffffff) - 0x10000000
    return 1023 - 126 <= ex <= 1023 + 127 and abs(dist) > ulps


GUARD_ULPS = 2 * K + 8                         # BAMD_F64_GUARD_ULPS(K)


def guarded_mean32(t):
    """what the GPU computes since round 3: the tree sum, unless its f32 mean could depend on the order — then the sequential sum"""
    st = sum_tree(t)
    if rounding_safe(st / K, GUARD_ULPS):
        return mean32(st), False
    return mean32(sum_seq(t)), True


def random_trials(n, rng):
    diff = 0
    for _ in range(n):
        x = (rng.standard_normal(K) * np.exp(rng.standard_normal(K) * 2.0)).astype(np.float32)      # wide dynamic range
        t = terms(x)
        if mean32(sum_seq(t)).view(np.uint32) != mean32(sum_tree(t)).view(np.uint32):
            diff += 1
    return diff


def construct(rng, tries=200):
    for attempt in range(tries):
        x = (rng.standard_normal(K) * np.exp(rng.standard_normal(K) * 2.0)).astype(np.float32)
        x[K - 1] = 0.0
        j1 = int(np.argmin(np.abs(np.abs(x[:K - 1]) - 1.0))); j2 = int(np.argmin(np.abs(np.abs(x[:K - 1]) - 1.0 / 64)))
        if j1 == j2:
            continue
        s0 = sum_seq(terms(x))
        m = mean32(s0)
        mid = (float(m) + float(np.nextafter(m, np.float32(np.inf)))) / 2.0 * K          # the sum at which the f32 mean rounds up
        for j, span in ((j1, 400000), (j2, 400000)):                                    # coarse, then fine: land just below `mid`
            best = None
            base = x[j]
            for d in range(-span, span, max(1, span // 4000)):
                x[j] = np.float32(base).view(np.int32).__add__(d).astype(np.int32).view(np.float32) if False else (np.array([base], np.float32).view(np.int32) + d).view(np.float32)[0]
                s = sum_seq(terms(x))
                gap = mid - s
                if gap > 0 and (best is None or gap < best[0]):
                    best = (gap, x[j])
            if best is None:
                break
            x[j] = best[1]
        if best is None:
            continue
        # walk across with the tiny last element: its square moves the sum by far less than one double-ulp per step of its f32 value
        v = np.float32(2.0 ** -22)
        for step in range(200000):
            x[K - 1] = v
            t = terms(x)
            a, b = mean32(sum_seq(t)), mean32(sum_tree(t))
            if a.view(np.uint32) != b.view(np.uint32):
                return x.copy(), a, b
            if a != m:
                break                                     # both crossed together: try another vector
            v = np.float32(v * np.float32(1.0 + 2.0 ** -9))
    return None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(17)
    d = random_trials(n, rng)
    print("random vectors: %d of %d have a different f32 mean under the two orders" % (d, n))
    r = construct(rng)
    if r is None:
        print("constructive search: no differing vector found")
        return
    x, a, b = r
    print("constructed: sequential mean %r (bits %08x), tree mean %r (bits %08x)" % (float(a), a.view(np.uint32), float(b), b.view(np.uint32)))
    if len(sys.argv) > 2:
        np.savez_compressed(sys.argv[2], x=x, mean_seq_bits=np.uint32(a.view(np.uint32)), mean_tree_bits=np.uint32(b.view(np.uint32)))
        print("wrote", sys.argv[2])


if __name__ == "__main__":
    main()
