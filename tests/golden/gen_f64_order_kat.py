#!/usr/bin/env python3
"""tests/golden/f64_order_kat.npz: the constructed activation vector on which the ONE documented numerics deviation is visible
(tools/f64_order_search.py): the reference sums the squares sequentially in double (ggml.c:11874-11879), the GPU in a fixed tree
order; on this vector the two sums straddle a rounding boundary of the f32 mean, which differs by one ulp.  Stored: x, both means,
and the Q8_K blocks of rms_norm(x) * 1 under either order (quantised by the oracle).  CPU only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import f64_order_search as fs  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

rng = np.random.default_rng(17)
eps = np.float32(1e-5)


def q8k(x, mean):
    scale = np.float32(1.0) / np.sqrt(np.float32(mean + eps), dtype=np.float32)
    y = ((x * scale).astype(np.float32) * np.float32(1.0)).astype(np.float32)
    return po.quantize_q8_K(y)


# a one-ulp difference of the mean survives 1 / sqrtf(mean + eps) only about every second time: construct vectors until the
# quantised activations (the block scales d) actually differ
for attempt in range(40):
    r = fs.construct(rng)
    assert r is not None
    x, m_seq, m_tree = r
    b_seq, b_tree = q8k(x, m_seq), q8k(x, m_tree)
    print("attempt", attempt, "means", m_seq, m_tree, "differing Q8_K bytes:", int((b_seq != b_tree).sum()))
    if not np.array_equal(b_seq, b_tree):
        break
assert not np.array_equal(b_seq, b_tree)
assert np.array_equal(po.quantize_q8_K(po.rms_norm(x, float(eps))), b_seq), "the oracle's rms_norm is the sequential order"
np.savez_compressed(os.path.join(HERE, "f64_order_kat.npz"), x=x, eps=eps, mean_seq=np.float32(m_seq), mean_tree=np.float32(m_tree), q8k_seq=b_seq, q8k_tree=b_tree)
