"""Procedural inputs of the Janus sampler fixtures (tests/golden/janus_kats.npz): base logits are a pure integer hash of
(seed, token id), so a fixture stores a seed and a few overrides instead of n_vocab floats per case.  Shared by the generator
(tests/golden/gen_janus_kats.py, build container) and the test (tests/test_janus.py)."""
import hashlib

import numpy as np

N_LAST = 64


def base_logits(seed, V, negative=False):
    """float32 logits in [-2, 10): splitmix64 of (seed, id), top 24 bits"""
    with np.errstate(over="ignore"):
        z = np.arange(V, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    x = (u * 12.0 - 2.0).astype(np.float32)
    return (-np.abs(x) - np.float32(1.0)).astype(np.float32) if negative else x


def case_logits(seed, V, negative, ov_ids, ov_vals):
    x = base_logits(int(seed), V, bool(negative))
    ids = np.asarray(ov_ids)
    keep = ids >= 0
    x[ids[keep]] = np.asarray(ov_vals, np.float32)[keep]
    return x


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()
