"""CPU: the weight-stream engine's planner (csrc/bamd_wse_plan.cpp) through bamd_wse_plan_describe — no device needed.

The plan cuts every matrix of build_llama's graph (cpp/src/llama.cpp:8781-8925) into contiguous runs of row-groups, one run per CU.  Checked
here for the BASELINE shapes: every record of every matrix is streamed by exactly one CU, every output row is produced exactly once, the record /
slot numbering is consistent, and the LDS budget holds."""
import numpy as np
import pytest

import booster_amd as b

RECB = {12: 1152, 13: 1408, 14: 1680}
SHAPES = {
    # name: (E, H, Hkv, hd, F, V, types [q k v o gate up down], head type)
    "llama3-8b-q4km": (4096, 32, 8, 128, 14336, 128256, [12, 12, 14, 12, 12, 12, 14], 14),
    "llama3-70b-q4km": (8192, 64, 8, 128, 28672, 128256, [12, 12, 13, 12, 12, 12, 14], 14),
    "mistral-7b-q6k": (4096, 32, 8, 128, 14336, 32000, [14] * 7, 14),
}


def stream_bytes(t, K, rows):
    return (rows + 7) // 8 * (K // 256) * RECB[t]


@pytest.mark.parametrize("name", sorted(SHAPES))
@pytest.mark.parametrize("nc", [10, 12])
def test_plan_covers_every_record_once(name, nc):
    E, H, Hkv, hd, F, V, types, ht = SHAPES[name]
    L, n_cu = 2, 256
    head, rows = b.wse_plan_describe(n_cu, E, H, Hkv, hd, F, L, V, types, ht, 512, nc)
    assert head["rc"] == 0 and head["n_cu"] == n_cu and 3 <= head["ns"] <= 8 and head["tr"] % 8 == 0 and head["lds_bytes"] <= 160 * 1024 - 2144
    mv = rows[rows[:, 1] == 1]
    at = rows[rows[:, 1] == 2]
    assert at.shape[0] == n_cu * L                        # one attention op per CU and layer
    # per CU: slots and records are numbered consecutively in program order; records in multiples of 8 (term chunks)
    for c in range(n_cu):
        r = rows[rows[:, 0] == c]
        gs = grec = 0
        for o in r:
            assert o[8] == gs and o[10] == grec, (c, o)
            if o[1] == 1:
                nrec = o[4] * o[5]
                assert nrec % 8 == 0 and o[4] % 8 == 0
                assert 1 <= o[9] <= 16384 // RECB[int(o[2])]
                gs += -(-nrec // o[9]); grec += nrec
    # every matrix: the byte ranges [src, src + ntask * nb * recb) of its pieces tile its stream exactly
    Ekv = Hkv * hd
    shapes = [(E, E), (Ekv, E), (Ekv, E), (E, E), (F, E), (F, E), (E, F)]
    # the planner's fake address map (bamd_wse_plan_describe): matrices follow one another from 1 MiB, 4 KiB aligned + 4 KiB
    base = 1 << 20
    mats = []
    for l in range(L):
        for (nr, K), t in zip(shapes, types):
            mats.append((base, t, nr, K)); base += (stream_bytes(t, K, nr) + 4095) // 4096 * 4096 + 4096
    mats.append((base, ht, V, E))
    for (mb, t, nr, K) in mats:
        sb = stream_bytes(t, K, nr)
        p = mv[(mv[:, 3] >= mb) & (mv[:, 3] < mb + sb)]
        assert p.shape[0] >= 1
        assert (p[:, 2] == t).all() and (p[:, 4] == K // 256).all()
        iv = sorted((int(o[3]), int(o[3] + o[5] * o[4] * RECB[t])) for o in p)
        assert iv[0][0] == mb and iv[-1][1] == mb + sb
        for (a0, a1), (b0, b1) in zip(iv, iv[1:]):
            assert a1 == b0
    # output rows: each op's pieces cover [0, rows) of its output vector exactly once (padding rows excluded)
    for l in range(L):
        for out_vec, n in ((2, E + 2 * Ekv), (4, E), (5, F)):
            p = mv[((mv[:, 15] >> 8) == out_vec) & ((mv[:, 15] & 255) == l) & (mv[:, 13] != 2)]
            seen = np.zeros(n, np.int32)
            for o in p:
                lo, hi = int(o[6]), min(int(o[6] + 8 * o[5]), int(o[7]))
                seen[lo:hi] += 1
            assert (seen == 1).all(), (l, out_vec)
    # gate and up pieces of a CU pair up: same rows, the up piece reuses the gate piece's activations
    g = mv[mv[:, 13] == 2]; u = mv[mv[:, 13] == 3]
    assert g.shape == u.shape and (g[:, 6] == u[:, 6]).all() and (g[:, 5] == u[:, 5]).all() and (u[:, 11] == 0).all() and (g[:, 11] == 3).all()


def test_plan_refuses_shapes_without_a_program():
    # Llama-2-7B: n_ff = 11008 = 43 super-blocks: the chainer's 8-record chunks do not fit -> the engine is not used (launch sequence instead)
    with pytest.raises(b.BamdError, match="multiple of 8"):
        b.wse_plan_describe(256, 4096, 32, 32, 128, 11008, 1, 32000, [12] * 7, 14, 512, 10)
    # a tiny model (E = 256: one super-block per row)
    with pytest.raises(b.BamdError):
        b.wse_plan_describe(256, 256, 4, 2, 64, 512, 1, 1000, [12] * 7, 14, 512, 10)
