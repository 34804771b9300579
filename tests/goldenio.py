"""Reader for the flat 'BGLD0001' containers written by oracle/harness/gen_golden.cpp.

Record = u32 name_len, name, u32 dtype (0 f32, 1 f16 raw, 2 i32, 3 u8), u32 ndims, u64 dims[ndims] (ggml order:
dims[0] is the fastest-varying axis), u64 nbytes, payload.  Arrays are returned with numpy shape = reversed(dims),
i.e. C-order with the ggml ne[0] axis last.
"""
import struct
import numpy as np

_DT = {0: np.float32, 1: np.float16, 2: np.int32, 3: np.uint8}


def load_bgld(path):
    out = {}
    with open(path, "rb") as f:
        buf = f.read()
    assert buf[:8] == b"BGLD0001", "bad magic"
    o = 8
    while o < len(buf):
        (nl,) = struct.unpack_from("<I", buf, o); o += 4
        name = buf[o:o + nl].decode(); o += nl
        dt, nd = struct.unpack_from("<II", buf, o); o += 8
        dims = struct.unpack_from("<%dQ" % nd, buf, o); o += 8 * nd
        (nb,) = struct.unpack_from("<Q", buf, o); o += 8
        arr = np.frombuffer(buf, dtype=_DT[dt], count=nb // np.dtype(_DT[dt]).itemsize, offset=o).reshape(tuple(reversed(dims)))
        o += nb
        # the same node name can occur once per micro-batch; keep the first
        out.setdefault(name, arr)
    return out
