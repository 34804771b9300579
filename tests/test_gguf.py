"""CPU: numpy GGUF reader/writer (host utility) — round trip and the reference-written fixture."""
import os

import numpy as np

from conftest import GOLDEN
from booster_amd import gguf


def test_read_reference_written_file():
    r = gguf.GGUFReader(os.path.join(GOLDEN, "tiny_a.gguf"))
    assert r.kv["general.architecture"] == "llama"
    assert r.kv["llama.embedding_length"] == 512 and r.kv["llama.block_count"] == 2
    t = r.tensors["blk.0.attn_v.weight"]
    assert t["type"] == gguf.Q6_K and t["shape"] == [512, 128] and t["data"].size == 128 * 2 * 210
    assert r.tensors["blk.1.attn_v.weight"]["type"] == gguf.Q5_K
    assert r.tensors["output_norm.weight"]["data"].view(np.float32).shape == (512,)


def test_writer_roundtrip(tmp_path):
    p = str(tmp_path / "syn.gguf")
    gguf.write_synthetic_llama(p, E=256, H=2, Hkv=1, L=3, F=512, V=64, seed=5, rope_freqs=True)
    r = gguf.GGUFReader(p)
    assert r.kv["llama.attention.head_count_kv"] == 1 and r.kv["tokenizer.ggml.model"] == "no_vocab"
    assert len(r.tensors) == 3 + 1 + 3 * 9
    assert r.tensors["output.weight"]["type"] == gguf.Q6_K
    assert r.tensors["blk.2.ffn_down.weight"]["shape"] == [512, 256]
    assert abs(float(r.kv["llama.rope.freq_base"]) - 500000.0) < 1e-3


def test_q4_k_m_recipe():
    # 8B: 16 of 32 layers carry Q6_K attn_v / ffn_down (SURVEY §8 header)
    n = sum(gguf.q4_k_m_type("ffn_down", il, 32) == gguf.Q6_K for il in range(32))
    assert n == 16
    assert gguf.q4_k_m_type("attn_q", 0, 32) == gguf.Q4_K and gguf.q4_k_m_type("output", 0, 32) == gguf.Q6_K


def test_split_model_reads_like_the_single_file(tmp_path):
    """gguf-split shards: the C++ reader opened on the first shard sees the same tensors (names, types, shapes, bytes) as on
    the un-split file; opening a later shard, a wrongly named shard or an incomplete set fails."""
    import ctypes as C
    import booster_amd
    L = booster_amd.lib()
    L.bamd_gguf_probe.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]

    def probe(path):
        n, b, d = C.c_int64(0), C.c_int64(0), C.c_uint64(0)
        rc = L.bamd_gguf_probe(path.encode(), C.byref(n), C.byref(b), C.byref(d))
        return rc, n.value, b.value, d.value

    single = str(tmp_path / "one.gguf")
    gguf.write_synthetic_llama(single, E=256, H=2, Hkv=1, L=3, F=512, V=64, seed=5, rope_freqs=True)
    shards = gguf.write_synthetic_llama(str(tmp_path / "model"), E=256, H=2, Hkv=1, L=3, F=512, V=64, seed=5, rope_freqs=True, n_split=3)
    assert [os.path.basename(p) for p in shards] == ["model-00001-of-00003.gguf", "model-00002-of-00003.gguf", "model-00003-of-00003.gguf"]
    want = probe(single)
    assert want[0] == 0 and want[1] == 3 + 1 + 3 * 9
    assert probe(shards[0]) == want
    assert probe(shards[1])[0] == 1                              # must be loaded with the first split
    os.rename(shards[2], shards[2] + ".away")
    assert probe(shards[0])[0] == 1                              # a shard is missing
    os.rename(shards[2] + ".away", shards[2])
    odd = str(tmp_path / "renamed.gguf"); os.rename(shards[0], odd)
    assert probe(odd)[0] == 1                                    # the first shard must carry the -00001-of-0000N.gguf name


def _patch_u32(blob, key, value):
    """overwrite the UINT32 value of one metadata key of a GGUF image (length-prefixed key, u32 type tag 4, u32 value)"""
    import struct
    tag = struct.pack("<Q", len(key)) + key.encode() + struct.pack("<I", 4)
    at = blob.index(tag) + len(tag)
    return blob[:at] + struct.pack("<I", value) + blob[at + 4:]


def test_model_header_is_validated_before_any_device(tmp_path):
    """a GGUF whose hyper-parameters the kernels cannot take (or that would divide by zero) is refused by name at load, on any machine:
    the header is read and checked before the first HIP call; only a well-formed file gets as far as 'no HIP device'"""
    import booster_amd
    good = str(tmp_path / "ok.gguf")
    gguf.write_synthetic_llama(good, E=256, H=2, Hkv=1, L=1, F=512, V=64, seed=5)
    blob = open(good, "rb").read()

    def load_error(image):
        p = str(tmp_path / "bad.gguf")
        open(p, "wb").write(image)
        try:
            booster_amd.Model(p).close()
        except booster_amd.BamdError as e:
            return str(e)
        return ""

    assert "head counts must be positive" in load_error(_patch_u32(blob, "llama.attention.head_count", 0))
    assert "head counts must be positive" in load_error(_patch_u32(blob, "llama.attention.head_count_kv", 0))
    assert "head counts must be positive" in load_error(_patch_u32(blob, "llama.block_count", 0))
    assert "not a multiple of llama.attention.head_count" in load_error(_patch_u32(blob, "llama.attention.head_count", 3))
    hd64 = _patch_u32(blob, "llama.rope.dimension_count", 64)
    assert "n_head % n_head_kv" in load_error(_patch_u32(_patch_u32(hd64, "llama.attention.head_count", 4), "llama.attention.head_count_kv", 3))
    assert "GQA ratio" in load_error(_patch_u32(_patch_u32(hd64, "llama.embedding_length", 1024), "llama.attention.head_count", 16))
    assert "rope.dimension_count" in load_error(_patch_u32(blob, "llama.rope.dimension_count", 64))
    assert "general.architecture" in load_error(blob.replace(b"\x05\x00\x00\x00\x00\x00\x00\x00llama", b"\x05\x00\x00\x00\x00\x00\x00\x00qwen2", 1))
    assert "GGUF" in load_error(b"GGML" + blob[4:]) or "magic" in load_error(b"GGML" + blob[4:])
    assert load_error(blob[: len(blob) // 2]) != ""                  # truncated file
    if booster_amd.device_count() <= 0:
        assert "no HIP device" in load_error(blob)                    # the well-formed file: refused only for want of a GPU


def test_reused_layers_differ_in_the_file(tmp_path):
    """the multi-GB synthetic models generate a layer's bytes once (reuse_layers) but must NOT be byte-identical layer to layer in the file: the writer XORs
    (layer + 1) into one quant byte of up to 64 blocks of every matrix, so that a loader aliasing one layer onto another shows in the full-size fixtures.
    The patched bytes are low-bit quants (any value is a well-formed quant): scales and f16 block scales are untouched."""
    import numpy as np
    from booster_amd import gguf
    p = str(tmp_path / "lid.gguf")
    gguf.write_synthetic_llama(p, E=512, H=8, Hkv=2, L=4, F=768, V=512, seed=7, reuse_layers=True, type_fn=lambda name, il: {"attn_v": gguf.Q6_K, "ffn_down": gguf.Q5_K}.get(name, gguf.Q4_K))
    r = gguf.GGUFReader(p)
    qoff = {gguf.Q4_K: (16, 144), gguf.Q5_K: (48, 176), gguf.Q6_K: (0, 210)}
    for nm in ("attn_q", "attn_k", "attn_v", "attn_output", "ffn_gate", "ffn_up", "ffn_down"):
        t = [r.tensors["blk.%d.%s.weight" % (i, nm)] for i in range(4)]
        d = [np.asarray(x["data"]) for x in t]
        lo, bb = qoff[t[0]["type"]]
        for i in range(4):
            for j in range(i + 1, 4):
                diff = np.flatnonzero(d[i] != d[j])
                assert 1 <= diff.size <= 64, (nm, i, j, diff.size)
                assert ((diff % bb) >= lo).all() and ((diff % bb) < lo + 128).all(), "a patched byte is not a low-bit quant"
