"""Minimal GGUF v3 reader / writer in numpy.

Host-side utility: metadata inspection and writing synthetic Llama-family GGUFs (tests, bench).  The product
loader that feeds the GPU is the C++ one in csrc/gguf.cpp; this module mirrors the same on-disk format
(reference: cpp/ggml/src/ggml.c:20753-21260 reader, :21485-21990 writer; type ids ggml.h:360-375,
KV value types ggml.h:2257-2272, default alignment 32 ggml.h:251).
"""
import struct
import numpy as np

GGUF_MAGIC = b"GGUF"
GGUF_VERSION = 3
DEFAULT_ALIGNMENT = 32

# ggml tensor types we care about: id -> (block_elems, block_bytes)
GGML_TYPES = {
    0: (1, 4),        # F32
    1: (1, 2),        # F16
    12: (256, 144),   # Q4_K
    13: (256, 176),   # Q5_K
    14: (256, 210),   # Q6_K
}
F32, F16, Q4_K, Q5_K, Q6_K = 0, 1, 12, 13, 14
TYPE_NAMES = {0: "F32", 1: "F16", 12: "Q4_K", 13: "Q5_K", 14: "Q6_K"}

# gguf KV value types
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)
_SCALAR_FMT = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i", T_F32: "<f",
               T_BOOL: "<?", T_U64: "<Q", T_I64: "<q", T_F64: "<d"}


def tensor_nbytes(ttype, shape):
    be, bb = GGML_TYPES[ttype]
    n = int(np.prod(shape))
    assert shape[0] % be == 0, "row length must be a multiple of the block size"
    return n // be * bb


class GGUFReader:
    """Parses header, KV pairs and tensor infos; tensor data are numpy views into a memory map."""

    def __init__(self, path):
        self.path = path
        self.mm = np.memmap(path, dtype=np.uint8, mode="r")
        buf = self.mm
        self.o = 0
        assert bytes(buf[0:4]) == GGUF_MAGIC, "not a GGUF file"
        self.o = 4
        self.version = self._rd("<I")
        n_tensors = self._rd("<Q")
        n_kv = self._rd("<Q")
        self.kv = {}
        for _ in range(n_kv):
            key = self._rd_str()
            vt = self._rd("<I")
            self.kv[key] = self._rd_val(vt)
        self.alignment = int(self.kv.get("general.alignment", DEFAULT_ALIGNMENT))
        infos = []
        for _ in range(n_tensors):
            name = self._rd_str()
            nd = self._rd("<I")
            shape = [self._rd("<Q") for _ in range(nd)]       # ggml order: shape[0] = row length
            ttype = self._rd("<I")
            off = self._rd("<Q")
            infos.append((name, shape, ttype, off))
        self.data_offset = (self.o + self.alignment - 1) // self.alignment * self.alignment
        self.tensors = {}
        for name, shape, ttype, off in infos:
            nb = tensor_nbytes(ttype, shape)
            start = self.data_offset + off
            self.tensors[name] = dict(shape=shape, type=ttype, data=self.mm[start:start + nb])

    def _rd(self, fmt):
        v = struct.unpack_from(fmt, self.mm, self.o)[0]
        self.o += struct.calcsize(fmt)
        return v

    def _rd_str(self):
        n = self._rd("<Q")
        s = bytes(self.mm[self.o:self.o + n]).decode("utf-8", errors="replace")
        self.o += n
        return s

    def _rd_val(self, vt):
        if vt == T_STR:
            return self._rd_str()
        if vt == T_ARR:
            et = self._rd("<I")
            n = self._rd("<Q")
            if et == T_STR:
                return [self._rd_str() for _ in range(n)]
            fmt = _SCALAR_FMT[et]
            sz = struct.calcsize(fmt)
            arr = np.frombuffer(self.mm, dtype=np.dtype(fmt[1:]).newbyteorder("<"), count=n, offset=self.o).copy()
            self.o += sz * n
            return arr
        return self._rd(_SCALAR_FMT[vt])


class GGUFWriter:
    def __init__(self):
        self.kv = []          # (key, type, value)
        self.tensors = []     # (name, shape(ggml order), type, bytes ndarray)

    def add_u16(self, k, v): self.kv.append((k, T_U16, int(v)))
    def add_u32(self, k, v): self.kv.append((k, T_U32, int(v)))
    def add_i32(self, k, v): self.kv.append((k, T_I32, int(v)))
    def add_f32(self, k, v): self.kv.append((k, T_F32, float(v)))
    def add_str(self, k, v): self.kv.append((k, T_STR, str(v)))
    def add_bool(self, k, v): self.kv.append((k, T_BOOL, bool(v)))
    def add_arr_str(self, k, v): self.kv.append((k, T_ARR, (T_STR, list(v))))
    def add_arr(self, k, et, v): self.kv.append((k, T_ARR, (et, list(v))))

    def add_tensor(self, name, shape, ttype, data, patch=None):
        """patch: (byte offsets, xor value) applied while the tensor is written — several tensors may share ONE data array (the multi-GB synthetic
        models reuse a layer's bytes) and still differ in the file"""
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        assert data.size == tensor_nbytes(ttype, shape), (name, data.size, tensor_nbytes(ttype, shape))
        self.tensors.append((name, list(shape), ttype, data, patch))

    @staticmethod
    def _s(s):
        b = s.encode("utf-8")
        return struct.pack("<Q", len(b)) + b

    def write(self, path):
        out = bytearray()
        out += GGUF_MAGIC + struct.pack("<I", GGUF_VERSION) + struct.pack("<Q", len(self.tensors)) + struct.pack("<Q", len(self.kv))
        for k, t, v in self.kv:
            out += self._s(k) + struct.pack("<I", t)
            if t == T_STR:
                out += self._s(v)
            elif t == T_ARR:
                et, items = v
                out += struct.pack("<I", et) + struct.pack("<Q", len(items))
                for it in items:
                    out += self._s(it) if et == T_STR else struct.pack(_SCALAR_FMT[et], it)
            else:
                out += struct.pack(_SCALAR_FMT[t], v)
        off = 0
        offs = []
        for name, shape, ttype, data, _patch in self.tensors:
            out += self._s(name) + struct.pack("<I", len(shape))
            for d in shape:
                out += struct.pack("<Q", d)
            out += struct.pack("<I", ttype) + struct.pack("<Q", off)
            offs.append(off)
            off += (data.size + DEFAULT_ALIGNMENT - 1) // DEFAULT_ALIGNMENT * DEFAULT_ALIGNMENT
        pad = (-len(out)) % DEFAULT_ALIGNMENT
        out += b"\0" * pad
        with open(path, "wb") as f:
            f.write(out)
            for (name, shape, ttype, data, patch) in self.tensors:
                start = f.tell()
                f.write(memoryview(data))
                f.write(b"\0" * ((-data.size) % DEFAULT_ALIGNMENT))
                if patch is not None:
                    end = f.tell()
                    offs, x = patch
                    for o in offs:
                        f.seek(start + int(o)); f.write(bytes([int(data[int(o)]) ^ (int(x) & 0xff)]))
                    f.seek(end)


    def write_split(self, path_prefix, n_split):
        """gguf-split layout (llama.cpp:3659-3714): <prefix>-0000k-of-0000n.gguf, every shard with its own tensor table and data section and the
        keys split.no / split.count / split.tensors.count; the model's own keys live in the first shard.  Returns the shard paths."""
        per = (len(self.tensors) + n_split - 1) // n_split
        paths = []
        for k in range(n_split):
            w = GGUFWriter()
            if k == 0:
                w.kv = list(self.kv)
            w.add_u16("split.no", k); w.add_u16("split.count", n_split); w.add_i32("split.tensors.count", len(self.tensors))
            w.tensors = self.tensors[k * per:(k + 1) * per]
            path = "%s-%05d-of-%05d.gguf" % (path_prefix, k + 1, n_split)
            w.write(path); paths.append(path)
        return paths


# ---------------------------------------------------------------------------------------------------------
# synthetic K-quant tensors: random but well-formed blocks (finite f16 scales), for throughput / parity runs
# ---------------------------------------------------------------------------------------------------------
def random_kquant_tensor(ttype, row_len, n_rows, rng, amp=1.0):
    """Random raw blocks of type `ttype` for an [n_rows, row_len] matrix whose dequantised values are roughly
    zero-mean with standard deviation ~ amp / sqrt(row_len)."""
    be, bb = GGML_TYPES[ttype]
    nblk = n_rows * (row_len // be)
    blk = rng.integers(0, 256, size=(nblk, bb), dtype=np.uint8)
    sigma = amp / np.sqrt(row_len)
    if ttype in (Q4_K, Q5_K):
        # w = d*sc*q - dmin*m ; q in [0,15]/[0,31], sc,m in [0,63]
        qmax = 15.0 if ttype == Q4_K else 31.0
        d = (sigma / (32.0 * qmax * 0.3)) * rng.uniform(0.5, 1.5, size=nblk)
        dmin = d * qmax * 0.5 * rng.uniform(0.8, 1.2, size=nblk)
        blk[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
        blk[:, 2:4] = dmin.astype(np.float16).view(np.uint8).reshape(-1, 2)
    elif ttype == Q6_K:
        d = (sigma / (64.0 * 18.0)) * rng.uniform(0.5, 1.5, size=nblk)
        blk[:, 208:210] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    else:
        raise ValueError(ttype)
    return blk.reshape(-1)


def q4_k_m_type(name, il, n_layer):
    """Tensor type under llama.cpp's Q4_K_M recipe for the tensors we use (reference: llama.cpp:15442-15444,
    :15547-15555, :15603-15610, use_more_bits :15466-15480; SURVEY §8 header)."""
    more = il < n_layer // 8 or il >= 7 * n_layer // 8 or (il - n_layer // 8) % 3 == 2
    if name == "output":
        return Q6_K
    if name in ("attn_v", "ffn_down"):
        return Q6_K if more else Q4_K
    return Q4_K


def q4_k_m_type_70b(name, il, n_layer=80):
    """The Q4_K_M recipe for the 70B shape: as q4_k_m_type, but attn_v (8 heads share it) is Q5_K where the recipe leaves it at
    Q4_K (llama.cpp:15552-15555)."""
    t = q4_k_m_type(name, il, n_layer)
    return Q5_K if (name == "attn_v" and t == Q4_K) else t


def write_synthetic_llama(path, E, H, Hkv, L, F, V, theta=500000.0, eps=1e-5, n_ctx_train=8192, seed=7,
                          type_fn=None, rope_freqs=False, embd_type=Q4_K, reuse_layers=False, vocab=None, n_split=0, rope_scaling=None, tied=False):
    """Write a synthetic Llama-architecture GGUF (tokenizer.ggml.model = no_vocab) with random K-quant blocks.
    reuse_layers: generate each (tensor kind, type) once and reuse the bytes in every layer (fast path for the
    multi-GB benchmark model; the arithmetic and the bytes streamed per token are unchanged).
    tied: no output.weight — the reader falls back to token_embd.weight (Llama-3.2; llama.cpp:6070-6076)."""
    rng = np.random.default_rng(seed)
    _cache = {}

    def kq(t, cols, rows, amp, tag):
        if not reuse_layers:
            return random_kquant_tensor(t, cols, rows, rng, amp)
        key = (tag, t, cols, rows)
        if key not in _cache:
            _cache[key] = random_kquant_tensor(t, cols, rows, rng, amp)
        return _cache[key]

    def layer_patch(t, cols, rows, il):
        """reuse_layers: the layers share their bytes in memory, but NOT in the file — layer il has (il + 1) XORed into one quant byte of 64 blocks
        spread over each of its matrices (rows 0, rows/64, ...: first block of the row, a byte of the low-bit quants: any value is a well-formed
        quant).  A loader that aliases one layer onto another — a layer-index or a > 4 GiB offset slip — then reads different weights and the
        full-size fixtures (every logit digest of the reference's run) catch it; generation stays O(1) per layer."""
        if not reuse_layers:
            return None
        bb = GGML_TYPES[t][1]; nb = cols // 256
        qoff = {Q4_K: 16, Q5_K: 48, Q6_K: 0}[t]                # qs / qs / ql
        n = min(64, rows)
        return ([(k * rows // n) * nb * bb + qoff + (k % 32) for k in range(n)], il + 1)

    type_fn = type_fn or (lambda name, il: q4_k_m_type(name, il, L))
    hd = E // H
    w = GGUFWriter()
    w.add_str("general.architecture", "llama")
    w.add_str("general.name", "booster-amd-synthetic")
    w.add_u32("llama.context_length", n_ctx_train)
    w.add_u32("llama.embedding_length", E)
    w.add_u32("llama.block_count", L)
    w.add_u32("llama.feed_forward_length", F)
    w.add_u32("llama.attention.head_count", H)
    w.add_u32("llama.attention.head_count_kv", Hkv)
    w.add_f32("llama.attention.layer_norm_rms_epsilon", eps)
    w.add_u32("llama.rope.dimension_count", hd)
    w.add_f32("llama.rope.freq_base", theta)
    if rope_scaling:                                        # dict(type=..., factor=..., orig_ctx=..., attn_factor=...): llama.rope.scaling.*
        if "type" in rope_scaling:
            w.add_str("llama.rope.scaling.type", rope_scaling["type"])
        if "factor" in rope_scaling:
            w.add_f32("llama.rope.scaling.factor", float(rope_scaling["factor"]))
        if "orig_ctx" in rope_scaling:
            w.add_u32("llama.rope.scaling.original_context_length", int(rope_scaling["orig_ctx"]))
        if "attn_factor" in rope_scaling:
            w.add_f32("llama.rope.scaling.attn_factor", float(rope_scaling["attn_factor"]))
    w.add_u32("llama.vocab_size", V)
    if vocab is None:
        w.add_str("tokenizer.ggml.model", "no_vocab")
    else:
        assert len(vocab["tokens"]) == V
        w.add_str("tokenizer.ggml.model", vocab["model"])
        if "pre" in vocab:
            w.add_str("tokenizer.ggml.pre", vocab["pre"])
        w.add_arr_str("tokenizer.ggml.tokens", vocab["tokens"])
        if "scores" in vocab:
            w.add_arr("tokenizer.ggml.scores", T_F32, [float(x) for x in vocab["scores"]])
        w.add_arr("tokenizer.ggml.token_type", T_I32, [int(x) for x in vocab["types"]])
        if "merges" in vocab:
            w.add_arr_str("tokenizer.ggml.merges", vocab["merges"])
        for key in ("bos_token_id", "eos_token_id", "unknown_token_id", "eot_token_id"):
            if key in vocab:
                w.add_u32("tokenizer.ggml." + key, vocab[key])
        for key in ("add_bos_token", "add_eos_token", "add_space_prefix"):
            if key in vocab:
                w.add_bool("tokenizer.ggml." + key, vocab[key])

    def norm():
        return (1.0 + 0.1 * rng.standard_normal(E)).astype(np.float32)

    w.add_tensor("token_embd.weight", [E, V], embd_type, random_kquant_tensor(embd_type, E, V, rng, amp=np.sqrt(E)))
    w.add_tensor("output_norm.weight", [E], F32, norm())
    if not tied:
        t = type_fn("output", 0)
        w.add_tensor("output.weight", [E, V], t, random_kquant_tensor(t, E, V, rng, amp=3.0))
    if rope_freqs:
        ff = np.ones(hd // 2, np.float32)
        ff[hd // 4:] = 1.0 + 7.0 * np.arange(hd // 2 - hd // 4, dtype=np.float32) / (hd // 4)
        w.add_tensor("rope_freqs.weight", [hd // 2], F32, ff)
    for il in range(L):
        p = "blk.%d." % il
        w.add_tensor(p + "attn_norm.weight", [E], F32, norm())
        for nm, rows, cols, amp in (("attn_q", E, E, 2.0), ("attn_k", Hkv * hd, E, 2.0), ("attn_v", Hkv * hd, E, 1.0),
                                    ("attn_output", E, E, 1.0)):
            t = type_fn(nm, il)
            w.add_tensor(p + nm + ".weight", [cols, rows], t, kq(t, cols, rows, amp, nm), layer_patch(t, cols, rows, il))
        w.add_tensor(p + "ffn_norm.weight", [E], F32, norm())
        for nm, rows, cols, amp in (("ffn_gate", F, E, 1.5), ("ffn_up", F, E, 1.5), ("ffn_down", E, F, 1.0)):
            t = type_fn(nm, il)
            w.add_tensor(p + nm + ".weight", [cols, rows], t, kq(t, cols, rows, amp, nm), layer_patch(t, cols, rows, il))
    if n_split > 1:                  # `path` is then the prefix; returns the shard paths (open the first)
        return w.write_split(path, n_split)
    w.write(path)


# ---------------------------------------------------------------------------------------------------------
# synthetic vocabularies (tests of the tokenizer side of the bridge)
# ---------------------------------------------------------------------------------------------------------
def synthetic_spm_vocab(n_extra=200, seed=1):
    """SentencePiece-style vocab: <unk> <s> </s>, 256 byte tokens, single characters, random multi-character pieces."""
    import random
    rnd = random.Random(seed)
    toks, scores, types = ["<unk>", "<s>", "</s>"], [0.0, 0.0, 0.0], [2, 3, 3]
    for b in range(256):
        toks.append("<0x%02X>" % b); scores.append(0.0); types.append(6)
    alphabet = list("abcdefghijklmnopqrstuvwxyzABCDE0123456789.,!?'-") + ["\u2581", "\u00e9", "\u0436", "\u4e2d"]
    for ch in alphabet:
        toks.append(ch); scores.append(-10.0 - rnd.random()); types.append(1)
    seen = set(toks)
    while len(toks) < 3 + 256 + len(alphabet) + n_extra:
        n = rnd.choice([2, 2, 3, 3, 4, 5])
        piece = "".join(rnd.choice(alphabet[:26] + ["\u2581", "\u2581", "e", "t", "a"]) for _ in range(n))
        if piece in seen:
            continue
        seen.add(piece); toks.append(piece); scores.append(-rnd.random() * 8.0 - n * 0.01); types.append(1)
    toks.append("<|user|>"); scores.append(0.0); types.append(4)
    return dict(model="llama", tokens=toks, scores=scores, types=types, bos_token_id=1, eos_token_id=2, unknown_token_id=0,
                add_bos_token=True, add_space_prefix=True)


def synthetic_janus_vocab(n_total=640, seed=5):
    """SPM vocab of n_total tokens for the Janus sampler tests: the synthetic_spm_vocab pieces, the punctuation / whitespace /
    'pedantic' pieces cpp/janus.cpp names (:381-392, :536-560), then random Latin, Cyrillic and other-script pieces with and without
    the leading space, interleaved over the whole id range — so that every token class (LANG_/SPACE_ EN, RU, OTHER, ZERO), every
    scale formula, the id-range rules of the Llama-3 branch (n_total > 128000) and the id table of the Llama-2 branch occur.
    Pieces stay below 20 bytes (Latin) / 40 bytes (Cyrillic): the reference indexes a 20-entry table by piece length without a
    bound.  "<|im_end|>" gives the vocabulary an EOT id (the reference writes scales[eot] unchecked)."""
    import random
    rnd = random.Random(seed)
    v = synthetic_spm_vocab(n_extra=60, seed=seed)
    toks, scores, types = v["tokens"][:-1], v["scores"][:-1], v["types"][:-1]
    ru = "абвгдежзиклмнопрстуфхцчшщыэюяё"
    ru_up = "АБВГДЕЖЗИКЛМНОПРСТУФ"
    en = "abcdefghijklmnopqrstuvwxyz"
    other = "\u00e9\u00fc\u00f1\u4e2d\u6587\u3042\u0391\u05d0"
    seen = set(toks)
    sp = "\u2581"
    extra = [sp + "*", sp + "=", sp + "-", sp + "+", "{", "}", "[", "]", sp + "{", sp + "}", sp + "[", sp + "]", "```", "<|end_of_text|>",
             "12", "345", sp, sp + sp, sp + sp + sp + sp, sp + "\u2014", ":", ";", sp + "(", ").", sp + ")", ")", "(", "\n\n",
             "\u00e9t\u00e9", sp + "\u00e9", "\u4e2d\u6587", sp + "\u4e2d", "\u041f\u0440\u0438", sp + "\u041c\u0438\u0440", "\u0401\u0436", "ab\u0436"]
    for piece in extra:
        if piece not in seen:
            seen.add(piece); toks.append(piece); scores.append(-rnd.random() * 8.0); types.append(1)
    while len(toks) < n_total - 2:
        script = rnd.choice(["ru", "ru", "ru", "en", "en", "other", "RU", "EN"])
        n = rnd.choice([1, 2, 2, 3, 4, 6, 9])
        if script == "ru":
            body = "".join(rnd.choice(ru) for _ in range(n))
        elif script == "RU":
            body = rnd.choice(ru_up) + "".join(rnd.choice(ru) for _ in range(n - 1))
        elif script == "en":
            body = "".join(rnd.choice(en) for _ in range(n))
        elif script == "EN":
            body = rnd.choice(en).upper() + "".join(rnd.choice(en) for _ in range(n - 1))
        else:
            body = "".join(rnd.choice(other) for _ in range(min(n, 4)))
        piece = (sp if rnd.random() < 0.4 else "") + body
        if piece in seen:
            continue
        seen.add(piece); toks.append(piece); scores.append(-rnd.random() * 8.0 - n * 0.01); types.append(1)
    toks.append("<|im_end|>"); scores.append(0.0); types.append(3)
    toks.append("<|user|>"); scores.append(0.0); types.append(4)
    v.update(tokens=toks, scores=scores, types=types)
    return v


def _bytes_to_unicode():
    bs = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def synthetic_bpe_vocab(n_merges=300, seed=2, pre="llama-bpe", extra_merges=None):
    """Byte-level BPE vocab (GPT-2 / Llama-3 style): 256 byte symbols, random merges, a few control tokens.
    extra_merges: [(left bytes, right bytes), ...] appended behind the random merges (digit-digit / cross-character merges, so that a different
    pre-tokeniser split changes the token stream)."""
    import random
    rnd = random.Random(seed)
    b2u = _bytes_to_unicode()
    toks = [b2u[b] for b in range(256)]
    merges = []
    seen = set(toks)
    common = [b2u[ord(c)] for c in "etaoinshrdlu"] + [b2u[ord(" ")]]
    while len(merges) < n_merges:
        a = rnd.choice(toks if rnd.random() < 0.5 else common)
        b = rnd.choice(toks if rnd.random() < 0.3 else common)
        if a + b in seen or " " in a or " " in b:
            continue
        seen.add(a + b); toks.append(a + b); merges.append(a + " " + b)
    for la, rb in (extra_merges or []):
        a = "".join(b2u[x] for x in la); b = "".join(b2u[x] for x in rb)
        if a + b in seen:
            continue
        seen.add(a + b); toks.append(a + b); merges.append(a + " " + b)
    types = [1] * len(toks)
    for sp in ("<|begin_of_text|>", "<|end_of_text|>", "<|eot_id|>", "<|start_header_id|>"):
        toks.append(sp); types.append(3)
    n = len(toks)
    return dict(model="gpt2", pre=pre, tokens=toks, types=types, merges=merges, bos_token_id=n - 4, eos_token_id=n - 3,
                add_bos_token=False)


def synthetic_bpe_vocab_holes(n_merges=300, seed=2):
    """An INCONSISTENT byte-level BPE vocabulary: every fourth merge's result is not a token (its text is replaced by a placeholder), and a few
    byte symbols are missing too.  The reference then falls back to single raw bytes of the byte-level text and drops what has no token
    (llama-vocab.cpp:575-584) — the path a well-formed vocabulary never takes."""
    v = synthetic_bpe_vocab(n_merges=n_merges, seed=seed)
    b2u = _bytes_to_unicode()
    toks = list(v["tokens"])
    for i in range(0, n_merges, 4):
        toks[256 + i] = "<hole_%d>" % i
    for ch in "zq":
        toks[ord(ch)] = "<nobyte_%s>" % ch
    toks[toks.index(b2u[ord("\n")])] = "<nobyte_nl>"
    v["tokens"] = toks
    return v

