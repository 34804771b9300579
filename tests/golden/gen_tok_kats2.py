"""Generates tests/golden/tok_kats2.npz with the GENUINE reference tokenizer (oracle/_ref/tok_ref): the hardened tokenizer pins.

Vocabularies (booster_amd.gguf, deterministic): a 32 768-merge byte-level BPE with the llama-3 pre-tokeniser, a small BPE with the
GPT-2 pre-tokeniser, the SentencePiece vocabulary.  Strings: > 2 000 per vocabulary — every boundary of the reference's \\p{L}, \\p{N}
and \\s classes (tests/golden/unicode_classes.json: the code points just outside, at the start, at the end of and just past every
range) in letter / digit / space / punctuation / newline contexts, random mixtures over a broad alphabet, and the hand-written cases
of gen_tok_kats.py.  Stored per vocabulary: the strings (UTF-8, concatenated + offsets) and the reference's token ids (flat + offsets)
from llama_tokenize(add_special=false, parse_special=true), the call of cpp/bridge.cpp:278.  Build container only."""
import json
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
from booster_amd import gguf  # noqa: E402

TOK_REF = os.path.join(ROOT, "oracle", "_ref", "tok_ref")


def vocabs():
    return {"bpe32k": gguf.synthetic_bpe_vocab(n_merges=32768, seed=11), "gpt2": gguf.synthetic_bpe_vocab(n_merges=600, seed=12, pre="gpt-2"),
            "spm": gguf.synthetic_spm_vocab(),
            # the regex chains around the GPT-2 regex and the qwen2 form of the llama-3 regex (llama-vocab.cpp:379-443)
            "qwen2": gguf.synthetic_bpe_vocab(n_merges=600, seed=13, pre="qwen2"), "starcoder": gguf.synthetic_bpe_vocab(n_merges=600, seed=14, pre="starcoder"),
            "default": gguf.synthetic_bpe_vocab(n_merges=600, seed=15, pre="default"), "falcon": gguf.synthetic_bpe_vocab(n_merges=600, seed=16, pre="falcon"),
            "poro": gguf.synthetic_bpe_vocab(n_merges=600, seed=17, pre="poro-chat"), "viking": gguf.synthetic_bpe_vocab(n_merges=600, seed=18, pre="viking"),
            "dscoder": gguf.synthetic_bpe_vocab(n_merges=600, seed=19, pre="deepseek-coder"), "tekken": gguf.synthetic_bpe_vocab(n_merges=600, seed=20, pre="tekken"),
            "dsllm": gguf.synthetic_bpe_vocab(n_merges=600, seed=21, pre="deepseek-llm"),
            # merges ACROSS the places where the chains split (the random merges above almost never join two digits or two characters): the default chain's
            # fourth regex "[0-9][0-9][0-9]" (1234567 -> 123 | 456 | 7) and deepseek-coder's CJK class, whose std::wregex form has holes at the
            # non-ASCII white space (U+3000 between two ideographs ends the run)
            "default_digits": gguf.synthetic_bpe_vocab(n_merges=600, seed=15, pre="default", extra_merges=DIGIT_MERGES),
            "falcon_digits": gguf.synthetic_bpe_vocab(n_merges=600, seed=16, pre="falcon", extra_merges=DIGIT_MERGES),
            "dscoder_cjk": gguf.synthetic_bpe_vocab(n_merges=600, seed=19, pre="deepseek-coder", extra_merges=DIGIT_MERGES + CJK_MERGES),
            "dsllm_cjk": gguf.synthetic_bpe_vocab(n_merges=600, seed=21, pre="deepseek-llm", extra_merges=DIGIT_MERGES + CJK_MERGES)}


DIGIT_MERGES = [(b"1", b"2"), (b"12", b"3"), (b"3", b"4"), (b"123", b"4"), (b"4", b"5"), (b"5", b"6"), (b"45", b"6"), (b"6", b"7"), (b"7", b"8"), (b"8", b"9"), (b"0", b"0"),
                (b"00", b"0"), (b"9", b"0"), (b"1", b"0"), (b"7", b"1"), (b"56", b"7"), (b"34", b"56"), (b".", b"."), (b"..", b"."), (b"!", b"!"), (b"?", b"!")]
# last byte of an ideograph + first byte of U+3000 / U+2003 / U+1680 / U+2028, and their last byte + the first byte of an ideograph or a Hangul syllable
CJK_MERGES = [(b"\xad", b"\xe3"), (b"\x87", b"\xe3"), (b"\x80", b"\xe4"), (b"\x80", b"\xe6"), (b"\xad", b"\xe2"), (b"\x83", b"\xe6"), (b"\xa8", b"\xe4"),
              (b"\xad", b"\xe1"), (b"\x80", b"\xea"), (b"\xe4\xb8", b"\xad"), (b"\xe3\x80", b"\x80")]


def valid(cp):
    return 0 < cp < 0x110000 and not (0xD800 <= cp <= 0xDFFF)


def strings():
    rnd = random.Random(21)
    cls = json.load(open(os.path.join(HERE, "unicode_classes.json")))
    edge = []
    for key in ("letter", "number", "whitespace", "punctuation"):
        for lo, hi in cls[key]:
            for cp in (lo - 1, lo, hi, hi + 1):
                if valid(cp):
                    edge.append(chr(cp))
    out = ["Hello world", " Hello  world!!", "it's he'll we'Re DON'T I'M you'D", "x = 12345 + 678;\n\n\ty++", "a\n\n b \r\n c   ", "   ", "\t\t\n", "1234567890",
           "<|begin_of_text|>abc<|eot_id|>def <|start_header_id|>", "<s>hi</s> there<unk>", "ab<0x41>cd", "\x1c\x1d\x1e\x1f a\x1cb", "  x　y", "", " ", "\n",
           "a+=b<<2;c^=~d|e$f`g`", "1234567 12 123 1234 ٣٣٣٣ 12a345", "f(x)=[1,2]{3}...!?", "x$$+y==z>=w<=v^^u~~t||s", "  ...  !!\n\n??", "100%done#tag@me", "a。b，c…d«e»", "HELLOworld helloWORLD HeLLo ÉCOLEécole ÀBc aÀB ABC abc ŻÓŁĆ żółć a/b//c\n/ x1Y2", "McDonald's iPhone XMLHttpRequest ΑΒΓαβγ", "aÀa bÖc ＡＢＣabc！？ 中文字가나다 x  \n y   ", "tail   ", "‘quoted’ 。，、 １２３ 123abc"]
    ctx = ["a%sb", " %s%s ", "1%s2", "%s\n%s", "x %s", "%s's", "'%s", "..%s!!", " %s1", "%s \r\n %s", "  %s", "%s\t"]
    rnd.shuffle(edge)
    for i in range(0, len(edge), 4):
        grp = edge[i:i + 4]
        s = ""
        for ch in grp:
            c = ctx[rnd.randrange(len(ctx))]
            s += c % ((ch,) * c.count("%s"))
        out.append(s)
    alphabet = "abcdehtAZ  \n\r\t'.,!?0189éжЖ中文ßıİǅ٣५๓½²ⅷ  ​\x1c_-+=#$<>^~|`«…。()，、।۔،가龥ࠀ"
    for _ in range(1200):
        n = rnd.randint(1, 48)
        out.append("".join(rnd.choice(alphabet) for _ in range(n)))
    # (appended behind the random strings, so that everything above keeps its index) long digit runs and ideographs around the non-ASCII white space
    out += ["1234567", "12345678901234567890", " 1234 5678 9", "a1234567b", "12 345 6789 0", "007 1000000 3.14159", "x=1234567890;y=100000", "٣٣٣٣1234", "123", "1234", "12345", "123456",
            "中\u3000文", "\u2192\u3000\u2192", "中\u2003文\u2028中", "가\u1680龥", "中文\u3000\u3000中文123456", "中\u202f文\u205f中", "a中\u3000文b 12345", "中 文", "中\t文", "...!!?!..", "1234567...!!"]
    for _ in range(200):
        n = rnd.randint(1, 24)
        out.append("".join(rnd.choice("0123456789012345 .中文\u3000가\u2003a") for _ in range(n)))
    return out


def main():
    strs = strings()
    res = {}
    for name, vocab in vocabs().items():
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, name + ".gguf")
            gguf.write_synthetic_llama(path, E=256, H=2, Hkv=1, L=1, F=256, V=len(vocab["tokens"]), seed=3, vocab=vocab)
            lines = os.path.join(td, "lines.txt")
            with open(lines, "w") as f:
                for s in strs:
                    f.write(s.encode("utf-8").hex() + "\n")
            out = subprocess.run([TOK_REF, path, lines], capture_output=True, text=True, check=True).stdout.splitlines()
        toks = [[int(x) for x in l.split()[1:]] for l in out if l.startswith("T")]
        assert len(toks) == len(strs)
        res[name + "_tok"] = np.concatenate([np.asarray(t, np.int32) for t in toks] + [np.zeros(0, np.int32)])
        res[name + "_off"] = np.cumsum([0] + [len(t) for t in toks]).astype(np.int64)
        print(name, len(vocab["tokens"]), "tokens in vocab;", len(strs), "strings;", int(res[name + "_off"][-1]), "token ids")
    raw = [s.encode("utf-8") for s in strs]
    res["str_bytes"] = np.frombuffer(b"".join(raw), np.uint8)
    res["str_off"] = np.cumsum([0] + [len(r) for r in raw]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "tok_kats2.npz"), **res)


if __name__ == "__main__":
    main()
