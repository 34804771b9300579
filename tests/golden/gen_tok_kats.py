"""Generates tests/golden/tok_kats.json with the GENUINE reference tokenizer (oracle/_ref/tok_ref, built by
`make -C oracle _ref/tok_ref`): synthetic SPM and byte-level BPE (llama-3 pre-tokeniser) vocabularies written by
booster_amd.gguf, a set of test strings, and for each the reference's token ids (llama_tokenize(add_special=false,
parse_special=true), the call of cpp/bridge.cpp:278), every token's piece, eos/eot ids.  Build container only."""
import json
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
from booster_amd import gguf  # noqa: E402

TOK_REF = os.path.join(ROOT, "oracle", "_ref", "tok_ref")


def strings(rnd):
    base = ["Hello world", " Hello  world!!", "the quick brown fox", "it's he'll we'Re DON'T", "x = 12345 + 678;\n\n\ty++", "a\n\n b \r\n c   ", "   ",
            "café жж 中文 ok", "<|user|>hi</s> there<s>", "<|begin_of_text|>abc<|eot_id|>def <|start_header_id|>", "tab\there", "end ", " ",
            "100000 2 33 4444", "a'b'c''d", "éé's", "??!! ... --", "mixed123abc456", "\n", "", "ab<0x41>cd"]
    alphabet = "abcdeht  \n'.,!?019éж中"
    base += ["the rain in the land is his", "zq zzz quiz\n\nnext", "tea tree street letter  settle"]
    for _ in range(40):
        base.append("".join(rnd.choice("etaoinshrdlu zq\n") for _ in range(rnd.randint(1, 40))))
    for _ in range(120):
        n = rnd.randint(1, 40)
        base.append("".join(rnd.choice(alphabet) for _ in range(n)))
    return base


def main():
    rnd = random.Random(9)
    out = {}
    for name, vocab in (("spm", gguf.synthetic_spm_vocab()), ("bpe", gguf.synthetic_bpe_vocab()), ("bpe_holes", gguf.synthetic_bpe_vocab_holes())):
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, name + ".gguf")
            gguf.write_synthetic_llama(path, E=256, H=2, Hkv=1, L=1, F=256, V=len(vocab["tokens"]), seed=3, vocab=vocab)
            strs = strings(rnd)
            lines = os.path.join(td, "lines.txt")
            with open(lines, "w") as f:
                for s in strs:
                    f.write(s.encode("utf-8").hex() + "\n")
            res = subprocess.run([TOK_REF, path, lines], capture_output=True, text=True, check=True).stdout.splitlines()
        toks = [[int(x) for x in l.split()[1:]] for l in res if l.startswith("T")]
        pieces = {int(l.split()[1]): "".join(l.split()[2:]) for l in res if l.startswith("P")}
        eos, eot = [int(x) for x in [l for l in res if l.startswith("E")][0].split()[1:]]
        assert len(toks) == len(strs)
        out[name] = dict(strings=[s.encode("utf-8").hex() for s in strs], tokens=toks, pieces=[pieces[i] for i in range(len(pieces))], eos=eos, eot=eot)
    with open(os.path.join(HERE, "tok_kats.json"), "w") as f:
        json.dump(out, f)
    print("wrote tok_kats.json")


if __name__ == "__main__":
    main()
