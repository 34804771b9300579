#!/usr/bin/env python3
"""tests/golden/unicode_classes.json["ds_word" / "ds_punct" / "ds_cjk"]: the code points that the three literal character classes of the
reference's deepseek-llm pre-tokeniser (regexes 2, 3 and 5 of cpp/src/llama-vocab.cpp:360-369) match — RECORDED FROM THE REFERENCE'S
BEHAVIOUR: the regexes are read from the reference's source at generation time and run, one at a time, through the reference's own
unicode_regex_split (oracle/_ref/regex_ref = oracle/harness/regex_ref.cpp linked against oracle/_ref/libggml_ref.so) over one probe string
per code point X, "X<m>X" with <m> a known member of the class: three pieces -> X is no member, one piece -> it is.
(The classes are what std::wregex makes of the literals in this build, which is not what they say on paper: regex 2 lists several hundred
letter ranges and matches ASCII and full-width letters only.)  Data only.  Build container only."""
import ast
import json
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
REGEX_REF = os.path.join(ROOT, "oracle", "_ref", "regex_ref")
SRC = "/root/reference/cpp/src/llama-vocab.cpp"


def reference_regexes():
    src = open(SRC, encoding="utf-8").read()
    body = src[src.index("case LLAMA_VOCAB_PRE_TYPE_DEEPSEEK_LLM:"):]
    body = body[:body.index("break;")]
    return [ast.literal_eval('"' + m + '"') for m in re.findall(r'^\s*"((?:[^"\\]|\\.)*)",\s*$', body, re.M)]


def members(regex, marker):
    cps = [cp for cp in range(1, 0x110000) if not (0xD800 <= cp <= 0xDFFF)]
    with tempfile.TemporaryDirectory() as td:
        rf, lf = os.path.join(td, "re.txt"), os.path.join(td, "lines.txt")
        open(rf, "w").write(regex.encode("utf-8").hex() + "\n")
        with open(lf, "w") as f:
            for cp in cps:
                f.write((chr(cp) + marker + chr(cp)).encode("utf-8").hex() + "\n")
        rows = subprocess.run([REGEX_REF, rf, lf], capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(rows) == len(cps)
    ok = {}
    for cp, l in zip(cps, rows):
        n = len(l.split()) - 1
        assert n in (1, 2, 3), (hex(cp), l)                # (2: X is white space and joins the marker through the optional \\s in front)
        ok[cp] = n == 1
    ranges, start = [], None
    for cp in range(0, 0x110001):
        m = ok.get(cp, False)
        if m and start is None: start = cp
        if not m and start is not None: ranges.append([start, cp - 1]); start = None
    return ranges


def main():
    res = reference_regexes()
    assert len(res) == 6 and res[0] == "[\r\n]" and res[3] == "\\s+$" and res[5] == "\\p{N}+", res
    p = os.path.join(HERE, "unicode_classes.json")
    d = json.load(open(p))
    d.pop("deepseek_llm_word", None)
    for key, rx, marker in (("ds_word", res[1], "a"), ("ds_punct", res[2], "!"), ("ds_cjk", res[4], "一")):
        d[key] = members(rx, marker)
        print(key, len(d[key]), "ranges,", sum(b - a + 1 for a, b in d[key]), "code points; first", d[key][:8])
    json.dump(d, open(p, "w"), separators=(",", ":"))


if __name__ == "__main__":
    main()
