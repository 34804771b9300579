// unicode_ref.cpp — OUR code linked against the genuine reference (oracle/_ref/libggml_ref.so): prints, for every code point, the
// classes the reference's pre-tokeniser regexes see (unicode_cpt_flags, cpp/src/unicode.h:59: \p{L}, \p{N}, \s, \p{P}) as ranges, one line
// per class: "<class> lo-hi lo-hi ...".  Build container only; tests/golden/gen_unicode_kats.py turns it into tests/golden/unicode_classes.json.
#include <cstdio>
#include <cstdint>
#include "unicode.h"
int main() {
    const char * names[4] = { "letter", "number", "whitespace", "punctuation" };
    for (int k = 0; k < 4; ++k) {
        printf("%s", names[k]);
        long start = -1;
        for (uint32_t cp = 0; cp <= 0x110000; ++cp) {
            bool ok = false;
            if (cp < 0x110000) { const codepoint_flags f = unicode_cpt_flags(cp); ok = k == 0 ? f.is_letter : k == 1 ? f.is_number : k == 2 ? f.is_whitespace : f.is_punctuation; }
            if (ok && start < 0) start = cp;
            if (!ok && start >= 0) { printf(" %lx-%x", start, cp - 1); start = -1; }
        }
        printf("\n");
    }
    return 0;
}
