// regex_ref.cpp — OUR code linked against the genuine reference (oracle/_ref/libggml_ref.so): runs the reference's unicode_regex_split
// (cpp/src/unicode.cpp:645) with the regexes of a file (one per line, hex-encoded UTF-8) over hex-encoded text lines and prints the byte length of
// every piece.  Build container only; tests/golden/gen_deepseek_class.py records the character classes of the deepseek-llm
// pre-tokeniser with it.  usage: regex_ref regexes.txt lines.txt
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include "unicode.h"
int main(int argc, char ** argv) {
    if (argc < 3) return 2;
    std::vector<std::string> res; std::string line;
    auto unhex = [](const std::string & l) { std::string t; for (size_t i = 0; i + 1 < l.size(); i += 2) t += (char) strtol(l.substr(i, 2).c_str(), nullptr, 16); return t; };
    { std::ifstream f(argv[1]); while (std::getline(f, line)) if (!line.empty()) res.push_back(unhex(line)); }
    std::ifstream f(argv[2]);
    while (std::getline(f, line)) {
        const std::string text = unhex(line);
        const std::vector<std::string> pieces = unicode_regex_split(text, res);
        printf("P"); for (const auto & p : pieces) printf(" %zu", p.size()); printf("\n");
    }
    return 0;
}
