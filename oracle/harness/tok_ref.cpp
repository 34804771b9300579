// tok_ref.cpp — OUR code linked against the genuine reference (oracle/_ref/libggml_ref.so): tokenises each line of a text file
// with llama_tokenize(add_special=false, parse_special=true) — the call cpp/bridge.cpp:278 makes — and prints the ids, then the
// piece of every token id (llama_token_to_piece, special=true).  Build container only.
// usage: tok_ref model.gguf lines.txt
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>
#include "llama.h"
int main(int argc, char ** argv) {
    if (argc < 3) return 2;
    llama_backend_init();
    llama_log_set([](ggml_log_level, const char *, void *) {}, nullptr);
    llama_model_params mp = llama_model_default_params();
    mp.vocab_only = true;
    llama_model * m = llama_load_model_from_file(argv[1], mp);
    if (!m) { fprintf(stderr, "load failed\n"); return 1; }
    std::ifstream f(argv[2]); std::string line;
    while (std::getline(f, line)) {
        // lines are hex-encoded so that any byte sequence (incl. newlines) can be carried
        std::string text; for (size_t i = 0; i + 1 < line.size(); i += 2) text += (char) strtol(line.substr(i, 2).c_str(), nullptr, 16);
        std::vector<llama_token> t(text.size() + 16);
        int n = llama_tokenize(m, text.c_str(), (int) text.size(), t.data(), (int) t.size(), false, true);
        printf("T"); for (int i = 0; i < n; ++i) printf(" %d", t[i]); printf("\n");
    }
    const int V = llama_n_vocab(m);
    for (int id = 0; id < V; ++id) {
        char buf[256]; int n = llama_token_to_piece(m, id, buf, sizeof buf, 0, true);
        printf("P %d", id); for (int i = 0; i < n; ++i) printf(" %02x", (unsigned char) buf[i]); printf("\n");
    }
    printf("E %d %d\n", llama_token_eos(m), llama_token_eot(m));
    llama_free_model(m);
    return 0;
}
