// bamd_vocab.cpp — see bamd_vocab.h.  The tokenizer side of the bridge (llama_tokenize / llama_token_to_piece of the reference,
// cpp/src/llama-vocab.cpp), reproduced on its own data structures: flat piece tables + one candidate heap for the merges
// (join_greedily), hand-written splitters for the two pre-tokeniser regexes, code-point classes from the reference's own table
// (bamd_unicode_tables.h <- tests/golden/unicode_classes.json).  Pinned token for token by tests/test_tokenizer.py.
#include "bamd_vocab.h"
#include "bamd_gguf.h"
#include "bamd_unicode_tables.h"

#include <string.h>
#include <algorithm>

// ---- UTF-8 helpers --------------------------------------------------------------------------------------------------
static size_t utf8_len(unsigned char c) {              // unicode_len_utf8 (unicode.cpp): by the high nibble
    static const size_t lookup[] = { 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4 };
    return lookup[c >> 4];
}
static std::vector<uint32_t> utf8_to_cpts(const std::string & s) {
    std::vector<uint32_t> out; size_t i = 0;
    while (i < s.size()) {
        const unsigned char c = (unsigned char) s[i];
        size_t n = utf8_len(c); if (i + n > s.size()) n = 1;
        uint32_t cp = c;
        if (n == 2) cp = ((c & 0x1F) << 6) | ((unsigned char) s[i + 1] & 0x3F);
        else if (n == 3) cp = ((c & 0x0F) << 12) | (((unsigned char) s[i + 1] & 0x3F) << 6) | ((unsigned char) s[i + 2] & 0x3F);
        else if (n == 4) cp = ((c & 0x07) << 18) | (((unsigned char) s[i + 1] & 0x3F) << 12) | (((unsigned char) s[i + 2] & 0x3F) << 6) | ((unsigned char) s[i + 3] & 0x3F);
        out.push_back(cp); i += n;
    }
    return out;
}
static void append_utf8(std::string & s, uint32_t cp) {
    if (cp < 0x80) s += (char) cp;
    else if (cp < 0x800) { s += (char) (0xC0 | (cp >> 6)); s += (char) (0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { s += (char) (0xE0 | (cp >> 12)); s += (char) (0x80 | ((cp >> 6) & 0x3F)); s += (char) (0x80 | (cp & 0x3F)); }
    else { s += (char) (0xF0 | (cp >> 18)); s += (char) (0x80 | ((cp >> 12) & 0x3F)); s += (char) (0x80 | ((cp >> 6) & 0x3F)); s += (char) (0x80 | (cp & 0x3F)); }
}
static bool in_ranges(const uint32_t (*r)[2], int n, uint32_t cp) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) { const int mid = (lo + hi) / 2; if (cp < r[mid][0]) hi = mid - 1; else if (cp > r[mid][1]) lo = mid + 1; else return true; }
    return false;
}
static bool is_letter(uint32_t c) { return in_ranges(BAMD_UNI_LETTER, BAMD_UNI_LETTER_N, c); }
static bool is_number(uint32_t c) { return in_ranges(BAMD_UNI_NUMBER, BAMD_UNI_NUMBER_N, c); }
static bool is_space(uint32_t c)  { return in_ranges(BAMD_UNI_SPACE, BAMD_UNI_SPACE_N, c); }
static bool is_punct(uint32_t c)  { return in_ranges(BAMD_UNI_PUNCT, BAMD_UNI_PUNCT_N, c); }

// GPT-2 byte <-> unicode map (bytes_to_unicode): printable bytes map to themselves, the rest to 256+n
static void byte_maps(uint32_t b2u[256], std::unordered_map<uint32_t, uint8_t> & u2b) {
    int n = 0;
    for (int b = 0; b < 256; ++b) {
        const bool keep = (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255);
        b2u[b] = keep ? (uint32_t) b : (uint32_t) (256 + n++);
        u2b[b2u[b]] = (uint8_t) b;
    }
}

// ---- loading --------------------------------------------------------------------------------------------------------
bool BamdVocab::load(const GgufFile & g, std::string & err) {
    // the tokenizer serves the files the engine loads (bamd_engine.cpp: general.architecture "llama" — Llama-2/3, Mistral, deepseek-llm / -coder, SmolLM,
    // Mistral-Nemo (tekken), Viking / Poro ... as GGUF names them); the pre-tokenisers of other architectures' families (qwen2, falcon, starcoder ...) are
    // restated and pinned as well — llm_load_vocab keys on tokenizer.ggml.pre alone
    // (the architecture check lives in the model loader, bamd_engine.cpp: the vocabulary is keyed on tokenizer.ggml.model / .pre alone, as llm_load_vocab is —
    //  vocab-only use on files of other families keeps working)
    std::string model;
    if (!g.get_str("tokenizer.ggml.model", model)) { err = "missing tokenizer.ggml.model"; return false; }
    if (model == "no_vocab") { type = BAMD_VOCAB_NONE; return true; }
    if (model == "llama") { type = BAMD_VOCAB_SPM; bos = 1; eos = 2; unk = 0; add_bos = true; add_space_prefix = true; }
    else if (model == "gpt2") { type = BAMD_VOCAB_BPE; bos = 11; eos = 11; add_bos = false; }
    else { err = "tokenizer.ggml.model \"" + model + "\" is not supported (llama / gpt2 / no_vocab)"; return false; }
    const GgufValue * toks = g.find("tokenizer.ggml.tokens");
    if (!toks || toks->arr_s.empty()) { err = "missing tokenizer.ggml.tokens"; return false; }
    text = toks->arr_s;
    const size_t n = text.size();
    score.assign(n, 0.f); attr.assign(n, BAMD_ATTR_NORMAL);
    if (const GgufValue * sc = g.find("tokenizer.ggml.scores")) if (sc->arr_data && sc->arr_n == n) memcpy(score.data(), sc->arr_data, n * 4);
    if (const GgufValue * tt = g.find("tokenizer.ggml.token_type")) if (tt->arr_data && tt->arr_n == n) {
        const int32_t * t = (const int32_t *) tt->arr_data;
        for (size_t i = 0; i < n; ++i) {      // llama_token_type -> attr (llama.cpp:5715-5730)
            switch (t[i]) {
                case 2: attr[i] = BAMD_ATTR_UNKNOWN; break; case 3: attr[i] = BAMD_ATTR_CONTROL; break;
                case 4: attr[i] = BAMD_ATTR_USER_DEFINED; break; case 5: attr[i] = BAMD_ATTR_UNUSED; break;
                case 6: attr[i] = BAMD_ATTR_BYTE; break; default: attr[i] = BAMD_ATTR_NORMAL; break;
            }
        }
    }
    for (size_t i = 0; i < n; ++i) token_to_id[text[i]] = (int) i;
    if (type == BAMD_VOCAB_BPE) {
        if (const GgufValue * mg = g.find("tokenizer.ggml.merges")) {
            for (size_t i = 0; i < mg->arr_s.size(); ++i) {
                const std::string & w = mg->arr_s[i];
                const size_t p = w.find(' ', 1);
                if (p == std::string::npos) continue;
                bpe_ranks[std::make_pair(w.substr(0, p), w.substr(p + 1))] = (int) i;
            }
        }
        // pre-tokeniser (llm_load_vocab, llama.cpp:5375-5472; regex sets llama-vocab.cpp:340-443): hand-written splitters for the llama-3 regex
        // (and its qwen2 form), the GPT-2 regex and the chains built around it (starcoder family, default, falcon), poro / viking and
        // deepseek-coder / deepseek-llm, tekken — every name llm_load_vocab knows; any other value FAILS the load instead of silently
        // producing a different token stream
        std::string pre;
        g.get_str("tokenizer.ggml.pre", pre);
        if (pre == "llama3" || pre == "llama-v3" || pre == "llama-bpe") { pre_llama3 = true; ignore_merges = true; add_bos = true; }
        else if (pre == "dbrx" || pre == "smaug-bpe" || pre == "chatglm-bpe") pre_llama3 = true;              // same regex, no flags
        else if (pre == "gpt-2" || pre == "phi-2" || pre == "jina-es" || pre == "jina-de" || pre == "jina-v2-es" || pre == "jina-v2-de" ||
                 pre == "jina-v2-code" || pre == "mpt" || pre == "olmo" || pre == "jais") pre_llama3 = false;
        else if (pre == "qwen2" || pre == "stablelm2") { pre_llama3 = true; pre_maxdigits = 1; }
        else if (pre == "starcoder" || pre == "refact" || pre == "command-r" || pre == "smollm" || pre == "codeshell") pre_chain = 1;
        else if (pre.empty() || pre == "default") pre_chain = 2;                 // (a missing key: the reference warns and uses "default")
        else if (pre == "falcon") pre_chain = 3;
        else if (pre == "poro-chat") pre_chain = 4;
        else if (pre == "viking") pre_chain = 5;
        else if (pre == "deepseek-coder") pre_chain = 6;
        else if (pre == "tekken") { pre_chain = 7; ignore_merges = true; add_bos = true; }
        else if (pre == "deepseek-llm") pre_chain = 8;
        else { err = "tokenizer.ggml.pre \"" + pre + "\" is not supported (llama-3 / qwen2, GPT-2, starcoder, default, falcon, poro / viking, deepseek and tekken pre-tokenisers only)"; return false; }
    }
    uint32_t u;
    if (g.get_u32("tokenizer.ggml.bos_token_id", u)) bos = (int) u;
    if (g.get_u32("tokenizer.ggml.eos_token_id", u)) eos = (int) u;
    if (g.get_u32("tokenizer.ggml.unknown_token_id", u)) unk = (int) u;
    if (g.get_u32("tokenizer.ggml.eot_token_id", u)) eot = (int) u;
    if (const GgufValue * v = g.find("tokenizer.ggml.add_bos_token")) add_bos = v->b;
    if (const GgufValue * v = g.find("tokenizer.ggml.add_eos_token")) add_eos = v->b;
    if (const GgufValue * v = g.find("tokenizer.ggml.add_space_prefix")) add_space_prefix = v->b;
    if (eot == -1) {                                      // llama.cpp:5648-5665
        for (const char * cand : { "<|eot_id|>", "<|im_end|>", "<|end|>", "<end_of_turn>", "<|endoftext|>" }) {
            auto it = token_to_id.find(cand);
            if (it != token_to_id.end()) { eot = it->second; break; }
        }
    }
    for (size_t i = 0; i < n; ++i) if (attr[i] & (BAMD_ATTR_CONTROL | BAMD_ATTR_USER_DEFINED | BAMD_ATTR_UNKNOWN)) special.push_back((int) i);
    std::stable_sort(special.begin(), special.end(), [&](int a, int b) { return text[(size_t) a].size() > text[(size_t) b].size(); });
    // token -> piece cache, special = true (llama.cpp:5696-5710, llama_token_to_piece_impl)
    uint32_t b2u[256]; std::unordered_map<uint32_t, uint8_t> u2b; byte_maps(b2u, u2b);
    piece.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const int a = attr[i];
        if (a & (BAMD_ATTR_UNKNOWN | BAMD_ATTR_CONTROL | BAMD_ATTR_USER_DEFINED)) piece[i] = text[i];
        else if (type == BAMD_VOCAB_SPM) {
            if (a & BAMD_ATTR_NORMAL) {
                std::string r = text[i]; size_t p = 0;
                while ((p = r.find("\xe2\x96\x81", p)) != std::string::npos) { r.replace(p, 3, " "); p += 1; }
                piece[i] = r;
            } else if (a & BAMD_ATTR_BYTE) piece[i] = std::string(1, (char) strtol(text[i].substr(3, 2).c_str(), nullptr, 16));   // "<0xAB>"
        } else if (a & BAMD_ATTR_NORMAL) {
            std::string r;
            for (uint32_t cp : utf8_to_cpts(text[i])) { auto it = u2b.find(cp); if (it != u2b.end()) r += (char) it->second; else append_utf8(r, cp); }
            piece[i] = r;
        }
    }
    return true;
}

const std::string & BamdVocab::token_to_piece(int id) const {
    static const std::string empty;
    return id >= 0 && (size_t) id < piece.size() ? piece[(size_t) id] : empty;
}

// ---- greedy pair merging, shared by the SentencePiece and the byte-level BPE tokenizers ------------------------------------------
// Behaviour to reproduce (llm_tokenizer_spm / llm_tokenizer_bpe, llama-vocab.cpp:190-300, :488-590): start from the UTF-8 characters
// of the text; repeatedly join the adjacent pair with the best key — SPM: the highest score among pairs whose concatenation is a
// token; BPE: the lowest merge rank — ties going to the leftmost pair; a join creates up to two new candidate pairs with its
// neighbours.  Own data structures: the pieces are runs of characters described by three flat arrays (no linked list of symbols),
// candidates sit in one binary min-heap keyed by (key, position) and carry the extents they were made for, so a candidate that
// has been overtaken by another join is recognised by comparing extents.
namespace {
struct PieceTable {
    std::vector<uint32_t> off;       // off[c] .. off[c+1]: the bytes of character c
    std::vector<int32_t> last;       // last[h]: last character of the piece that starts at character h   (valid while starts[h])
    std::vector<int32_t> first;      // first[t]: first character of the piece that ends at character t   (valid at piece ends)
    std::vector<uint8_t> starts;     // starts[c]: a piece begins at character c
    int n = 0;
    void init(const char * s, size_t len, size_t (*char_len)(unsigned char)) {
        off.clear(); size_t o = 0;
        while (o < len) { off.push_back((uint32_t) o); o += std::min(len - o, char_len((unsigned char) s[o])); }
        n = (int) off.size(); off.push_back((uint32_t) len);
        last.resize((size_t) n); first.resize((size_t) n); starts.assign((size_t) n, 1);
        for (int c = 0; c < n; ++c) { last[(size_t) c] = c; first[(size_t) c] = c; }
    }
    void whole(size_t len) {                                    // one piece covering everything (BPE ignore_merges)
        off.assign({ 0u, (uint32_t) len }); n = 1; last.assign(1, 0); first.assign(1, 0); starts.assign(1, 1);
    }
    int before(int h) const { return h == 0 ? -1 : first[(size_t) h - 1]; }
    int after(int h) const { const int r = last[(size_t) h] + 1; return r < n ? r : -1; }
    uint32_t lo(int h) const { return off[(size_t) h]; }
    uint32_t hi(int h) const { return off[(size_t) last[(size_t) h] + 1]; }
    void join(int l, int r) { const int t = last[(size_t) r]; last[(size_t) l] = t; first[(size_t) t] = l; starts[(size_t) r] = 0; }
};

struct Cand { double key; int32_t l, r, l_last, r_last; };      // a candidate join of the pieces starting at l and r, as they were when it was made
struct CandHeap {                                                // binary min-heap on (key, l)
    std::vector<Cand> a;
    static bool less(const Cand & x, const Cand & y) { return x.key < y.key || (x.key == y.key && x.l < y.l); }
    bool empty() const { return a.empty(); }
    void push(const Cand & c) {
        a.push_back(c); size_t i = a.size() - 1;
        while (i > 0) { const size_t p = (i - 1) / 2; if (!less(a[i], a[p])) break; std::swap(a[i], a[p]); i = p; }
    }
    Cand pop() {
        const Cand top = a[0]; a[0] = a.back(); a.pop_back();
        size_t i = 0; const size_t m = a.size();
        for (;;) {
            size_t b = i; const size_t l = 2 * i + 1, r = l + 1;
            if (l < m && less(a[l], a[b])) b = l;
            if (r < m && less(a[r], a[b])) b = r;
            if (b == i) break;
            std::swap(a[i], a[b]); i = b;
        }
        return top;
    }
};

// KEY(l, r, &key) -> is the pair of pieces (l, r) joinable, and with which key (smaller joins first)
template <typename KEY>
static void join_greedily(PieceTable & pt, KEY && key_of) {
    CandHeap heap;
    auto offer = [&](int l, int r) {
        if (l < 0 || r < 0) return;
        double k;
        if (key_of(l, r, k)) heap.push(Cand{ k, l, r, pt.last[(size_t) l], pt.last[(size_t) r] });
    };
    for (int c = 0; c + 1 < pt.n; ++c) offer(c, c + 1);
    while (!heap.empty()) {
        const Cand c = heap.pop();
        // still the two pieces this candidate was made for?  (either may have been swallowed, or have grown, since)
        if (!pt.starts[(size_t) c.l] || !pt.starts[(size_t) c.r] || pt.last[(size_t) c.l] != c.l_last || pt.last[(size_t) c.r] != c.r_last) continue;
        pt.join(c.l, c.r);
        offer(pt.before(c.l), c.l);
        offer(c.l, pt.after(c.l));
    }
}

// ---- SentencePiece (llama-vocab.cpp:190-300) ---------------------------------------------------------------------------------
static int spm_byte_token(const BamdVocab & v, unsigned char ch) {      // llama_byte_to_token_impl: "<0xXX>", else the raw byte as a token
    static const char * hex = "0123456789ABCDEF";
    const char buf[7] = { '<', '0', 'x', hex[ch >> 4], hex[ch & 15], '>', 0 };
    auto it = v.token_to_id.find(buf);
    if (it != v.token_to_id.end()) return it->second;
    auto it2 = v.token_to_id.find(std::string(1, (char) ch));
    return it2 != v.token_to_id.end() ? it2->second : (v.unk >= 0 ? v.unk : 0);
}
static void spm_tokenize(const BamdVocab & v, const std::string & text, std::vector<int> & out) {
    if (text.empty()) return;
    PieceTable pt; pt.init(text.data(), text.size(), utf8_len);
    std::string tmp;
    join_greedily(pt, [&](int l, int r, double & key) {
        tmp.assign(text, pt.lo(l), pt.hi(r) - pt.lo(l));
        auto it = v.token_to_id.find(tmp);
        if (it == v.token_to_id.end() || (size_t) it->second >= v.text.size()) return false;
        key = -(double) v.score[(size_t) it->second];            // highest score first (exact: float -> double -> negation)
        return true;
    });
    for (int h = 0; h != -1; h = pt.after(h)) {                  // pieces left to right; a piece that is no token falls back to its bytes
        tmp.assign(text, pt.lo(h), pt.hi(h) - pt.lo(h));
        auto it = v.token_to_id.find(tmp);
        if (it != v.token_to_id.end()) out.push_back(it->second);
        else for (unsigned char ch : tmp) out.push_back(spm_byte_token(v, ch));
    }
}

// ---- byte-level BPE (llama-vocab.cpp:488-590) ----------------------------------------------------------------------------------
// split of the llama-3 pre-tokeniser regex (llama-vocab.cpp:345-349):
// (?:'[sS]|'[tT]|'[rR][eE]|'[vV][eE]|'[mM]|'[lL][lL]|'[dD])|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+
static std::vector<std::pair<size_t, size_t>> split_llama3(const std::vector<uint32_t> & c, size_t maxdigits) {     // maxdigits 3: llama-3; 1: qwen2 (\\p{N})
    std::vector<std::pair<size_t, size_t>> out; const size_t n = c.size(); size_t i = 0;
    auto lower = [](uint32_t x) { return x >= 'A' && x <= 'Z' ? x + 32 : x; };
    while (i < n) {
        size_t j = i;
        const uint32_t x = c[i];
        if (x == '\'' && i + 1 < n) {
            const uint32_t a = lower(c[i + 1]);
            if (a == 's' || a == 't' || a == 'm' || a == 'd') j = i + 2;
            else if (i + 2 < n) { const uint32_t b = lower(c[i + 2]); if ((a == 'r' && b == 'e') || (a == 'v' && b == 'e') || (a == 'l' && b == 'l')) j = i + 3; }
        }
        if (j == i) {                                                         // [^\r\n\p{L}\p{N}]?\p{L}+
            size_t k = i;
            if (!(x == '\r' || x == '\n' || is_letter(x) || is_number(x)) && i + 1 < n && is_letter(c[i + 1])) k = i + 1;
            if (k < n && is_letter(c[k])) { while (k < n && is_letter(c[k])) ++k; j = k; }
        }
        if (j == i && is_number(x)) { size_t k = i; while (k < n && k < i + maxdigits && is_number(c[k])) ++k; j = k; }     // \p{N}{1,3} / \p{N}
        if (j == i) {                                                         //  ?[^\s\p{L}\p{N}]+[\r\n]*
            size_t k = i; if (x == ' ' && i + 1 < n) k = i + 1;
            if (k < n && !is_space(c[k]) && !is_letter(c[k]) && !is_number(c[k])) {
                while (k < n && !is_space(c[k]) && !is_letter(c[k]) && !is_number(c[k])) ++k;
                while (k < n && (c[k] == '\r' || c[k] == '\n')) ++k;
                j = k;
            }
        }
        if (j == i && is_space(x)) {
            size_t e = i; while (e < n && is_space(c[e])) ++e;                // whitespace run [i, e)
            size_t last_nl = (size_t) -1; for (size_t k = i; k < e; ++k) if (c[k] == '\r' || c[k] == '\n') last_nl = k;
            if (last_nl != (size_t) -1) j = last_nl + 1;                      // \s*[\r\n]+
            else if (e == n) j = e;                                           // \s+(?!\S) at end of text
            else if (e - i >= 2) j = e - 1;                                   // \s+(?!\S): leave one for the next word
            else j = e;                                                       // \s+
        }
        if (j == i) j = i + 1;
        out.emplace_back(i, j); i = j;
    }
    return out;
}
// tekken (Mistral-Nemo; llama-vocab.cpp:428-434): the case classes of the original regex are approximated in the reference by look-aheads,
//   U = (?=[\p{L}])([^a-z]) = a letter that is not ASCII lower case,  W = (?=[\p{L}])([^A-Z]) = a letter that is not ASCII upper case
// (a non-ASCII letter is both), and the regex is  P?U*W+ | P?U+W* | \p{N} |  ?[^\s\p{L}\p{N}]+[\r\n/]* | \s*[\r\n]+ | \s+(?!\S) | \s+
// with P = [^\r\n\p{L}\p{N}].  ECMAScript semantics: first alternative that matches, greedy quantifiers with backtracking — U* gives
// characters back until a W can follow.
static std::vector<std::pair<size_t, size_t>> split_tekken(const std::vector<uint32_t> & c) {
    std::vector<std::pair<size_t, size_t>> out; const size_t n = c.size(); size_t i = 0;
    auto isU = [](uint32_t x) { return is_letter(x) && !(x >= 'a' && x <= 'z'); };
    auto isW = [](uint32_t x) { return is_letter(x) && !(x >= 'A' && x <= 'Z'); };
    while (i < n) {
        size_t j = i;
        const uint32_t x = c[i];
        const bool isP = !(x == '\r' || x == '\n' || is_letter(x) || is_number(x));
        for (int p = isP ? 1 : 0; p >= 0 && j == i; --p) {                    // P?U*W+
            const size_t k = i + (size_t) p;
            size_t umax = 0; while (k + umax < n && isU(c[k + umax])) ++umax;
            for (size_t u = umax + 1; u-- > 0 && j == i; ) {
                size_t w = 0; while (k + u + w < n && isW(c[k + u + w])) ++w;
                if (w >= 1) j = k + u + w;
            }
        }
        for (int p = isP ? 1 : 0; p >= 0 && j == i; --p) {                    // P?U+W*
            const size_t k = i + (size_t) p;
            size_t u = 0; while (k + u < n && isU(c[k + u])) ++u;
            if (u >= 1) { size_t w = 0; while (k + u + w < n && isW(c[k + u + w])) ++w; j = k + u + w; }
        }
        if (j == i && is_number(x)) j = i + 1;                                // \p{N}
        if (j == i) {                                                         //  ?[^\s\p{L}\p{N}]+[\r\n/]*
            size_t k = i; if (x == ' ' && i + 1 < n) k = i + 1;
            if (k < n && !is_space(c[k]) && !is_letter(c[k]) && !is_number(c[k])) {
                while (k < n && !is_space(c[k]) && !is_letter(c[k]) && !is_number(c[k])) ++k;
                while (k < n && (c[k] == '\r' || c[k] == '\n' || c[k] == '/')) ++k;
                j = k;
            }
        }
        if (j == i && is_space(x)) {
            size_t e = i; while (e < n && is_space(c[e])) ++e;                // whitespace run [i, e)
            size_t last_nl = (size_t) -1; for (size_t k = i; k < e; ++k) if (c[k] == '\r' || c[k] == '\n') last_nl = k;
            if (last_nl != (size_t) -1) j = last_nl + 1;                      // \s*[\r\n]+
            else if (e == n) j = e;                                           // \s+(?!\S) at end of text
            else if (e - i >= 2) j = e - 1;                                   // \s+(?!\S): leave one for the next word
            else j = e;                                                       // \s+
        }
        if (j == i) j = i + 1;
        out.emplace_back(i, j); i = j;
    }
    return out;
}
// default GPT-2 style: 's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+
static std::vector<std::pair<size_t, size_t>> split_gpt2(const std::vector<uint32_t> & c) {
    std::vector<std::pair<size_t, size_t>> out; const size_t n = c.size(); size_t i = 0;
    while (i < n) {
        size_t j = i; const uint32_t x = c[i];
        if (x == '\'' && i + 1 < n) {
            const uint32_t a = c[i + 1];
            if (a == 's' || a == 't' || a == 'm' || a == 'd') j = i + 2;
            else if (i + 2 < n) { const uint32_t b = c[i + 2]; if ((a == 'r' && b == 'e') || (a == 'v' && b == 'e') || (a == 'l' && b == 'l')) j = i + 3; }
        }
        if (j == i) {
            size_t k = i; if (x == ' ' && i + 1 < n) k = i + 1;
            if (k < n && is_letter(c[k])) { while (k < n && is_letter(c[k])) ++k; j = k; }
            else if (k < n && is_number(c[k])) { while (k < n && is_number(c[k])) ++k; j = k; }
            else if (k < n && !is_space(c[k])) { while (k < n && !is_space(c[k]) && !is_letter(c[k]) && !is_number(c[k])) ++k; j = k; }
        }
        if (j == i && is_space(x)) {
            size_t e = i; while (e < n && is_space(c[e])) ++e;
            j = (e == n || e - i < 2) ? e : e - 1;
        }
        if (j == i) j = i + 1;
        out.emplace_back(i, j); i = j;
    }
    return out;
}

// ---- regex CHAINS (unicode_regex_split, unicode.cpp:645-800): every regex splits every piece the previous ones left — its matches become
//      pieces and so do the gaps between them (unicode_regex_split_stl, :485-513) ----
typedef std::vector<std::pair<size_t, size_t>> Spans;
// one simple regex over [lo, hi): `len(i)` = length of the match that starts at i (0 = none)
template <typename F> static void split_by(const std::vector<uint32_t> & c, const Spans & in, Spans & out, F len) {
    out.clear();
    for (const auto & sp : in) {
        size_t start = sp.first, i = sp.first;
        while (i < sp.second) {
            const size_t m = len(i, sp.second);
            if (m == 0) { ++i; continue; }
            if (i > start) out.emplace_back(start, i);
            out.emplace_back(i, i + m);
            i += m; start = i;
        }
        if (start < sp.second) out.emplace_back(start, sp.second);
    }
}
static Spans split_chain(const std::vector<uint32_t> & c, int chain) {
    Spans a = { { 0, c.size() } }, b;
    if (c.empty()) return Spans();
    auto gpt2 = [&](const Spans & in, Spans & out) {           // the GPT-2 regex inside every piece (unicode_regex_split_custom_gpt2 works per piece)
        out.clear();
        for (const auto & sp : in) {
            const std::vector<uint32_t> sub(c.begin() + (long) sp.first, c.begin() + (long) sp.second);
            for (const auto & q : split_gpt2(sub)) out.emplace_back(sp.first + q.first, sp.first + q.second);
        }
    };
    if (chain == 4 || chain == 5) {                            // poro / viking: " ?[^(\\s|.,!?…。，、।۔،)]+" (viking: then "\\p{N}")
        auto ok = [&](uint32_t x) { return !(is_space(x) || x == '(' || x == '|' || x == '.' || x == ',' || x == '!' || x == '?' || x == ')' || x == 0x2026 || x == 0x3002 ||
                                             x == 0xFF0C || x == 0x3001 || x == 0x0964 || x == 0x06D4 || x == 0x060C); };
        split_by(c, a, b, [&](size_t i, size_t hi) { size_t k = i; if (c[i] == ' ' && i + 1 < hi && ok(c[i + 1])) k = i + 1; size_t e = k; while (e < hi && ok(c[e])) ++e; return e > k ? e - i : (size_t) 0; });
        if (chain == 4) return b;
        split_by(c, b, a, [&](size_t i, size_t) { return is_number(c[i]) ? (size_t) 1 : (size_t) 0; });
        return a;
    }
    if (chain == 6) {                                          // deepseek-coder: "[\r\n]", "\\s?\\p{L}+", "\\s?\\p{P}+", "[一-龥ࠀ-一가-퟿]+", "\\p{N}"
        split_by(c, a, b, [&](size_t i, size_t) { return c[i] == '\r' || c[i] == '\n' ? (size_t) 1 : (size_t) 0; });
        split_by(c, b, a, [&](size_t i, size_t hi) { size_t k = i; if (is_space(c[i]) && i + 1 < hi && is_letter(c[i + 1])) k = i + 1; size_t e = k; while (e < hi && is_letter(c[e])) ++e; return e > k ? e - i : (size_t) 0; });
        split_by(c, a, b, [&](size_t i, size_t hi) { size_t k = i; if (is_space(c[i]) && i + 1 < hi && is_punct(c[i + 1])) k = i + 1; size_t e = k; while (e < hi && is_punct(c[e])) ++e; return e > k ? e - i : (size_t) 0; });
        // "[一-龥ࠀ-一가-퟿]+" as std::wregex sees it on the reference's wtext, where non-ASCII white space has become 0x0B (unicode.cpp:786-792): the class
        // recorded code point by code point for deepseek-llm's identical regex (U+1680, U+2000-200A, U+2028/9, U+202F, U+205F, U+3000 are holes)
        split_by(c, b, a, [&](size_t i, size_t hi) { size_t e = i; while (e < hi && in_ranges(BAMD_UNI_DSCJK, BAMD_UNI_DSCJK_N, c[e])) ++e; return e - i; });
        split_by(c, a, b, [&](size_t i, size_t) { return is_number(c[i]) ? (size_t) 1 : (size_t) 0; });
        return b;
    }
    if (chain == 8) {                                          // deepseek-llm: "[\r\n]", "\\s?[W]+", "\\s?[P]+", "\\s+$", "[C]+", "\\p{N}+" — W, P, C: what std::wregex makes of
        // the literal classes of the reference's regexes, recorded code point by code point through its own unicode_regex_split
        // (tests/golden/gen_deepseek_class.py; P includes ':'..'~', so the ASCII letters: regex 3 re-splits the words of regex 2)
        auto in = [](const uint32_t (*r)[2], int n, uint32_t x) { return in_ranges(r, n, x); };
        auto cls_run = [&](const uint32_t (*r)[2], int nr, bool space_first) {
            return [&, r, nr, space_first](size_t i, size_t hi) {
                size_t k = i; if (space_first && is_space(c[i]) && i + 1 < hi && in(r, nr, c[i + 1])) k = i + 1;
                size_t e = k; while (e < hi && in(r, nr, c[e])) ++e;
                return e > k ? e - i : (size_t) 0;
            };
        };
        split_by(c, a, b, [&](size_t i, size_t) { return c[i] == '\r' || c[i] == '\n' ? (size_t) 1 : (size_t) 0; });
        split_by(c, b, a, cls_run(BAMD_UNI_DSWORD, BAMD_UNI_DSWORD_N, true));
        split_by(c, a, b, cls_run(BAMD_UNI_DSPUNCT, BAMD_UNI_DSPUNCT_N, true));
        split_by(c, b, a, [&](size_t i, size_t hi) { for (size_t k = i; k < hi; ++k) if (!is_space(c[k])) return (size_t) 0; return hi - i; });      // "\\s+$": to the end of the piece
        split_by(c, a, b, cls_run(BAMD_UNI_DSCJK, BAMD_UNI_DSCJK_N, false));
        split_by(c, b, a, [&](size_t i, size_t hi) { size_t k = i; while (k < hi && is_number(c[k])) ++k; return k - i; });
        return a;
    }
    if (chain == 1) {                                          // "\\p{N}" then GPT-2
        split_by(c, a, b, [&](size_t i, size_t) { return is_number(c[i]) ? (size_t) 1 : (size_t) 0; });
        gpt2(b, a);
        return a;
    }
    const bool falcon = chain == 3;
    auto punct = [&](uint32_t x) { return is_punct(x) || x == '$' || x == '+' || x == '<' || x == '=' || x == '>' || x == '^' || x == '~' || x == '|' || (falcon && x == '`'); };
    split_by(c, a, b, [&](size_t i, size_t hi) { size_t k = i; while (k < hi && punct(c[k])) ++k; return k - i; });       // "[\\p{P}\\$\\+<=>\\^~\\|]+" (falcon: + `)
    gpt2(b, a);
    auto three_digits = [&](size_t i, size_t hi) { return i + 3 <= hi && c[i] >= '0' && c[i] <= '9' && c[i + 1] >= '0' && c[i + 1] <= '9' && c[i + 2] >= '0' && c[i + 2] <= '9' ? (size_t) 3 : (size_t) 0; };
    if (falcon) { split_by(c, a, b, three_digits); return b; }                                                                                          // "[0-9][0-9][0-9]"
    // default (also: no tokenizer.ggml.pre): FOUR regexes, llama-vocab.cpp:437-442 — "\\p{N}+" and then "[0-9][0-9][0-9]": 1234567 -> 123 | 456 | 7
    split_by(c, a, b, [&](size_t i, size_t hi) { size_t k = i; while (k < hi && is_number(c[k])) ++k; return k - i; });
    split_by(c, b, a, three_digits);
    return a;
}

static void bpe_tokenize(const BamdVocab & v, const std::string & text, std::vector<int> & out) {
    uint32_t b2u[256]; std::unordered_map<uint32_t, uint8_t> u2b; byte_maps(b2u, u2b);
    const std::vector<uint32_t> cps = utf8_to_cpts(text);
    const auto spans = v.pre_chain == 7 ? split_tekken(cps) : v.pre_chain ? split_chain(cps, v.pre_chain) : v.pre_llama3 ? split_llama3(cps, (size_t) v.pre_maxdigits) : split_gpt2(cps);
    std::string raw, word, lt, rt;
    PieceTable pt;
    for (const auto & sp : spans) {
        raw.clear(); for (size_t k = sp.first; k < sp.second; ++k) append_utf8(raw, cps[k]);                   // the word as UTF-8 ...
        word.clear(); for (unsigned char ch : raw) append_utf8(word, b2u[ch]);                                 // ... then as byte-level unicode text
        if (v.ignore_merges && v.token_to_id.find(word) != v.token_to_id.end()) pt.whole(word.size());       // llama-3: a word that is a token stays whole
        else {
            pt.init(word.data(), word.size(), utf8_len);
            join_greedily(pt, [&](int l, int r, double & key) {
                lt.assign(word, pt.lo(l), pt.hi(l) - pt.lo(l)); rt.assign(word, pt.lo(r), pt.hi(r) - pt.lo(r));
                auto it = v.bpe_ranks.find(std::make_pair(lt, rt));
                if (it == v.bpe_ranks.end() || it->second < 0) return false;
                key = (double) it->second;                       // lowest merge rank first
                return true;
            });
        }
        if (pt.n == 0) continue;
        for (int h = 0; h != -1; h = pt.after(h)) {
            lt.assign(word, pt.lo(h), pt.hi(h) - pt.lo(h));
            auto it = v.token_to_id.find(lt);
            if (it != v.token_to_id.end()) { out.push_back(it->second); continue; }
            for (size_t k = 0; k < lt.size(); ++k) {             // unknown piece: one raw BYTE of the byte-level text at a time (llama-vocab.cpp:575-584:
                auto bt = v.token_to_id.find(lt.substr(k, 1));   // std::string(1, *j)), dropping what has no token — so the two-byte byte-level
                if (bt != v.token_to_id.end()) out.push_back(bt->second);    // characters such as U+0120 never match there, as in the reference
            }
        }
    }
}
}  // namespace

// ---- llama_tokenize_internal ------------------------------------------------------------------------------------------
std::vector<int> BamdVocab::tokenize(const std::string & raw, bool add_special, bool parse_special) const {
    std::vector<int> out;
    if (type == BAMD_VOCAB_NONE) return out;
    struct Frag { bool is_token; int token; size_t off, len; };
    std::vector<Frag> frags;
    if (!raw.empty()) frags.push_back(Frag{ false, -1, 0, raw.size() });
    for (int sid : special) {                                                   // tokenizer_st_partition
        const std::string & st = text[(size_t) sid];
        if (st.empty()) continue;
        if (!parse_special && (attr[(size_t) sid] & (BAMD_ATTR_CONTROL | BAMD_ATTR_UNKNOWN))) continue;
        std::vector<Frag> next;
        for (const Frag & f : frags) {
            if (f.is_token) { next.push_back(f); continue; }
            size_t base = f.off, len = f.len;
            while (true) {
                const size_t m = raw.find(st, base);
                if (m == std::string::npos || m + st.size() > base + len) { if (len > 0) next.push_back(Frag{ false, -1, base, len }); break; }
                if (m > base) next.push_back(Frag{ false, -1, base, m - base });
                next.push_back(Frag{ true, sid, 0, 0 });
                const size_t consumed = m + st.size() - base;
                base += consumed; len -= consumed;
                if (len == 0) break;
            }
        }
        frags.swap(next);
    }
    if (type == BAMD_VOCAB_SPM) {
        bool prev_special = true;
        if (add_special && add_bos && bos != -1) out.push_back(bos);
        for (const Frag & f : frags) {
            if (f.is_token) { out.push_back(f.token); prev_special = true; continue; }
            std::string t = raw.substr(f.off, f.len);
            if (add_space_prefix && prev_special) t = " " + t;
            size_t p = 0; while ((p = t.find(' ', p)) != std::string::npos) { t.replace(p, 1, "\xe2\x96\x81"); p += 3; }   // llama_escape_whitespace
            spm_tokenize(*this, t, out);
            prev_special = false;
        }
        if (add_special && add_eos && eos != -1) out.push_back(eos);
    } else {
        if (add_special && add_bos && bos != -1) out.push_back(bos);
        for (const Frag & f : frags) {
            if (f.is_token) { out.push_back(f.token); continue; }
            bpe_tokenize(*this, raw.substr(f.off, f.len), out);
        }
        if (add_special && add_eos && eos != -1) out.push_back(eos);
    }
    return out;
}
