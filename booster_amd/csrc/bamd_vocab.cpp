// bamd_vocab.cpp — see bamd_vocab.h.  Written from the behaviour of the reference's tokenizer, not from its code.
#include "bamd_vocab.h"
#include "bamd_gguf.h"
#include "bamd_unicode_tables.h"

#include <string.h>
#include <algorithm>
#include <queue>

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
        std::string pre;
        if (g.get_str("tokenizer.ggml.pre", pre) && (pre == "llama3" || pre == "llama-v3" || pre == "llama-bpe")) { pre_llama3 = true; ignore_merges = true; }
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

// ---- SPM (llm_tokenizer_spm) ------------------------------------------------------------------------------------------
namespace {
struct Sym { int prev, next; const char * text; size_t n; };

struct SpmBigram { int left, right; float score; size_t size; };
struct SpmCmp { bool operator()(const SpmBigram & l, const SpmBigram & r) const { return (l.score < r.score) || (l.score == r.score && l.left > r.left); } };

struct SpmSession {
    const BamdVocab & v; std::vector<Sym> syms; std::priority_queue<SpmBigram, std::vector<SpmBigram>, SpmCmp> q;
    std::map<std::string, std::pair<int, int>> rev;
    explicit SpmSession(const BamdVocab & vv) : v(vv) {}
    void try_add(int l, int r) {
        if (l == -1 || r == -1) return;
        const std::string t(syms[(size_t) l].text, syms[(size_t) l].n + syms[(size_t) r].n);
        auto it = v.token_to_id.find(t);
        if (it == v.token_to_id.end() || (size_t) it->second >= v.text.size()) return;
        q.push(SpmBigram{ l, r, v.score[(size_t) it->second], t.size() });
        rev[t] = std::make_pair(l, r);
    }
    int byte_token(unsigned char ch) const {              // llama_byte_to_token_impl: "<0xXX>", else the raw byte as a token
        static const char * hex = "0123456789ABCDEF";
        const char buf[7] = { '<', '0', 'x', hex[ch >> 4], hex[ch & 15], '>', 0 };
        auto it = v.token_to_id.find(buf);
        if (it != v.token_to_id.end()) return it->second;
        auto it2 = v.token_to_id.find(std::string(1, (char) ch));
        return it2 != v.token_to_id.end() ? it2->second : (v.unk >= 0 ? v.unk : 0);
    }
    void resegment(const Sym & s, std::vector<int> & out) {
        const std::string t(s.text, s.n);
        auto it = v.token_to_id.find(t);
        if (it != v.token_to_id.end()) { out.push_back(it->second); return; }
        auto p = rev.find(t);
        if (p == rev.end()) { for (size_t j = 0; j < s.n; ++j) out.push_back(byte_token((unsigned char) s.text[j])); return; }
        resegment(syms[(size_t) p->second.first], out); resegment(syms[(size_t) p->second.second], out);
    }
    void tokenize(const std::string & text, std::vector<int> & out) {
        int index = 0; size_t offs = 0;
        while (offs < text.size()) {
            Sym s; const size_t len = utf8_len((unsigned char) text[offs]);
            s.text = text.c_str() + offs; s.n = std::min(len, text.size() - offs); offs += s.n;
            s.prev = index - 1; s.next = offs == text.size() ? -1 : index + 1; ++index;
            syms.push_back(s);
        }
        for (size_t i = 1; i < syms.size(); ++i) try_add((int) i - 1, (int) i);
        while (!q.empty()) {
            const SpmBigram b = q.top(); q.pop();
            Sym & L = syms[(size_t) b.left]; Sym & R = syms[(size_t) b.right];
            if (L.n == 0 || R.n == 0 || L.n + R.n != b.size) continue;
            L.n += R.n; R.n = 0; L.next = R.next;
            if (R.next >= 0) syms[(size_t) R.next].prev = b.left;
            try_add(L.prev, b.left); try_add(b.left, L.next);
        }
        if (syms.empty()) return;
        for (int i = 0; i != -1; i = syms[(size_t) i].next) resegment(syms[(size_t) i], out);
    }
};

// ---- BPE (llm_tokenizer_bpe) --------------------------------------------------------------------------------------------
struct BpeBigram { int left, right; std::string text; int rank; size_t size; };
struct BpeCmp { bool operator()(const BpeBigram & l, const BpeBigram & r) const { return l.rank > r.rank || (l.rank == r.rank && l.left > r.left); } };

// split of the llama-3 pre-tokeniser regex (llama-vocab.cpp:345-349):
// (?:'[sS]|'[tT]|'[rR][eE]|'[vV][eE]|'[mM]|'[lL][lL]|'[dD])|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+
static std::vector<std::pair<size_t, size_t>> split_llama3(const std::vector<uint32_t> & c) {
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
        if (j == i && is_number(x)) { size_t k = i; while (k < n && k < i + 3 && is_number(c[k])) ++k; j = k; }     // \p{N}{1,3}
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

struct BpeSession {
    const BamdVocab & v; std::vector<Sym> syms, fin;
    std::priority_queue<BpeBigram, std::vector<BpeBigram>, BpeCmp> q;
    explicit BpeSession(const BamdVocab & vv) : v(vv) {}
    void add(int l, int r) {
        if (l == -1 || r == -1) return;
        const std::string a(syms[(size_t) l].text, syms[(size_t) l].n), b(syms[(size_t) r].text, syms[(size_t) r].n);
        auto it = v.bpe_ranks.find(std::make_pair(a, b));
        if (it == v.bpe_ranks.end() || it->second < 0) return;
        q.push(BpeBigram{ l, r, a + b, it->second, a.size() + b.size() });
    }
    void tokenize(const std::string & text, std::vector<int> & out) {
        uint32_t b2u[256]; std::unordered_map<uint32_t, uint8_t> u2b; byte_maps(b2u, u2b);
        const std::vector<uint32_t> cps = utf8_to_cpts(text);
        const auto spans = v.pre_llama3 ? split_llama3(cps) : split_gpt2(cps);
        std::vector<std::string> words;
        for (auto & sp : spans) {                                             // word -> UTF-8 -> byte-level unicode text
            std::string raw; for (size_t k = sp.first; k < sp.second; ++k) append_utf8(raw, cps[k]);
            std::string enc; for (unsigned char ch : raw) append_utf8(enc, b2u[ch]);
            words.push_back(enc);
        }
        int final_prev = -1;
        for (const std::string & word : words) {
            q = decltype(q)(); syms.clear();
            int index = 0; size_t offset = 0;
            if (v.ignore_merges && v.token_to_id.find(word) != v.token_to_id.end()) { syms.push_back(Sym{ -1, -1, word.c_str(), word.size() }); offset = word.size(); }
            while (offset < word.size()) {
                Sym s; const size_t cl = std::min(word.size() - offset, utf8_len((unsigned char) word[offset]));
                s.text = word.c_str() + offset; s.n = cl; offset += cl;
                s.prev = index - 1; s.next = offset == word.size() ? -1 : index + 1; ++index;
                syms.push_back(s);
            }
            for (size_t i = 1; i < syms.size(); ++i) add((int) i - 1, (int) i);
            while (!q.empty()) {
                const BpeBigram b = q.top(); q.pop();
                Sym & L = syms[(size_t) b.left]; Sym & R = syms[(size_t) b.right];
                if (L.n == 0 || R.n == 0) continue;
                if (std::string(L.text, L.n) + std::string(R.text, R.n) != b.text) continue;
                L.n += R.n; R.n = 0; L.next = R.next;
                if (R.next >= 0) syms[(size_t) R.next].prev = b.left;
                add(L.prev, b.left); add(b.left, L.next);
            }
            for (const Sym & s : syms) {
                if (s.n == 0) continue;
                const std::string str(s.text, s.n);
                auto it = v.token_to_id.find(str);
                if (it != v.token_to_id.end()) out.push_back(it->second);
                else for (size_t k = 0; k < str.size(); ) {                   // unknown piece: byte by byte (llama-vocab.cpp:575-590)
                    const size_t cl = std::min(str.size() - k, utf8_len((unsigned char) str[k]));
                    auto bt = v.token_to_id.find(str.substr(k, cl));
                    if (bt != v.token_to_id.end()) out.push_back(bt->second);
                    k += cl;
                }
            }
            (void) final_prev;
        }
    }
};
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
            SpmSession s(*this); s.tokenize(t, out);
            prev_special = false;
        }
        if (add_special && add_eos && eos != -1) out.push_back(eos);
    } else {
        if (add_special && add_bos && bos != -1) out.push_back(bos);
        for (const Frag & f : frags) {
            if (f.is_token) { out.push_back(f.token); continue; }
            BpeSession s(*this); s.tokenize(raw.substr(f.off, f.len), out);
        }
        if (add_special && add_eos && eos != -1) out.push_back(eos);
    }
    return out;
}
