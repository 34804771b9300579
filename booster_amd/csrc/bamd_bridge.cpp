// bamd_bridge.cpp — level 2 of the C-ABI: the nine cgo symbols of gotzmann/booster (include/booster_bridge.h) on top of the
// MI355X runtime (include/bamd.h), plus the Janus sampler.  Restates the behaviour of cpp/bridge.cpp (init_context :118-171,
// do_inference :175-658, C wrappers :697-835) and cpp/janus.cpp (sample_janus_token :191-331, initJanus :410-490, tokType
// :723-800, isLower :826-850, isPedantic :381-392) — written from their behaviour, sharing no code with them.
#include "../../include/bamd.h"
#include "../../include/booster_bridge.h"
#include "bamd_gguf.h"
#include "bamd_vocab.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <exception>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#define BAMD_API extern "C" __attribute__((visibility("default")))
const GgufFile * bamd_model_gguf(const bamd_model * m);

namespace {
const int EOS_HARDCODED = 2;                     // cpp/janus.h:18 (hard-coded id, kept as is)
enum { LANG_ZERO = 0, LANG_EN = 2, SPACE_EN = 20, LANG_RU = 3, SPACE_RU = 30, LANG_OTHER = 4, SPACE_OTHER = 40 };

struct JanusParams { int32_t janus = 1, depth = 200; float scale = 0.96f, hi = 0.99f, lo = 0.96f; };

// one layer-split stage: its slice of the model on one device, the context's own (non-blocking) stream as the stage stream, the
// hand-off buffer the previous stage writes over xGMI, and the event that orders the neighbours behind this stage's work
// hand-off buffers and events come in PAIRS, used alternately by consecutive evaluations (parity of Pod::seq): stage s may start on
// micro-batch k + 1 while stage s + 1 still reads micro-batch k from the other buffer — the prompt pipelines across the stages the way the
// reference's scheduler does with its input copies (GGML_SCHED_MAX_COPIES, cpp/ggml/src/ggml-backend.c:1030, :1751-1844)
struct Stage {
    bamd_model * model = nullptr; bamd_context * ctx = nullptr; int device = 0, vdevice = 0, layer_first = 0, layer_last = 0; void * hidden_in[2] = { nullptr, nullptr }; hipEvent_t done[2] = { nullptr, nullptr };
    void * stream() const { return bamd_context_stream(ctx); }
};

struct Pod {
    std::vector<Stage> stages;
    BamdVocab vocab;
    int n_ctx = 0, n_predict = 0, n_batch = 512, n_vocab = 0, n_embd = 0;
    unsigned seq = 0;                            // evaluations handed through the stages so far (parity picks the hand-off buffer / event)
    JanusParams jp;
    std::vector<float> scales, types;            // per-vocab Janus tables (cpp/janus.cpp:36-37 keeps them global)
    bool janus_ready = false;
    std::mt19937 rng;
    std::atomic<bool> stop{ false };
    std::vector<float> logits;                   // host copy the sampler may modify in place
    bool gpu_sampler = false;                    // Janus penalties + shortlist on the device (bamd_logits_shortlist); env BAMD_JANUS_GPU=0 disables
    std::chrono::steady_clock::time_point eval_t0; int eval_n = 0; bool eval_open = false;   // an evaluation whose wait happens in the sampler (pod_decode)
    int64_t n_sample_dev = 0, n_sample_host = 0; // tokens sampled from the device shortlist / through the host path
    // llama_timings equivalents (llama.cpp:18527-18551)
    double t_p_eval_ms = 0, t_eval_ms = 0; int64_t n_p_eval = 0, n_eval = 0;
};

// The text of a job (prompt echo + generated pieces), append-only.  status() hands out a pointer into the current buffer; a poller may
// hold it for as long as the job exists (Go copies it at once, but the contract must not depend on that): a buffer is never freed or
// moved — when the text outgrows it, a buffer of twice the capacity takes over and the old one stays behind (retired memory <= the
// final text).  Appending inside a buffer keeps every reader's view NUL-terminated: the new terminator is written first, then the
// piece from its last byte to its first, which overwrites the old terminator last.  That ordering is what an UNLOCKED reader (Go's
// C.GoString on the pointer status() returned) relies on; the writer enforces it with release stores (append), not with the host's store order
// (the reference hands out std::string::c_str() of a string it keeps appending to, with no ordering at all).
// Growth: a job keeps <= 2x its text; the table itself is bounded — beyond BAMD_MAX_JOBS entries the oldest FINISHED jobs are dropped, 1/4 of the
// table at a time (their status() pointers die with them: a server that polls a job it started tens of thousands of jobs ago gets "").
#define BAMD_MAX_JOBS 65536
struct Job {
    std::vector<std::unique_ptr<char[]>> bufs;
    char * cur_buf = nullptr; size_t cap = 0, len = 0;
    int64_t prompt_eval = 0, timing = 0, prompt_tokens = 0; uint32_t seed = 0;
    bool finished = false; uint64_t serial = 0;  // set when doInference returns / insertion order (retirement of old jobs)
    const char * c_str() { if (!cur_buf) grow(64); return cur_buf; }
    void grow(size_t need) {
        size_t ncap = cap ? cap : 64; while (ncap < need) ncap *= 2;
        std::unique_ptr<char[]> nb(new char[ncap]);
        if (cur_buf) memcpy(nb.get(), cur_buf, len);
        nb[len] = 0;
        cur_buf = nb.get(); cap = ncap; bufs.push_back(std::move(nb));
    }
    void append(const std::string & piece) {
        if (piece.empty()) return;
        if (!cur_buf || len + piece.size() + 1 > cap) grow(len + piece.size() + 1);
        // every byte is a RELEASE store (round 6: rounds 3-5 used volatile stores and relied on x86-64's store order): each one is ordered behind the new
        // terminator and the bytes after it on any host, so a reader that sees byte i also sees a terminator somewhere behind it.  (The reader is
        // C.GoString on a bare char *: its side of the contract cannot be made atomic from here; the reference does not order anything.)
        char * b = cur_buf;
        __atomic_store_n(&b[len + piece.size()], (char) 0, __ATOMIC_RELEASE);
        for (size_t i = piece.size(); i-- > 0; ) __atomic_store_n(&b[len + i], piece[i], __ATOMIC_RELEASE);
        len += piece.size();
    }
};

std::mutex g_mu;
std::unordered_map<std::string, Job> g_jobs;
uint64_t g_job_serial = 0;
// call with g_mu held, before a job starts: keeps the table bounded
void retire_old_jobs() {
    if (g_jobs.size() < BAMD_MAX_JOBS) return;
    std::vector<std::pair<uint64_t, std::string>> done;
    for (auto & kv : g_jobs) if (kv.second.finished) done.push_back({ kv.second.serial, kv.first });
    std::sort(done.begin(), done.end());
    const size_t n = std::min(done.size(), (size_t) BAMD_MAX_JOBS / 4);
    for (size_t i = 0; i < n; ++i) g_jobs.erase(done[i].second);
}
Pod * g_pods[8] = { nullptr };
std::string g_debug;
bool dbg(const char * what) { return g_debug.find(what) != std::string::npos; }

// ---- Janus ------------------------------------------------------------------------------------------------------------
int tok_type(const std::string & in) {            // cpp/janus.cpp:723-800
    int en = 0, ru = 0, other = 0; bool space = false;
    const unsigned char * b = (const unsigned char *) in.data(); const size_t n = in.size();
    if (n > 0 && b[0] == 0x20) space = true;
    for (size_t i = 0; i < n; i++) {
        if ((b[i] >= 0x41 && b[i] <= 0x5A) || (b[i] >= 0x61 && b[i] <= 0x7A)) { en++; continue; }
        if (b[i] < 0x80) continue;
        if (b[i] == 0xD0 && i + 1 < n) { i++; if ((b[i] >= 0x90 && b[i] <= 0xBF) || b[i] == 0x81) ru++; else other++; continue; }
        if (b[i] == 0xD1 && i + 1 < n) { i++; if ((b[i] >= 0x80 && b[i] <= 0x8F) || b[i] == 0x91) ru++; else other++; continue; }
        if (b[i] >= 0xC3 && b[i] < 0xE3) { i++; other++; continue; }
        if (b[i] >= 0xE3 && b[i] < 0xF0) { i += 2; other++; continue; }
        if (b[i] >= 0xF0) { i += 3; continue; }
    }
    if (space) { if (other) return SPACE_OTHER; if (en) return SPACE_EN; if (ru) return SPACE_RU; }
    if (other) return LANG_OTHER; if (en) return LANG_EN; if (ru) return LANG_RU;
    return LANG_ZERO;
}
bool is_lower(const std::string & in) {           // cpp/janus.cpp:826-850
    const unsigned char * b = (const unsigned char *) in.data(); const size_t n = in.size();
    if (n == 0) return false;
    if (b[0] >= 0x61 && b[0] <= 0x7A) return true;
    if (b[0] == 0xD0 && n >= 2 && b[1] >= 0xB0 && b[1] <= 0xBF) return true;
    if (b[0] == 0xD1 && n >= 2 && ((b[1] >= 0x80 && b[1] <= 0x8F) || b[1] == 0x91)) return true;
    return false;
}
bool is_pedantic(const std::string & t) {         // cpp/janus.cpp:381-392 (an empty piece parses as a number there, too)
    char * end; strtol(t.c_str(), &end, 10);
    if (*end == 0) return true;
    if (t == " *" || t == " =" || t == " -" || t == " +") return true;
    if (t == "{" || t == "}" || t == "[" || t == "]") return true;
    if (t == " {" || t == " }" || t == " [" || t == " ]") return true;
    if (t == "<|end_of_text|>" || t == "```") return true;
    return false;
}
void init_janus(Pod & p) {                        // cpp/janus.cpp:410-490
    JanusParams & jp = p.jp;
    if (jp.depth <= 0) jp.depth = 200;
    if (jp.scale <= 0.0 || jp.scale > 1.0) jp.scale = 0.97f;
    if (jp.hi <= 0.0 || jp.hi > 1.0) jp.hi = 0.99f;
    if (jp.lo <= 0.0 || jp.lo > 1.0) jp.lo = 0.96f;
    const float scale = jp.scale;
    static const float probes[] = { 0.20f, 0.22f, 0.25f, 0.28f, 0.30f, 0.32f, 0.33f, 0.35f, 0.36f, 0.38f,
                                    0.40f, 0.42f, 0.44f, 0.45f, 0.46f, 0.48f, 0.50f, 0.52f, 0.53f, 0.55f };
    p.scales.assign((size_t) p.n_vocab, 0.f); p.types.assign((size_t) p.n_vocab, 0.f);
    for (int id = 0; id < p.n_vocab; id++) {
        const std::string & piece = p.vocab.token_to_piece(id);
        const int type = tok_type(piece); const bool lower = is_lower(piece); const size_t len = piece.size();
        p.types[(size_t) id] = (float) type;
        if (is_pedantic(piece)) { p.scales[(size_t) id] = (float) (1.0 - (1.0 - scale) * 0.20); continue; }
        // the reference indexes probes[len/2] / probes[len] without a bound (UB past 20 entries): clamped here
        if (type == LANG_RU && lower) { p.scales[(size_t) id] = (float) (1.0 - (1.0 - scale) * probes[std::min<size_t>(len / 2, 19)]); continue; }
        if (type == LANG_EN && lower) { p.scales[(size_t) id] = (float) (1.0 - (1.0 - scale) * probes[std::min<size_t>(len, 19)]); continue; }
        p.scales[(size_t) id] = scale;
    }
    // second half of initJanus (cpp/janus.cpp:528-700): fixed overrides.  The reference writes them without bounds (an id past the
    // vocabulary, or eot = -1, is a wild heap write there); here such ids are skipped.
    const int V = p.n_vocab;
    auto put = [&](int id, float v) { if (id >= 0 && id < V) p.scales[(size_t) id] = v; };
    auto part = [&](double k) { return (float) (1.0 - (1.0 - scale) * k); };
    put(0, 1.0f); put(p.vocab.eos, scale); put(p.vocab.eot, scale);
    // (the reference tests llama_model_desc for "llama" / "mistral": always true for the one architecture this engine loads)
    if (V > 128000) {                              // Llama-3 vocabulary: by piece, then by id range and class
        for (int id = 0; id < V; id++) {
            const std::string & t = p.vocab.token_to_piece(id);
            if (t == "\n" || t == "\n\n" || t == " " || t == "," || t == ".") { put(id, part(0.10)); continue; }
            if (t == "  " || t == "    ") { put(id, part(0.20)); continue; }
            if (t == " \xe2\x80\x94" || t == "-" || t == ":" || t == ";" || t == " (" || t == ")." || t == " )" || t == ")" || t == "(") { put(id, part(0.30)); continue; }
            const int type = (int) p.types[(size_t) id];
            if (type == SPACE_RU && id < 50000) { put(id, part(id < 20000 ? 0.30 : id < 35000 ? 0.40 : 0.50)); continue; }
            if (type == SPACE_EN && id < 1100) { put(id, part(id < 500 ? 0.30 : id < 800 ? 0.40 : 0.50)); continue; }
        }
    } else {                                       // Llama-2 / Mistral vocabulary: a table of token ids
        static const struct { double k; int ids[20]; } table[] = {
            { 0.10, { 13, 29871, 29892, -1 } },
            { 0.20, { 259, 268, 29889, -1 } },
            { 0.30, { 813, 29899, 29901, 29936, 313, 467, 1723, 29897, 29898, 490, 531, 606, 614, 263, 278, 297, 304, 310, 322, -1 } },
            { 0.35, { 665, 733, 863, 363, 372, 373, 385, 393, 408, 411, -1 } },
            { 0.40, { 1077, 1097, 1186, 470, 472, 526, -1 } },
            { 0.45, { 1447, 1538, 1604, 1685, -1 } },
            { 0.50, { 4281, 857, 939, 1651, 319, -1 } },
        };
        put(EOS_HARDCODED, scale);
        for (const auto & row : table) for (int i = 0; i < 20 && row.ids[i] >= 0; i++) put(row.ids[i], part(row.k));
    }
    p.janus_ready = true;
}

struct Cand { int id; float logit, p; };
float janus_cutoff(const Pod & p, int topToken);
// the per-vocabulary tables of the device sampler, on the stage that owns the output layer; false = stay on the host sampler
bool upload_sampler_tables(Pod & p) {
    const char * e = getenv("BAMD_JANUS_GPU");
    p.gpu_sampler = false;
    if ((e && atoi(e) == 0) || p.stages.empty()) return false;
    std::vector<uint8_t> cls((size_t) p.n_vocab); std::vector<float> cut((size_t) p.n_vocab);
    for (int id = 0; id < p.n_vocab; id++) {
        const float t = p.types[(size_t) id];
        cls[(size_t) id] = (t == LANG_EN || t == LANG_OTHER) ? 1 : 0;
        cut[(size_t) id] = janus_cutoff(p, id);
    }
    if (bamd_sampler_tables(p.stages.back().ctx, cls.data(), cut.data(), p.n_vocab)) return false;
    if (p.stages.size() == 1) bamd_set_logits_readback(p.stages[0].ctx, 0);
    p.gpu_sampler = true;
    return true;
}

// cpp/janus.cpp:191-331 + llama_sample_token (llama-sampling.cpp:32-58, :610-631)
// The shortlist of sample_janus_token (janus.cpp:262-300): ALL candidates sorted by logit (std::sort, descending), the cut-off
// chosen from the top token, then everything from the first candidate with logit / topLogit < cutoff dropped.
//   slow path: exactly that — a full sort of the vocabulary (≈10 ms at V = 128256, several times a decode step on the GPU);
//   fast path: when the top logit is positive and unique, logit / top is monotone along the sorted order, so the shortlist is the
//     set { logit / top >= cutoff } — two linear passes and a sort of the (typically tiny) shortlist.  Equal logits inside the
//     shortlist would make the order depend on std::sort's internals over the whole array: then the slow path runs.
// Same result in every case (bamd_janus_shortlist_test compares the two).
template <typename CutoffFn>
static void janus_shortlist(const float * logits, size_t V, CutoffFn cutoff_of, bool allow_fast, std::vector<Cand> & cand) {
    cand.clear();
    if (allow_fast && V > 0) {
        size_t top = 0; size_t ntop = 1;
        for (size_t id = 1; id < V; id++) {
            if (logits[id] > logits[top]) { top = id; ntop = 1; }
            else if (logits[id] == logits[top]) ++ntop;
        }
        const float topLogit = logits[top];
        if (topLogit > 0.0f && ntop == 1) {
            const float cutoff = cutoff_of((int) top);
            for (size_t id = 0; id < V; id++) if (!(logits[id] / topLogit < cutoff)) cand.push_back(Cand{ (int) id, logits[id], 0.0f });
            std::sort(cand.data(), cand.data() + cand.size(), [](const Cand & a, const Cand & b) { return a.logit > b.logit; });
            bool ties = cand.empty() || cand[0].id != (int) top;
            for (size_t i = 1; i < cand.size() && !ties; i++) ties = !(cand[i].logit < cand[i - 1].logit);   // equal, or a NaN that broke the ordering
            if (!ties) return;
            cand.clear();
        }
    }
    cand.reserve(V);
    for (int id = 0; id < (int) V; id++) cand.push_back(Cand{ id, logits[id], 0.0f });
    std::sort(cand.data(), cand.data() + cand.size(), [](const Cand & a, const Cand & b) { return a.logit > b.logit; });
    const float topLogit = cand[0].logit;
    const float cutoff = cutoff_of(cand[0].id);
    for (size_t i = 1; i < cand.size(); i++) if (cand[i].logit / topLogit < cutoff) { cand.resize(i); break; }
}

// the cut-off of the shortlist, chosen by the top token (janus.cpp:303-306)
float janus_cutoff(const Pod & p, int topToken) {
    const float topType = p.types[(size_t) topToken];
    return (is_pedantic(p.vocab.token_to_piece(topToken)) || topType == LANG_RU || topType == LANG_EN) ? p.jp.hi : p.jp.lo;
}
// llama_sample_token (llama-sampling.cpp:610-631): softmax over the sorted shortlist, one draw from std::discrete_distribution on the pod's mt19937
static int janus_draw(Pod & p, std::vector<Cand> & cand) {
    const float max_l = cand[0].logit; float cum = 0.0f;
    for (auto & c : cand) { c.p = expf(c.logit - max_l); cum += c.p; }
    std::vector<float> probs; probs.reserve(cand.size());
    for (auto & c : cand) { c.p /= cum; probs.push_back(c.p); }
    std::discrete_distribution<> dist(probs.begin(), probs.end());
    return cand[(size_t) dist(p.rng)].id;
}

int sample_janus(Pod & p, float * logits, const std::vector<int> & last_tokens, size_t promptLen, size_t pos, size_t max) {
    const size_t V = (size_t) p.n_vocab, ctxSize = last_tokens.size();
    const int lastToken = last_tokens[ctxSize - 1];
    const float lastType = p.types[(size_t) lastToken];
    if (V > (size_t) EOS_HARDCODED) logits[EOS_HARDCODED] *= 1.0 + log(1.0 + float(pos - promptLen) / float(max)) * 0.05;
    const size_t depth = std::min((size_t) p.jp.depth, pos - promptLen);
    for (size_t i = 0; i < depth; i++) {
        const int id = last_tokens[ctxSize - 1 - i];
        const float curType = p.types[(size_t) id];
        if ((lastType == SPACE_RU || lastType == LANG_RU) && curType == LANG_RU) { logits[id] *= 1.0 - (1.0 - p.scales[(size_t) id]) * 0.20; continue; }
        logits[id] *= p.scales[(size_t) id];
    }
    for (size_t id = 0; id < V; id++) {
        const float curType = p.types[id];
        if ((lastType == SPACE_RU || lastType == LANG_RU) && (curType == LANG_EN || curType == LANG_OTHER)) logits[id] *= 0.5;
    }
    std::vector<Cand> cand;
    janus_shortlist(logits, V, [&](int topToken) { return janus_cutoff(p, topToken); }, true, cand);
    return janus_draw(p, cand);
}

// The same sampler with the O(n_vocab) passes on the device: the penalties become a short list of per-token factors
// (a token that occurs k times in the window is multiplied k times, in sequence, as the reference's loop does), the x0.5 pass and
// the ratio test run where the logits are, and only the shortlist comes back.  Whenever the order of the result could depend on the
// reference's full sort the (already penalised) logits are read back and the host shortlist runs on them: same result either way.
static void pod_eval_done(Pod & p) {
    if (!p.eval_open) return;
    p.eval_open = false;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p.eval_t0).count();
    if (p.eval_n == 1) { p.t_eval_ms += ms; p.n_eval += 1; } else { p.t_p_eval_ms += ms; p.n_p_eval += p.eval_n; }
}
int sample_janus_device(Pod & p, const std::vector<int> & last_tokens, size_t promptLen, size_t pos, size_t max) {
    const size_t V = (size_t) p.n_vocab, ctxSize = last_tokens.size();
    Stage & st = p.stages.back();
    const float lastType = p.types[(size_t) last_tokens[ctxSize - 1]];
    const bool ru_context = lastType == SPACE_RU || lastType == LANG_RU;
    std::vector<bamd_logit_penalty> pen;
    std::unordered_map<int, size_t> slot;
    auto entry = [&](int id) -> bamd_logit_penalty & {
        auto it = slot.find(id);
        if (it == slot.end()) { it = slot.emplace(id, pen.size()).first; pen.push_back(bamd_logit_penalty{ id, 0, 0, 1.0f, 1.0, 0.0 }); }
        return pen[it->second];
    };
    if (V > (size_t) EOS_HARDCODED) entry(EOS_HARDCODED).pre = 1.0 + log(1.0 + float(pos - promptLen) / float(max)) * 0.05;
    const size_t depth = std::min((size_t) p.jp.depth, pos - promptLen);
    for (size_t i = 0; i < depth; i++) {
        const int id = last_tokens[ctxSize - 1 - i];
        bamd_logit_penalty & e = entry(id);
        if (ru_context && p.types[(size_t) id] == LANG_RU) { e.kind = 1; e.d = 1.0 - (1.0 - p.scales[(size_t) id]) * 0.20; }
        else { e.kind = 0; e.f = p.scales[(size_t) id]; }
        e.count++;
    }
    std::vector<Cand> cand;
    bool host_path = pen.size() > (size_t) BAMD_PENALTY_CAP;
    if (!host_path) {
        bamd_shortlist_head head;
        std::vector<int32_t> ids((size_t) BAMD_SHORTLIST_CAP); std::vector<float> vals((size_t) BAMD_SHORTLIST_CAP);
        void * stream = st.stream();
        if (bamd_logits_shortlist(st.ctx, pen.data(), (int) pen.size(), ru_context ? 1 : 0, &head, ids.data(), vals.data(), stream)) return -1;
        pod_eval_done(p);
        host_path = head.nan || !(head.top_logit > 0.0f) || head.ntop != 1 || head.count < 1 || head.count > BAMD_SHORTLIST_CAP;
        if (!host_path) {
            for (int i = 0; i < head.count; i++) cand.push_back(Cand{ ids[(size_t) i], vals[(size_t) i], 0.0f });
            std::sort(cand.data(), cand.data() + cand.size(), [](const Cand & a, const Cand & b) { return a.logit > b.logit; });
            host_path = cand[0].id != head.top_id;
            for (size_t i = 1; i < cand.size() && !host_path; i++) host_path = !(cand[i].logit < cand[i - 1].logit);
        }
        if (host_path) {                          // the device logits already carry the penalties: shortlist them on the host
            const float * lg = p.stages.size() == 1 ? bamd_get_logits(st.ctx) : bamd_stage_get_logits(st.ctx, st.stream());
            if (!lg) return -1;
            janus_shortlist(lg, V, [&](int topToken) { return janus_cutoff(p, topToken); }, true, cand);
        }
    } else {                                      // more distinct penalised tokens than the device list holds: the host sampler
        const float * lg = p.stages.size() == 1 ? bamd_get_logits(st.ctx) : bamd_stage_get_logits(st.ctx, st.stream());
        pod_eval_done(p);
        if (!lg) return -1;
        memcpy(p.logits.data(), lg, V * 4);
        p.n_sample_host++;
        return sample_janus(p, p.logits.data(), last_tokens, promptLen, pos, max);
    }
    if (host_path) p.n_sample_host++; else p.n_sample_dev++;
    return janus_draw(p, cand);
}

// ---- model placement: Booster's gpus: split (cpp/bridge.cpp:745-750, llama.cpp:5932-5969) ----------------------------------
// llm_load_tensors (llama.cpp:5932-5969) with the bridge's settings (bridge.cpp:745-750: n_gpu_layers = gpu1+..+gpu4, tensor_split = gpuN):
// only the first `device_count` entries of the split count; if those are all zero the reference splits by free device memory (equal
// here: identical GPUs); layer i lives on upper_bound(cumulative normalised splits, i / act), the output layer with fraction (act-1)/act.
// BAMD_MAX_GPUS = 8: the nine symbols carry four weights (gpu1..gpu4, cpp/bridge.cpp:745-750 -> tensor_split[0..3]); the environment
// variable BOOSTER_GPUS="w0,w1,...,w7"  replaces them with up to eight — same rule, more devices.
#define BAMD_MAX_GPUS 8
bool plan_stages(int n_layer, const int * gpu, int n_gpu, int device_count, std::vector<std::pair<int, std::pair<int, int>>> & out, std::string & err) {
    if (n_layer < 1) { err = "model has no layers"; return false; }
    int n_gpu_layers = 0;
    for (int i = 0; i < n_gpu; ++i) n_gpu_layers += gpu[i];
    if (n_gpu_layers <= 0) { err = "the gpu weights are all zero: this build has no CPU path"; return false; }
    if (n_gpu_layers <= n_layer) { err = "sum(gpuN) must exceed the layer count (partial CPU offload is not supported: no CPU path)"; return false; }
    const int dc = std::min(device_count, n_gpu);
    if (dc < 1) { err = "no HIP device"; return false; }
    const int act = std::min(n_gpu_layers, n_layer + 1);
    bool all_zero = true;
    for (int i = 0; i < dc; ++i) all_zero = all_zero && gpu[i] == 0;
    float splits[BAMD_MAX_GPUS], sum = 0.f;
    for (int i = 0; i < dc; ++i) { sum += all_zero ? 1.0f : (float) gpu[i]; splits[i] = sum; }
    for (int i = 0; i < dc; ++i) splits[i] /= sum;
    auto dev_of = [&](int i) { const float f = (float) i / (float) act; int d = 0; while (d < dc - 1 && !(f < splits[d])) ++d; return d; };   // std::upper_bound
    int start = 0, cur = dev_of(0);
    for (int il = 1; il <= n_layer; ++il) {
        const int d = il < n_layer ? dev_of(il) : -1;
        if (d != cur) { out.push_back({ cur, { start, il } }); start = il; cur = d; }
    }
    // the output layer lives with the last fraction (llama.cpp:5961-5966); it must be the device of the last layers
    const int dout = dev_of(act - 1);
    if (dout != out.back().first) out.push_back({ dout, { n_layer, n_layer } });
    return true;
}
// gpu1..gpu4, or BOOSTER_GPUS (comma-separated non-negative integers, up to eight); returns the number of weights
int gpu_weights(int gpu1, int gpu2, int gpu3, int gpu4, int * w) {
    w[0] = gpu1; w[1] = gpu2; w[2] = gpu3; w[3] = gpu4;
    const char * e = getenv("BOOSTER_GPUS");
    if (!e || !*e) return 4;
    int n = 0;
    while (*e && n < BAMD_MAX_GPUS) {
        char * end; const long v = strtol(e, &end, 10);
        if (end == e) break;
        w[n++] = (int) std::max(0l, v);
        e = *end == ',' ? end + 1 : end;
        if (*end != ',') break;
    }
    return n > 0 ? n : 4;
}

// need_logits = false (a prompt micro-batch that is not the last one before sampling): nothing is read back and the host does not wait — the
// next micro-batch is enqueued behind this one and the stages overlap; the next call that does need the logits waits for everything
int pod_decode(Pod & p, const int * tokens, int n, int n_past, bool need_logits = true) {          // llama_decode for one micro-batch (<= 512 tokens)
    const auto t0 = std::chrono::steady_clock::now();
    if (p.stages.size() == 1) {
        if (bamd_decode(p.stages[0].ctx, tokens, n, n_past)) return 1;
        if (!p.gpu_sampler) memcpy(p.logits.data(), bamd_get_logits(p.stages[0].ctx), (size_t) p.n_vocab * 4);     // else the logits stay on the device
    } else {
        // prompt micro-batches go through every stage as ONE batch (hidden state [n][n_embd] handed to the next device); single tokens,
        // and shapes without batched kernels, step token by token
        // Stream-ordered hand-off, no host synchronisation per hop: stage s works on its own stream, behind the event of stage s-1 for THIS
        // evaluation (whose work ends with the peer write of the hidden state into this stage's hand-off buffer of this parity) and behind
        // the event of stage s+1 for the evaluation before last (which read the buffer of this parity that this stage is about to
        // overwrite).  A single decoded token therefore crosses the devices as one chain of device-side dependencies — each stage
        // replaying its captured graph — and the host waits once, at the end (llama_decode's own synchronisation point).
        auto ordered = [&](size_t s, int par) -> int {
            Stage & st = p.stages[s];
            if (hipSetDevice(st.device) != hipSuccess) return 1;
            if (s > 0 && hipStreamWaitEvent((hipStream_t) st.stream(), p.stages[s - 1].done[par], 0) != hipSuccess) return 1;
            if (s + 1 < p.stages.size() && hipStreamWaitEvent((hipStream_t) st.stream(), p.stages[s + 1].done[par], 0) != hipSuccess) return 1;   // never recorded yet: no-op
            return 0;
        };
        auto stamp = [&](size_t s, int par) -> int { return hipEventRecord(p.stages[s].done[par], (hipStream_t) p.stages[s].stream()) != hipSuccess; };
        bool batched = n > 1 && n <= 512;
        if (batched) {
            const int par = (int) (p.seq & 1u);
            for (size_t s = 0; s < p.stages.size() && batched; ++s) {
                Stage & st = p.stages[s];
                const bool last = s + 1 == p.stages.size();
                void * hout = last ? nullptr : p.stages[s + 1].hidden_in[par];
                if (ordered(s, par)) return 1;
                const int rc = bamd_stage_prefill(st.ctx, s == 0 ? tokens : nullptr, n, n_past, st.hidden_in[par], hout, last && need_logits ? 1 : 0, st.stream());
                if (rc == 2 && s == 0) { batched = false; break; }            // no batched kernels for this model: per-token path below
                if (rc || stamp(s, par)) return 1;
            }
            if (batched) p.seq += 1;
        }
        const int prefill = n > 1;
        for (int t = 0; t < n && !batched; ++t) {
            const int par = (int) (p.seq & 1u);
            for (size_t s = 0; s < p.stages.size(); ++s) {
                Stage & st = p.stages[s];
                const bool last = s + 1 == p.stages.size();
                void * hout = last ? nullptr : p.stages[s + 1].hidden_in[par];       // lives on the NEXT device; peer write
                if (ordered(s, par)) return 1;
                if (bamd_stage_step(st.ctx, tokens[t], nullptr, n_past + t, st.hidden_in[par], hout, last && t == n - 1 && need_logits, prefill, st.stream())) return 1;
                if (stamp(s, par)) return 1;
            }
            p.seq += 1;
        }
        Stage & lst = p.stages.back();
        if (!need_logits) { }                      // nothing to wait for: the next evaluation queues behind this one on every stage
        else if (!p.gpu_sampler) {
            const float * lg = bamd_stage_get_logits(lst.ctx, lst.stream());      // synchronises the last stage's stream: everything before it is done
            if (!lg) return 1;
            memcpy(p.logits.data(), lg, (size_t) p.n_vocab * 4);
        } else {                                  // the logits stay on the last device; llama_decode's synchronisation still applies
            hipSetDevice(lst.device);
            if (hipStreamSynchronize((hipStream_t) lst.stream()) != hipSuccess) return 1;
        }
    }
    if (p.stages.size() == 1 && p.gpu_sampler) {
        // bamd_decode did not wait (no read-back: the sampler prefilter follows on the same stream): the evaluation's time ends where sample_janus_device's one
        // wait returns — the clock keeps running until then (pod_eval_done; the micro-batches of a prompt share one interval), so that timing() / promptEval()
        // still report wall time per token
        if (!p.eval_open) { p.eval_t0 = t0; p.eval_n = 0; p.eval_open = true; }
        p.eval_n += n;
        return 0;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (n == 1) { p.t_eval_ms += ms; p.n_eval += 1; } else { p.t_p_eval_ms += ms; p.n_p_eval += n; }    // llama_synchronize, llama.cpp:18527-18551
    return 0;
}

void pod_free(Pod * p) {
    if (!p) return;
    for (auto & s : p->stages) {
        hipSetDevice(s.device);
        for (int k = 0; k < 2; ++k) { if (s.done[k]) hipEventDestroy(s.done[k]); if (s.hidden_in[k]) hipFree(s.hidden_in[k]); }
        if (s.ctx) bamd_context_free(s.ctx);
        if (s.model) bamd_model_free(s.model);
    }
    delete p;
}
}  // namespace

// ===========================================================================================================================
BAMD_API void init(char * swap, char * debug) {
    (void) swap;
    std::lock_guard<std::mutex> lk(g_mu);
    g_debug = debug ? debug : "";
    bamd_backend_init();
}

static void * init_context_impl(int idx, char * modelName, int batch_size, int gpu1, int gpu2, int gpu3, int gpu4, int context, int predict,
                                int32_t janus, int32_t depth, float scale, float hi, float lo, char * debug) {
    if (idx < 0 || idx >= 8 || !modelName) return nullptr;
    { std::lock_guard<std::mutex> lk(g_mu); g_debug = debug ? debug : ""; }
    const std::string path = modelName;                       // the Go side leaks its C.CString; we keep our own copy anyway
    std::unique_ptr<Pod, void (*)(Pod *)> pod(new Pod(), pod_free);
    const int ndev = bamd_device_count();
    if (ndev <= 0) { fprintf(stderr, "initContext: %s\n", bamd_last_error()); return nullptr; }
    // first look at the file to learn the layer count (a stage with no layers and no tensors is cheap to load)
    bamd_model * probe = bamd_model_load(path.c_str(), 0, 0, 0, 0, 0);
    if (!probe) { fprintf(stderr, "initContext: error: failed to load model '%s': %s\n", path.c_str(), bamd_last_error()); return nullptr; }
    const int n_layer = bamd_model_n_layer(probe);
    std::string err;
    if (!pod->vocab.load(*bamd_model_gguf(probe), err)) { fprintf(stderr, "initContext: tokenizer: %s\n", err.c_str()); bamd_model_free(probe); return nullptr; }
    pod->n_vocab = bamd_model_n_vocab(probe); pod->n_embd = bamd_model_n_embd(probe);
    const int n_ctx_train = bamd_model_n_ctx_train(probe);
    bamd_model_free(probe);
    int gpu[BAMD_MAX_GPUS] = { 0 };
    const int n_gpu = gpu_weights(gpu1, gpu2, gpu3, gpu4, gpu);
    // BAMD_VIRTUAL_DEVICES=N (tests): plan the split as if N devices were present and place device d on physical device d mod ndev —
    // the whole multi-stage path (stage streams, events, hand-off copies, stage graphs) then runs on a box with fewer GPUs
    int ndev_plan = ndev;
    if (const char * e = getenv("BAMD_VIRTUAL_DEVICES")) { const int v = atoi(e); if (v >= 1 && v <= BAMD_MAX_GPUS) ndev_plan = v; }
    std::vector<std::pair<int, std::pair<int, int>>> plan;
    if (!plan_stages(n_layer, gpu, n_gpu, ndev_plan, plan, err)) { fprintf(stderr, "initContext: %s\n", err.c_str()); return nullptr; }
    int n_ctx = context > 0 ? context : n_ctx_train;          // n_ctx 0 = from model (llama.cpp:16640)
    n_ctx = (n_ctx + 31) / 32 * 32;
    pod->n_ctx = n_ctx; pod->n_predict = predict;
    pod->n_batch = (batch_size > 0 && batch_size <= n_ctx) ? batch_size : 512;          // cpp/bridge.cpp:152-160 (GPU branch)
    pod->jp.janus = janus; pod->jp.depth = depth; pod->jp.scale = scale; pod->jp.hi = hi; pod->jp.lo = lo;
    for (size_t s = 0; s < plan.size(); ++s) {
        Stage st; st.device = ndev_plan != ndev ? plan[s].first % ndev : plan[s].first;
        st.vdevice = plan[s].first; st.layer_first = plan[s].second.first; st.layer_last = plan[s].second.second;
        if (st.device >= ndev) { fprintf(stderr, "initContext: gpu%d requested but only %d HIP device(s) present\n", st.device + 1, ndev); return nullptr; }
        const bool first = s == 0, last = s + 1 == plan.size();
        st.model = bamd_model_load(path.c_str(), st.device, plan[s].second.first, plan[s].second.second, first, last);
        if (!st.model) { fprintf(stderr, "initContext: error: failed to load model '%s': %s\n", path.c_str(), bamd_last_error()); return nullptr; }
        pod->stages.push_back(st);
        Stage & ref = pod->stages.back();
        ref.ctx = bamd_context_new(ref.model, n_ctx);
        if (!ref.ctx) { fprintf(stderr, "initContext: error: failed to create context: %s\n", bamd_last_error()); return nullptr; }
        hipSetDevice(ref.device);
        for (int k = 0; k < 2; ++k) if (hipEventCreateWithFlags(&ref.done[k], hipEventDisableTiming) != hipSuccess) return nullptr;
        if (!first) {
            hipSetDevice(ref.device);
            for (int k = 0; k < 2; ++k) if (hipMalloc(&ref.hidden_in[k], (size_t) 512 * pod->n_embd * 4) != hipSuccess) return nullptr;     // a prompt micro-batch of hidden states, twice
            // let the producer's device write the hand-off buffer directly over xGMI (the reference: cudaDeviceEnablePeerAccess, ggml-cuda.cu:1304)
            const int prev = pod->stages[s - 1].device;
            if (prev != ref.device) { int can = 0; hipDeviceCanAccessPeer(&can, prev, ref.device); if (can) { hipSetDevice(prev); hipDeviceEnablePeerAccess(ref.device, 0); } }
        }
    }
    pod->logits.assign((size_t) pod->n_vocab, 0.f);
    Pod * raw = pod.release();
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_pods[idx]) pod_free(g_pods[idx]);
    g_pods[idx] = raw;
    return raw;
}

// the cgo symbol: sampling parameters other than Janus' are accepted and ignored exactly as the reference's bridge ignores them
// (cpp/bridge.cpp:118-171 stores them, do_inference only ever calls the Janus sampler); no C++ exception may cross into Go
BAMD_API void * initContext(int idx, char * modelName, int threads, int batch_size, int gpu1, int gpu2, int gpu3, int gpu4, int context, int predict,
                            int32_t mirostat, float mirostat_tau, float mirostat_eta, float temperature, int top_k, float top_p, float typical_p,
                            float repetition_penalty, int penalty_last_n, int32_t janus, int32_t depth, float scale, float hi, float lo, uint32_t seed,
                            char * debug) {
    (void) threads; (void) mirostat; (void) mirostat_tau; (void) mirostat_eta; (void) temperature; (void) top_k; (void) top_p; (void) typical_p;
    (void) repetition_penalty; (void) penalty_last_n; (void) seed;
    try { return init_context_impl(idx, modelName, batch_size, gpu1, gpu2, gpu3, gpu4, context, predict, janus, depth, scale, hi, lo, debug); }
    catch (const std::exception & e) { fprintf(stderr, "initContext: %s\n", e.what()); return nullptr; }
}

static int64_t do_inference_impl(int idx, void * ctx, char * jobID, char * prompt) {
    if (idx < 0 || idx >= 8 || !ctx || !jobID || !prompt) return 1;
    Pod & p = *(Pod *) ctx;
    const std::string job = jobID, text = prompt;
    p.t_p_eval_ms = p.t_eval_ms = 0; p.n_p_eval = p.n_eval = 0; p.eval_open = false;   // llama_reset_timings
    // an evaluation that no sampler call followed (stopInference during the prompt, the n_ctx - 4 exit, an error return) was never waited for: wait on every
    // way out, close its timing interval and report a give-up of that evaluation
    struct OpenEval { Pod & p; bool failed = false; void close() { if (p.eval_open) { if (bamd_synchronize(p.stages[0].ctx)) { fprintf(stderr, "doInference: %s\n", bamd_last_error()); failed = true; } pod_eval_done(p); } }
                      ~OpenEval() { close(); } } open_eval{ p };
    p.stop.store(false);
    if (!p.janus_ready) { init_janus(p); upload_sampler_tables(p); }                                       // the reference rebuilds (and leaks) the tables per request
    const uint32_t seed = (uint32_t) time(nullptr);
    p.rng.seed(seed);                                                        // llama_set_rng_seed
    { std::lock_guard<std::mutex> lk(g_mu); retire_old_jobs(); Job & nj = g_jobs[job]; nj.seed = seed; nj.finished = false; nj.serial = ++g_job_serial; }
    if (p.vocab.type == BAMD_VOCAB_NONE) { fprintf(stderr, "doInference: model has no tokenizer (tokenizer.ggml.model = no_vocab)\n"); return 1; }
    const std::vector<int> embd_inp = p.vocab.tokenize(text, false, true);
    if (dbg("tokenizer")) { fprintf(stderr, "TOKENS: ["); for (int t : embd_inp) fprintf(stderr, " %d,", t); fprintf(stderr, " ]\n"); }
    const int n_ctx = p.n_ctx;
    { std::lock_guard<std::mutex> lk(g_mu); g_jobs[job].prompt_tokens = (int64_t) embd_inp.size(); }
    if ((int) embd_inp.size() > n_ctx - 4) { fprintf(stderr, "doInference: error: prompt is too long (%d tokens, max %d)\n", (int) embd_inp.size(), n_ctx - 4); return 0; }
    std::vector<int> last_tokens((size_t) n_ctx, 0);
    int n_past = 0, n_consumed = 0, n_remain = p.n_predict;
    std::vector<int> embd;
    for (auto & s : p.stages) bamd_kv_cache_clear(s.ctx);
    const int max_embd_size = n_ctx - 4;
    while (n_remain && n_past < max_embd_size && !p.stop.load()) {
        if (!embd.empty()) {
            if ((int) embd.size() > max_embd_size) embd.resize((size_t) max_embd_size);
            // context shift, bridge.cpp:482-507 (ga_n == 1, params.n_keep = 0).  Kept for fidelity: the loop guard stops at n_ctx - 4 positions
            // and an evaluation is at most one generated token or a prompt slice that fits, so the condition cannot hold — the level-1
            // calls behind it are pinned against the reference where it does (tests/test_gpu_fullsize_ref.py, fixture "shift")
            if (n_past + (int) embd.size() > n_ctx) {
                if (p.n_predict == -2) break;
                const int n_keep = 0, n_discard = (n_past - n_keep) / 2;
                for (auto & st : p.stages)
                    if (bamd_kv_seq_rm(st.ctx, n_keep, n_keep + n_discard) || bamd_kv_seq_add(st.ctx, n_keep + n_discard, n_past, -n_discard)) return 1;
                n_past -= n_discard;
            }
            for (int i = 0; i < (int) embd.size(); i += p.n_batch) {
                int n_eval = std::min((int) embd.size() - i, p.n_batch);
                for (int u = 0; u < n_eval; u += 512) {                      // llama_decode's n_ubatch = 512 micro-batches (llama.cpp:14615)
                    const int nu = std::min(512, n_eval - u);
                    // the logits are sampled from only once the whole prompt is in (below): earlier micro-batches are not waited for
                    const bool need = (int) embd_inp.size() <= n_consumed && i + u + nu >= (int) embd.size();
                    if (pod_decode(p, &embd[(size_t) (i + u)], nu, n_past + u, need)) return 1;
                }
                n_past += n_eval;
            }
        }
        embd.clear();
        if ((int) embd_inp.size() <= n_consumed) {
            const int id = p.gpu_sampler ? sample_janus_device(p, last_tokens, embd_inp.size(), (size_t) n_past, (size_t) p.n_predict)
                                         : sample_janus(p, p.logits.data(), last_tokens, embd_inp.size(), (size_t) n_past, (size_t) p.n_predict);
            if (id < 0) return 1;
            last_tokens.erase(last_tokens.begin()); last_tokens.push_back(id);
            embd.push_back(id);
            --n_remain;
        } else {
            while ((int) embd_inp.size() > n_consumed) {
                embd.push_back(embd_inp[(size_t) n_consumed]); ++n_consumed;
                if ((int) embd.size() >= p.n_batch) break;
            }
        }
        {
            std::string add; for (int id : embd) add += p.vocab.token_to_piece(id);
            std::lock_guard<std::mutex> lk(g_mu);
            g_jobs[job].append(add);
        }
        if (p.vocab.is_eog(embd.back())) break;
    }
    open_eval.close();
    if (open_eval.failed) return 1;
    std::lock_guard<std::mutex> lk(g_mu);
    Job & j = g_jobs[job];
    j.prompt_eval = p.n_p_eval ? (int64_t) (p.t_p_eval_ms / (double) p.n_p_eval) : 0;
    j.timing = p.n_eval ? (int64_t) (p.t_eval_ms / (double) p.n_eval) : 0;
    j.finished = true;
    return p.n_p_eval + p.n_eval;
}

BAMD_API int64_t doInference(int idx, void * ctx, char * jobID, char * sessionID, char * prompt) {
    (void) sessionID;
    int64_t rc;
    try { rc = do_inference_impl(idx, ctx, jobID, prompt); }
    catch (const std::exception & e) { fprintf(stderr, "doInference: %s\n", e.what()); rc = 1; }
    if (jobID) { std::lock_guard<std::mutex> lk(g_mu); auto it = g_jobs.find(jobID); if (it != g_jobs.end()) it->second.finished = true; }   // (also on the early-return paths)
    return rc;
}

BAMD_API void stopInference(int idx) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx >= 0 && idx < 8 && g_pods[idx]) g_pods[idx]->stop.store(true);
}
// Read-only accessors: an unknown (never started, or retired) job id answers "" / 0 as the reference's map default does (cpp/bridge.cpp:662-695), but
// WITHOUT inserting an entry: an inserted entry would never be "finished", so it could never be retired and a client polling stale ids would
// grow the table past BAMD_MAX_JOBS.  Only doInference inserts.
static Job * find_job(const char * jobID) { auto it = g_jobs.find(jobID ? jobID : ""); return it == g_jobs.end() ? nullptr : &it->second; }
BAMD_API const char * status(char * jobID) { std::lock_guard<std::mutex> lk(g_mu); Job * j = find_job(jobID); return j ? j->c_str() : ""; }
BAMD_API int64_t promptEval(char * jobID) { std::lock_guard<std::mutex> lk(g_mu); Job * j = find_job(jobID); return j ? j->prompt_eval : 0; }
BAMD_API int64_t getPromptTokenCount(char * jobID) { std::lock_guard<std::mutex> lk(g_mu); Job * j = find_job(jobID); return j ? j->prompt_tokens : 0; }
BAMD_API int64_t timing(char * jobID) { std::lock_guard<std::mutex> lk(g_mu); Job * j = find_job(jobID); return j ? j->timing : 0; }
BAMD_API uint32_t getSeed(char * jobID) { std::lock_guard<std::mutex> lk(g_mu); Job * j = find_job(jobID); return j ? j->seed : 0; }

// ---- test hooks (not part of the cgo surface): tokenizer and Janus as pure functions -------------------------------------------
// test hook: the Janus shortlist of `logits` with a fixed cut-off, through the fast (1) or the full-sort (0) path -> ids, count
BAMD_API int bamd_janus_shortlist_test(const float * logits, int V, float cutoff, int fast, int32_t * ids, int cap) {
    std::vector<Cand> cand;
    janus_shortlist(logits, (size_t) V, [&](int) { return cutoff; }, fast != 0, cand);
    for (size_t i = 0; i < cand.size() && (int) i < cap; ++i) ids[i] = cand[i].id;
    return (int) cand.size();
}

// test hook: one draw of the pod's sampler on scripted logits (n_vocab floats, uploaded to the device of the output layer);
// device = 1: penalties + shortlist on the device (sample_janus_device), 0: the host sampler.  logits_after (may be null) receives
// the logits after the penalties.  counts[2] (may be null): draws served by the device shortlist / by the host path so far.
BAMD_API int bamd_bridge_sample_test(void * ctx, const float * logits, const int32_t * last, int n_last, int prompt_len, int pos, int max, uint32_t seed,
                                     int device, float * logits_after, int64_t * counts) {
    Pod & p = *(Pod *) ctx;
    if (!p.janus_ready) { init_janus(p); upload_sampler_tables(p); }
    p.rng.seed(seed);
    const std::vector<int> last_tokens(last, last + n_last);
    int id;
    if (device) {
        if (!p.gpu_sampler || bamd_set_logits_test(p.stages.back().ctx, logits)) return -1;
        id = sample_janus_device(p, last_tokens, (size_t) prompt_len, (size_t) pos, (size_t) max);
        if (logits_after) {
            const float * lg = p.stages.size() == 1 ? bamd_get_logits(p.stages.back().ctx) : bamd_stage_get_logits(p.stages.back().ctx, p.stages.back().stream());
            if (!lg) return -1;
            memcpy(logits_after, lg, (size_t) p.n_vocab * 4);
        }
    } else {
        memcpy(p.logits.data(), logits, (size_t) p.n_vocab * 4);
        id = sample_janus(p, p.logits.data(), last_tokens, (size_t) prompt_len, (size_t) pos, (size_t) max);
        if (logits_after) memcpy(logits_after, p.logits.data(), (size_t) p.n_vocab * 4);
    }
    if (counts) { counts[0] = p.n_sample_dev; counts[1] = p.n_sample_host; }
    return id;
}

BAMD_API int bamd_bridge_tokenize(void * ctx, const char * text, int add_special, int parse_special, int32_t * out, int cap) {
    Pod & p = *(Pod *) ctx;
    const std::vector<int> t = p.vocab.tokenize(text, add_special != 0, parse_special != 0);
    for (int i = 0; i < (int) t.size() && i < cap; ++i) out[i] = t[(size_t) i];
    return (int) t.size();
}
// test hook: the stages a pod was split into — {device of the plan (virtual or real), layer_first, layer_last} per stage (what plan_stages + BOOSTER_GPUS + BAMD_VIRTUAL_DEVICES produced)
BAMD_API int bamd_bridge_stage_layout(void * ctx, int32_t * out, int cap_stages) {
    Pod & p = *(Pod *) ctx;
    for (size_t s = 0; s < p.stages.size() && (int) s < cap_stages; ++s) { out[3 * s] = p.stages[s].vdevice; out[3 * s + 1] = p.stages[s].layer_first; out[3 * s + 2] = p.stages[s].layer_last; }
    return (int) p.stages.size();
}
BAMD_API const char * bamd_bridge_token_to_piece(void * ctx, int id, int * len) {
    Pod & p = *(Pod *) ctx; const std::string & s = p.vocab.token_to_piece(id); if (len) *len = (int) s.size(); return s.c_str();
}

// ---- vocabulary-only entry points (no GPU needed): llama_tokenize / llama_token_to_piece / llama_token_is_eog of a GGUF ----------
struct bamd_vocab { GgufFile file; BamdVocab v; };
BAMD_API bamd_vocab * bamd_vocab_load(const char * gguf_path) {
    std::unique_ptr<bamd_vocab> h(new bamd_vocab());
    std::string err;
    if (!h->file.open(gguf_path, err) || !h->v.load(h->file, err)) { fprintf(stderr, "bamd_vocab_load: %s\n", err.c_str()); return nullptr; }
    return h.release();
}
BAMD_API void bamd_vocab_free(bamd_vocab * h) { delete h; }
// test hook, CPU only: the device of every layer and of the output layer (index n_layer) for a gpu1..gpu4 setting on a box with
// `device_count` GPUs; returns 0, or 1 when the setting is refused (no CPU path)
BAMD_API int bamd_plan_stages_test(int n_layer, int g1, int g2, int g3, int g4, int device_count, int32_t * device_of) {
    int gpu[BAMD_MAX_GPUS] = { 0 };
    const int n_gpu = gpu_weights(g1, g2, g3, g4, gpu);       // honours BOOSTER_GPUS like initContext
    std::vector<std::pair<int, std::pair<int, int>>> plan; std::string err;
    if (!plan_stages(n_layer, gpu, n_gpu, device_count, plan, err)) return 1;
    for (const auto & st : plan) for (int il = st.second.first; il < st.second.second; ++il) device_of[il] = st.first;
    device_of[n_layer] = plan.back().first;
    return 0;
}
// CPU-only probe of the GGUF reader (single file or the first shard of a split model): tensor count, total tensor bytes and an
// FNV-1a digest over the tensors in name order (name, type, shape, data) — equal for a model and its gguf-split shards
BAMD_API int bamd_gguf_probe(const char * gguf_path, int64_t * n_tensors, int64_t * n_bytes, uint64_t * digest) {
    GgufFile f; std::string err;
    if (!f.open(gguf_path, err)) { fprintf(stderr, "bamd_gguf_probe: %s\n", err.c_str()); return 1; }
    std::vector<const GgufTensor *> order;
    for (const GgufTensor & t : f.tensors) order.push_back(&t);
    std::sort(order.begin(), order.end(), [](const GgufTensor * a, const GgufTensor * b) { return a->name < b->name; });
    uint64_t h = 1469598103934665603ull; int64_t bytes = 0;
    auto mix = [&](const void * p, size_t n) { const uint8_t * b = (const uint8_t *) p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    for (const GgufTensor * t : order) {
        mix(t->name.data(), t->name.size()); mix(&t->type, sizeof t->type);
        for (int64_t d : t->ne) mix(&d, sizeof d);
        mix(t->data, t->nbytes); bytes += (int64_t) t->nbytes;
    }
    if (n_tensors) *n_tensors = (int64_t) f.tensors.size();
    if (n_bytes) *n_bytes = bytes;
    if (digest) *digest = h;
    return 0;
}
// test hooks: the host Janus sampler (init_janus + sample_janus) on a vocabulary-only GGUF and scripted logits — no GPU involved
BAMD_API void * bamd_janus_test_new(const bamd_vocab * h, float scale, float hi, float lo, int depth) {
    Pod * p = new Pod();
    p->vocab = h->v; p->n_vocab = h->v.n_vocab();
    p->jp.scale = scale; p->jp.hi = hi; p->jp.lo = lo; p->jp.depth = depth;
    init_janus(*p);
    return p;
}
BAMD_API void bamd_janus_test_tables(void * pod, float * types, float * scales) {
    Pod & p = *(Pod *) pod;
    memcpy(types, p.types.data(), p.types.size() * 4); memcpy(scales, p.scales.data(), p.scales.size() * 4);
}
// logits: n_vocab floats, modified in place as the sampler does; last: the n_last most recent tokens (newest last)
BAMD_API int bamd_janus_test_sample(void * pod, float * logits, const int32_t * last, int n_last, int prompt_len, int pos, int max, uint32_t seed) {
    Pod & p = *(Pod *) pod;
    p.rng.seed(seed);
    const std::vector<int> last_tokens(last, last + n_last);
    return sample_janus(p, logits, last_tokens, (size_t) prompt_len, (size_t) pos, (size_t) max);
}
BAMD_API void bamd_janus_test_free(void * pod) { delete (Pod *) pod; }
BAMD_API int bamd_vocab_tokenize(const bamd_vocab * h, const char * text, int text_len, int add_special, int parse_special, int32_t * out, int cap) {
    const std::vector<int> t = h->v.tokenize(std::string(text, (size_t) text_len), add_special != 0, parse_special != 0);
    for (int i = 0; i < (int) t.size() && i < cap; ++i) out[i] = t[(size_t) i];
    return (int) t.size();
}
BAMD_API int bamd_vocab_piece(const bamd_vocab * h, int id, char * buf, int cap) {
    const std::string & s = h->v.token_to_piece(id);
    if ((int) s.size() > cap) return -(int) s.size();
    memcpy(buf, s.data(), s.size());
    return (int) s.size();
}
BAMD_API int bamd_vocab_n(const bamd_vocab * h) { return h->v.n_vocab(); }
BAMD_API int bamd_vocab_is_eog(const bamd_vocab * h, int id) { return h->v.is_eog(id) ? 1 : 0; }
BAMD_API int bamd_vocab_eos(const bamd_vocab * h) { return h->v.eos; }
BAMD_API int bamd_vocab_eot(const bamd_vocab * h) { return h->v.eot; }
