// ref_run.cpp — full-size fixture generator and CPU-baseline timer.  OUR code; links against oracle/_ref/libggml_ref.so, the
// GENUINE reference CPU path (ggml + llama.cpp as vendored by gotzmann/booster, compiled in place by oracle/Makefile).  Runs ONLY
// in the build container; what it writes is committed as data under tests/golden/ (tokens, a few logits, digests — no weights).
//
// It evaluates the deterministic synthetic prompt tok[i] = (7919 i + 13) mod V in micro-batches of 512 exactly as
// cpp/bridge.cpp feeds a prompt (n_batch chunks through llama_decode, bridge.cpp:560-600), then decodes greedily, and records for
// the prompt's last token and every generated token:
//   tokens        i32 [n_decode + 1]     arg-max of each logits vector (token fed to the next step)
//   digest        u64 [n_decode + 1]     sum_i (bits(logit_i) + 0x9E3779B97F4A7C15) * (2 i + 1)  mod 2^64   (all V logits)
//   probe_idx     i32 [32]               fixed vocabulary ids
//   probe_logits  f32 [n_decode + 1][32] the logits at those ids
//   top_logit     f32 [n_decode + 1]
//   timing        f64 [4]                prompt seconds, decode seconds, n_threads, n_decode
// Container: the flat BGLD0001 format of gen_golden.cpp (reader: tests/goldenio.py).
//
// usage: ref_run <model.gguf> <n_threads> <n_prompt> <n_decode> <n_ctx> <out.bgld> [n_keep]
// n_keep (optional, >= 0): generate PAST n_ctx with Booster's context shift — the statements of cpp/bridge.cpp:487-503, run whenever
// n_past + 1 > n_ctx (in the bridge itself its generation loop stops at n_ctx - 4 first, so the shift never fires there; here it does):
//     n_discard = (n_past - n_keep) / 2;  llama_kv_cache_seq_rm(ctx, 0, n_keep, n_keep + n_discard);
//     llama_kv_cache_seq_add(ctx, 0, n_keep + n_discard, n_past, -n_discard);  n_past -= n_discard;
// n_keep < -1 encodes Self-Extend instead: -(100 * ga_n + ga_w), e.g. -216 = group factor 2, window 16 — the loop of cpp/bridge.cpp:507-523
// (llama_kv_cache_seq_add / _seq_div / _seq_add per window) run before every evaluation, as the bridge would with ga_n > 1.
// recorded in addition:  shift_steps i32 [n_shifts] (decode steps before which a shift ran), n_past_of_step i32 [n_decode].
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "llama.h"

static FILE * g_f = nullptr;
static void rec(const std::string & name, uint32_t dtype, const std::vector<int64_t> & dims, const void * data, size_t nbytes) {
    uint32_t nl = (uint32_t) name.size();
    fwrite(&nl, 4, 1, g_f); fwrite(name.data(), 1, nl, g_f);
    fwrite(&dtype, 4, 1, g_f);
    uint32_t nd = (uint32_t) dims.size(); fwrite(&nd, 4, 1, g_f);
    for (auto d : dims) { uint64_t u = (uint64_t) d; fwrite(&u, 8, 1, g_f); }
    uint64_t nb = nbytes; fwrite(&nb, 8, 1, g_f);
    fwrite(data, 1, nbytes, g_f);
}

int main(int argc, char ** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s model.gguf n_threads n_prompt n_decode n_ctx out.bgld\n", argv[0]); return 2; }
    const int n_threads = atoi(argv[2]), n_prompt = atoi(argv[3]), n_decode = atoi(argv[4]), n_ctx = atoi(argv[5]);
    const int n_keep = argc > 7 ? atoi(argv[7]) : -1;
    llama_backend_init();
    llama_model_params mp = llama_model_default_params();
    mp.n_gpu_layers = 0; mp.use_mmap = true;
    llama_model * model = llama_load_model_from_file(argv[1], mp);
    if (!model) { fprintf(stderr, "ref_run: cannot load %s\n", argv[1]); return 1; }
    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = (uint32_t) n_ctx; cp.n_batch = 512; cp.n_ubatch = 512; cp.n_threads = n_threads; cp.n_threads_batch = n_threads;
    llama_context * ctx = llama_new_context_with_model(model, cp);
    if (!ctx) { fprintf(stderr, "ref_run: cannot create context\n"); return 1; }
    const int V = llama_n_vocab(model);
    std::vector<llama_token> prompt((size_t) n_prompt);
    for (int i = 0; i < n_prompt; ++i) prompt[(size_t) i] = (llama_token) ((7919ll * i + 13) % V);
    const int NP = 32;
    std::vector<int32_t> pidx(NP);
    for (int j = 0; j < NP; ++j) pidx[(size_t) j] = (int32_t) ((104729ll * j + 7) % V);
    std::vector<int32_t> toks; std::vector<uint64_t> dig; std::vector<float> probes, tops;
    auto take = [&]() {
        const float * lg = llama_get_logits(ctx);
        int b = 0; uint64_t d = 0;
        for (int i = 0; i < V; ++i) {
            if (lg[i] > lg[b]) b = i;
            uint32_t u; memcpy(&u, lg + i, 4);
            d += ((uint64_t) u + 0x9E3779B97F4A7C15ull) * (uint64_t) (2 * (uint64_t) i + 1);
        }
        toks.push_back(b); dig.push_back(d); tops.push_back(lg[b]);
        for (int j = 0; j < NP; ++j) probes.push_back(lg[pidx[(size_t) j]]);
        return (llama_token) b;
    };
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n_prompt; i += 512) {
        const int n = n_prompt - i < 512 ? n_prompt - i : 512;
        if (llama_decode(ctx, llama_batch_get_one(prompt.data() + i, n, i, 0))) { fprintf(stderr, "ref_run: prefill failed\n"); return 1; }
    }
    const double tp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    llama_token tok = take();
    double td = 0.0;
    int n_past = n_prompt;
    std::vector<int32_t> shift_steps, past_of_step;
    const int ga_n = n_keep < -1 ? (-n_keep) / 100 : 1, ga_w = n_keep < -1 ? (-n_keep) % 100 : 0;
    int ga_i = 0;
    for (int s = 0; s < n_decode; ++s) {
        if (ga_n > 1) {
            while (n_past >= ga_i + ga_w) {
                const int ib = (ga_n * ga_i) / ga_w, bd = (ga_w / ga_n) * (ga_n - 1), dd = (ga_w / ga_n) - ib * bd - ga_w;
                llama_kv_cache_seq_add(ctx, 0, ga_i, n_past, ib * bd);
                llama_kv_cache_seq_div(ctx, 0, ga_i + ib * bd, ga_i + ib * bd + ga_w, ga_n);
                llama_kv_cache_seq_add(ctx, 0, ga_i + ib * bd + ga_w, n_past + ib * bd, dd);
                n_past -= bd;
                ga_i += ga_w / ga_n;
                shift_steps.push_back(s);
            }
        }
        if (n_keep >= 0 && n_past + 1 > n_ctx) {
            const int n_left = n_past - n_keep, n_discard = n_left / 2;
            llama_kv_cache_seq_rm (ctx, 0, n_keep, n_keep + n_discard);
            llama_kv_cache_seq_add(ctx, 0, n_keep + n_discard, n_past, -n_discard);
            n_past -= n_discard;
            shift_steps.push_back(s);
        }
        past_of_step.push_back(n_past);
        auto t1 = std::chrono::steady_clock::now();
        if (llama_decode(ctx, llama_batch_get_one(&tok, 1, n_past, 0))) { fprintf(stderr, "ref_run: decode failed\n"); return 1; }
        n_past += 1;
        td += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        tok = take();
    }
    g_f = fopen(argv[6], "wb");
    if (!g_f) { perror(argv[6]); return 1; }
    fwrite("BGLD0001", 1, 8, g_f);
    const int32_t meta[4] = { n_prompt, n_decode, n_ctx, V };
    rec("meta", 2, { 4 }, meta, sizeof meta);
    rec("tokens", 2, { (int64_t) toks.size() }, toks.data(), toks.size() * 4);
    rec("digest", 3, { 8, (int64_t) dig.size() }, dig.data(), dig.size() * 8);
    rec("probe_idx", 2, { NP }, pidx.data(), pidx.size() * 4);
    rec("probe_logits", 0, { NP, (int64_t) toks.size() }, probes.data(), probes.size() * 4);
    rec("top_logit", 0, { (int64_t) tops.size() }, tops.data(), tops.size() * 4);
    if (n_keep >= 0 || n_keep < -1) {
        rec("shift_steps", 2, { (int64_t) shift_steps.size() }, shift_steps.data(), shift_steps.size() * 4);
        rec("n_past_of_step", 2, { (int64_t) past_of_step.size() }, past_of_step.data(), past_of_step.size() * 4);
    }
    const double timing[4] = { tp, td, (double) n_threads, (double) n_decode };
    rec("timing", 3, { 8, 4 }, timing, sizeof timing);
    fclose(g_f);
    printf("ref_run prompt_tokens_per_s=%.3f tokens_per_s=%.4f ms_per_token=%.3f threads=%d n_prompt=%d n_decode=%d last_token=%d\n",
           n_prompt / tp, n_decode > 0 ? n_decode / td : 0.0, n_decode > 0 ? td / n_decode * 1e3 : 0.0, n_threads, n_prompt, n_decode, (int) tok);
    llama_free(ctx); llama_free_model(model); llama_backend_free();
    return 0;
}
