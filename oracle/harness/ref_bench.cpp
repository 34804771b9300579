// ref_bench.cpp — times the GENUINE reference CPU path (oracle/_ref/libggml_ref.so = ggml + llama.cpp as vendored by
// gotzmann/booster, compiled in place by oracle/Makefile) on a GGUF: one prefill micro-batch, then n single-token llama_decode
// steps, greedy.  OUR code; test/measurement infrastructure only (bench.py's cpu_baseline leg, kind "reference").
// The metric is the reference's own: generated tokens / wall time of the single-token decode steps (llama.cpp:18533-18536).
//
// usage: ref_bench <model.gguf> <n_threads> <n_prompt> <n_decode> [n_ctx]
// prints one line:  ref_bench tokens_per_s=<f> ms_per_token=<f> prompt_tokens_per_s=<f> threads=<n> n_decode=<n> last_token=<id>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "llama.h"

int main(int argc, char ** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s model.gguf n_threads n_prompt n_decode [n_ctx]\n", argv[0]); return 2; }
    const int n_threads = atoi(argv[2]), n_prompt = atoi(argv[3]), n_decode = atoi(argv[4]);
    const int n_ctx = argc > 5 ? atoi(argv[5]) : 512;
    llama_backend_init();
    llama_model_params mp = llama_model_default_params();
    mp.n_gpu_layers = 0; mp.use_mmap = true;
    llama_model * model = llama_load_model_from_file(argv[1], mp);
    if (!model) { fprintf(stderr, "ref_bench: cannot load %s\n", argv[1]); return 1; }
    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = (uint32_t) n_ctx; cp.n_batch = 512; cp.n_ubatch = 512; cp.n_threads = n_threads; cp.n_threads_batch = n_threads;
    llama_context * ctx = llama_new_context_with_model(model, cp);
    if (!ctx) { fprintf(stderr, "ref_bench: cannot create context\n"); return 1; }
    const int V = llama_n_vocab(model);
    std::vector<llama_token> prompt((size_t) n_prompt);
    for (int i = 0; i < n_prompt; ++i) prompt[(size_t) i] = (llama_token) ((7919ll * i + 13) % V);   // SURVEY 8d synthetic prompt
    auto t0 = std::chrono::steady_clock::now();
    if (llama_decode(ctx, llama_batch_get_one(prompt.data(), n_prompt, 0, 0))) { fprintf(stderr, "ref_bench: prefill failed\n"); return 1; }
    const double tp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto argmax = [&]() { const float * lg = llama_get_logits(ctx); int b = 0; for (int i = 1; i < V; ++i) if (lg[i] > lg[b]) b = i; return (llama_token) b; };
    llama_token tok = argmax();
    double td = 0.0;
    for (int s = 0; s < n_decode; ++s) {
        auto t1 = std::chrono::steady_clock::now();
        if (llama_decode(ctx, llama_batch_get_one(&tok, 1, n_prompt + s, 0))) { fprintf(stderr, "ref_bench: decode failed\n"); return 1; }
        td += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        tok = argmax();
    }
    printf("ref_bench tokens_per_s=%.4f ms_per_token=%.3f prompt_tokens_per_s=%.3f threads=%d n_decode=%d last_token=%d\n",
           n_decode / td, td / n_decode * 1e3, n_prompt / tp, n_threads, n_decode, (int) tok);
    llama_free(ctx); llama_free_model(model); llama_backend_free();
    return 0;
}
