// janus_ref.cpp — OUR code linked against the genuine reference (oracle/_ref/libggml_ref.so + cpp/janus.cpp + cpp/common/common.cpp
// compiled in place by oracle/Makefile): runs the reference's Janus sampler (initJanus as cpp/bridge.cpp:196 calls it, then
// sample_janus_token as cpp/bridge.cpp:589 does) on scripted logits and writes what it did.  Build container only; test infrastructure.
//
// usage: janus_ref model.gguf cases.bin out.bin
//   cases.bin: i32 n_cases, f32 scale, hi, lo, i32 depth ; per case: i32 n_last, promptLen, pos, max, u32 seed, i32 last[n_last], f32 logits[V]
//   out.bin:   i32 V ; f32 types[V], f32 scales[V] (the per-token tables initJanus built) ; per case: i32 token, f32 logits_after[V]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "llama.h"
#include "common.h"
#include "sampling.h"
#include "janus.h"

extern float * scales;   // cpp/janus.cpp:36-37
extern float * types;

template <typename T> static T rd(FILE * f) { T v; if (fread(&v, sizeof v, 1, f) != 1) { fprintf(stderr, "janus_ref: short read\n"); exit(1); } return v; }

int main(int argc, char ** argv) {
    if (argc < 4) return 2;
    llama_backend_init();
    llama_log_set([](ggml_log_level, const char *, void *) {}, nullptr);
    llama_model * model = llama_load_model_from_file(argv[1], llama_model_default_params());
    if (!model) { fprintf(stderr, "janus_ref: load failed\n"); return 1; }
    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = 64; cp.n_batch = 32; cp.n_ubatch = 32; cp.n_threads = 1; cp.n_threads_batch = 1;
    llama_context * ctx = llama_new_context_with_model(model, cp);
    llama_token bos = llama_token_bos(model);
    if (llama_decode(ctx, llama_batch_get_one(&bos, 1, 0, 0))) { fprintf(stderr, "janus_ref: decode failed\n"); return 1; }   // allocates the logits buffer
    const int V = llama_n_vocab(model);
    FILE * in = fopen(argv[2], "rb"), * out = fopen(argv[3], "wb");
    if (!in || !out) return 1;
    const int n_cases = rd<int32_t>(in);
    janus_params jp;
    jp.scale = rd<float>(in); jp.hi = rd<float>(in); jp.lo = rd<float>(in); jp.depth = rd<int32_t>(in);
    static char debug[1] = { 0 };
    initJanus(ctx, jp, debug);
    int32_t v32 = V; fwrite(&v32, 4, 1, out);
    fwrite(::types, 4, (size_t) V, out); fwrite(::scales, 4, (size_t) V, out);
    llama_sampling_params sp;
    for (int c = 0; c < n_cases; ++c) {
        const int n_last = rd<int32_t>(in), promptLen = rd<int32_t>(in), pos = rd<int32_t>(in), max = rd<int32_t>(in);
        const uint32_t seed = rd<uint32_t>(in);
        std::vector<llama_token> last((size_t) n_last);
        if (fread(last.data(), 4, (size_t) n_last, in) != (size_t) n_last) return 1;
        float * logits = llama_get_logits(ctx);
        if (fread(logits, 4, (size_t) V, in) != (size_t) V) return 1;
        llama_set_rng_seed(ctx, seed);
        const int32_t tok = sample_janus_token(ctx, sp, jp, last, (size_t) promptLen, (size_t) pos, (size_t) max);
        fwrite(&tok, 4, 1, out);
        fwrite(logits, 4, (size_t) V, out);
    }
    fclose(in); fclose(out);
    llama_free(ctx); llama_free_model(model);
    return 0;
}
