// gen_golden.cpp — fixture generator.  OUR code; links against oracle/_ref/libggml_ref.so, which is the
// genuine reference CPU path (ggml + llama.cpp as vendored by gotzmann/booster) compiled in place by
// oracle/Makefile.  It is run ONLY in the build container (the reference cannot travel to the GPU box);
// its outputs are committed as data under tests/golden/.
//
// What it does:
//   1. writes a tiny synthetic Llama GGUF with the reference's own gguf_* writer and ggml_quantize_chunk
//      (reference: ggml.c gguf_init_empty/gguf_add_tensor/gguf_write_to_file, ggml_quantize_chunk);
//   2. loads it with llama_load_model_from_file, runs one prefill micro-batch and N greedy decode steps
//      through llama_decode (llama.cpp:18517), capturing every named graph node with cparams.cb_eval
//      (llama.cpp:14707) for selected steps, and the logits + arg-max token of every step;
//   3. writes everything into one flat binary container (see tests/goldenio.py for the reader).
//
// usage: gen_golden <out.gguf> <out.bgld> <variant: a|b> [n_decode]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <random>
#include <algorithm>

#include "ggml.h"
#include "ggml-backend.h"
#include "llama.h"

struct Dump {
    FILE * f = nullptr;
    void open(const char * path) { f = fopen(path, "wb"); if (!f) { perror(path); exit(1); } fwrite("BGLD0001", 1, 8, f); }
    void rec(const std::string & name, uint32_t dtype, const std::vector<int64_t> & dims, const void * data, size_t nbytes) {
        uint32_t nl = (uint32_t) name.size();
        fwrite(&nl, 4, 1, f); fwrite(name.data(), 1, nl, f);
        fwrite(&dtype, 4, 1, f);
        uint32_t nd = (uint32_t) dims.size(); fwrite(&nd, 4, 1, f);
        for (auto d : dims) { uint64_t u = (uint64_t) d; fwrite(&u, 8, 1, f); }
        uint64_t nb = nbytes; fwrite(&nb, 8, 1, f);
        fwrite(data, 1, nbytes, f);
    }
    void close() { fclose(f); }
};

static Dump g_dump;
static std::string g_prefix;
static bool g_capture = false;

// dtype codes: 0 f32, 1 f16(raw u16), 2 i32, 3 u8
static bool cb_eval(struct ggml_tensor * t, bool ask, void * /*ud*/) {
    if (ask) return g_capture;
    if (!g_capture) return true;
    if (!ggml_is_contiguous(t)) return true;
    uint32_t dt;
    if      (t->type == GGML_TYPE_F32) dt = 0;
    else if (t->type == GGML_TYPE_F16) dt = 1;
    else if (t->type == GGML_TYPE_I32) dt = 2;
    else return true;
    std::vector<int64_t> dims;
    int nd = ggml_n_dims(t);
    for (int i = 0; i < nd; ++i) dims.push_back(t->ne[i]);   // ggml order: ne[0] fastest
    g_dump.rec(g_prefix + t->name, dt, dims, t->data, ggml_nbytes(t));
    return true;
}

struct Cfg {
    int E, H, Hkv, L, F, V, n_ctx_train;
    float theta, eps;
    bool rope_freqs;
    // per-layer types
    std::vector<ggml_type> t_q, t_k, t_v, t_o, t_gate, t_up, t_down;
    ggml_type t_embd, t_out;
};

static void fill_normal(std::vector<float> & v, std::mt19937 & rng, float sigma, float mean = 0.f) {
    std::normal_distribution<float> nd(mean, sigma);
    for (auto & x : v) x = nd(rng);
}

static void add_tensor(gguf_context * g, ggml_context * ctx, const char * name, ggml_type type, int64_t ne0, int64_t ne1,
                       std::mt19937 & rng, float sigma, float mean = 0.f) {
    ggml_tensor * t = ne1 > 0 ? ggml_new_tensor_2d(ctx, type, ne0, ne1) : ggml_new_tensor_1d(ctx, type, ne0);
    ggml_set_name(t, name);
    int64_t rows = ne1 > 0 ? ne1 : 1;
    std::vector<float> src((size_t) ne0 * rows);
    fill_normal(src, rng, sigma, mean);
    if (type == GGML_TYPE_F32) memcpy(t->data, src.data(), src.size() * 4);
    else ggml_quantize_chunk(type, src.data(), t->data, 0, rows, ne0, nullptr);
    gguf_add_tensor(g, t);
}

static void write_gguf(const char * path, const Cfg & c, uint32_t seed) {
    ggml_init_params ip = { (size_t) 256 * 1024 * 1024, nullptr, false };
    ggml_context * ctx = ggml_init(ip);
    gguf_context * g = gguf_init_empty();
    gguf_set_val_str(g, "general.architecture", "llama");
    gguf_set_val_str(g, "general.name", "booster-amd-golden-tiny");
    gguf_set_val_u32(g, "llama.context_length", c.n_ctx_train);
    gguf_set_val_u32(g, "llama.embedding_length", c.E);
    gguf_set_val_u32(g, "llama.block_count", c.L);
    gguf_set_val_u32(g, "llama.feed_forward_length", c.F);
    gguf_set_val_u32(g, "llama.attention.head_count", c.H);
    gguf_set_val_u32(g, "llama.attention.head_count_kv", c.Hkv);
    gguf_set_val_f32(g, "llama.attention.layer_norm_rms_epsilon", c.eps);
    gguf_set_val_u32(g, "llama.rope.dimension_count", c.E / c.H);
    gguf_set_val_f32(g, "llama.rope.freq_base", c.theta);
    gguf_set_val_u32(g, "llama.vocab_size", c.V);
    gguf_set_val_str(g, "tokenizer.ggml.model", "no_vocab");

    std::mt19937 rng(seed);
    const int hd = c.E / c.H;
    const float sE = 1.0f / sqrtf((float) c.E), sF = 1.0f / sqrtf((float) c.F);
    add_tensor(g, ctx, "token_embd.weight", c.t_embd, c.E, c.V, rng, 1.0f);
    add_tensor(g, ctx, "output_norm.weight", GGML_TYPE_F32, c.E, 0, rng, 0.1f, 1.0f);
    add_tensor(g, ctx, "output.weight", c.t_out, c.E, c.V, rng, sE * 3.0f);
    if (c.rope_freqs) {
        // llama-3.1 style frequency factors (llama.cpp:8611-8624): mild, deterministic
        ggml_tensor * t = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, hd / 2);
        ggml_set_name(t, "rope_freqs.weight");
        for (int i = 0; i < hd / 2; ++i) ((float *) t->data)[i] = i < hd / 4 ? 1.0f : 1.0f + 7.0f * (float) (i - hd / 4) / (float) (hd / 4);
        gguf_add_tensor(g, t);
    }
    char nm[128];
    for (int il = 0; il < c.L; ++il) {
        auto N = [&](const char * s) { snprintf(nm, sizeof nm, "blk.%d.%s.weight", il, s); return nm; };
        add_tensor(g, ctx, N("attn_norm"), GGML_TYPE_F32, c.E, 0, rng, 0.1f, 1.0f);
        add_tensor(g, ctx, N("attn_q"), c.t_q[il], c.E, c.E, rng, sE * 2.0f);
        add_tensor(g, ctx, N("attn_k"), c.t_k[il], c.E, c.Hkv * hd, rng, sE * 2.0f);
        add_tensor(g, ctx, N("attn_v"), c.t_v[il], c.E, c.Hkv * hd, rng, sE);
        add_tensor(g, ctx, N("attn_output"), c.t_o[il], c.E, c.E, rng, sE);
        add_tensor(g, ctx, N("ffn_norm"), GGML_TYPE_F32, c.E, 0, rng, 0.1f, 1.0f);
        add_tensor(g, ctx, N("ffn_gate"), c.t_gate[il], c.E, c.F, rng, sE * 1.5f);
        add_tensor(g, ctx, N("ffn_up"), c.t_up[il], c.E, c.F, rng, sE * 1.5f);
        add_tensor(g, ctx, N("ffn_down"), c.t_down[il], c.F, c.E, rng, sF);
    }
    gguf_write_to_file(g, path, false);
    gguf_free(g);
    ggml_free(ctx);
}

int main(int argc, char ** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s out.gguf out.bgld a|b [n_decode]\n", argv[0]); return 2; }
    const char * gguf_path = argv[1];
    const char * dump_path = argv[2];
    const char variant = argv[3][0];
    const int n_decode = argc > 4 ? atoi(argv[4]) : 40;

    Cfg c;
    if (variant == 'a') {
        // Llama-3 shaped: hd 128, GQA 4:1, Q4_K_M-like mixture incl. Q5_K/Q6_K, F = 3 super-blocks
        c = { 512, 4, 1, 2, 768, 512, 512, 500000.0f, 1e-5f, false, {}, {}, {}, {}, {}, {}, {}, GGML_TYPE_Q4_K, GGML_TYPE_Q6_K };
        c.t_q = { GGML_TYPE_Q4_K, GGML_TYPE_Q4_K }; c.t_k = c.t_q; c.t_o = c.t_q; c.t_gate = c.t_q; c.t_up = c.t_q;
        c.t_v = { GGML_TYPE_Q6_K, GGML_TYPE_Q5_K };
        c.t_down = { GGML_TYPE_Q6_K, GGML_TYPE_Q4_K };
    } else {
        // Mistral-Q6_K shaped: hd 64, GQA 2:1, every matrix Q6_K, rope_freqs present, theta 10000
        c = { 256, 4, 2, 2, 512, 320, 512, 10000.0f, 1e-5f, true, {}, {}, {}, {}, {}, {}, {}, GGML_TYPE_Q6_K, GGML_TYPE_Q6_K };
        c.t_q = { GGML_TYPE_Q6_K, GGML_TYPE_Q6_K }; c.t_k = c.t_q; c.t_v = c.t_q; c.t_o = c.t_q; c.t_gate = c.t_q; c.t_up = c.t_q; c.t_down = c.t_q;
    }
    write_gguf(gguf_path, c, variant == 'a' ? 1234u : 4321u);

    llama_backend_init();
    llama_log_set([](ggml_log_level, const char *, void *) {}, nullptr);
    llama_model_params mp = llama_model_default_params();
    mp.n_gpu_layers = 0;
    mp.use_mmap = false;
    llama_model * model = llama_load_model_from_file(gguf_path, mp);
    if (!model) { fprintf(stderr, "load failed\n"); return 1; }

    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = 128; cp.n_batch = 512; cp.n_ubatch = 512;
    cp.n_threads = 2; cp.n_threads_batch = 2;     // outputs are thread-count invariant
    cp.cb_eval = cb_eval; cp.cb_eval_user_data = nullptr;
    cp.flash_attn = false;
    llama_context * ctx = llama_new_context_with_model(model, cp);
    if (!ctx) { fprintf(stderr, "ctx failed\n"); return 1; }

    g_dump.open(dump_path);
    {   // header: config + build flavour of the oracle
        int32_t cfg[10] = { c.E, c.H, c.Hkv, c.L, c.F, c.V, (int32_t) cp.n_ctx, c.rope_freqs, 0, 0 };
        g_dump.rec("meta/config", 2, { 10 }, cfg, sizeof cfg);
        float fcfg[2] = { c.theta, c.eps };
        g_dump.rec("meta/fconfig", 0, { 2 }, fcfg, sizeof fcfg);
        std::string sys = llama_print_system_info();
        g_dump.rec("meta/system_info", 3, { (int64_t) sys.size() }, sys.data(), sys.size());
    }

    const int n_prompt = 8;
    std::vector<llama_token> prompt(n_prompt);
    for (int i = 0; i < n_prompt; ++i) prompt[i] = (7919 * i + 13) % c.V;
    g_dump.rec("meta/prompt", 2, { n_prompt }, prompt.data(), n_prompt * 4);

    const int V = llama_n_vocab(model);
    std::vector<int32_t> toks;
    std::vector<float> all_logits;

    // prefill (one micro-batch of 8 tokens), all nodes captured
    g_prefix = "prefill/"; g_capture = true;
    if (llama_decode(ctx, llama_batch_get_one(prompt.data(), n_prompt, 0, 0))) { fprintf(stderr, "decode failed\n"); return 1; }
    g_capture = false;
    int n_past = n_prompt;
    for (int s = 0; s < n_decode; ++s) {
        const float * lg = llama_get_logits(ctx);
        all_logits.insert(all_logits.end(), lg, lg + V);
        llama_token id = (llama_token) (std::max_element(lg, lg + V) - lg);
        toks.push_back(id);
        // capture nodes for decode step 0 (n_kv 32), and the step where n_kv becomes 64 (n_past == 32)
        g_capture = (s == 0) || (n_past == 32);
        char pf[32]; snprintf(pf, sizeof pf, "decode%d/", n_past); g_prefix = pf;
        if (llama_decode(ctx, llama_batch_get_one(&id, 1, n_past, 0))) { fprintf(stderr, "decode failed\n"); return 1; }
        g_capture = false;
        n_past += 1;
    }
    {
        const float * lg = llama_get_logits(ctx);
        all_logits.insert(all_logits.end(), lg, lg + V);
    }
    g_dump.rec("greedy/tokens", 2, { (int64_t) toks.size() }, toks.data(), toks.size() * 4);
    g_dump.rec("greedy/logits", 0, { V, (int64_t) (n_decode + 1) }, all_logits.data(), all_logits.size() * 4);
    g_dump.close();

    llama_free(ctx);
    llama_free_model(model);
    llama_backend_free();
    fprintf(stderr, "wrote %s and %s (%d decode steps)\n", gguf_path, dump_path, n_decode);
    return 0;
}
