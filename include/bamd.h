/* bamd.h — C-ABI of libbooster_amd.so, level 1: the model-runtime calls that Booster's bridge makes.
 *
 * This is the layer the reference's cpp/bridge.cpp and cpp/janus.cpp call through cpp/include/llama.h
 * (SURVEY.md §8b "below the boundary"); every entry point names the llama.h function it replaces.  Plain C
 * types only: pointers are HOST pointers unless the name says `dev`.  All functions return 0 on success or a
 * non-zero error code unless documented otherwise; bamd_last_error() gives the message.  There is NO CPU
 * fallback: without a usable MI355X-class HIP device every compute entry point fails loudly.
 *
 * The nine cgo symbols of cpp/bridge.h are declared in booster_bridge.h and implemented on top of this.
 */
#ifndef BAMD_H
#define BAMD_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct bamd_model   bamd_model;
typedef struct bamd_context bamd_context;

/* ggml tensor type ids, as stored in GGUF (cpp/ggml/include/ggml.h:360-375) */
enum { BAMD_TYPE_F32 = 0, BAMD_TYPE_F16 = 1, BAMD_TYPE_Q4_K = 12, BAMD_TYPE_Q5_K = 13, BAMD_TYPE_Q6_K = 14 };

const char * bamd_last_error(void);

/* llama_backend_init (llama.h) / ggml_backend_cuda_get_device_count (ggml-cuda.h:30): number of HIP devices, <0 on error */
int bamd_backend_init(void);
int bamd_device_count(void);

/* llama_load_model_from_file (cpp/src/llama.cpp:16539).  Loads layers [layer_first, layer_last) of a GGUF Llama
 * model onto HIP device `device` (layer_last < 0: all).  with_embd / with_output say whether this stage owns the
 * token-embedding lookup and output_norm + lm_head (layer split: first / last stage — llama.cpp:5932-5969). */
bamd_model * bamd_model_load(const char * gguf_path, int device, int layer_first, int layer_last, int with_embd, int with_output);
void         bamd_model_free(bamd_model * m);                           /* llama_free_model */
int          bamd_model_n_vocab(const bamd_model * m);                  /* llama_n_vocab */
int          bamd_model_n_embd(const bamd_model * m);                   /* llama_n_embd */
int          bamd_model_n_layer(const bamd_model * m);                  /* llama_n_layer (whole model) */
int          bamd_model_n_ctx_train(const bamd_model * m);              /* llama_n_ctx_train */
int64_t      bamd_model_weight_bytes(const bamd_model * m);             /* bytes of matmul weights resident on this stage */
/* copy back the GGUF-layout bytes of one resident tensor (testing / oracle cross-checks); returns bytes or <0 */
int64_t      bamd_model_tensor_raw(const bamd_model * m, const char * name, void * dst, int64_t cap);

/* llama_new_context_with_model (llama.cpp:16592): f16 KV cache for n_ctx positions, scratch, RoPE table. */
bamd_context * bamd_context_new(bamd_model * m, int n_ctx);
void           bamd_context_free(bamd_context * c);                     /* llama_free */
int            bamd_n_ctx(const bamd_context * c);                      /* llama_n_ctx */
void           bamd_kv_cache_clear(bamd_context * c);                   /* llama_kv_cache_clear (llama.cpp:3230) */

/* llama_decode(ctx, llama_batch_get_one(tokens, n_tokens, n_past, 0)) (llama.cpp:18517, :14537).
 * Processes the tokens at positions n_past.. with the reference's semantics for a micro-batch of n_tokens
 * (n_tokens > 1: attention scores use the f16-rounded q of the reference's T>1 path) and leaves the logits of
 * the LAST token for bamd_get_logits.  More than 512 tokens are evaluated in micro-batches of 512 (n_ubatch, llama.cpp:14615).
 * Returns 0, or 1 on failure like llama_decode. */
int           bamd_decode(bamd_context * c, const int32_t * tokens, int n_tokens, int n_past);
const float * bamd_get_logits(bamd_context * c);                        /* llama_get_logits: host, n_vocab floats */

/* Position edits of the KV cache, sequence 0 — llama_kv_cache_seq_rm / llama_kv_cache_seq_add (cpp/src/llama.cpp:3150-3217, :3268-3313),
 * the two calls of Booster's context shift (cpp/bridge.cpp:487-503):
 *     bamd_kv_seq_rm (ctx, n_keep, n_keep + n_discard);  bamd_kv_seq_add(ctx, n_keep + n_discard, n_past, -n_discard);  n_past -= n_discard;
 * Cells whose position leaves the sequence are freed and refilled in cell order (llama_kv_cache_find_slot, :3028-3127); the K rows of
 * moved cells are re-rotated by their delta before the next evaluation (K-shift: build_k_shift :8482-8512, ggml.c:14169-14290); the
 * attention then runs over cells and masks by the position each holds — all as the reference does, bit for bit.  While cells and
 * positions differ, tokens are evaluated one per bamd_decode / bamd_stage_step call (multi-token calls return an error) or by the device
 * loop bamd_generate_greedy, which runs find_slot for all its steps ahead on the host (it does not depend on the tokens); an edit that leaves cell i holding position i again (a plain truncation, seq_rm(n, -1)) ends that mode at once.  The calls wait
 * for the device to go idle first (hipDeviceSynchronize): work of this context still in flight on any stream is finished before the cell
 * metadata changes.
 * Negative p0 / p1 mean 0 / infinity as in llama.h.  Layer-split: call on every stage's context.  Return 0 on success. */
int bamd_kv_seq_rm(bamd_context * c, int p0, int p1);
int bamd_kv_seq_add(bamd_context * c, int p0, int p1, int delta);
/* llama_kv_cache_seq_div(ctx, 0, p0, p1, d) (llama.cpp:3315-3350): positions in [p0, p1) divided by d, the difference added to the pending
 * rotation — with bamd_kv_seq_add the three calls of Self-Extend (cpp/bridge.cpp:507-523; ga_n is fixed to 1 there, so Booster never issues them). */
int bamd_kv_seq_div(bamd_context * c, int p0, int p1, int d);

/* Greedy decode entirely on the device: n_steps single-token steps starting at position n_past; step 0 consumes
 * the arg-max of the logits left by the previous bamd_decode/bamd_generate_greedy call.  out_tokens receives
 * n_steps+1 ids: the token fed to each step, then the arg-max after the last step.  No host round trip between
 * steps: the steps are replayed as AQL packets with fence scope NONE from the library's own HSA queue (round 6; csrc/bamd_aql.h) where
 * the step's kernels allow it, else as one hipGraph per step on the context's stream (bamd_set_aql(0) / BAMD_AQL=0 forces that).
 * *elapsed_ms (optional) = time of the n_steps steps (HIP events; host clock from submission to completion on the own queue). */
int bamd_generate_greedy(bamd_context * c, int n_past, int n_steps, int32_t * out_tokens, float * elapsed_ms);
void bamd_set_aql(int on);                 /* process-wide: 1 (default, also env BAMD_AQL) = own AQL queue where possible; 0 = hipGraph replays */
int bamd_aql_runs(const bamd_context * c); /* bamd_generate_greedy calls + single-token bamd_decode steps of this context that ran on the own queue so far */

/* ---- layer-split stage interface (one process per GPU; hidden state moves between stages) ---- */
/* Run this stage's layers on one token.  The token id comes from `token`, or — when token_dev is non-NULL — from that
 * device int32 (no host round trip; used on the first stage).  hidden_in_dev: f32 [n_embd] device pointer (ignored on
 * the first stage, which embeds the token); hidden_out_dev: f32 [n_embd] device pointer (ignored on the last stage,
 * which computes logits + arg-max instead).  Work is enqueued on `hip_stream` (a hipStream_t; NULL = the default
 * stream) and NOT synchronised.  prefill_mode as in bamd_decode (n_tokens > 1).  Returns 0 or an error code. */
int bamd_stage_step(bamd_context * c, int32_t token, const void * token_dev, int pos, const void * hidden_in_dev, void * hidden_out_dev,
                    int want_logits, int prefill_mode, void * hip_stream);
/* Batched counterpart of bamd_stage_step for prompt micro-batches of 2..512 tokens: `tokens` (host) on the stage that owns the
 * embedding, hidden_in_dev [n_tokens][n_embd] f32 on the others; writes hidden_out_dev [n_tokens][n_embd] on every stage but the
 * last, which computes the logits of the LAST token when want_logits (bamd_stage_get_logits).  Returns 2 when this model / context
 * shape has no batched kernels: the caller then falls back to bamd_stage_step per token.  Bit-identical to that path. */
int bamd_stage_prefill(bamd_context * c, const int32_t * tokens, int n_tokens, int n_past, const void * hidden_in_dev, void * hidden_out_dev,
                       int want_logits, void * hip_stream);
/* last stage: write the arg-max of the last bamd_stage_step(want_logits=1) into a device int32 (stream-ordered). */
int bamd_stage_token_to(bamd_context * c, void * token_dev, void * hip_stream);
/* host logits (n_vocab floats) of the last bamd_stage_step(want_logits=1) on the last stage; synchronises `hip_stream`. */
const float * bamd_stage_get_logits(bamd_context * c, void * hip_stream);
int bamd_model_device(const bamd_model * m);
/* arg-max token of the last bamd_stage_step(want_logits=1) on the last stage; synchronises `hip_stream`. */
int bamd_stage_argmax(bamd_context * c, void * hip_stream, int32_t * token);

/* ---- sampler prefilter on the device: the Janus penalties and shortlist where the lm_head left the logits ---- */
/* one distinct penalised token: logit = (float)(logit * pre) when pre != 0 (the EOS boost, janus.cpp:236), then `count` times
 * logit *= f (kind 0: float x float, janus.cpp:263) or logit = (float)(logit * d) (kind 1: float x double, janus.cpp:255) */
typedef struct { int32_t id, count, kind; float f; double d, pre; } bamd_logit_penalty;
typedef struct { unsigned long long top_key; int32_t count, ntop, nan, top_id; float top_logit, cutoff; } bamd_shortlist_head;
#define BAMD_SHORTLIST_CAP 1024
#define BAMD_PENALTY_CAP 512
/* per-vocabulary tables, uploaded once: halve_class[id] != 0 = token of a class the x0.5 pass hits (janus.cpp:269-283);
 * cutoff_of[id] = the shortlist cut-off when `id` is the top token (janus.cpp:303-306) */
int bamd_sampler_tables(bamd_context * c, const uint8_t * halve_class, const float * cutoff_of, int n_vocab);
/* Applies the penalties to the logits of the last decode / stage step IN PLACE on the device, then collects the candidates with
 * !(logit / top < cutoff_of[top]) (unordered) into ids / vals (up to BAMD_SHORTLIST_CAP) and fills *head: count (may exceed the cap),
 * ntop (logits equal to the top), nan, top_id, top_logit, cutoff.  Nothing is collected when top <= 0.  Runs on `hip_stream` —
 * pass bamd_context_stream(c) after bamd_decode, the stage's stream after bamd_stage_step — and synchronises it.  Returns 0 or 1. */
int bamd_logits_shortlist(bamd_context * c, const bamd_logit_penalty * pen, int n_pen, int halve, bamd_shortlist_head * head,
                          int32_t * ids, float * vals, void * hip_stream);
void * bamd_context_stream(bamd_context * c);
/* 0: bamd_decode leaves the logits on the device; bamd_get_logits then copies them on demand (default 1: copied by every decode) */
void bamd_set_logits_readback(bamd_context * c, int on);
/* waits for the context's stream and reports what a bamd_decode without read-back left unchecked (llama_synchronize, llama.cpp:18527-18551): for a caller
 * that evaluates and then does NOT sample (a request stopped during its prompt).  Returns 0 or 1. */
int bamd_synchronize(bamd_context * c);
/* test hook: overwrite the device logits (n_vocab floats) */
int bamd_set_logits_test(bamd_context * c, const float * logits);

/* ---- tokenizer of a GGUF, CPU only: llama_tokenize / llama_token_to_piece / llama_token_is_eog ---- */
typedef struct bamd_vocab bamd_vocab;
bamd_vocab * bamd_vocab_load(const char * gguf_path);                   /* llm_load_vocab, llama.cpp:5250 */
void bamd_vocab_free(bamd_vocab * v);
/* llama_tokenize (llama-vocab.cpp:1243); returns the token count (may exceed cap) */
int bamd_vocab_tokenize(const bamd_vocab * v, const char * text, int text_len, int add_special, int parse_special, int32_t * out, int cap);
/* llama_token_to_piece with special = true (llama-vocab.cpp:1539); returns bytes written, or -needed */
int bamd_vocab_piece(const bamd_vocab * v, int id, char * buf, int cap);
int bamd_vocab_n(const bamd_vocab * v);
int bamd_vocab_is_eog(const bamd_vocab * v, int id);                    /* llama_token_is_eog */
int bamd_vocab_eos(const bamd_vocab * v);
int bamd_vocab_eot(const bamd_vocab * v);

/* Test hook, CPU only: Booster's `gpus:` split (bridge.cpp:745-750 + llm_load_tensors, llama.cpp:5932-5969) — the device of every layer and,
 * at index n_layer, of the output layer, on a box with device_count GPUs.  Returns 1 when the setting is refused (it would need a CPU path). */
int bamd_plan_stages_test(int n_layer, int gpu1, int gpu2, int gpu3, int gpu4, int device_count, int32_t * device_of);

/* CPU only: open a GGUF (or the FIRST shard of a gguf-split model, llama.cpp:3659-3714) with the library's reader and report the tensor
 * count, the total tensor bytes and an FNV-1a digest over (name, type, shape, data) in name order.  Returns 0, or 1 with a message on stderr. */
int bamd_gguf_probe(const char * gguf_path, int64_t * n_tensors, int64_t * n_bytes, uint64_t * digest);

/* Test hook, CPU only: the candidate shortlist of the Janus sampler (janus.cpp:262-300 — full descending sort, cut at the first
 * candidate with logit / top < cutoff) through the linear-time path (fast = 1; falls back by itself when ties or a non-positive top
 * make the order depend on the full sort) or the full-sort path (fast = 0).  Writes up to `cap` ids in order, returns the count. */
int bamd_janus_shortlist_test(const float * logits, int n_vocab, float cutoff, int fast, int32_t * ids, int cap);
/* Test hooks, CPU only: the host Janus sampler on a vocabulary (initJanus, janus.cpp:410-700; sample_janus_token, janus.cpp:191-331;
 * llama_sample_token, llama-sampling.cpp:610-631).  _new builds the per-token type / scale tables; _sample applies the penalties to
 * `logits` (n_vocab floats) in place and draws with std::mt19937(seed); last = the most recent tokens, newest last. */
void * bamd_janus_test_new(const bamd_vocab * v, float scale, float hi, float lo, int depth);
void bamd_janus_test_tables(void * janus, float * types, float * scales);
int bamd_janus_test_sample(void * janus, float * logits, const int32_t * last, int n_last, int prompt_len, int pos, int max, uint32_t seed);
void bamd_janus_test_free(void * janus);
/* Test hook, GPU: one draw of a pod's sampler (ctx = what initContext returned) on scripted logits; device = 1 runs the penalties and
 * the shortlist on the device (bamd_logits_shortlist), 0 the host sampler.  logits_after / counts[2] (device draws, host-path draws)
 * may be NULL.  Returns the token, or -1. */
int bamd_bridge_sample_test(void * ctx, const float * logits, const int32_t * last, int n_last, int prompt_len, int pos, int max, uint32_t seed,
                            int device, float * logits_after, int64_t * counts);

/* Test hook: the stages initContext split a pod into (Booster's gpus: rule, BOOSTER_GPUS, BAMD_VIRTUAL_DEVICES): out[3 s .. 3 s + 2] = {device of the plan,
 * first layer, last layer + 1} for up to cap_stages stages; returns the number of stages. */
int bamd_bridge_stage_layout(void * ctx, int32_t * out, int cap_stages);

/* Prompt evaluation mode, process-wide: 1 (default, also env BAMD_PREFILL_BATCH) = bamd_decode with 2..512 tokens runs the batched
 * prefill kernels (every layer once per micro-batch, like llama_decode with n_tokens > 1); 0 = token by token through the decode
 * kernels.  Bit-identical results.  Contexts with n_ctx > 8192 use the token-by-token path regardless (round 1). */
void bamd_set_prefill_batch(int on);   /* 2 = batched, but the mat-muls on the integer-dot kernel instead of the matrix-core kernels (the path of a model whose
                                        * side tables did not fit: they are built at model load, all matrices or none, BAMD_PREFILL_AUX_RESERVE_GB of the device left free) */

/* ---- measurement -------------------------------------------------------------------------------------- */
/* One eager single-token step at position `pos` with a HIP-event pair around every kernel launch.
 * classes: 0 matvec (all weight streaming), 1 attention (qk+softmax+pv), 2 step-begin/other.
 * For each class: launches[], ms[] (sum of durations), bytes[] (algorithmic bytes: weight records streamed /
 * KV bytes read).  Entry 3: ms[3] = what an EMPTY event pair reads on that stream (to subtract per launch).
 * Arrays must hold 4 entries. */
int bamd_profile_step(bamd_context * c, int pos, int * launches, double * ms, double * bytes);
/* the same per launch kind — arrays of 8: [0] fused QKV, [1] attention, [2] other, [3] wo, [4] gate/up, [5] ffn_down, [6] lm_head, [7] an empty event pair */
int bamd_profile_step_kinds(bamd_context * c, int pos, int * launches, double * ms, double * bytes);

/* Micro-benchmark: `iters` back-to-back launches of one mat-vec shape on random resident weights (pro: 0 plain,
 * 1 RMSNorm prologue; epi: 0 store, 1 +residual, 2 silu(gate)*up pair, 3 store+arg-max; mode as below). */
int bamd_bench_matvec(int type, int nrows, int k, int pro, int epi, int mode, int iters, float * us_per_launch);

/* ---- op-level entry points (parity tests call the kernels through these; host pointers in, host out) ---- */
/* quantize_row_q8_K (ggml-quants.c:3593) of norm_w ? rms_norm(x)*norm_w : x ; out = k/256 block_q8_K (292 B each) */
int bamd_op_quantize_q8_K(const float * x, int64_t k, const float * norm_w, float eps, void * out_blocks);
/* y[nrows] = W . Q8_K(act) (+ residual), W = GGUF-layout blocks [nrows][k] of `type`  (ggml_compute_forward_mul_mat, ggml.c:12277) */
int bamd_op_mul_mat_vec(int type, const void * w_raw, int nrows, int k, const float * x, const float * norm_w, float eps,
                        const float * residual, float * y, int mode /* 0 auto, 1 wave-per-row-group, 2 split-K */);
/* Y[T][nrows] = rows of W . Q8_K(act_t) (+ residual[T][nrows]) for T activation rows x[T][k] through the batched prefill kernels
 * (ggml_compute_forward_mul_mat with ne11 = T): impl 0 = integer-dot kernel (any K-quant), 1 = MFMA kernel (Q4_K, k % 1024 == 0) */
int bamd_op_mul_mat_batch(int type, const void * w_raw, int nrows, int k, const float * x, int T, const float * norm_w, float eps,
                          const float * residual, float * y, int impl);
/* y[nrows] = silu(Wg . a) * (Wu . a)  (llm_build_ffn LLM_FFN_SILU/LLM_FFN_PAR, llama.cpp:7960-8085) */
int bamd_op_ffn_gate_up(int type, const void * wg_raw, const void * wu_raw, int nrows, int k, const float * x, const float * norm_w,
                        float eps, float * y);
/* one row of a GGUF matrix dequantised (ggml_compute_forward_get_rows_q, ggml.c:13186) */
int bamd_op_get_row(int type, const void * w_raw, int nrows, int k, int row, float * y);
/* single-token attention of one layer (llm_build_kv, llama.cpp:8318): ropes q/k with the table row `pos`, stores
 * k/v (f16) into the caches, returns out[H*hd]; caches are host arrays in the reference's layouts and are updated. */
int bamd_op_attention(const float * q, const float * k, const float * v, uint16_t * k_cache, uint16_t * v_cache_t,
                      const float * rope_row, int H, int Hkv, int hd, int n_ctx, int pos, int prefill_mode, float * out,
                      float * probs_h0);
/* RoPE (cos,sin) table row as built on the host for position pos (ggml_rope_cache_init, ggml.c:14017) */
int bamd_op_rope_row(int pos, int n_dims, float freq_base, float freq_scale, const float * freq_factors, float * row);

#ifdef __cplusplus
}
#endif
#endif
