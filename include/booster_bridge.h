/* booster_bridge.h — C-ABI of libbooster_amd.so, level 2: the NINE cgo symbols of gotzmann/booster.
 *
 * Byte-compatible with the `extern "C"` block of the reference's cpp/bridge.h:132-165; the Go side
 * (pkg/server/server.go:7-36, pkg/booster/booster.go:15-21) declares exactly these prototypes in its cgo preamble.
 * Semantics follow cpp/bridge.cpp (file:line cited per symbol) with the deviations listed in INTEGRATION.md:
 *   - the model runs on MI355X GPUs through include/bamd.h (no CPU path: gpu1..gpu4 all zero is an error);
 *   - status() returns a pointer that stays valid and NUL-terminated while a decode appends (the reference's is racy);
 *   - stop flags are atomics.
 */
#ifndef BOOSTER_BRIDGE_H
#define BOOSTER_BRIDGE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* cpp/bridge.cpp:711-719.  Idempotent; may be called before or after initContext (server.go:344-360 vs :532-553).
 * `swap` (session path) is accepted and unused, as in the reference (session code is commented out). */
void init(char * swap, char * debug);

/* cpp/bridge.cpp:723-786.  Loads the GGUF model `modelName` for pod `idx` (0..7) and creates its context.
 * gpu1..gpu4: relative layer-split weights over HIP devices 0..3 (reference: tensor_split + n_gpu_layers); their sum
 * must be > 0.  context = n_ctx, predict = n_predict.  janus/depth/scale/hi/lo: Janus sampler parameters
 * (cpp/janus.h).  The other sampling scalars are stored and, like in the reference, unused (the standard sampler
 * branch is commented out, bridge.cpp:586-596).  Returns an opaque context pointer, NULL on failure. */
void * initContext(int idx, char * modelName, int threads, int batch_size, int gpu1, int gpu2, int gpu3, int gpu4,
                   int context, int predict, int32_t mirostat, float mirostat_tau, float mirostat_eta, float temperature,
                   int top_k, float top_p, float typical_p, float repetition_penalty, int penalty_last_n, int32_t janus,
                   int32_t depth, float scale, float hi, float lo, uint32_t seed, char * debug);

/* cpp/bridge.cpp:175-658 (do_inference) via :788-798.  Tokenises `prompt` (add_special=false, parse_special=true),
 * clears the KV cache, evaluates the prompt in batches of n_batch, then samples with Janus until n_predict tokens,
 * n_ctx-4 positions, an end-of-generation token or stopInference(idx).  Returns n_p_eval + n_eval; 0 if the prompt is
 * longer than n_ctx-4; 1 if a decode failed.  Blocks the calling thread; one call at a time per idx. */
int64_t doInference(int idx, void * ctx, char * jobID, char * sessionID, char * prompt);

/* cpp/bridge.cpp:802-804 */
void stopInference(int idx);
/* cpp/bridge.cpp:662-667, :806-809: prompt text + generated text so far.  Borrowed pointer, NUL-terminated at every instant and
 * valid for the lifetime of the process: the job's text lives in append-only buffers that are never freed or moved (a full buffer
 * is succeeded by one of twice the capacity; a pointer into the old one keeps reading the text as it was).  Go copies it at once
 * with C.GoString, the reference's own buffer (a std::string c_str()) is racy. */
const char * status(char * jobID);
/* cpp/bridge.cpp:669-674, :811-814: integer-truncated ms per prompt token (t_p_eval_ms / n_p_eval) */
int64_t promptEval(char * jobID);
/* cpp/bridge.cpp:676-681, :816-819 */
int64_t getPromptTokenCount(char * jobID);
/* cpp/bridge.cpp:683-688, :821-824: integer-truncated ms per generated token (t_eval_ms / n_eval) */
int64_t timing(char * jobID);
/* cpp/bridge.cpp:690-695, :826-829: the time(NULL) seed used for the job's RNG */
uint32_t getSeed(char * jobID);

#ifdef __cplusplus
}
#endif
#endif
