// bamd_sampler.hip — device side of the Janus sampler: the penalties of sample_janus_token (cpp/janus.cpp:236-280)
// applied to the logits where the lm_head left them, then the shortlist { logit / top >= cutoff } (janus.cpp:285-320), so that a
// few candidates instead of n_vocab floats cross PCIe per token.  Every float operation is the reference's: float x float for the
// plain scale, float x double -> float where the reference multiplies by a double expression, IEEE division for the ratio test.
// The host (bamd_bridge.cpp) sorts the shortlist, softmaxes and draws; when the order could depend on the reference's full sort
// (ties, NaN, a top logit <= 0, more candidates than the buffer holds) it reads the penalised logits back and runs the host path.
#include "bamd_kernels.h"

// order-preserving key of a non-NaN float, ties to the LOWEST id
__device__ __forceinline__ unsigned long long logit_key(float l, int id) {
    const unsigned int b = __float_as_uint(l);
    const unsigned int o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long) o << 32) | (0xffffffffu - (unsigned int) id);
}

// one thread per distinct penalised token: the EOS boost first (janus.cpp:236), then its repetition penalties in sequence (:245-265)
__global__ void sampler_penalty_kernel(float * logits, const bamd_logit_penalty * pen, int n, bamd_shortlist_head * head) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { head->top_key = 0ull; head->count = 0; head->ntop = 0; head->nan = 0; head->top_id = -1; head->top_logit = 0.f; head->cutoff = 0.f; }
    if (i >= n) return;
    const bamd_logit_penalty e = pen[i];
    float l = logits[e.id];
    if (e.pre != 0.0) l = (float) ((double) l * e.pre);
    for (int c = 0; c < e.count; ++c) l = e.kind ? (float) ((double) l * e.d) : l * e.f;
    logits[e.id] = l;
}

// the x0.5 pass over the incompatible classes (janus.cpp:269-283) fused with the search for the top logit
__global__ void __launch_bounds__(1024) sampler_max_kernel(float * logits, const uint8_t * halve_class, int halve, int V, bamd_shortlist_head * head) {
    __shared__ unsigned long long wk[16];
    __shared__ int wnan[16];
    unsigned long long key = 0ull; int nan = 0;
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < V; id += gridDim.x * blockDim.x) {
        float l = logits[id];
        if (halve && halve_class[id]) { l = l * 0.5f; logits[id] = l; }
        if (l != l) nan = 1;
        else { const unsigned long long k = logit_key(l, id); key = k > key ? k : key; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off);
        key = o > key ? o : key;
        nan |= __shfl_xor(nan, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { wk[wave] = key; wnan[wave] = nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int) (blockDim.x >> 6); ++w) { key = wk[w] > key ? wk[w] : key; nan |= wnan[w]; }
        atomicMax(&head->top_key, key);
        if (nan) atomicOr(&head->nan, 1);
    }
}

// candidates that survive the ratio test against the top logit, in any order; ntop = how many logits equal the top
__global__ void __launch_bounds__(1024) sampler_select_kernel(const float * logits, const float * cutoff_of, int V, bamd_shortlist_head * head,
                                                              int32_t * ids, float * vals, int cap) {
    const unsigned long long key = head->top_key;
    if (key == 0ull) return;                                        // nothing but NaN
    const int top_id = (int) (0xffffffffu - (unsigned int) (key & 0xffffffffull));
    const float top = logits[top_id];
    const float cutoff = cutoff_of[top_id];
    if (blockIdx.x == 0 && threadIdx.x == 0) { head->top_id = top_id; head->top_logit = top; head->cutoff = cutoff; }
    if (!(top > 0.0f)) return;                                      // the ratio test is not monotone then: host path
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < V; id += gridDim.x * blockDim.x) {
        const float l = logits[id];
        if (l == top) atomicAdd(&head->ntop, 1);
        if (!(l / top < cutoff)) {
            const int slot = atomicAdd(&head->count, 1);
            if (slot < cap) { ids[slot] = id; vals[slot] = l; }
        }
    }
}

void bamd_launch_sampler_shortlist(float * logits, const bamd_logit_penalty * pen, int n_pen, const uint8_t * halve_class, int halve,
                                   const float * cutoff_of, int V, bamd_shortlist_head * head, int32_t * ids, float * vals, int cap, hipStream_t s) {
    const int nb = (V + 1023) / 1024;
    hipLaunchKernelGGL(sampler_penalty_kernel, dim3((n_pen + 255) / 256 > 0 ? (n_pen + 255) / 256 : 1), dim3(256), 0, s, logits, pen, n_pen, head);
    hipLaunchKernelGGL(sampler_max_kernel, dim3(nb), dim3(1024), 0, s, logits, halve_class, halve, V, head);
    hipLaunchKernelGGL(sampler_select_kernel, dim3(nb), dim3(1024), 0, s, logits, cutoff_of, V, head, ids, vals, cap);
}
