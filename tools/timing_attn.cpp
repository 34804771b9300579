#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string.h>
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DBAMD_TIMING [-DK_DIM=..] -Ibooster_amd/csrc tools/timing_attn.cpp
#include "bamd_attention.hip"      // one translation unit: the kernels, their phase stamps and this driver
int main() {
    const int H = 32, Hkv = 8, hd = 128, n_ctx = 512, pos = 250, Ekv = Hkv * hd; float *probs;
    bamd_step_state h; memset(&h, 0, sizeof h); h.pos = pos; h.n_ctx = n_ctx; h.n_kv = 256;
    bamd_attn_args a; memset(&a, 0, sizeof a);
    bamd_step_state * st; hipMalloc(&st, sizeof h); hipMemcpy(st, &h, sizeof h, hipMemcpyHostToDevice); a.st = st;
    float *q, *k, *v, *rope, *scores, *out; unsigned short *kc, *vc;
    hipMalloc(&q, H*hd*4); hipMalloc(&k, Ekv*4); hipMalloc(&v, Ekv*4); hipMalloc(&rope, n_ctx*hd*4); hipMalloc(&scores, H*n_ctx*4); hipMalloc(&probs, H*n_ctx*4); hipMalloc(&out, H*hd*4);
    hipMalloc(&kc, n_ctx*Ekv*2); hipMalloc(&vc, n_ctx*Ekv*2);
    hipMemset(q, 0, H*hd*4); hipMemset(k, 0, Ekv*4); hipMemset(v, 0, Ekv*4); hipMemset(rope, 0, n_ctx*hd*4); hipMemset(kc, 0, n_ctx*Ekv*2); hipMemset(vc, 0, n_ctx*Ekv*2);
    a.q = q; a.k = k; a.v = v; a.kc = kc; a.vc = vc; a.rope = rope; a.scores = scores; a.probs = probs; a.out = out; a.hd = hd; a.Hkv = Hkv; a.n_ctx = n_ctx; a.kq_scale = 0.088f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) { bamd_launch_attention(a, 4, 8, nullptr); hipDeviceSynchronize(); }
    hipEventRecord(e0); for (int it = 0; it < 100; ++it) bamd_launch_attention(a, 4, 8, nullptr); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("fused: %.2f us per launch\n", ms * 10);
    hipEventRecord(e0); for (int it = 0; it < 100; ++it) bamd_launch_attention(a, 4, -8, nullptr); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1); printf("3-kernel: %.2f us per launch-set\n", ms * 10);
    bamd_launch_attention(a, 4, 8, nullptr); hipDeviceSynchronize();
    unsigned long long st2[64*16]; bamd_read_stamps(st2);
    for (int w = 0; w < 8; w += 7) { printf("wave %d:", w); for (int j = 0; j < 7; ++j) printf(" %6lld", (long long)(st2[w*16+j] - st2[0])); printf("\n"); }
    return 0;
}
