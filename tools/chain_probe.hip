// chain_probe.hip — latency of a dependent chain step on gfx950: v_fma_f32, v_fma_mix_f32 (f16 operand extended in the instruction), with one or two
// waves per SIMD.  usage: chain_probe   (GPU box)   prints ns per dependent step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define N_STEPS (1 << 20)
template <int MODE>
__global__ void __launch_bounds__(512) chain(float * out, const uint32_t * in, int n) {
    float acc = in[threadIdx.x & 63];
    uint32_t w0 = in[64 + (threadIdx.x & 63)], w1 = in[128 + (threadIdx.x & 63)];
    float p0 = __uint_as_float(in[192]), p1 = __uint_as_float(in[193]);
    for (int i = 0; i < n; i += 8) {
        if (MODE == 0) {           // plain f32 fma, operands ready
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(p0), "v"(p1));
        } else if (MODE == 1) {    // fma_mix, low / high halves alternating
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(w0), "v"(p0));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(w1), "v"(p1));
            }
        } else if (MODE == 2) {    // two independent chains interleaved (issue rate)
            float acc2 = acc;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(p0), "v"(p1));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc2) : "v"(p0), "v"(p1));
            }
            acc += acc2;
        } else if (MODE == 3) {    // fma_mix with the s_nop the compiler puts between some steps
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]\n\ts_nop 0" : "+v"(acc) : "v"(w0), "v"(p0));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0" : "+v"(acc) : "v"(w1), "v"(p1));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// shader clock: clock64() (s_memtime, core clock) against wall_clock64() (100 MHz) around a long dependent chain of one wave
__global__ void clocks(long long * out, const uint32_t * in, int n) {
    float acc = in[threadIdx.x & 63]; const float p0 = __uint_as_float(in[192]), p1 = __uint_as_float(in[193]);
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(p0), "v"(p1));
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long) acc; }
}
template <int MODE> static void run(const char * name, int threads, float * out, uint32_t * in) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(chain<MODE>, dim3(256), dim3(threads), 0, 0, out, in, 1024);
    hipEventRecord(a);
    hipLaunchKernelGGL(chain<MODE>, dim3(256), dim3(threads), 0, 0, out, in, N_STEPS);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const int steps = MODE == 2 ? N_STEPS : N_STEPS;
    printf("%-34s %4d threads/CU: %.3f ns per loop step (%.2f cycles at 2.4 GHz)\n", name, threads, ms * 1e6 / steps, ms * 1e6 / steps * 2.4);
}
int main() {
    float * out; uint32_t * in; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&in, 1024);
    uint32_t h[256]; for (int i = 0; i < 256; ++i) h[i] = 0x3c003c00u; h[192] = 0x3f800000u; h[193] = 0x3f800000u;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    for (int t : { 64, 256, 512 }) {
        if (t == 64) { run<0>("v_fma_f32 dependent", 64, out, in); run<1>("v_fma_mix_f32 dependent", 64, out, in); run<3>("v_fma_mix_f32 + s_nop", 64, out, in); run<2>("v_fma_f32 two chains (per pair/2)", 64, out, in); }
        if (t == 256) { run<0>("v_fma_f32 dependent", 256, out, in); run<1>("v_fma_mix_f32 dependent", 256, out, in); }
        if (t == 512) { run<0>("v_fma_f32 dependent", 512, out, in); run<1>("v_fma_mix_f32 dependent", 512, out, in); }
    }
    {   // does a dependent v_fma_mix_f32 need the wait state hipcc puts behind it?  The same chains with and without s_nop, on random operands
        uint32_t hr[256]; uint32_t sd = 99u;
        for (int i = 0; i < 192; ++i) { sd = sd * 1664525u + 1013904223u; const uint32_t a = 0x3000u + ((sd >> 8) & 0x0fffu), b = 0xb000u + ((sd >> 20) & 0x0fffu); hr[i] = a | (b << 16); }
        hr[192] = 0x3f7f0000u; hr[193] = 0xbf7e8000u;
        hipMemcpy(in, hr, sizeof hr, hipMemcpyHostToDevice);
        static float o1[64], o3[64];
        hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, out, in, 4096); hipMemcpy(o1, out, sizeof o1, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(chain<3>, dim3(1), dim3(64), 0, 0, out, in, 4096); hipMemcpy(o3, out, sizeof o3, hipMemcpyDeviceToHost);
        int diff = 0; for (int i = 0; i < 64; ++i) diff += memcmp(&o1[i], &o3[i], 4) != 0;
        printf("dependent v_fma_mix_f32 chains of 4096 steps, with vs without s_nop: %d of 64 lanes differ (lane 0: %g %g)\n", diff, o1[0], o3[0]);
        hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    }
    long long * cl; hipMalloc(&cl, 64); long long hc[3];
    for (int n : { 1 << 12, 1 << 16, 1 << 20 }) {
        hipLaunchKernelGGL(clocks, dim3(1), dim3(64), 0, 0, cl, in, n); hipMemcpy(hc, cl, 24, hipMemcpyDeviceToHost);
        printf("one wave, %7d dependent v_fma_f32: %lld core clocks, %lld x 10 ns -> %.2f clocks per step, core clock %.0f MHz\n", n, hc[0], hc[1], (double) hc[0] / n, hc[0] / (hc[1] * 10e-9) / 1e6);
    }
    return 0;
}
