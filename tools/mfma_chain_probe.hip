// tools/mfma_chain_probe.hip — what one v_mfma_f32_16x16x32_f16 costs a SIMD when every result feeds f32 chain FMAs on the VALU (the shape of the exact
// prefill mat-mul: isum = MFMA(A, B, 0); acc = fma(D, isum, acc)), by waves per SIMD and by what else rides along.  Round 5: three differently structured
// prefill kernels all land at ~19-21 % MFMA utilisation; this isolates the common core.
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/mfma_chain_probe tools/mfma_chain_probe.hip      run: tools/mfma_chain_probe
// Patterns (per loop iteration = 8 MFMAs, the eight e of a super-block):
//   0  MFMA only (C = 0, results summed by MFMA accumulation of a second chain so nothing is dead)
//   1  MFMA + chain FMAs on the PREVIOUS MFMA's result (software pipelined by one), packed f32 (v_pk_fma_f32 x 2)
//   2  the same with four v_fma_f32
//   3  pattern 1 + the A / B operands re-read from LDS for every MFMA (2 x ds_read_b128)
//   4  pattern 1, chain FMAs on the result of the SAME iteration's MFMA (no pipelining)
//   5  pattern 3 + 5 cheap VALU (v_perm / v_pk_fma_f16) per MFMA: the fragment build
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ void __launch_bounds__(1024) probe(float * out, int iters, const float * dsrc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // operands: small exact integers
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16) (float) ((lane + i) & 7); B[i] = (_Float16) (float) ((lane * 3 + i) & 3); }
    unsigned char * my = smem + wave * 4096 + lane * 16;
    for (int e = 0; e < 4; ++e) *(h8 *) (my + e * 1024) = e & 1 ? B : A;
    __syncthreads();
    f4 acc[8];
    for (int e = 0; e < 8; ++e) acc[e] = (f4) { 0.f, 0.f, 0.f, 0.f };
    float D[4]; for (int i = 0; i < 4; ++i) D[i] = dsrc[(lane + i) & 15];
    f4 sprev = { 0.f, 0.f, 0.f, 0.f };
    uint32_t junk = lane;
    union { uint32_t u; h2 h; } s0, n0; s0.u = 0x3c003c00u; n0.u = 0u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h8 a = A, b = B;
            if (PAT == 3 || PAT == 5) { a = *(const volatile h8 *) (my + (e & 1) * 2048); b = *(const volatile h8 *) (my + 1024 + (e & 1) * 2048); }
            const f4 z = { 0.f, 0.f, 0.f, 0.f };
            if (PAT == 0) { acc[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[e], 0, 0, 0); continue; }
            const f4 si = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);
            if (PAT == 5) {
                union { uint32_t u; h2 h; } c0, c1, r0, r1;
                const uint32_t lo = junk & 0x0f0f0f0fu;
                c0.u = __builtin_amdgcn_perm(0x64646464u, lo, 0x04010400u); c1.u = __builtin_amdgcn_perm(0x64646464u, lo, 0x04030402u);
                r0.h = __builtin_elementwise_fma(c0.h, s0.h, n0.h); r1.h = __builtin_elementwise_fma(c1.h, s0.h, n0.h);
                junk = junk * 3u + (r0.u ^ r1.u);
            }
            if (PAT == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[e][i] = fmaf(D[i], si[i], acc[e][i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[(e + 7) & 7][i] = fmaf(D[i], sprev[i], acc[(e + 7) & 7][i]);
                sprev = si;
            }
            if (PAT == 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = (float) (junk & 1);
    for (int e = 0; e < 8; ++e) for (int i = 0; i < 4; ++i) s += acc[e][i];
    s += sprev[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// pattern 6: pattern 1 with the A / B operands of e + DEPTH requested (plain ds_read_b128, addresses vary with e) while e is multiplied — the software
// pipeline of the prefill kernels (DEPTH = 2 there); DEPTH = 8: all sixteen reads of a super-block in flight before its first MFMA
template <int DEPTH>
__global__ void __launch_bounds__(1024) probe_depth(float * out, int iters, const float * dsrc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16) (float) ((lane + i) & 7); B[i] = (_Float16) (float) ((lane * 3 + i) & 3); }
    unsigned char * my = smem + wave * 4096 + lane * 16;
    for (int e = 0; e < 4; ++e) *(h8 *) (my + e * 1024) = e & 1 ? B : A;
    __syncthreads();
    f4 acc[8];
    for (int e = 0; e < 8; ++e) acc[e] = (f4) { 0.f, 0.f, 0.f, 0.f };
    float D[4]; for (int i = 0; i < 4; ++i) D[i] = dsrc[(lane + i) & 15];
    f4 sprev = { 0.f, 0.f, 0.f, 0.f };
    constexpr int R = DEPTH + 1;
    h8 ra[R], rb[R];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { ra[d] = *(const h8 *) (my + (d & 1) * 2048); rb[d] = *(const h8 *) (my + 1024 + (d & 1) * 2048); }
    for (int it = 0; it < iters; ++it) {
        const int sw = (it & 1) * 2048;                       // the addresses change from iteration to iteration (nothing to hoist)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ra[(e + DEPTH) % R] = *(const h8 *) (my + (((e + DEPTH) & 1) * 2048 ^ sw));
            rb[(e + DEPTH) % R] = *(const h8 *) (my + 1024 + (((e + DEPTH) & 1) * 2048 ^ sw));
            const f4 z = { 0.f, 0.f, 0.f, 0.f };
            const f4 si = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[e % R], rb[e % R], z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[(e + 7) & 7][i] = fmaf(D[i], sprev[i], acc[(e + 7) & 7][i]);
            sprev = si;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int e = 0; e < 8; ++e) for (int i = 0; i < 4; ++i) s += acc[e][i];
    s += sprev[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int DEPTH> static double run_depth(int waves, int iters, float * out, const float * dsrc, int ncu) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe_depth<DEPTH>, dim3(ncu), dim3(waves * 64), waves * 4096, 0, out, 16, dsrc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe_depth<DEPTH>, dim3(ncu), dim3(waves * 64), waves * 4096, 0, out, iters, dsrc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms;
}

template <int PAT> static double run(int waves, int iters, float * out, const float * dsrc, int ncu) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<PAT>, dim3(ncu), dim3(waves * 64), waves * 4096, 0, out, 16, dsrc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<PAT>, dim3(ncu), dim3(waves * 64), waves * 4096, 0, out, iters, dsrc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float * out, * dsrc; hipMalloc(&out, (size_t) ncu * 1024 * 4); hipMalloc(&dsrc, 64);
    std::vector<float> h(16); for (int i = 0; i < 16; ++i) h[i] = 1.0f + i * 0.125f;
    hipMemcpy(dsrc, h.data(), 64, hipMemcpyHostToDevice);
    const int iters = 20000;
    const double ghz = 2.4;
    printf("cycles per MFMA per SIMD (2.4 GHz assumed; 16 = the matrix pipe's rate), %d CUs, one workgroup per CU\n", ncu);
    printf("%-72s %10s %10s %10s\n", "pattern", "4 waves", "8 waves", "16 waves");
    const char * names[6] = { "0 MFMA only", "1 MFMA + 2 v_pk_fma_f32 on the previous result", "2 MFMA + 4 v_fma_f32 on the previous result",
                              "3 = 1 + A, B re-read from LDS per MFMA", "4 MFMA + 2 v_pk_fma_f32 on its OWN result", "5 = 3 + 5 cheap VALU per MFMA" };
    for (int pat = 0; pat < 6; ++pat) {
        printf("%-72s", names[pat]);
        for (int waves : { 4, 8, 16 }) {
            double ms = 0;
            switch (pat) {
                case 0: ms = run<0>(waves, iters, out, dsrc, ncu); break;
                case 1: ms = run<1>(waves, iters, out, dsrc, ncu); break;
                case 2: ms = run<2>(waves, iters, out, dsrc, ncu); break;
                case 3: ms = run<3>(waves, iters, out, dsrc, ncu); break;
                case 4: ms = run<4>(waves, iters, out, dsrc, ncu); break;
                default: ms = run<5>(waves, iters, out, dsrc, ncu); break;
            }
            const double mfma_per_simd = (double) iters * 8 * (waves / 4.0);
            printf(" %10.1f", ms * 1e-3 * ghz * 1e9 / mfma_per_simd);
        }
        printf("\n");
    }
    for (int depth : { 1, 2, 3, 4, 8 }) {
        printf("6 = 1 + A, B of e + %d read from LDS while e is multiplied%*s", depth, 17, "");
        for (int waves : { 4, 8, 16 }) {
            if (waves == 16 && depth == 8) { printf(" %10s", "(regs)"); continue; }
            double ms = depth == 1 ? run_depth<1>(waves, iters, out, dsrc, ncu) : depth == 2 ? run_depth<2>(waves, iters, out, dsrc, ncu) : depth == 3 ? run_depth<3>(waves, iters, out, dsrc, ncu)
                      : depth == 4 ? run_depth<4>(waves, iters, out, dsrc, ncu) : run_depth<8>(waves, iters, out, dsrc, ncu);
            printf(" %10.1f", ms * 1e-3 * ghz * 1e9 / ((double) iters * 8 * (waves / 4.0)));
        }
        printf("\n");
    }
    return 0;
}
