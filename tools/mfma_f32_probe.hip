// mfma_f32_probe.hip — in which order does v_mfma_f32_16x16x4_f32 accumulate?  D = A(16x4) B(4x16) + C, against candidate fmaf orders, bit for bit,
// on operands whose roundings differ between the orders (random f32 with wide exponents; and f16-valued operands as the attention kernels use).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_f32_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float * A, const float * B, const float * C, float * D, int ntile) {
    // per tile: A[16][4] row-major, B[4][16] row-major, C/D [16][16] row-major
    const int l = threadIdx.x;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const float a = A[t * 64 + (l % 16) * 4 + l / 16];          // lane l: A[m = l % 16][k = l / 16]
        const float b = B[t * 64 + (l / 16) * 16 + l % 16];         // lane l: B[k = l / 16][n = l % 16]
        f32x4 c;
        for (int r = 0; r < 4; ++r) c[r] = C[t * 256 + (4 * (l / 16) + r) * 16 + l % 16];   // reg r: [m = 4 (l / 16) + r][n = l % 16]
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[t * 256 + (4 * (l / 16) + r) * 16 + l % 16] = d[r];
    }
}
static float rnd(int mode) {
    if (mode == 0) { const float m = (float) rand() / RAND_MAX * 2.f - 1.f; return ldexpf(m, rand() % 24 - 12); }
    // f16-valued: 11-bit mantissa, modest exponent
    const int mant = rand() % 2048 - 1024; return ldexpf((float) mant, rand() % 10 - 12);
}
int main() {
    const int NT = 4096;
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<float> A(NT * 64), B(NT * 64), C(NT * 256), D(NT * 256);
        srand(7 + mode);
        for (auto & x : A) x = rnd(mode); for (auto & x : B) x = rnd(mode);
        for (auto & x : C) x = (rand() % 4 == 0) ? 0.f : rnd(0);
        float * dA, * dB, * dC, * dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, dA, dB, dC, dD, NT);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        long bad[4] = { 0, 0, 0, 0 }, n = 0;
        for (int t = 0; t < NT; ++t) for (int m = 0; m < 16; ++m) for (int nn = 0; nn < 16; ++nn) {
            const float * a = &A[t * 64 + m * 4]; const float c = C[t * 256 + m * 16 + nn];
            auto b = [&](int kk) { return B[t * 64 + kk * 16 + nn]; };
            float r0 = c; for (int kk = 0; kk < 4; ++kk) r0 = fmaf(a[kk], b(kk), r0);                // c first, k ascending: the reference's chain
            float r1 = c; for (int kk = 3; kk >= 0; --kk) r1 = fmaf(a[kk], b(kk), r1);               // k descending
            float r2 = fmaf(a[0], b(0), 0.f); for (int kk = 1; kk < 4; ++kk) r2 = fmaf(a[kk], b(kk), r2); r2 = r2 + c;   // products first, c last
            const double ex = (double) c + (double) a[0] * b(0) + (double) a[1] * b(1) + (double) a[2] * b(2) + (double) a[3] * b(3);
            const float r3 = (float) ex;                                                             // (nearly) exact sum, one rounding
            const float got = D[t * 256 + m * 16 + nn];
            uint32_t g, x; memcpy(&g, &got, 4);
            memcpy(&x, &r0, 4); bad[0] += g != x; memcpy(&x, &r1, 4); bad[1] += g != x; memcpy(&x, &r2, 4); bad[2] += g != x; memcpy(&x, &r3, 4); bad[3] += g != x;
            ++n;
        }
        printf("%s operands, %ld outputs: differ from [c, then k ascending fmaf chain] %ld | [k descending] %ld | [products first, + c] %ld | [one rounding of the exact sum] %ld\n",
               mode ? "f16-valued" : "random f32", n, bad[0], bad[1], bad[2], bad[3]);
    }
    return 0;
}
