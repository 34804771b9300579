// dot4_probe.cpp — does the VOP3P form `v_dot4_i32_i8 d, a, b, c` (inline-constant / register accumulator) agree with
// __builtin_amdgcn_sdot4 on gfx950?   hipcc --offload-arch=gfx950 -O3 tools/dot4_probe.cpp -o /tmp/dot4_probe && /tmp/dot4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint32_t * a, const uint32_t * b, int * o0, int * o1, int * o2, int * o3) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    o0[i] = __builtin_amdgcn_sdot4((int) a[i], (int) b[i], 0, false);
    int r1, r2, r3;
    asm volatile("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(r1) : "v"(a[i]), "v"(b[i]));
    asm volatile("v_dot4_i32_i8 %0, %1, %2, 0 neg_lo:[1,1,0]" : "=v"(r2) : "v"(a[i]), "v"(b[i]));
    asm volatile("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(r3) : "v"(a[i]), "v"(b[i]), "v"(1000));
    o1[i] = r1; o2[i] = r2; o3[i] = r3;
}
int main() {
    const int n = 1024; uint32_t ha[n], hb[n]; uint32_t s = 1;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; ha[i] = i < 512 ? (s & 0x0f0f0f0fu) : s; s = s * 1664525u + 1013904223u; hb[i] = s; }
    uint32_t * a, * b; int * o[4];
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); for (int j = 0; j < 4; ++j) hipMalloc(&o[j], n * 4);
    hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, b, o[0], o[1], o[2], o[3]);
    int h[4][n]; for (int j = 0; j < 4; ++j) hipMemcpy(h[j], o[j], n * 4, hipMemcpyDeviceToHost);
    int bad[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < n; ++i) {
        int ref = 0; for (int t = 0; t < 4; ++t) ref += (int) (int8_t) (ha[i] >> (8 * t)) * (int) (int8_t) (hb[i] >> (8 * t));
        bad[0] += h[0][i] != ref; bad[1] += h[1][i] != ref; bad[2] += h[2][i] != ref; bad[3] += h[3][i] != ref + 1000;
    }
    printf("mismatches: builtin %d  vop3p(0) %d  vop3p neg_lo %d  vop3p(reg acc) %d   (first: ref-ish %d vs %d %d)\n", bad[0], bad[1], bad[2], bad[3], h[0][600], h[1][600], h[2][600]);
    return 0;
}
