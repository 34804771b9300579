// which DPP row rotate/shift gives lane i <- lane i+4 ?  (GPU box only)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __global__ void k(int * out) { int v = threadIdx.x; out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> void run(const char * name) {
    int * d; hipMalloc(&d, 256); k<CTRL><<<1, 64>>>(d); int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%-12s:", name); for (int i = 0; i < 20; ++i) printf(" %2d", h[i]); printf("\n"); hipFree(d);
}
int main() { run<0x104>("row_shl:4"); run<0x114>("row_shr:4"); run<0x124>("row_ror:4"); run<0x12C>("row_ror:12"); run<0x141>("half_mirror"); run<0x140>("mirror"); run<0x4E>("qp xor2"); run<0xB1>("qp xor1"); return 0; }
