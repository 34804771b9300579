#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string.h>
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DBAMD_TIMING [-DK_DIM=..] -Ibooster_amd/csrc tools/timing_main.cpp
#include "bamd_matvec.hip"      // one translation unit: the kernels, their phase stamps and this driver
int main() {
    const int type = 12, nrows = 4096, k = K_DIM;
    size_t wb = bamd_row_bytes(type, k) * (size_t) nrows;
    std::vector<uint8_t> hw(wb, 0x11);
    for (size_t b = 0; b < wb / 144; ++b) { hw[b*144+0]=0; hw[b*144+1]=0x1c; hw[b*144+2]=0; hw[b*144+3]=0x1c; }
    std::vector<float> hx(k, 0.5f);
    void *raw, *str; float *dx, *dres, *dy; unsigned long long* key;
    hipMalloc(&raw, wb); hipMalloc(&str, wb); hipMalloc(&dx, k*4); hipMalloc(&dres, nrows*4); hipMalloc(&dy, nrows*4); hipMalloc(&key, 8);
    hipMemcpy(raw, hw.data(), wb, hipMemcpyHostToDevice); hipMemcpy(dx, hx.data(), k*4, hipMemcpyHostToDevice); hipMemset(dres, 0, nrows*4);
    bamd_launch_repack(raw, str, type, nrows, k, nullptr);
    bamd_mv_args a; memset(&a, 0, sizeof a);
    a.seg[0].w = str; a.seg[0].out = dy; a.seg[0].type = type; a.seg[0].nrows = nrows; a.nseg = 1;
    a.x = dx; a.normw = dx; a.eps = 1e-5f; a.K = k; a.res = dres; a.best_key = key; a.mode = 2;
    for (int it = 0; it < 5; ++it) { bamd_launch_matvec(a, 0, 1, 256, nullptr); hipDeviceSynchronize(); }
    unsigned long long st[64*16]; bamd_read_stamps(st);
    for (int w = 0; w < 8; ++w) { printf("wave %d:", w); for (int j = 0; j < 12; ++j) printf(" %6lld", st[w*16+j] ? (long long)(st[w*16+j] - st[0]) : -1); printf("\n"); }
    return 0;
}
