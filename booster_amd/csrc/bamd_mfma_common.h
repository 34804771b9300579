// bamd_mfma_common.h — vector types and the global -> LDS copy helpers shared by the matrix-core prefill kernels (bamd_prefill.hip, bamd_prefill2.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 bamd_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 bamd_h4 __attribute__((ext_vector_type(4)));
typedef float bamd_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void * bamd_lds_vp;
typedef const __attribute__((address_space(1))) void * bamd_glb_vp;
// global -> LDS copy without registers: each active lane moves 16 (4) bytes from ITS global address to LDS base + lane * 16 (4).
// Issued through inline asm on purpose: for the builtin the compiler cannot tell the destination buffer from the buffer being read
// (both index the same dynamic LDS array) and puts s_waitcnt vmcnt(0) in front of the next LDS read, which serialises the copy with
// the math it is meant to overlap.  The asm is invisible to the wait-count pass, so the consumer side waits explicitly
// (lds_dma_wait before the barrier that publishes the stage); the compiler's own counted waits stay valid (completion is in order).
__device__ __forceinline__ void lds_dma16(const void * gsrc, void * lds_wave_base) {
    uint32_t keep; const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t) (size_t) (bamd_lds_vp) lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void * gsrc, void * lds_wave_base) {
    uint32_t keep; const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t) (size_t) (bamd_lds_vp) lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// vmcnt(0) as the BUILTIN (imm: vmcnt 0, expcnt 7, lgkmcnt 15): the wait-count pass sees it and does not repeat it behind the next issue
__device__ __forceinline__ void lds_dma_wait() { __builtin_amdgcn_s_waitcnt(0x0f70); }
// the same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: no vector address arithmetic per copy
__device__ __forceinline__ void lds_dma16_s(const void * sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void lds_dma4_s(const void * sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
