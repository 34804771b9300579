// bamd_aql.h — a decode step replayed from the library's OWN AQL queue instead of a hipGraph (round 6).
//
// Why: between two dependent launches the HIP runtime (stream or graph alike) puts agent-scope acquire / release fences — an L2 write-back at the end of
// one kernel, an L1 / L2 / scalar-cache invalidate at the start of the next.  tools/aql_probe.hip measured that pair at 0.38 us per launch on a mat-vec
// shaped chain (profiles/r03_aql_probe.txt, `aqlbar agent/agent` 10.85 vs `aqlbar none/none` 10.47 us per kernel); a decode step is 130 dependent launches
// (402 at the 70B depth).  HIP has no switch for fence scope NONE, so the step's launch sequence is RECORDED once — every launcher of the decode kernels goes
// through BAMD_LAUNCH, which appends {kernel, grid, block, LDS bytes, packed arguments} to the recording instead of launching — the kernels' code objects
// are loaded a second time through HSA (the same ISA: built from the same sources with the same flags, embedded in this library), and a replay is N x P
// AQL packets written straight into the queue's ring: the first one acquires at system scope, the last one releases at system scope and carries the
// completion signal, everything in between runs with scope NONE.  What then carries data from launch to launch is spelled out in bamd_device.h
// ("Inter-kernel data"): sc1 stores and sc1 loads only.
//
// What it replaces in the reference: one CUDA graph per token, cpp/ggml/src/ggml-cuda.cu:2467-2722.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <vector>

struct bamd_aql_launch {                 // one recorded launch
    const void * host_fn;                // the kernel's host-side stub (what hipLaunchKernelGGL takes): resolved to its code-object symbol at build time
    uint32_t grid[3], block[3];
    uint32_t lds_bytes;                  // dynamic LDS
    std::vector<uint8_t> kernarg;        // explicit arguments, packed with the kernel ABI's natural alignment
};
struct bamd_aql_recording { std::vector<bamd_aql_launch> launches; };
extern thread_local bamd_aql_recording * bamd_aql_rec;          // non-null: BAMD_LAUNCH records instead of launching (bamd_aql.cpp)

template <typename P> inline void bamd_aql_pack1(std::vector<uint8_t> & buf, const P & v) {
    const size_t al = alignof(P);
    const size_t off = (buf.size() + al - 1) / al * al;
    buf.resize(off + sizeof(P));
    memcpy(buf.data() + off, &v, sizeof(P));
}
// the argument list is converted to the kernel's PARAMETER types first (what a launch does), then laid out as the AMDGPU kernel ABI lays out by-value
// parameters: each at the next multiple of its alignment
template <typename... P, typename... A>
inline void bamd_aql_record(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, A &&... args) {
    bamd_aql_launch l;
    l.host_fn = (const void *) kernel;
    l.grid[0] = grid.x; l.grid[1] = grid.y; l.grid[2] = grid.z; l.block[0] = block.x; l.block[1] = block.y; l.block[2] = block.z;
    l.lds_bytes = (uint32_t) lds;
    (bamd_aql_pack1<P>(l.kernarg, static_cast<P>(args)), ...);
    bamd_aql_rec->launches.push_back(std::move(l));
}
// every launcher of a kernel that can be part of a decode step uses this in place of hipLaunchKernelGGL
#define BAMD_LAUNCH(KERNEL, GRID, BLOCK, LDS, STREAM, ...) do { \
        if (bamd_aql_rec) bamd_aql_record(KERNEL, dim3(GRID), dim3(BLOCK), (size_t) (LDS), __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__); } while (0)

// ---- host side (bamd_aql.cpp) ---------------------------------------------------------------------------------------------------------------
struct bamd_aql_graph;
// 1 when the own-queue path can be used on `device` (HSA agent found, queue created, code objects loaded); the reason otherwise in *why (may be null)
int bamd_aql_available(int device, const char ** why);
// turns a finished recording into packets for `device`: kernel objects resolved, kernel arguments (explicit + the hidden block of code object v5) in
// kernarg memory.  nullptr + *why on failure (an unknown kernel, an allocation)
bamd_aql_graph * bamd_aql_build(int device, const bamd_aql_recording & rec, const char ** why);
// `replays` back-to-back replays of the graph; returns when the last packet has completed (or 1 after `timeout_s`); *seconds = submit -> completion on the host clock
int bamd_aql_run(bamd_aql_graph * g, int replays, double * seconds, const char ** why);
int bamd_aql_graph_launches(const bamd_aql_graph * g);
void bamd_aql_free(bamd_aql_graph * g);
