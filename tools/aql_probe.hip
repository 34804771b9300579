// aql_probe.hip — what does a dependent launch cost on MI355X, by hand-over mechanism?  (round 3 experiment behind DESIGN §7b)
//
// A chain of N "mat-vec shaped" kernels (256 workgroups x 512 threads; each streams its own slice of a weight buffer, reads ALL of the
// previous kernel's 16 KB output vector, writes its share of the next one) is run five ways:
//   hip      : ordinary launches on one HIP stream (kernel boundary = the dependency)                      — what the product does today
//   hipany   : hipExtLaunchKernel(..., hipExtAnyOrderLaunch) + device flags
//   aqlbar   : our own HSA queue, AQL packets WITH the barrier bit (acquire/release fence scopes selectable)
//   aqlflag  : our own HSA queue, AQL packets WITHOUT the barrier bit: kernel n+1 is dispatched while kernel n runs, requests the first
//              PRE chunks of its weights, then waits on kernel n's arrival counters (1 counter, or 8 shards), then reads the vector with
//              sc1 loads (the producer stored it sc1: write-through) — "run-ahead launches"
// Every spin is bounded (give-up code in err[]); the chain is value-checked (out = in[perm] + 1, so after N kernels every element = N).
//
// build (here, no GPU needed):   tools/build_aql_probe.sh        run (GPU box): tools/aql_probe [N] [slice_MB]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <chrono>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char * m_ = ""; hsa_status_string(s_, &m_); printf("HSA error %s at line %d\n", m_, __LINE__); exit(1); } } while (0)

struct ChainArgs {
    const float * in; float * out;                 // 4096 floats each
    const uint8_t * w;                             // this kernel's weight slice
    int pre, post;                                 // chunks (8 KB per workgroup each) requested before / after the wait
    const unsigned * done_prev; unsigned * done_mine; unsigned target; int nshard;   // nshard 0: no flag wait (a kernel boundary orders us)
    int ashard;                                    // arrival counters of this kernel (0: none)
    int delay_ticks;                               // emulated latency-bound work (100 MHz ticks) instead of streaming
    unsigned long long * stamps;                   // [256 workgroups][4]: entry, wait done, exit
    unsigned * err; float * sink;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define MAXPRE 16
extern "C" __global__ void __launch_bounds__(512) chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float * red = (float *) smem;
    const int tid = threadIdx.x, wg = blockIdx.x;
    if (tid == 0) a.stamps[wg * 4 + 0] = wall_clock64();
    const u32x4 * wp = (const u32x4 *) (a.w + (size_t) wg * (size_t) (a.pre + a.post) * 8192) + tid;
    u32x4 ring[MAXPRE];
#pragma unroll
    for (int i = 0; i < MAXPRE; ++i) ring[i] = __builtin_nontemporal_load(wp + (i < a.pre ? i : 0) * 512);     // run-ahead requests
    if (a.nshard > 0) {
        if (tid < a.nshard) {
            unsigned spins = 0;
            while (__hip_atomic_load(a.done_prev + tid * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 21)) { atomicAdd(a.err, 1u); break; }
            }
        }
        __syncthreads();
    }
    if (tid == 0) a.stamps[wg * 4 + 1] = wall_clock64();
    if (a.delay_ticks > 0) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < (unsigned long long) a.delay_ticks) __builtin_amdgcn_s_sleep(4); }
    // the whole 16 KB vector, coherently (sc1 loads: L1 bypassed; the producer wrote through)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += __hip_atomic_load(a.in + j * 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    float tot = 0.f;
    for (int k = 0; k < 8; ++k) tot += red[k];
    // consume the ring, then stream the rest (8 requests in flight per thread)
    u32x4 acc = { 0u, 0u, 0u, 0u };
#pragma unroll
    for (int i = 0; i < MAXPRE; ++i) acc ^= ring[i];
    for (int c = a.pre; c < a.pre + a.post; c += 8) {
        u32x4 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_nontemporal_load(wp + (c + i < a.pre + a.post ? c + i : c) * 512);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= t[i];
    }
    if (tid < 16) {
        const int i = wg * 16 + tid;
        const float v = __hip_atomic_load(a.in + ((i * 17 + 5) & 4095), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.out + i, v + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tot == 1234567.f && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) a.sink[tid] = tot;     // keeps the loads alive
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (a.ashard > 0) __hip_atomic_fetch_add(a.done_mine + (wg % a.ashard) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.stamps[wg * 4 + 2] = wall_clock64();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct Hsa {
    hsa_agent_t gpu{}, cpu{}; hsa_queue_t * q = nullptr; hsa_amd_memory_pool_t kernarg_pool{}; bool have_pool = false;
    uint64_t kobj = 0; uint32_t kernarg_size = 0, group_size = 0, private_size = 0;
    hsa_signal_t done{};
};
static hsa_status_t agent_cb(hsa_agent_t ag, void * data) {
    Hsa * h = (Hsa *) data; hsa_device_type_t t;
    hsa_agent_get_info(ag, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && h->gpu.handle == 0) h->gpu = ag;
    if (t == HSA_DEVICE_TYPE_CPU && h->cpu.handle == 0) h->cpu = ag;
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t pool_cb(hsa_amd_memory_pool_t pool, void * data) {
    Hsa * h = (Hsa *) data; hsa_amd_segment_t seg; uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !h->have_pool) { h->kernarg_pool = pool; h->have_pool = true; }
    return HSA_STATUS_SUCCESS;
}
static std::vector<char> slurp(const char * path) {
    std::vector<char> v; FILE * f = fopen(path, "rb"); if (!f) return v;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != (size_t) n) v.clear(); fclose(f); return v;
}
static void hsa_setup(Hsa & h, const char * hsaco) {
    HK(hsa_init());
    HK(hsa_iterate_agents(agent_cb, &h));
    HK(hsa_amd_agent_iterate_memory_pools(h.cpu, pool_cb, &h));
    if (!h.have_pool) { printf("no kernarg pool\n"); exit(1); }
    HK(hsa_queue_create(h.gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &h.q));
    static std::vector<char> img = slurp(hsaco);
    if (img.empty()) { printf("cannot read %s\n", hsaco); exit(1); }
    hsa_code_object_reader_t rd; hsa_executable_t ex;
    HK(hsa_code_object_reader_create_from_memory(img.data(), img.size(), &rd));
    HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    HK(hsa_executable_load_agent_code_object(ex, h.gpu, rd, nullptr, nullptr));
    HK(hsa_executable_freeze(ex, nullptr));
    hsa_executable_symbol_t sym;
    HK(hsa_executable_get_symbol_by_name(ex, "chain_kernel.kd", &h.gpu, &sym));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &h.kobj));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &h.kernarg_size));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &h.group_size));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &h.private_size));
    HK(hsa_signal_create(1, 0, nullptr, &h.done));
    printf("hsa: kernel object %#llx kernarg %u B group %u B private %u B, queue size %u\n", (unsigned long long) h.kobj, h.kernarg_size, h.group_size, h.private_size, h.q->size);
}

struct Bufs {
    float * va = nullptr, * vb = nullptr, * sink = nullptr; uint8_t * w = nullptr; size_t wbytes = 0;
    unsigned * flags = nullptr; unsigned long long * stamps = nullptr; unsigned * err = nullptr;
    int N = 0;
};
static void reset(Bufs & b) {
    CK(hipMemset(b.va, 0, 4096 * 4)); CK(hipMemset(b.vb, 0, 4096 * 4));
    CK(hipMemset(b.flags, 0, (size_t) (b.N + 1) * 8 * 128)); CK(hipMemset(b.err, 0, 64));
    CK(hipMemset(b.stamps, 0, (size_t) b.N * 256 * 32));
    CK(hipDeviceSynchronize());
}
static ChainArgs args_for(const Bufs & b, int n, int pre, int post, int nshard, size_t slice) {
    ChainArgs a; memset(&a, 0, sizeof a);
    a.in = (n & 1) ? b.vb : b.va; a.out = (n & 1) ? b.va : b.vb;
    a.w = b.w + ((size_t) n * slice) % (b.wbytes - slice + 1);
    a.pre = pre; a.post = post;
    a.done_prev = b.flags + (size_t) n * 8 * 32; a.done_mine = b.flags + (size_t) (n + 1) * 8 * 32;
    a.nshard = n == 0 ? 0 : nshard; a.ashard = nshard; a.target = nshard > 0 ? 256u / (unsigned) nshard : 0u;
    a.stamps = b.stamps + (size_t) n * 256 * 4; a.err = b.err; a.sink = b.sink;
    return a;
}
static void report(const char * name, Bufs & b, double host_us, int N) {
    std::vector<float> v(4096); std::vector<unsigned long long> raw((size_t) N * 256 * 4), st((size_t) N * 4); unsigned err[2] = { 0, 0 };
    CK(hipMemcpy(v.data(), (N & 1) ? b.vb : b.va, 4096 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(raw.data(), b.stamps, raw.size() * 8, hipMemcpyDeviceToHost));
    for (int n = 0; n < N; ++n) {                            // first entry, last wait-done, last exit over the workgroups
        unsigned long long e = ~0ull, w = 0, x = 0;
        for (int g = 0; g < 256; ++g) { const unsigned long long * r = &raw[((size_t) n * 256 + g) * 4]; e = std::min(e, r[0]); w = std::max(w, r[1]); x = std::max(x, r[2]); }
        st[(size_t) n * 4] = e; st[(size_t) n * 4 + 1] = w; st[(size_t) n * 4 + 2] = x;
    }
    CK(hipMemcpy(err, b.err, 8, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 4096; ++i) if (v[i] != (float) N) ++bad;
    // device clock: 100 MHz wall clock
    const double dev_us = (double) (st[(size_t) (N - 1) * 4 + 2] - st[0]) / 100.0;
    double overlap = 0, waitsum = 0; int inorder = 1;
    for (int n = 1; n < N; ++n) {
        const long long ov = (long long) st[(size_t) (n - 1) * 4 + 2] - (long long) st[(size_t) n * 4];     // previous exit - my first entry: > 0 = ran ahead
        overlap += (double) ov / 100.0;
        waitsum += (double) ((long long) st[(size_t) n * 4 + 1] - (long long) st[(size_t) n * 4]) / 100.0;
        if (st[(size_t) n * 4] < st[(size_t) (n - 1) * 4]) inorder = 0;
    }
    printf("%-28s host %8.2f us/kernel  device %8.2f us/kernel  run-ahead %6.2f us  entry->wait-done %6.2f us  %s%s  wrong %d  give-ups %u\n", name, host_us / N, dev_us / N,
           overlap / (N - 1), waitsum / (N - 1), inorder ? "entries in order" : "ENTRIES OUT OF ORDER", "", bad, err[0]);
    fflush(stdout);
}

int main(int argc, char ** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 200;
    const int slice_mb = argc > 2 ? atoi(argv[2]) : 32;
    const char * hsaco = argc > 3 ? argv[3] : "tools/aql_probe.hsaco";
    const size_t slice = (size_t) slice_mb << 20;
    const int chunks = (int) (slice / 256 / 8192);           // 8 KB chunks per workgroup
    Bufs b; b.N = N; b.wbytes = (size_t) 2 << 30;
    CK(hipSetDevice(0));
    CK(hipMalloc(&b.va, 4096 * 4)); CK(hipMalloc(&b.vb, 4096 * 4)); CK(hipMalloc(&b.sink, 4096)); CK(hipMalloc(&b.w, b.wbytes));
    CK(hipMalloc(&b.flags, (size_t) (N + 1) * 8 * 128)); CK(hipMalloc(&b.stamps, (size_t) N * 256 * 32)); CK(hipMalloc(&b.err, 64));
    CK(hipMemset(b.w, 1, b.wbytes));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t lds = 64 * 1024;                            // two workgroups per CU at most
    CK(hipFuncSetAttribute((const void *) chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    printf("chain of %d kernels, 256 x 512 threads, %d MB of weights per kernel (%d chunks of 8 KB per workgroup), 16 KB vector all-to-all\n", N, slice_mb, chunks);

    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    for (int rep = 0; rep < 2; ++rep) {
        // ---- hip: plain launches -------------------------------------------------------------------------------------
        for (int pre : { 0, 8, 16 }) {
            reset(b);
            auto t0 = now();
            for (int n = 0; n < N; ++n) { ChainArgs a = args_for(b, n, pre, chunks - pre, 0, slice); hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(512), lds, s, a); }
            const double enq = us_since(t0);
            CK(hipStreamSynchronize(s));
            char nm[64]; snprintf(nm, sizeof nm, "hip pre=%d (enqueue %.2f us/launch)", pre, enq / N); report(nm, b, us_since(t0), N);
        }
        // ---- hip graph ---------------------------------------------------------------------------------------------------
        {
            reset(b);
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int n = 0; n < N; ++n) { ChainArgs a = args_for(b, n, 8, chunks - 8, 0, slice); hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(512), lds, s, a); }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            auto t0 = now();
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            report("hipgraph pre=8", b, us_since(t0), N);
            (void) hipGraphExecDestroy(ge); (void) hipGraphDestroy(g);
        }
        // ---- hip any-order + flags --------------------------------------------------------------------------------------
        for (int pre : { 8, 16 }) for (int nshard : { 1, 8 }) {
            reset(b);
            auto t0 = now();
            for (int n = 0; n < N; ++n) {
                ChainArgs a = args_for(b, n, pre, chunks - pre, nshard, slice); void * pa[1] = { &a };
                CK(hipExtLaunchKernel((const void *) chain_kernel, dim3(256), dim3(512), pa, lds, s, nullptr, nullptr, hipExtAnyOrderLaunch));
            }
            const double enq = us_since(t0);
            CK(hipStreamSynchronize(s));
            char nm[96]; snprintf(nm, sizeof nm, "hipany shards=%d pre=%d (enq %.2f)", nshard, pre, enq / N); report(nm, b, us_since(t0), N);
        }
        // ---- the same captured into a hipGraph: does the replay keep the any-order property? -----------------------------------
        {
            reset(b);
            hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
            hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            for (int n = 0; n < N && e == hipSuccess; ++n) {
                ChainArgs a = args_for(b, n, 8, chunks - 8, 8, slice); void * pa[1] = { &a };
                e = hipExtLaunchKernel((const void *) chain_kernel, dim3(256), dim3(512), pa, lds, s, nullptr, nullptr, hipExtAnyOrderLaunch);
            }
            hipError_t e2 = hipStreamEndCapture(s, &g);
            if (e == hipSuccess && e2 == hipSuccess && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
                auto t0 = now();
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                report("hipany shards=8 pre=8 GRAPH", b, us_since(t0), N);
                (void) hipGraphExecDestroy(ge);
            } else printf("capturing hipExtLaunchKernel(any-order) failed: %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2));
            if (g) (void) hipGraphDestroy(g);
            (void) hipGetLastError();
        }
    }

    // ---- attention || wo: a 32-workgroup latency-bound kernel (4.5 us) followed by a 224-workgroup kernel that streams 9.2 MB and needs ALL of
    //      the first one's output: (a) both ordinary launches, (b) the second launched any-order: it requests its weights while the first runs,
    //      then waits on the first one's arrival counter -----------------------------------------------------------------------------------
    for (int rep = 0; rep < 2; ++rep) for (int co = 0; co < 2; ++co) {
        reset(b);
        const int NP = N / 2;
        auto t0 = now();
        for (int n = 0; n < NP; ++n) {
            ChainArgs a = args_for(b, 2 * n, 0, 0, 0, slice); a.ashard = 1; a.delay_ticks = 450; a.nshard = 0;
            hipLaunchKernelGGL(chain_kernel, dim3(32), dim3(512), lds, s, a);
            ChainArgs w = args_for(b, 2 * n + 1, 5, 0, co ? 1 : 0, slice); w.ashard = 0; w.target = 32; w.nshard = co ? 1 : 0;
            void * pa[1] = { &w };
            CK(hipExtLaunchKernel((const void *) chain_kernel, dim3(224), dim3(512), pa, lds, s, nullptr, nullptr, co ? hipExtAnyOrderLaunch : 0));
        }
        CK(hipStreamSynchronize(s));
        const double us = us_since(t0);
        std::vector<unsigned long long> raw((size_t) N * 256 * 4);
        CK(hipMemcpy(raw.data(), b.stamps, raw.size() * 8, hipMemcpyDeviceToHost));
        unsigned err[2]; CK(hipMemcpy(err, b.err, 8, hipMemcpyDeviceToHost));
        auto first_entry = [&](int n, int g) { unsigned long long e = ~0ull; for (int i = 0; i < g; ++i) e = std::min(e, raw[((size_t) n * 256 + i) * 4]); return e; };
        auto last_exit = [&](int n, int g) { unsigned long long e = 0; for (int i = 0; i < g; ++i) e = std::max(e, raw[((size_t) n * 256 + i) * 4 + 2]); return e; };
        double pair = 0, tailb = 0, ahead = 0;
        for (int n = 1; n < NP; ++n) {
            pair += (double) (first_entry(2 * n, 32) - first_entry(2 * n - 2, 32)) / 100.0;
            tailb += (double) ((long long) last_exit(2 * n + 1, 224) - (long long) last_exit(2 * n, 32)) / 100.0;
            ahead += (double) ((long long) last_exit(2 * n, 32) - (long long) first_entry(2 * n + 1, 224)) / 100.0;
        }
        printf("pair %-22s host %7.2f us/pair  device %7.2f us/pair  second kernel: enters %5.2f us before the first ends, exits %5.2f us after it  give-ups %u\n",
               co ? "co-launched (any-order)" : "two ordinary launches", us / NP, pair / (NP - 1), ahead / (NP - 1), tailb / (NP - 1), err[0]);
        fflush(stdout);
    }

    // ---- two HIP streams: per iteration  A (32 workgroups, 4.5 us latency-bound: "attention")  ->  B (224 workgroups, 9.2 MB: "wo")  ->  C (256
    //      workgroups, 32 MB: "gate/up").  (a) one stream, three ordinary launches.  (b) A and C on stream 1, B on stream 2: B is released by an
    //      event behind C of the previous iteration (so that it never takes CUs from C), requests its weights while A runs and waits on A's
    //      counter; C follows A in stream order and waits on B's counters after requesting its first weights ------------------------------------
    {
        hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        const int NI = N / 3;
        std::vector<hipEvent_t> evs(NI);
        for (auto & e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int rep = 0; rep < 2; ++rep) for (int co = 0; co < 2; ++co) {
            reset(b);
            auto t0 = now();
            for (int n = 0; n < NI; ++n) {
                ChainArgs a = args_for(b, 3 * n, 0, 0, 0, slice); a.ashard = 1; a.delay_ticks = 450; a.nshard = 0;
                ChainArgs w = args_for(b, 3 * n + 1, 5, 0, 1, slice); w.ashard = 8; w.target = 32; w.nshard = co ? 1 : 0;     // arrives on 8 shards (28 each)
                ChainArgs c = args_for(b, 3 * n + 2, 8, chunks - 8, 8, slice); c.ashard = 0; c.target = 28; c.nshard = co ? 8 : 0;
                hipLaunchKernelGGL(chain_kernel, dim3(32), dim3(512), lds, s, a);
                if (co) { if (n > 0) CK(hipStreamWaitEvent(s2, evs[n - 1], 0)); hipLaunchKernelGGL(chain_kernel, dim3(224), dim3(512), lds, s2, w); }
                else hipLaunchKernelGGL(chain_kernel, dim3(224), dim3(512), lds, s, w);
                hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(512), lds, s, c);
                if (co) CK(hipEventRecord(evs[n], s));
            }
            double enq = us_since(t0);
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
            double us = us_since(t0);
            if (rep == 1) {          // second repetition: the same schedule captured into ONE hipGraph (fork / join by events) and replayed
                reset(b);
                hipGraph_t g; hipGraphExec_t ge; hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int n = 0; n < NI; ++n) {
                    ChainArgs a = args_for(b, 3 * n, 0, 0, 0, slice); a.ashard = 1; a.delay_ticks = 450; a.nshard = 0;
                    ChainArgs w = args_for(b, 3 * n + 1, 5, 0, 1, slice); w.ashard = 8; w.target = 32; w.nshard = co ? 1 : 0;
                    ChainArgs c = args_for(b, 3 * n + 2, 8, chunks - 8, 8, slice); c.ashard = 0; c.target = 28; c.nshard = co ? 8 : 0;
                    if (co) { CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0)); }      // fork: B depends on what precedes A (the previous C), not on A
                    hipLaunchKernelGGL(chain_kernel, dim3(32), dim3(512), lds, s, a);
                    hipLaunchKernelGGL(chain_kernel, dim3(224), dim3(512), lds, co ? s2 : s, w);
                    if (co) { CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s, ej, 0)); }      // join: C depends on A (stream order) and on B
                    hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(512), lds, s, c);
                }
                CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                t0 = now();
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                us = us_since(t0); enq = 0;
                (void) hipGraphExecDestroy(ge); (void) hipGraphDestroy(g);
                printf("(captured into one hipGraph) ");
            }
            std::vector<unsigned long long> raw((size_t) N * 256 * 4);
            CK(hipMemcpy(raw.data(), b.stamps, raw.size() * 8, hipMemcpyDeviceToHost));
            unsigned err[2]; CK(hipMemcpy(err, b.err, 8, hipMemcpyDeviceToHost));
            auto first_entry = [&](int n, int g) { unsigned long long e = ~0ull; for (int i = 0; i < g; ++i) e = std::min(e, raw[((size_t) n * 256 + i) * 4]); return e; };
            auto last_exit = [&](int n, int g) { unsigned long long e = 0; for (int i = 0; i < g; ++i) e = std::max(e, raw[((size_t) n * 256 + i) * 4 + 2]); return e; };
            double it = 0, bA = 0, bx = 0, cx = 0, ce = 0;
            for (int n = 1; n < NI; ++n) {
                it += (double) (first_entry(3 * n, 32) - first_entry(3 * n - 3, 32)) / 100.0;
                bA += (double) ((long long) first_entry(3 * n + 1, 224) - (long long) first_entry(3 * n, 32)) / 100.0;
                bx += (double) ((long long) last_exit(3 * n + 1, 224) - (long long) last_exit(3 * n, 32)) / 100.0;
                ce += (double) ((long long) first_entry(3 * n + 2, 256) - (long long) last_exit(3 * n, 32)) / 100.0;
                cx += (double) ((long long) last_exit(3 * n + 2, 256) - (long long) last_exit(3 * n, 32)) / 100.0;
            }
            printf("triple %-26s host %7.2f us/iter (enqueue %5.2f)  device %7.2f us/iter; B enters %5.2f us after A's entry, exits %5.2f us after A's end; C enters %5.2f, exits %5.2f us after A's end; give-ups %u\n",
                   co ? "two streams, soft edges" : "one stream, three launches", us / NI, enq / NI, it / (NI - 1), bA / (NI - 1), bx / (NI - 1), ce / (NI - 1), cx / (NI - 1), err[0]);
            fflush(stdout);
        }
    }

    // ---- our own AQL queue ------------------------------------------------------------------------------------------------
    Hsa h; hsa_setup(h, hsaco);
    ChainArgs * karg = nullptr;
    const size_t kstride = std::max<size_t>(256, (h.kernarg_size + 63) & ~63u);
    HK(hsa_amd_memory_pool_allocate(h.kernarg_pool, (size_t) N * kstride, 0, (void **) &karg));
    memset(karg, 0, (size_t) N * kstride);
    HK(hsa_amd_agents_allow_access(1, &h.gpu, nullptr, karg));
    char * karg_dev = nullptr; CK(hipMalloc(&karg_dev, (size_t) N * kstride)); CK(hipMemset(karg_dev, 0, (size_t) N * kstride));
    auto run_aql = [&](const char * name, bool barrier, int acq, int rel, int pre, int nshard, bool devargs = true) {
        reset(b);
        for (int n = 0; n < N; ++n) { ChainArgs a = args_for(b, n, pre, chunks - pre, nshard, slice); memcpy((char *) karg + (size_t) n * kstride, &a, sizeof a); }
        if (devargs) { CK(hipMemcpy(karg_dev, karg, (size_t) N * kstride, hipMemcpyHostToDevice)); CK(hipDeviceSynchronize()); }
        hsa_signal_store_relaxed(h.done, 1);
        const uint32_t mask = h.q->size - 1;
        auto t0 = std::chrono::steady_clock::now();
        const uint64_t base = hsa_queue_add_write_index_relaxed(h.q, (uint64_t) N);
        while (base + N - hsa_queue_load_read_index_scacquire(h.q) > h.q->size) { }
        for (int n = 0; n < N; ++n) {
            hsa_kernel_dispatch_packet_t * p = (hsa_kernel_dispatch_packet_t *) h.q->base_address + ((base + n) & mask);
            p->workgroup_size_x = 512; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
            p->grid_size_x = 256 * 512; p->grid_size_y = 1; p->grid_size_z = 1;
            p->private_segment_size = h.private_size; p->group_segment_size = h.group_size + (uint32_t) lds;
            p->kernel_object = h.kobj; p->kernarg_address = (devargs ? karg_dev : (char *) karg) + (size_t) n * kstride; p->reserved2 = 0;
            p->completion_signal.handle = n == N - 1 ? h.done.handle : 0;
            // the last packet always carries the barrier bit + system release so that the completion signal means "all done"
            const bool bar = barrier || n == N - 1;
            const int a2 = n == N - 1 ? 2 : acq, r2 = n == N - 1 ? 2 : rel;
            const uint16_t header = (uint16_t) ((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((bar ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                                (a2 << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (r2 << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
            const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            __atomic_store_n((uint32_t *) p, (uint32_t) header | ((uint32_t) setup << 16), __ATOMIC_RELEASE);
        }
        hsa_signal_store_screlease(h.q->doorbell_signal, (hsa_signal_value_t) (base + N - 1));
        const hsa_signal_value_t v = hsa_signal_wait_scacquire(h.done, HSA_SIGNAL_CONDITION_LT, 1, 20ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (v >= 1) { printf("%s: TIMED OUT waiting for the completion signal\n", name); exit(2); }
        report(name, b, us, N);
    };
    // ---- the pair experiment on our own queue: first kernel (32 workgroups, emulated latency-bound work) WITH the barrier bit, second (224
    //      workgroups, 9.2 MB of weights, waits on the first one's counter) WITHOUT; the emulated work and the LDS request are varied to see
    //      what the second kernel's entry is tied to ---------------------------------------------------------------------------------------
    auto run_pair = [&](int delay, uint32_t lds_b, bool second_barrier) {
        reset(b);
        const int NP = N / 2;
        for (int n = 0; n < NP; ++n) {
            ChainArgs a = args_for(b, 2 * n, 0, 0, 0, slice); a.ashard = 1; a.delay_ticks = delay; a.nshard = 0;
            ChainArgs w = args_for(b, 2 * n + 1, 5, 0, 1, slice); w.ashard = 0; w.target = 32; w.nshard = second_barrier ? 0 : 1;
            memcpy((char *) karg + (size_t) (2 * n) * kstride, &a, sizeof a); memcpy((char *) karg + (size_t) (2 * n + 1) * kstride, &w, sizeof w);
        }
        CK(hipMemcpy(karg_dev, karg, (size_t) N * kstride, hipMemcpyHostToDevice)); CK(hipDeviceSynchronize());
        hsa_signal_store_relaxed(h.done, 1);
        const uint32_t mask = h.q->size - 1;
        const uint64_t base = hsa_queue_add_write_index_relaxed(h.q, (uint64_t) (2 * NP));
        while (base + 2 * NP - hsa_queue_load_read_index_scacquire(h.q) > h.q->size) { }
        for (int n = 0; n < 2 * NP; ++n) {
            hsa_kernel_dispatch_packet_t * p = (hsa_kernel_dispatch_packet_t *) h.q->base_address + ((base + n) & mask);
            const bool first = (n & 1) == 0, lastp = n == 2 * NP - 1;
            p->workgroup_size_x = 512; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
            p->grid_size_x = (first ? 32 : 224) * 512; p->grid_size_y = 1; p->grid_size_z = 1;
            p->private_segment_size = h.private_size; p->group_segment_size = h.group_size + lds_b;
            p->kernel_object = h.kobj; p->kernarg_address = karg_dev + (size_t) n * kstride; p->reserved2 = 0;
            p->completion_signal.handle = lastp ? h.done.handle : 0;
            const bool bar = first || second_barrier || lastp;
            const int sc = lastp ? 2 : 1;
            const uint16_t header = (uint16_t) ((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((bar ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                                (sc << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (sc << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
            __atomic_store_n((uint32_t *) p, (uint32_t) header | ((uint32_t) (1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16), __ATOMIC_RELEASE);
        }
        hsa_signal_store_screlease(h.q->doorbell_signal, (hsa_signal_value_t) (base + 2 * NP - 1));
        if (hsa_signal_wait_scacquire(h.done, HSA_SIGNAL_CONDITION_LT, 1, 20ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE) >= 1) { printf("pair: TIMED OUT\n"); exit(2); }
        std::vector<unsigned long long> raw((size_t) N * 256 * 4);
        CK(hipMemcpy(raw.data(), b.stamps, raw.size() * 8, hipMemcpyDeviceToHost));
        unsigned err[2]; CK(hipMemcpy(err, b.err, 8, hipMemcpyDeviceToHost));
        auto first_entry = [&](int n, int g) { unsigned long long e = ~0ull; for (int i = 0; i < g; ++i) e = std::min(e, raw[((size_t) n * 256 + i) * 4]); return e; };
        auto last_entry = [&](int n, int g) { unsigned long long e = 0; for (int i = 0; i < g; ++i) e = std::max(e, raw[((size_t) n * 256 + i) * 4]); return e; };
        auto last_exit = [&](int n, int g) { unsigned long long e = 0; for (int i = 0; i < g; ++i) e = std::max(e, raw[((size_t) n * 256 + i) * 4 + 2]); return e; };
        double pair = 0, tailb = 0, e1 = 0, e2 = 0, alen = 0;
        for (int n = 1; n < NP; ++n) {
            pair += (double) (first_entry(2 * n, 32) - first_entry(2 * n - 2, 32)) / 100.0;
            tailb += (double) ((long long) last_exit(2 * n + 1, 224) - (long long) last_exit(2 * n, 32)) / 100.0;
            e1 += (double) ((long long) first_entry(2 * n + 1, 224) - (long long) first_entry(2 * n, 32)) / 100.0;
            e2 += (double) ((long long) last_entry(2 * n + 1, 224) - (long long) first_entry(2 * n, 32)) / 100.0;
            alen += (double) ((long long) last_exit(2 * n, 32) - (long long) first_entry(2 * n, 32)) / 100.0;
        }
        printf("aqlpair delay %4.1f us lds %3u KB %s: %6.2f us/pair; first kernel lasts %5.2f us; second enters %5.2f .. %5.2f us after the first's entry, exits %5.2f us after the first's end; give-ups %u\n",
               delay / 100.0, lds_b >> 10, second_barrier ? "second WITH barrier bit   " : "second WITHOUT barrier bit", pair / (NP - 1), alen / (NP - 1), e1 / (NP - 1), e2 / (NP - 1), tailb / (NP - 1), err[0]);
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run_pair(450, 64 << 10, true); run_pair(450, 64 << 10, false); run_pair(900, 64 << 10, false); run_pair(450, 1 << 10, false); run_pair(900, 1 << 10, false);
    }
    for (int rep = 0; rep < 2; ++rep) {
        run_aql("aqlbar sys/sys hostargs", true, 2, 2, 8, 0, false);
        run_aql("aqlbar sys/sys", true, 2, 2, 8, 0);
        run_aql("aqlbar agent/agent", true, 1, 1, 8, 0);
        run_aql("aqlbar none/none", true, 0, 0, 8, 0);
        run_aql("aqlbar agent/agent pre=16", true, 1, 1, 16, 0);
        run_aql("aqlflag shards=1 pre=8", false, 0, 0, 8, 1);
        run_aql("aqlflag shards=8 pre=8", false, 0, 0, 8, 8);
        run_aql("aqlflag shards=8 pre=16", false, 0, 0, 16, 8);
        run_aql("aqlflag shards=8 pre=0", false, 0, 0, 0, 8);
        run_aql("aqlflag shards=8 pre=8 acq=agent", false, 1, 0, 8, 8);
    }
    return 0;
}
