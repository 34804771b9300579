// bamd_aql.cpp — host side of the own-queue replay of a decode step (see bamd_aql.h): HSA agent / queue / code objects per device, recorded launches ->
// AQL packets, submission and completion.  No reference counterpart below the level named in bamd_aql.h (the reference replays a CUDA graph per token).
#include "bamd_aql.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <map>
#include <mutex>
#include <string>

thread_local bamd_aql_recording * bamd_aql_rec = nullptr;

// the decode kernels' code objects (gfx950 ELF images), embedded by booster_amd/build.py: bamd_hsaco_blob.S defines the table
struct bamd_hsaco_entry { const unsigned char * begin, * end; const char * name; };
extern "C" const bamd_hsaco_entry bamd_hsaco_table[];
extern "C" const int bamd_hsaco_count;

namespace {

thread_local std::string t_why;
const char * why_(const std::string & s) { t_why = s; return t_why.c_str(); }
std::string hsa_err(const char * what, hsa_status_t st) { const char * m = ""; hsa_status_string(st, &m); return std::string(what) + ": " + (m ? m : "?"); }

struct Kernel { uint64_t object = 0; uint32_t kernarg_size = 0, group_size = 0, private_size = 0; };

struct Device {
    bool tried = false, ok = false; std::string why;
    hsa_agent_t agent{}; hsa_queue_t * q = nullptr; hsa_signal_t done{};
    std::vector<hsa_executable_t> exes;
    std::map<std::string, Kernel> kernels;          // by code-object symbol name (mangled kernel name + ".kd"), filled on demand
    uint64_t widx = 0;                              // packets written so far (this library is the queue's only producer)
    std::mutex mu;                                  // one replay at a time per device
};
Device g_dev[16];
std::mutex g_mu;

struct AgentPick { std::vector<hsa_agent_t> gpus; };
hsa_status_t agent_cb(hsa_agent_t ag, void * data) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(ag, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_GPU) ((AgentPick *) data)->gpus.push_back(ag);
    return HSA_STATUS_SUCCESS;
}

// the HSA agent of HIP device `device`: matched by PCI domain / bus / device / function; enumeration order when the match is not unique
bool find_agent(int device, hsa_agent_t & out, std::string & why) {
    AgentPick pick;
    hsa_status_t st = hsa_iterate_agents(agent_cb, &pick);
    if (st != HSA_STATUS_SUCCESS || pick.gpus.empty()) { why = "no HSA GPU agent"; return false; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { why = "hipGetDeviceProperties failed"; return false; }
    int hits = 0;
    for (hsa_agent_t ag : pick.gpus) {
        uint32_t bdf = 0, dom = 0;
        if (hsa_agent_get_info(ag, (hsa_agent_info_t) HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) continue;
        hsa_agent_get_info(ag, (hsa_agent_info_t) HSA_AMD_AGENT_INFO_DOMAIN, &dom);
        if ((int) ((bdf >> 8) & 0xff) == prop.pciBusID && (int) ((bdf >> 3) & 0x1f) == prop.pciDeviceID && (int) dom == prop.pciDomainID) { out = ag; ++hits; }
    }
    if (hits == 1) return true;
    if (device < (int) pick.gpus.size()) { out = pick.gpus[(size_t) device]; return true; }
    why = "no HSA agent matches HIP device " + std::to_string(device);
    return false;
}

Device * device_get(int device) {
    if (device < 0 || device >= 16) return nullptr;
    Device & d = g_dev[device];
    std::lock_guard<std::mutex> lk(g_mu);
    if (d.tried) return &d;
    d.tried = true;
    if (const char * e = getenv("BAMD_AQL")) if (e[0] == '0') { d.why = "switched off (BAMD_AQL=0)"; return &d; }
    hsa_status_t st = hsa_init();                    // reference-counted: the HIP runtime holds its own reference
    if (st != HSA_STATUS_SUCCESS) { d.why = hsa_err("hsa_init", st); return &d; }
    if (!find_agent(device, d.agent, d.why)) return &d;
    uint32_t qmax = 0;
    hsa_agent_get_info(d.agent, HSA_AGENT_INFO_QUEUE_MAX_SIZE, &qmax);
    uint32_t qsize = 16384;
    while (qmax && qsize > qmax) qsize >>= 1;
    st = hsa_queue_create(d.agent, qsize, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &d.q);
    if (st != HSA_STATUS_SUCCESS) { d.why = hsa_err("hsa_queue_create", st); return &d; }
    st = hsa_signal_create(1, 0, nullptr, &d.done);
    if (st != HSA_STATUS_SUCCESS) { d.why = hsa_err("hsa_signal_create", st); return &d; }
    for (int i = 0; i < bamd_hsaco_count; ++i) {
        const bamd_hsaco_entry & e = bamd_hsaco_table[i];
        hsa_code_object_reader_t rd; hsa_executable_t ex;
        st = hsa_code_object_reader_create_from_memory(e.begin, (size_t) (e.end - e.begin), &rd);
        if (st != HSA_STATUS_SUCCESS) { d.why = hsa_err(e.name, st); return &d; }
        st = hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex);
        if (st == HSA_STATUS_SUCCESS) st = hsa_executable_load_agent_code_object(ex, d.agent, rd, nullptr, nullptr);
        if (st == HSA_STATUS_SUCCESS) st = hsa_executable_freeze(ex, nullptr);
        if (st != HSA_STATUS_SUCCESS) { d.why = hsa_err((std::string("loading code object ") + e.name).c_str(), st); return &d; }
        d.exes.push_back(ex);
    }
    if (d.exes.empty()) { d.why = "no embedded code objects"; return &d; }
    d.widx = hsa_queue_load_write_index_relaxed(d.q);
    d.ok = true;
    return &d;
}

bool kernel_lookup(Device & d, const std::string & name, Kernel & k) {
    auto it = d.kernels.find(name);
    if (it != d.kernels.end()) { k = it->second; return true; }
    const std::string sym = name + ".kd";
    for (hsa_executable_t ex : d.exes) {
        hsa_executable_symbol_t s;
        if (hsa_executable_get_symbol_by_name(ex, sym.c_str(), &d.agent, &s) != HSA_STATUS_SUCCESS) continue;
        Kernel kk;
        if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kk.object) != HSA_STATUS_SUCCESS) continue;
        hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kk.kernarg_size);
        hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &kk.group_size);
        hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &kk.private_size);
        d.kernels[name] = kk; k = kk;
        return true;
    }
    return false;
}

// the hidden kernel arguments of code object v5 / v6 behind the explicit ones (LLVM AMDGPUUsage, "Code object V5 implicit kernel arguments"): block counts,
// group sizes, remainders, global offsets, grid dimensions; the rest (printf / hostcall buffers, heap, queue pointers) stays zero — the decode kernels use none
void fill_hidden(uint8_t * ka, size_t explicit_bytes, size_t total, const bamd_aql_launch & l, uint32_t dyn_lds) {
    const size_t h = (explicit_bytes + 7) & ~(size_t) 7;
    auto put = [&](size_t off, const void * v, size_t n) { if (h + off + n <= total) memcpy(ka + h + off, v, n); };
    for (int i = 0; i < 3; ++i) { const uint32_t bc = l.grid[i]; put(4 * i, &bc, 4); }
    for (int i = 0; i < 3; ++i) { const uint16_t gs = (uint16_t) l.block[i]; put(12 + 2 * i, &gs, 2); }
    for (int i = 0; i < 3; ++i) { const uint16_t rm = 0; put(18 + 2 * i, &rm, 2); }
    for (int i = 0; i < 3; ++i) { const uint64_t go = 0; put(40 + 8 * i, &go, 8); }
    const uint16_t dims = l.grid[2] > 1 ? 3 : l.grid[1] > 1 ? 2 : 1; put(64, &dims, 2);
    put(120, &dyn_lds, 4);
}

}  // namespace

struct bamd_aql_graph {
    int device = 0;
    std::vector<hsa_kernel_dispatch_packet_t> packets;   // headers filled at submission
    void * kernarg_dev = nullptr;                        // device memory (kernel arguments in host memory cost 24 us per launch: profiles/r03_aql_probe.txt, `hostargs`)
};

int bamd_aql_available(int device, const char ** why) {
    Device * d = device_get(device);
    if (!d) { if (why) *why = "bad device index"; return 0; }
    if (!d->ok && why) *why = d->why.c_str();
    return d->ok ? 1 : 0;
}

bamd_aql_graph * bamd_aql_build(int device, const bamd_aql_recording & rec, const char ** why) {
    const char * w = nullptr;
    if (!bamd_aql_available(device, &w)) { if (why) *why = w; return nullptr; }
    Device & d = g_dev[device];
    if (rec.launches.empty()) { if (why) *why = "empty recording"; return nullptr; }
    std::lock_guard<std::mutex> lk(d.mu);
    std::vector<Kernel> ks(rec.launches.size());
    std::vector<size_t> off(rec.launches.size());
    size_t total = 0;
    for (size_t i = 0; i < rec.launches.size(); ++i) {
        const bamd_aql_launch & l = rec.launches[i];
        const char * name = hipKernelNameRefByPtr(l.host_fn, nullptr);
        if (!name || !*name) { if (why) *why = why_("a recorded kernel has no registered name (launch " + std::to_string(i) + ")"); return nullptr; }
        if (!kernel_lookup(d, name, ks[i])) { if (why) *why = why_(std::string("kernel not in the embedded code objects: ") + name); return nullptr; }
        if (ks[i].kernarg_size < l.kernarg.size()) { if (why) *why = why_(std::string("kernel argument block larger than the code object's segment: ") + name); return nullptr; }
        off[i] = total;
        total += ((size_t) ks[i].kernarg_size + 255) & ~(size_t) 255;
    }
    std::vector<uint8_t> host(total, 0);
    bamd_aql_graph * g = new bamd_aql_graph();
    g->device = device;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&g->kernarg_dev, total) != hipSuccess) { (void) hipGetLastError(); if (why) *why = "kernel argument memory"; delete g; return nullptr; }
    g->packets.resize(rec.launches.size());
    for (size_t i = 0; i < rec.launches.size(); ++i) {
        const bamd_aql_launch & l = rec.launches[i];
        uint8_t * ka = host.data() + off[i];
        memcpy(ka, l.kernarg.data(), l.kernarg.size());
        fill_hidden(ka, l.kernarg.size(), ks[i].kernarg_size, l, l.lds_bytes);
        hsa_kernel_dispatch_packet_t & p = g->packets[i];
        memset(&p, 0, sizeof p);
        p.workgroup_size_x = (uint16_t) l.block[0]; p.workgroup_size_y = (uint16_t) l.block[1]; p.workgroup_size_z = (uint16_t) l.block[2];
        p.grid_size_x = l.grid[0] * l.block[0]; p.grid_size_y = l.grid[1] * l.block[1]; p.grid_size_z = l.grid[2] * l.block[2];
        p.private_segment_size = ks[i].private_size; p.group_segment_size = ks[i].group_size + l.lds_bytes;
        p.kernel_object = ks[i].object;
        p.kernarg_address = (uint8_t *) g->kernarg_dev + off[i];
    }
    if (hipMemcpy(g->kernarg_dev, host.data(), total, hipMemcpyHostToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void) hipGetLastError(); if (why) *why = "kernel argument upload"; bamd_aql_free(g); return nullptr;
    }
    return g;
}

int bamd_aql_graph_launches(const bamd_aql_graph * g) { return g ? (int) g->packets.size() : 0; }

void bamd_aql_free(bamd_aql_graph * g) {
    if (!g) return;
    if (g->kernarg_dev) { hipSetDevice(g->device); hipFree(g->kernarg_dev); }
    delete g;
}

// fence scopes of the packets between the first and the last of a run: 0 = none (default), 1 = agent (what a HIP stream issues: BAMD_AQL_SCOPE=agent, the A/B)
static int inner_scope() {
    static const int v = [] { const char * e = getenv("BAMD_AQL_SCOPE"); return e && (e[0] == 'a' || e[0] == '1') ? 1 : e && (e[0] == 's' || e[0] == '2') ? 2 : 0; }();
    return v;
}

int bamd_aql_run(bamd_aql_graph * g, int replays, double * seconds, const char ** why) {
    if (!g || replays < 1) { if (why) *why = "nothing to run"; return 1; }
    Device & d = g_dev[g->device];
    std::lock_guard<std::mutex> lk(d.mu);
    hsa_queue_t * q = d.q;
    const uint32_t mask = q->size - 1;
    const size_t P = g->packets.size();
    if (P > q->size / 2) { if (why) *why = "launch sequence longer than half the queue"; return 1; }
    const int sc = inner_scope();
    const uint16_t setup = 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    auto header = [&](int acq, int rel) {
        return (uint16_t) ((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                           (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    };
    hsa_signal_store_relaxed(d.done, 1);
    const auto t0 = std::chrono::steady_clock::now();
    const auto deadline = t0 + std::chrono::seconds(60);
    for (int r = 0; r < replays; ++r) {
        // room for one replay: the ring holds q->size packets, the packet processor has consumed everything below the read index
        while (d.widx + P - hsa_queue_load_read_index_scacquire(q) > q->size) {
            if (std::chrono::steady_clock::now() > deadline) { if (why) *why = "the queue stopped consuming packets"; return 1; }
        }
        hsa_kernel_dispatch_packet_t * ring = (hsa_kernel_dispatch_packet_t *) q->base_address;
        for (size_t i = 0; i < P; ++i) {
            hsa_kernel_dispatch_packet_t * p = ring + ((d.widx + i) & mask);
            const hsa_kernel_dispatch_packet_t & src = g->packets[i];
            const bool first = r == 0 && i == 0, last = r == replays - 1 && i == P - 1;
            // body first (everything behind the 4 bytes of header + setup), the header last with release semantics: the packet processor may look at the slot at once
            memcpy((char *) p + 4, (const char *) &src + 4, sizeof src - 4);
            p->completion_signal.handle = last ? d.done.handle : 0;
            const uint16_t h = header(first ? HSA_FENCE_SCOPE_SYSTEM : sc, last ? HSA_FENCE_SCOPE_SYSTEM : sc);
            __atomic_store_n((uint32_t *) p, (uint32_t) h | ((uint32_t) setup << 16), __ATOMIC_RELEASE);
            // one doorbell never covers packets on both sides of the ring's wrap: under rocprofv3 the queue is the tool's intercept queue, whose handler is
            // given the run of new packets as ONE pointer + count and reads past the end of the ring (SIGSEGV in librocprofiler-sdk at the 1 MiB boundary,
            // reproduced with tools/aql_under_profiler.py); the hardware queue does not care, and it is one more doorbell per 16384 packets
            if (((d.widx + i) & mask) == mask && i + 1 < P) {
                hsa_queue_store_write_index_screlease(q, d.widx + i + 1);
                hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t) (d.widx + i));
            }
        }
        d.widx += P;
        hsa_queue_store_write_index_screlease(q, d.widx);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t) (d.widx - 1));
    }
    const hsa_signal_value_t v = hsa_signal_wait_scacquire(d.done, HSA_SIGNAL_CONDITION_LT, 1, 60ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (v >= 1) { if (why) *why = "timed out waiting for the completion signal of the own queue"; return 1; }
    return 0;
}
