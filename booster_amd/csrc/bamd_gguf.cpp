// bamd_gguf.cpp — see bamd_gguf.h
#include "bamd_gguf.h"
#include "bamd_formats.h"
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
enum { T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 };

struct Cursor {
    const uint8_t * p; const uint8_t * end; bool ok = true;
    template <typename T> T rd() {
        T v{};
        if (p + sizeof(T) > end) { ok = false; return v; }
        memcpy(&v, p, sizeof(T)); p += sizeof(T); return v;
    }
    size_t left() const { return (size_t) (end - p); }
    std::string str() {
        uint64_t n = rd<uint64_t>();
        if (!ok || n > left()) { ok = false; return std::string(); }
        std::string s((const char *) p, (size_t) n); p += n; return s;
    }
};

size_t scalar_size(uint32_t t) {
    switch (t) {
        case T_U8: case T_I8: case T_BOOL: return 1;
        case T_U16: case T_I16: return 2;
        case T_U32: case T_I32: case T_F32: return 4;
        case T_U64: case T_I64: case T_F64: return 8;
    }
    return 0;
}

bool read_scalar(Cursor & c, uint32_t t, GgufValue & v) {
    switch (t) {
        case T_U8:   v.u = c.rd<uint8_t>();  v.i = (int64_t) v.u; v.f = (double) v.u; break;
        case T_I8:   v.i = c.rd<int8_t>();   v.u = (uint64_t) v.i; v.f = (double) v.i; break;
        case T_U16:  v.u = c.rd<uint16_t>(); v.i = (int64_t) v.u; v.f = (double) v.u; break;
        case T_I16:  v.i = c.rd<int16_t>();  v.u = (uint64_t) v.i; v.f = (double) v.i; break;
        case T_U32:  v.u = c.rd<uint32_t>(); v.i = (int64_t) v.u; v.f = (double) v.u; break;
        case T_I32:  v.i = c.rd<int32_t>();  v.u = (uint64_t) v.i; v.f = (double) v.i; break;
        case T_F32:  v.f = c.rd<float>();    v.i = (int64_t) v.f; v.u = (uint64_t) v.i; break;
        case T_BOOL: v.b = c.rd<uint8_t>() != 0; v.u = v.b; v.i = v.b; break;
        case T_U64:  v.u = c.rd<uint64_t>(); v.i = (int64_t) v.u; v.f = (double) v.u; break;
        case T_I64:  v.i = c.rd<int64_t>();  v.u = (uint64_t) v.i; v.f = (double) v.i; break;
        case T_F64:  v.f = c.rd<double>();   v.i = (int64_t) v.f; v.u = (uint64_t) v.i; break;
        default: return false;
    }
    return c.ok;
}
}  // namespace

GgufFile::~GgufFile() {
    if (map_) munmap((void *) map_, size_);
    if (fd_ >= 0) close(fd_);
}

bool GgufFile::open_one(const std::string & path, std::string & err) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) { err = "cannot open " + path; return false; }
    struct stat st;
    if (fstat(fd_, &st) != 0) { err = "fstat failed"; return false; }
    size_ = (size_t) st.st_size;
    void * m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) { err = "mmap failed"; map_ = nullptr; return false; }
    map_ = (const uint8_t *) m;
    Cursor c{map_, map_ + size_};
    if (size_ < 24 || memcmp(map_, "GGUF", 4) != 0) { err = "not a GGUF file"; return false; }
    c.p += 4;
    version = c.rd<uint32_t>();
    if (version < 2 || version > 3) { err = "unsupported GGUF version"; return false; }
    const uint64_t n_tensors = c.rd<uint64_t>(), n_kv = c.rd<uint64_t>();
    // a corrupt header must come back as an error, not as a bad_alloc / length_error thrown across the C boundary: every count is
    // bounded by the bytes that are left (a KV pair takes >= 12 bytes, a tensor entry >= 24, a string >= 8)
    if (n_kv > c.left() / 12 || n_tensors > c.left() / 24) { err = "corrupt GGUF header (counts exceed the file size)"; return false; }
    for (uint64_t k = 0; k < n_kv && c.ok; ++k) {
        std::string key = c.str();
        GgufValue v; v.type = c.rd<uint32_t>();
        if (v.type == T_STR) v.s = c.str();
        else if (v.type == T_ARR) {
            v.arr_type = c.rd<uint32_t>(); v.arr_n = c.rd<uint64_t>();
            if (v.arr_type == T_STR) {
                if (v.arr_n > c.left() / 8) { err = "bad array length in KV " + key; return false; }
                v.arr_s.reserve((size_t) v.arr_n);
                for (uint64_t i = 0; i < v.arr_n && c.ok; ++i) v.arr_s.push_back(c.str());
            } else {
                const size_t sz = scalar_size(v.arr_type);
                if (!sz || v.arr_n > c.left() / sz) { err = "bad array in KV " + key; return false; }
                v.arr_data = c.p; c.p += sz * v.arr_n;
            }
        } else if (!read_scalar(c, v.type, v)) { err = "bad KV type for " + key; return false; }
        kv[key] = std::move(v);
    }
    if (!c.ok) { err = "truncated GGUF header"; return false; }
    { uint32_t a; if (get_u32("general.alignment", a) && a) alignment = a; }
    tensors.resize((size_t) n_tensors);
    for (auto & t : tensors) {
        t.name = c.str();
        const uint32_t nd = c.rd<uint32_t>();
        if (nd > 4) { err = "bad n_dims"; return false; }
        for (uint32_t d = 0; d < nd; ++d) {
            const uint64_t n = c.rd<uint64_t>();
            if (n == 0 || n > (uint64_t) 1 << 40) { err = "tensor " + t.name + ": bad dimension"; return false; }
            t.ne.push_back((int64_t) n);
        }
        t.type = (int) c.rd<uint32_t>();
        t.offset = c.rd<uint64_t>();
    }
    if (!c.ok) { err = "truncated GGUF tensor table"; return false; }
    const size_t meta = (size_t) (c.p - map_);
    const size_t data_off = (meta + alignment - 1) / alignment * alignment;
    for (size_t i = 0; i < tensors.size(); ++i) {
        auto & t = tensors[i];
        unsigned __int128 rows = 1; for (size_t d = 1; d < t.ne.size(); ++d) rows *= (unsigned __int128) t.ne[d];
        const bool known = t.type == BAMD_F32 || t.type == BAMD_F16 || bamd_is_kquant(t.type);
        const unsigned __int128 nbytes = known ? (unsigned __int128) bamd_row_bytes(t.type, t.ne.empty() ? 0 : t.ne[0]) * rows : 0;
        if (data_off > size_ || t.offset > size_ - data_off || nbytes > (unsigned __int128) (size_ - data_off - t.offset)) { err = "tensor " + t.name + " out of file bounds"; return false; }
        t.nbytes = (size_t) nbytes;
        t.data = map_ + data_off + t.offset;
        index_[t.name] = i;
    }
    return true;
}

// the model file, or the first shard of a split model (llama_model_loader, llama.cpp:3659-3714: the KV pairs of the first shard describe
// the model; every shard carries its own tensor table and data section)
bool GgufFile::open(const std::string & path, std::string & err) {
    if (!open_one(path, err)) return false;
    uint32_t n_split = 0, idx = 0, n_expected = 0;
    if (!get_u32("split.count", n_split) || n_split <= 1) return true;
    get_u32("split.no", idx);
    if (idx != 0) { err = "illegal split file: model must be loaded with the first split"; return false; }
    char postfix[32]; snprintf(postfix, sizeof postfix, "-%05d-of-%05d.gguf", 1, (int) n_split);
    const size_t pl = strlen(postfix);
    if (path.size() <= pl || path.compare(path.size() - pl, pl, postfix) != 0) { err = "invalid split file name: " + path; return false; }
    const std::string prefix = path.substr(0, path.size() - pl);
    for (uint32_t k = 1; k < n_split; ++k) {
        char name[64]; snprintf(name, sizeof name, "-%05d-of-%05d.gguf", (int) k + 1, (int) n_split);
        std::unique_ptr<GgufFile> part(new GgufFile());
        if (!part->open_one(prefix + name, err)) { err = "failed to load GGUF split " + prefix + name + ": " + err; return false; }
        for (const GgufTensor & t : part->tensors) {
            if (index_.count(t.name)) { err = "tensor " + t.name + " appears in two splits"; return false; }
            index_[t.name] = tensors.size();
            tensors.push_back(t);                            // data points into the shard's mapping, kept alive by parts_
        }
        parts_.push_back(std::move(part));
    }
    if (get_u32("split.tensors.count", n_expected) && n_expected != (uint32_t) tensors.size()) {
        err = "corrupted model: " + std::to_string(n_expected) + " tensors expected but " + std::to_string(tensors.size()) + " found"; return false;
    }
    return true;
}

const GgufValue * GgufFile::find(const std::string & key) const { auto it = kv.find(key); return it == kv.end() ? nullptr : &it->second; }
bool GgufFile::get_u32(const std::string & key, uint32_t & v) const {
    const GgufValue * x = find(key); if (!x || x->type == T_STR || x->type == T_ARR) return false; v = (uint32_t) x->u; return true;
}
bool GgufFile::get_f32(const std::string & key, float & v) const {
    const GgufValue * x = find(key); if (!x || x->type == T_STR || x->type == T_ARR) return false; v = (float) x->f; return true;
}
bool GgufFile::get_str(const std::string & key, std::string & v) const {
    const GgufValue * x = find(key); if (!x || x->type != T_STR) return false; v = x->s; return true;
}
const GgufTensor * GgufFile::tensor(const std::string & name) const {
    auto it = index_.find(name); return it == index_.end() ? nullptr : &tensors[it->second];
}
