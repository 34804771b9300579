// bamd_gguf.h — GGUF v2/v3 reader (mmap).  Same on-disk format the reference reads in
// cpp/ggml/src/ggml.c:20896-21260 (gguf_init_from_file): magic "GGUF", u32 version, u64 n_tensors, u64 n_kv,
// KV pairs {string key, u32 type, value}, tensor infos {string name, u32 n_dims, u64 ne[], u32 type, u64 offset},
// data section aligned to general.alignment (default 32).  Split models (gguf-split: <prefix>-00001-of-0000N.gguf, split.count / split.no /
// split.tensors.count, llama.cpp:3659-3714): open the FIRST shard; the others are mapped alongside and their tensors join the table.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <memory>

struct GgufValue {
    uint32_t type = 0;                 // gguf_type
    uint64_t u = 0; int64_t i = 0; double f = 0; bool b = false;
    std::string s;
    uint32_t arr_type = 0;
    std::vector<std::string> arr_s;    // string arrays
    const uint8_t * arr_data = nullptr; uint64_t arr_n = 0;   // numeric arrays: pointer into the mapping
};

struct GgufTensor {
    std::string name;
    std::vector<int64_t> ne;           // ggml order: ne[0] = row length
    int type = 0;
    uint64_t offset = 0;               // relative to data section
    const uint8_t * data = nullptr;
    size_t nbytes = 0;
};

class GgufFile {
  public:
    ~GgufFile();
    bool open(const std::string & path, std::string & err);
    const GgufValue * find(const std::string & key) const;
    bool get_u32(const std::string & key, uint32_t & v) const;
    bool get_f32(const std::string & key, float & v) const;
    bool get_str(const std::string & key, std::string & v) const;
    const GgufTensor * tensor(const std::string & name) const;

    std::map<std::string, GgufValue> kv;
    std::vector<GgufTensor> tensors;
    uint32_t version = 0;
    size_t alignment = 32;

  private:
    bool open_one(const std::string & path, std::string & err);
    std::vector<std::unique_ptr<GgufFile>> parts_;   // shards 2..n of a split model (split.count > 1): their tensors are merged into `tensors`
    int fd_ = -1;
    const uint8_t * map_ = nullptr;
    size_t size_ = 0;
    std::map<std::string, size_t> index_;
};
