// bamd_vocab.h — tokenizer side of the bridge: GGUF vocabulary, SPM and byte-level BPE tokenisers,
// token -> piece, end-of-generation test.  CPU-only host code; restates the behaviour of cpp/src/llama-vocab.cpp
// (llm_tokenizer_spm :190-330, llm_tokenizer_bpe :340-560, tokenizer_st_partition :1119-1240, llama_tokenize_internal
// :1243-1345, llama_token_to_piece_impl :1539-1608) and the vocab loading of cpp/src/llama.cpp:5250-5760.
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

class GgufFile;

enum { BAMD_VOCAB_NONE = 0, BAMD_VOCAB_SPM = 1, BAMD_VOCAB_BPE = 2 };
enum {   // llama_token_attr (llama.h)
    BAMD_ATTR_UNKNOWN = 1 << 0, BAMD_ATTR_UNUSED = 1 << 1, BAMD_ATTR_NORMAL = 1 << 2, BAMD_ATTR_CONTROL = 1 << 3,
    BAMD_ATTR_USER_DEFINED = 1 << 4, BAMD_ATTR_BYTE = 1 << 5,
};

struct BamdVocab {
    int type = BAMD_VOCAB_NONE;
    bool pre_llama3 = false;                 // tokenizer.ggml.pre in {llama3, llama-v3, llama-bpe} and the others with that regex
    int pre_chain = 0;                       // 0: the single llama-3 / GPT-2 regex; else a regex chain around the GPT-2 regex (llama-vocab.cpp:379-443):
                                             // 1 = {\p{N}} + GPT-2 (starcoder, refact, command-r, smollm, codeshell); 2 = default ({[\p{P}$+<=>^~|]+} + GPT-2 + {\p{N}+});
                                             // 3 = falcon ({[\p{P}$+<=>^~|`]+} + GPT-2 + {[0-9][0-9][0-9]})
    int pre_maxdigits = 3;                   // llama-3 regex: \p{N}{1,3}; qwen2 / stablelm2: \p{N}
    bool ignore_merges = false;
    bool add_space_prefix = true;
    bool add_bos = false, add_eos = false;
    int bos = -1, eos = -1, eot = -1, unk = -1;
    std::vector<std::string> text;
    std::vector<float> score;
    std::vector<int> attr;
    std::unordered_map<std::string, int> token_to_id;
    std::map<std::pair<std::string, std::string>, int> bpe_ranks;
    std::vector<int> special;                // CONTROL | USER_DEFINED | UNKNOWN ids, longest text first
    std::vector<std::string> piece;          // token -> piece with special = true (llama.cpp:5696-5710)

    bool load(const GgufFile & g, std::string & err);
    std::vector<int> tokenize(const std::string & text, bool add_special, bool parse_special) const;
    const std::string & token_to_piece(int id) const;
    bool is_eog(int id) const { return id != -1 && (id == eos || id == eot); }
    int n_vocab() const { return (int) text.size(); }
};
