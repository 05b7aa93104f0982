// Host-side text plumbing behind the C ABI: vocabulary, SentencePiece-style tokenizer, detokenizer, sampler chain.
// Behavioural restatement of reference models/ggml/llama.cpp: llm_load_vocab :1648-1760, llm_tokenizer_spm
// :3080-3210 (+ byte fallback :3030-3042, whitespace escape :3044-3050), llama_tokenize_internal :3390-3427,
// llama_token_to_piece :6151-6187, samplers :3805-3889, :4013-4052, :4281-4302 and the chain order of
// models/llms/llama.cc:53-84.  All of it runs on the CPU in the reference too (O(n_vocab) per token).
#pragma once
#include <stdint.h>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

namespace ctamd {

class GgufFile;

enum VocabType { VOCAB_SPM = 0, VOCAB_BPE = 1, VOCAB_GPT = 2 };   // GPT: legacy models (models/common.h gpt_vocab / gpt_tokenize)
enum TokenType { TT_UNDEFINED = 0, TT_NORMAL = 1, TT_UNKNOWN = 2, TT_CONTROL = 3, TT_USER = 4, TT_UNUSED = 5, TT_BYTE = 6 };

struct Vocab {
    VocabType type = VOCAB_SPM;
    std::vector<std::string> text;
    std::vector<float> score;
    std::vector<int> ttype;
    std::unordered_map<std::string, int> to_id;
    std::unordered_map<std::string, int> bpe_rank;   // "left\x01right" -> merge rank (BPE vocabularies)
    int bos_id = 1, eos_id = 2, unk_id = 0;
    // legacy StarCoder files: pieces the text is split at before the word regex (gpt_vocab::special_tokens, models/common.h:35-39)
    std::vector<std::string> special;

    bool load(const GgufFile& f, std::string& err);
    // legacy GGML files: raw pieces, eos = bos = id of "<|endoftext|>" or 0 (models/llm.h:104-110)
    void load_legacy(const std::vector<std::string>& pieces);
    // the StarChat / fill-in-the-middle markers the reference registers for starcoder when the file's vocabulary holds them
    // (starcoder_model_load, models/llms/starcoder.cc:123-138)
    void mark_starcoder_specials();
    int size() const { return (int)text.size(); }
    std::vector<int> tokenize(const std::string& text, bool add_bos) const;
    std::string piece(int token) const;
};

// repetition penalty -> top-k -> top-p -> temperature -> multinomial draw with std::mt19937(seed)
int sample_token(const float* logits, int n_vocab, const int* last_tokens, int n_last, int top_k, float top_p,
                 float temperature, float repetition_penalty, int seed);

// The legacy models' sampler (reference models/common.h:127-205 gpt_sample_top_k_top_p, called from LLM::Sample
// models/llm.h:74-90): temperature scale, repetition penalty over the SET of recent tokens, partial sort of the top k,
// double-precision softmax, top-p cut, std::mt19937(seed) + std::discrete_distribution.
int sample_token_gpt(const float* logits, int n_vocab, const int* last_tokens, int n_last, int top_k, float top_p,
                     float temperature, float repetition_penalty, int seed);

}  // namespace ctamd
