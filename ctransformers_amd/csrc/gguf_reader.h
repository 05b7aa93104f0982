// GGUF v1/v2/v3 reader (mmap, zero-copy tensor views).  Restates the container format the reference parses in
// models/ggml/ggml.c:19561-19880 (gguf_init_from_file): header :19473-19493, KV pairs :19626-19707 (value type ids
// ggml.h:1830-1845), tensor infos :19709-19742, data section aligned to general.alignment (default 32) :19744-19759.
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

namespace ctamd {

enum GgufValueType : uint32_t {
    GV_U8 = 0, GV_I8, GV_U16, GV_I16, GV_U32, GV_I32, GV_F32, GV_BOOL, GV_STR, GV_ARR, GV_U64, GV_I64, GV_F64,
};

struct GgufValue {
    uint32_t type = 0;
    uint32_t elem_type = 0;     // for arrays
    uint64_t u = 0;             // integer / bool payload
    double f = 0.0;             // float payload
    std::string s;              // string payload
    uint64_t n = 0;             // array length
    const uint8_t* arr = nullptr;       // start of packed array payload (non-string arrays)
    std::vector<std::string> strs;      // string arrays
};

struct GgufTensor {
    std::string name;
    int n_dims = 0;
    int64_t ne[4] = {1, 1, 1, 1};
    int type = 0;
    uint64_t offset = 0;
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
};

class GgufFile {
   public:
    ~GgufFile();
    // returns false (and sets error()) on malformed input
    bool open(const std::string& path);
    const std::string& error() const { return err_; }
    uint32_t version() const { return version_; }

    const GgufValue* find(const std::string& key) const;
    bool get_u32(const std::string& key, uint32_t& out) const;
    bool get_f32(const std::string& key, float& out) const;
    bool get_str(const std::string& key, std::string& out) const;
    const GgufTensor* tensor(const std::string& name) const;
    const std::vector<GgufTensor>& tensors() const { return tensors_; }
    // the open file and its mapping: the load pipeline reads tensor bytes with pread() into pinned staging (engine.cc:stage_file)
    int fd() const { return fd_; }
    const uint8_t* map_base() const { return map_; }
    size_t map_size() const { return size_; }

   private:
    bool fail(const std::string& m) { err_ = m; return false; }
    std::string err_;
    int fd_ = -1;
    const uint8_t* map_ = nullptr;
    size_t size_ = 0;
    uint32_t version_ = 0;
    std::map<std::string, GgufValue> kv_;
    std::vector<GgufTensor> tensors_;
    std::map<std::string, size_t> tindex_;
};

// Pre-GGUF GGML container of the reference's legacy loaders (gpt2: models/llms/gpt2.cc:61-381): magic 0x67676d6c, six i32
// hparams, vocabulary (i32 count; u32 length + bytes each), then tensors until EOF: i32 n_dims, i32 name_len, i32 type,
// i32 dims, name, data (no alignment).
class LegacyGgmlFile {
   public:
    ~LegacyGgmlFile();
    // mpt: the MPT header (d_model, max_seq_len, n_heads, n_layers, n_vocab, alibi_bias_max, clip_qkv, ftype; no vocabulary count;
    // pieces converted from UTF-8 to one byte per code point — reference mpt_model_load, models/llms/mpt.cc:70-112)
    bool open(const std::string& path, bool mpt = false);
    const std::string& error() const { return err_; }
    int32_t hparams[6] = {0, 0, 0, 0, 0, 0};   // n_vocab, n_ctx, n_embd, n_head, n_layer, ftype (quantization version stripped)
    float alibi_bias_max = 0.0f, clip_qkv = 0.0f;   // MPT only
    std::vector<std::string> vocab;
    const GgufTensor* tensor(const std::string& name) const;

   private:
    bool fail(const std::string& m) { err_ = m; return false; }
    std::string err_;
    int fd_ = -1;
    const uint8_t* map_ = nullptr;
    size_t size_ = 0;
    std::vector<GgufTensor> tensors_;
    std::map<std::string, size_t> tindex_;
};

}  // namespace ctamd
