#include "gguf_reader.h"

#include <codecvt>
#include <locale>

#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "quant.h"

namespace ctamd {

namespace {
struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    bool v1 = false;
    size_t left() const { return (size_t)(end - p); }   // sizes are compared against this, never by pointer addition (which can wrap)
    template <class T> T get() {
        T v{};
        if (sizeof(T) > left()) { ok = false; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    uint64_t len() { return v1 ? (uint64_t)get<uint32_t>() : get<uint64_t>(); }
    std::string str() {
        const uint64_t n = len();
        if (!ok || n > left()) { ok = false; return std::string(); }
        std::string s((const char*)p, (size_t)n);
        p += n;
        return s;
    }
};
size_t scalar_size(uint32_t t) {
    switch (t) {
        case GV_U8: case GV_I8: case GV_BOOL: return 1;
        case GV_U16: case GV_I16: return 2;
        case GV_U32: case GV_I32: case GV_F32: return 4;
        case GV_U64: case GV_I64: case GV_F64: return 8;
        default: return 0;
    }
}
}  // namespace

GgufFile::~GgufFile() {
    if (map_) munmap((void*)map_, size_);
    if (fd_ >= 0) close(fd_);
}

bool GgufFile::open(const std::string& path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) return fail("cannot open " + path);
    struct stat st;
    if (fstat(fd_, &st) != 0) return fail("cannot stat " + path);
    size_ = (size_t)st.st_size;
    if (size_ < 16) return fail("file too small");
    void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) return fail("mmap failed");
    map_ = (const uint8_t*)m;
    Cursor c{map_, map_ + size_};
    if (c.get<uint32_t>() != 0x46554747u) return fail("bad magic (not GGUF)");
    version_ = c.get<uint32_t>();
    if (version_ < 1 || version_ > 3) return fail("unsupported GGUF version");
    c.v1 = version_ == 1;
    const uint64_t n_tensors = c.len();
    const uint64_t n_kv = c.len();
    // every tensor entry / key-value pair takes at least a length field: counts beyond the file size are corrupt headers
    if (!c.ok || n_tensors > size_ / 8 || n_kv > size_ / 8) return fail("corrupt header (tensor / key counts exceed the file size)");
    for (uint64_t i = 0; i < n_kv && c.ok; ++i) {
        std::string key = c.str();
        GgufValue v;
        v.type = c.get<uint32_t>();
        switch (v.type) {
            case GV_U8: v.u = c.get<uint8_t>(); break;
            case GV_I8: v.u = (uint64_t)(int64_t)c.get<int8_t>(); break;
            case GV_U16: v.u = c.get<uint16_t>(); break;
            case GV_I16: v.u = (uint64_t)(int64_t)c.get<int16_t>(); break;
            case GV_U32: v.u = c.get<uint32_t>(); break;
            case GV_I32: v.u = (uint64_t)(int64_t)c.get<int32_t>(); break;
            case GV_U64: v.u = c.get<uint64_t>(); break;
            case GV_I64: v.u = (uint64_t)c.get<int64_t>(); break;
            case GV_BOOL: v.u = c.get<uint8_t>(); break;
            case GV_F32: v.f = c.get<float>(); break;
            case GV_F64: v.f = c.get<double>(); break;
            case GV_STR: v.s = c.str(); break;
            case GV_ARR: {
                v.elem_type = c.get<uint32_t>();
                v.n = c.len();
                if (!c.ok || v.n > c.left()) return fail("corrupt array length for key " + key);
                if (v.elem_type == GV_STR) {
                    v.strs.reserve((size_t)v.n);
                    for (uint64_t k = 0; k < v.n && c.ok; ++k) v.strs.push_back(c.str());
                } else {
                    const size_t es = scalar_size(v.elem_type);
                    if (es == 0) return fail("bad array element type for key " + key);
                    v.arr = c.p;
                    if (v.n > c.left() / es) return fail("truncated array " + key);
                    c.p += es * v.n;
                }
                break;
            }
            default: return fail("bad value type for key " + key);
        }
        kv_[key] = std::move(v);
    }
    if (!c.ok) return fail("truncated metadata");
    tensors_.resize((size_t)n_tensors);
    for (uint64_t i = 0; i < n_tensors && c.ok; ++i) {
        GgufTensor& t = tensors_[(size_t)i];
        t.name = c.str();
        t.n_dims = (int)c.get<uint32_t>();
        if (t.n_dims < 1 || t.n_dims > 4) return fail("bad n_dims for tensor " + t.name);
        size_t n_rows = 1;
        for (int d = 0; d < t.n_dims; ++d) {
            const uint64_t ne = c.len();
            if (ne == 0 || ne > (1ull << 40)) return fail("bad dimension in tensor " + t.name);
            t.ne[d] = (int64_t)ne;
            if (d > 0) {
                if (n_rows > size_ / (size_t)ne + 1) return fail("tensor larger than the file: " + t.name);
                n_rows *= (size_t)ne;
            }
        }
        t.type = (int)c.get<uint32_t>();
        t.offset = c.get<uint64_t>();
        if (ggml_block_elems(t.type) == 0) return fail("unsupported tensor type in " + t.name);
        if (t.ne[0] % ggml_block_elems(t.type) != 0) return fail("row length not a multiple of the block size: " + t.name);
        if (ggml_row_bytes(t.type, t.ne[0]) != 0 && n_rows > size_ / ggml_row_bytes(t.type, t.ne[0]) + 1) return fail("tensor larger than the file: " + t.name);
        t.nbytes = ggml_row_bytes(t.type, t.ne[0]) * n_rows;
        tindex_[t.name] = (size_t)i;
    }
    if (!c.ok) return fail("truncated tensor table");
    uint32_t align = 32;
    get_u32("general.alignment", align);
    if (align == 0 || (align & (align - 1)) != 0) return fail("general.alignment must be a power of two");
    size_t base = (size_t)(c.p - map_);
    base = (base + align - 1) / align * align;
    for (auto& t : tensors_) {
        if (base > size_ || t.offset > size_ - base || t.nbytes > size_ - base - t.offset) return fail("tensor data out of file bounds: " + t.name);
        t.data = map_ + base + t.offset;
    }
    return true;
}

const GgufValue* GgufFile::find(const std::string& key) const {
    auto it = kv_.find(key);
    return it == kv_.end() ? nullptr : &it->second;
}
bool GgufFile::get_u32(const std::string& key, uint32_t& out) const {
    const GgufValue* v = find(key);
    if (!v || v->type == GV_STR || v->type == GV_ARR || v->type == GV_F32 || v->type == GV_F64) return false;
    out = (uint32_t)v->u;
    return true;
}
bool GgufFile::get_f32(const std::string& key, float& out) const {
    const GgufValue* v = find(key);
    if (!v || (v->type != GV_F32 && v->type != GV_F64)) return false;
    out = (float)v->f;
    return true;
}
bool GgufFile::get_str(const std::string& key, std::string& out) const {
    const GgufValue* v = find(key);
    if (!v || v->type != GV_STR) return false;
    out = v->s;
    return true;
}
const GgufTensor* GgufFile::tensor(const std::string& name) const {
    auto it = tindex_.find(name);
    return it == tindex_.end() ? nullptr : &tensors_[it->second];
}

LegacyGgmlFile::~LegacyGgmlFile() {
    if (map_) munmap((void*)map_, size_);
    if (fd_ >= 0) close(fd_);
}

bool LegacyGgmlFile::open(const std::string& path, bool mpt) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) return fail("cannot open " + path);
    struct stat st;
    if (fstat(fd_, &st) != 0) return fail("fstat failed");
    size_ = (size_t)st.st_size;
    void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) return fail("mmap failed");
    map_ = (const uint8_t*)m;
    Cursor c{map_, map_ + size_};
    if (c.get<uint32_t>() != 0x67676d6cu) return fail("not a legacy GGML file (bad magic)");
    int32_t nv = 0;
    if (mpt) {
        const int32_t d_model = c.get<int32_t>(), max_seq_len = c.get<int32_t>(), n_heads = c.get<int32_t>(), n_layers = c.get<int32_t>();
        nv = c.get<int32_t>();
        alibi_bias_max = c.get<float>();
        clip_qkv = c.get<float>();
        const int32_t ftype = c.get<int32_t>();
        const int32_t h[6] = {nv, max_seq_len, d_model, n_heads, n_layers, ftype % 1000};
        for (int i = 0; i < 6; ++i) hparams[i] = h[i];
    } else {
        for (int i = 0; i < 6; ++i) hparams[i] = c.get<int32_t>();
        hparams[5] %= 1000;   // GGML_QNT_VERSION_FACTOR (gpt2.cc:88-90)
        nv = c.get<int32_t>();
    }
    if (!c.ok || nv != hparams[0] || nv <= 0 || (size_t)nv > size_ / 4) return fail("bad vocabulary size");
    vocab.reserve((size_t)nv);
    for (int i = 0; i < nv; ++i) {
        const uint32_t len = c.get<uint32_t>();
        if (!c.ok || len > c.left()) return fail("truncated vocabulary");
        vocab.emplace_back((const char*)c.p, (size_t)len);
        c.p += len;
        if (mpt) {   // mpt.cc:101-107: UTF-8 -> code points -> the low byte of each
            std::string& w = vocab.back();
            try {
                std::wstring_convert<std::codecvt_utf8<wchar_t>> conv;
                const std::wstring wide = conv.from_bytes(w);
                w.resize(wide.size());
                for (size_t k = 0; k < wide.size(); ++k) w[k] = (char)(uint8_t)wide[k];
            } catch (const std::exception&) {
                return fail("vocabulary piece " + std::to_string(i) + " is not valid UTF-8");
            }
        }
    }
    while (c.p < c.end) {
        GgufTensor t;
        t.n_dims = c.get<int32_t>();
        const int32_t name_len = c.get<int32_t>();
        t.type = c.get<int32_t>();
        if (!c.ok || t.n_dims < 1 || t.n_dims > 4 || name_len < 0) return fail("corrupt tensor header");
        int64_t n_el = 1;
        for (int i = 0; i < t.n_dims; ++i) {
            t.ne[i] = c.get<int32_t>();
            if (!c.ok || t.ne[i] <= 0 || n_el > (int64_t)size_ * 8 / t.ne[i] + 1) return fail("corrupt tensor dimensions");
            n_el *= t.ne[i];
        }
        if (!c.ok || (size_t)name_len > c.left()) return fail("corrupt tensor name");
        t.name.assign((const char*)c.p, (size_t)name_len);
        c.p += name_len;
        const int be = ggml_block_elems(t.type), bb = ggml_block_bytes(t.type);
        if (be == 0 || n_el % be) return fail("tensor " + t.name + ": unsupported type " + std::to_string(t.type));
        t.nbytes = (size_t)(n_el / be) * (size_t)bb;
        if (t.nbytes > c.left()) return fail("tensor " + t.name + ": truncated data");
        t.data = c.p;
        t.offset = (uint64_t)(c.p - map_);
        c.p += t.nbytes;
        tindex_[t.name] = tensors_.size();
        tensors_.push_back(t);
    }
    return true;
}

const GgufTensor* LegacyGgmlFile::tensor(const std::string& name) const {
    auto it = tindex_.find(name);
    return it == tindex_.end() ? nullptr : &tensors_[it->second];
}

}  // namespace ctamd
