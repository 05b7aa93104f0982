#include "gguf_reader.h"

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
    template <class T> T get() {
        T v{};
        if (p + sizeof(T) > end) { ok = false; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    uint64_t len() { return v1 ? (uint64_t)get<uint32_t>() : get<uint64_t>(); }
    std::string str() {
        const uint64_t n = len();
        if (!ok || p + n > end) { ok = false; return std::string(); }
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
                if (v.elem_type == GV_STR) {
                    v.strs.reserve((size_t)v.n);
                    for (uint64_t k = 0; k < v.n && c.ok; ++k) v.strs.push_back(c.str());
                } else {
                    const size_t es = scalar_size(v.elem_type);
                    if (es == 0) return fail("bad array element type for key " + key);
                    v.arr = c.p;
                    if (c.p + es * v.n > c.end) return fail("truncated array " + key);
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
        for (int d = 0; d < t.n_dims; ++d) t.ne[d] = (int64_t)c.len();
        t.type = (int)c.get<uint32_t>();
        t.offset = c.get<uint64_t>();
        if (ggml_block_elems(t.type) == 0) return fail("unsupported tensor type in " + t.name);
        if (t.ne[0] % ggml_block_elems(t.type) != 0) return fail("row length not a multiple of the block size: " + t.name);
        t.nbytes = ggml_row_bytes(t.type, t.ne[0]) * (size_t)(t.ne[1] * t.ne[2] * t.ne[3]);
        tindex_[t.name] = (size_t)i;
    }
    if (!c.ok) return fail("truncated tensor table");
    uint32_t align = 32;
    get_u32("general.alignment", align);
    size_t base = (size_t)(c.p - map_);
    base = (base + align - 1) / align * align;
    for (auto& t : tensors_) {
        if (base + t.offset + t.nbytes > size_) return fail("tensor data out of file bounds: " + t.name);
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

bool LegacyGgmlFile::open(const std::string& path) {
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
    for (int i = 0; i < 6; ++i) hparams[i] = c.get<int32_t>();
    hparams[5] %= 1000;   // GGML_QNT_VERSION_FACTOR (gpt2.cc:88-90)
    const int32_t nv = c.get<int32_t>();
    if (!c.ok || nv != hparams[0] || nv <= 0) return fail("bad vocabulary size");
    vocab.reserve((size_t)nv);
    for (int i = 0; i < nv; ++i) {
        const uint32_t len = c.get<uint32_t>();
        if (!c.ok || c.p + len > c.end) return fail("truncated vocabulary");
        vocab.emplace_back((const char*)c.p, (size_t)len);
        c.p += len;
    }
    while (c.p < c.end) {
        GgufTensor t;
        t.n_dims = c.get<int32_t>();
        const int32_t name_len = c.get<int32_t>();
        t.type = c.get<int32_t>();
        if (!c.ok || t.n_dims < 1 || t.n_dims > 4 || name_len < 0) return fail("corrupt tensor header");
        int64_t n_el = 1;
        for (int i = 0; i < t.n_dims; ++i) { t.ne[i] = c.get<int32_t>(); n_el *= t.ne[i]; }
        if (!c.ok || c.p + name_len > c.end) return fail("corrupt tensor name");
        t.name.assign((const char*)c.p, (size_t)name_len);
        c.p += name_len;
        const int be = ggml_block_elems(t.type), bb = ggml_block_bytes(t.type);
        if (be == 0 || n_el % be) return fail("tensor " + t.name + ": unsupported type " + std::to_string(t.type));
        t.nbytes = (size_t)(n_el / be) * (size_t)bb;
        if (c.p + t.nbytes > c.end) return fail("tensor " + t.name + ": truncated data");
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
