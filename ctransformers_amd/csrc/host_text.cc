#include "host_text.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <map>
#include <queue>
#include <regex>
#include <unordered_set>

#include "gguf_reader.h"

namespace ctamd {

bool Vocab::load(const GgufFile& f, std::string& err) {
    std::string model = "llama";
    f.get_str("tokenizer.ggml.model", model);
    if (model == "llama") {
        type = VOCAB_SPM;
        bos_id = 1; eos_id = 2; unk_id = 0;
    } else if (model == "gpt2") {
        type = VOCAB_BPE;
        bos_id = 11; eos_id = 11; unk_id = -1;
    } else {
        err = "unknown tokenizer model '" + model + "'";
        return false;
    }
    const GgufValue* toks = f.find("tokenizer.ggml.tokens");
    if (!toks || toks->type != GV_ARR || toks->elem_type != GV_STR) { err = "tokenizer.ggml.tokens missing"; return false; }
    const size_t n = toks->strs.size();
    text = toks->strs;
    score.assign(n, 0.0f);
    ttype.assign(n, TT_NORMAL);
    const GgufValue* sc = f.find("tokenizer.ggml.scores");
    if (sc && sc->type == GV_ARR && sc->elem_type == GV_F32 && sc->n == n) memcpy(score.data(), sc->arr, n * 4);
    const GgufValue* tt = f.find("tokenizer.ggml.token_type");
    if (tt && tt->type == GV_ARR && (tt->elem_type == GV_I32 || tt->elem_type == GV_U32) && tt->n == n)
        memcpy(ttype.data(), tt->arr, n * 4);
    to_id.reserve(n * 2);
    for (size_t i = 0; i < n; ++i) to_id[text[i]] = (int)i;
    uint32_t v;
    if (f.get_u32("tokenizer.ggml.bos_token_id", v)) bos_id = (int)v;
    if (f.get_u32("tokenizer.ggml.eos_token_id", v)) eos_id = (int)v;
    if (f.get_u32("tokenizer.ggml.unknown_token_id", v)) unk_id = (int)v;
    if (type == VOCAB_BPE) {   // merges -> ranks (reference llama.cpp:1691-1716: split at the first space after position 1)
        const GgufValue* mg = f.find("tokenizer.ggml.merges");
        if (!mg || mg->type != GV_ARR || mg->elem_type != GV_STR) { err = "cannot find tokenizer merges in model file"; return false; }
        bpe_rank.reserve(mg->strs.size() * 2);
        for (size_t i = 0; i < mg->strs.size(); ++i) {
            const std::string& word = mg->strs[i];
            std::string first, second;
            const size_t pos = word.find(' ', 1);
            if (pos != std::string::npos) { first = word.substr(0, pos); second = word.substr(pos + 1); }
            bpe_rank.emplace(first + '\x01' + second, (int)i);
        }
    }
    return true;
}

void Vocab::load_legacy(const std::vector<std::string>& pieces) {
    type = VOCAB_GPT;
    text = pieces;
    score.assign(text.size(), 0.0f);
    ttype.assign(text.size(), TT_NORMAL);
    to_id.clear();
    to_id.reserve(text.size() * 2);
    for (size_t i = 0; i < text.size(); ++i) to_id[text[i]] = (int)i;
    auto it = to_id.find("<|endoftext|>");
    eos_id = bos_id = it == to_id.end() ? 0 : it->second;
    unk_id = -1;
}

void Vocab::mark_starcoder_specials() {
    special.clear();
    for (const char* t : {"<|system|>", "<|user|>", "<|assistant|>", "<|end|>", "<fim-prefix>", "<fim-middle>", "<fim-suffix>", "<fim-pad>",
                          "<|end_of_turn|>"})
        if (to_id.count(t)) special.push_back(t);
}

namespace {
size_t utf8_len(char c) {
    static const size_t lookup[] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4};
    return lookup[(uint8_t)c >> 4];
}
void replace_all(std::string& s, const std::string& a, const std::string& b) {
    std::string out;
    size_t pos = 0;
    for (;;) {
        const size_t hit = s.find(a, pos);
        if (hit == std::string::npos) { out.append(s, pos, std::string::npos); break; }
        out.append(s, pos, hit - pos);
        out += b;
        pos = hit + a.size();
    }
    s.swap(out);
}

// Greedy highest-score bigram merging over a linked list of UTF-8 characters, then byte fallback for leftovers.
struct Sym { int prev, next; const char* p; size_t n; };
struct Bigram { int left, right; float score; size_t size; };
struct BigramLess {
    bool operator()(const Bigram& l, const Bigram& r) const {
        return l.score < r.score || (l.score == r.score && l.left > r.left);
    }
};

class SpmRun {
   public:
    explicit SpmRun(const Vocab& v) : v_(v) {}
    void run(const std::string& text, std::vector<int>& out) {
        size_t off = 0;
        int idx = 0;
        while (off < text.size()) {
            size_t len = std::min(utf8_len(text[off]), text.size() - off);
            Sym s{idx - 1, -1, text.data() + off, len};
            off += len;
            s.next = off == text.size() ? -1 : idx + 1;
            syms_.push_back(s);
            ++idx;
        }
        for (size_t i = 1; i < syms_.size(); ++i) try_add((int)i - 1, (int)i);
        while (!q_.empty()) {
            const Bigram b = q_.top();
            q_.pop();
            Sym& l = syms_[b.left];
            Sym& r = syms_[b.right];
            if (l.n == 0 || r.n == 0 || l.n + r.n != b.size) continue;
            l.n += r.n;
            r.n = 0;
            l.next = r.next;
            if (r.next >= 0) syms_[r.next].prev = b.left;
            try_add(l.prev, b.left);
            try_add(b.left, l.next);
        }
        if (syms_.empty()) return;
        for (int i = 0; i != -1; i = syms_[i].next) emit(syms_[i], out);
    }

   private:
    void emit(const Sym& s, std::vector<int>& out) {
        const std::string t(s.p, s.n);
        auto it = v_.to_id.find(t);
        if (it != v_.to_id.end()) { out.push_back(it->second); return; }
        auto rm = rev_.find(t);
        if (rm == rev_.end()) {
            for (size_t j = 0; j < s.n; ++j) {
                char buf[8];
                snprintf(buf, sizeof(buf), "<0x%02X>", (unsigned)(uint8_t)s.p[j]);
                auto bt = v_.to_id.find(buf);
                out.push_back(bt == v_.to_id.end() ? v_.unk_id : bt->second);
            }
            return;
        }
        emit(syms_[rm->second.first], out);
        emit(syms_[rm->second.second], out);
    }
    void try_add(int left, int right) {
        if (left == -1 || right == -1) return;
        const std::string t(syms_[left].p, syms_[left].n + syms_[right].n);
        auto it = v_.to_id.find(t);
        if (it == v_.to_id.end() || it->second >= v_.size()) return;
        q_.push(Bigram{left, right, v_.score[it->second], t.size()});
        rev_[t] = std::make_pair(left, right);
    }
    const Vocab& v_;
    std::vector<Sym> syms_;
    std::priority_queue<Bigram, std::vector<Bigram>, BigramLess> q_;
    std::map<std::string, std::pair<int, int>> rev_;
};
}  // namespace

namespace {
// Byte-level BPE as the reference runs it (llm_tokenizer_bpe, llama.cpp:3228-3388): GPT-2 pre-split with the same
// std::regex pattern, symbols = UTF-8 characters, merge the adjacent pair of lowest rank first (ties: leftmost), ranks
// looked up with ' ' -> U+0120 and '\n' -> U+010A substituted (find_bpe_rank :962-974), then whole-symbol vocabulary
// lookup with per-byte fallback.
class BpeRun {
   public:
    explicit BpeRun(const Vocab& v) : v_(v) {}
    void run(const std::string& text, std::vector<int>& out) {
        static const std::regex re(R"('s|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s[:alpha:][:digit:]]+|\s+(?!\S)|\s+)");
        std::string rest = text;
        std::smatch m;
        std::vector<std::string> words;
        while (std::regex_search(rest, m, re)) {
            for (auto x : m) words.push_back(x);
            rest = m.suffix();
        }
        for (const std::string& word : words) {
            struct S { int prev, next; size_t off, n; };
            std::vector<S> sym;
            size_t off = 0;
            int idx = 0;
            while (off < word.size()) {
                const size_t n = std::min(word.size() - off, utf8_len(word[off]));
                sym.push_back(S{idx - 1, off + n == word.size() ? -1 : idx + 1, off, n});
                off += n;
                ++idx;
            }
            struct B { int left, right, rank; std::string text; };
            auto worse = [](const B& l, const B& r) { return l.rank > r.rank || (l.rank == r.rank && l.left > r.left); };
            std::priority_queue<B, std::vector<B>, decltype(worse)> q(worse);
            auto piece = [&](int i) { return word.substr(sym[i].off, sym[i].n); };
            auto push = [&](int l, int r) {
                if (l == -1 || r == -1) return;
                const std::string lt = piece(l), rt = piece(r);
                const int rank = rank_of(lt, rt);
                if (rank < 0) return;
                q.push(B{l, r, rank, lt + rt});
            };
            for (size_t i = 1; i < sym.size(); ++i) push((int)i - 1, (int)i);
            while (!q.empty()) {
                const B b = q.top();
                q.pop();
                S& L = sym[b.left];
                S& R = sym[b.right];
                if (L.n == 0 || R.n == 0) continue;
                if (piece(b.left) + piece(b.right) != b.text) continue;   // outdated bigram
                L.n += R.n;
                R.n = 0;
                L.next = R.next;
                if (R.next >= 0) sym[R.next].prev = b.left;
                push(L.prev, b.left);
                push(b.left, L.next);
            }
            for (size_t i = 0; i < sym.size(); ++i) {
                if (sym[i].n == 0) continue;
                const std::string str = word.substr(sym[i].off, sym[i].n);
                auto it = v_.to_id.find(str);
                if (it != v_.to_id.end()) { out.push_back(it->second); continue; }
                for (char c : str) {
                    auto bt = v_.to_id.find(std::string(1, c));
                    if (bt != v_.to_id.end()) out.push_back(bt->second);
                    else fprintf(stderr, "ERROR: byte not found in vocab: '%c'\n", c);
                }
            }
        }
    }

   private:
    int rank_of(std::string l, std::string r) const {
        replace_all(l, " ", "\xc4\xa0");
        replace_all(l, "\n", "\xc4\x8a");
        replace_all(r, " ", "\xc4\xa0");
        replace_all(r, "\n", "\xc4\x8a");
        auto it = v_.bpe_rank.find(l + '\x01' + r);
        return it == v_.bpe_rank.end() ? -1 : it->second;
    }
    const Vocab& v_;
};
}  // namespace

std::vector<int> Vocab::tokenize(const std::string& raw, bool add_bos) const {
    std::vector<int> out;
    if (type == VOCAB_GPT) {   // gpt_tokenize (models/common.h:66-125; add_bos is ignored, models/llm.h:27-30): GPT-2 regex
        static const std::regex re(R"('s|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s[:alpha:][:digit:]]+|\s+(?!\S)|\s+)");
        std::vector<std::string> words;   // split, then the LONGEST vocabulary piece at every position of every word
        auto split_words = [&](std::string rest) {
            std::smatch m;
            while (std::regex_search(rest, m, re)) {
                for (auto x : m) words.push_back(x);
                rest = m.suffix();
            }
        };
        // Special pieces first (models/common.h:76-101): the reference searches an alternation of the escaped pieces, i.e. the
        // leftmost position where any of them starts and, there, the first of them in registration order; each becomes a word of
        // its own and the text between them goes through the word regex.
        size_t from = 0;
        for (size_t p = 0; !special.empty() && p < raw.size();) {
            const std::string* hit = nullptr;
            for (const std::string& t : special)
                if (raw.compare(p, t.size(), t) == 0) { hit = &t; break; }
            if (!hit) { ++p; continue; }
            split_words(raw.substr(from, p - from));
            words.push_back(*hit);
            from = p = p + hit->size();
        }
        split_words(raw.substr(from));
        for (const std::string& word : words) {
            for (int i = 0; i < (int)word.size();) {
                for (int j = (int)word.size() - 1; j >= i; --j) {
                    auto it = to_id.find(word.substr(i, j - i + 1));
                    if (it != to_id.end()) { out.push_back(it->second); i = j + 1; break; }
                    if (j == i) { fprintf(stderr, "gpt_tokenize: unknown token '%c'\n", word[i]); ++i; }
                }
            }
        }
        return out;
    }
    if (add_bos && bos_id != -1) out.push_back(bos_id);
    if (raw.empty()) return out;
    if (type == VOCAB_SPM) {
        std::string t = " " + raw;
        replace_all(t, " ", "\xe2\x96\x81");
        SpmRun run(*this);
        run.run(t, out);
    } else if (type == VOCAB_BPE) {
        BpeRun run(*this);
        run.run(raw, out);
    }
    return out;
}

std::string Vocab::piece(int token) const {
    if (token < 0 || token >= size()) return std::string();
    if (type == VOCAB_GPT) return text[token];
    const int tt = ttype[token];
    if (tt == TT_NORMAL) {
        std::string r = text[token];
        if (type == VOCAB_SPM) replace_all(r, "\xe2\x96\x81", " ");
        return r;
    }
    if (tt == TT_UNKNOWN) return "\xe2\x96\x85";
    if (tt == TT_BYTE) {
        const std::string& t = text[token];  // "<0xXX>"
        if (t.size() >= 5) return std::string(1, (char)strtol(t.substr(3, 2).c_str(), nullptr, 16));
    }
    return std::string();
}

// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Cand { int id; float logit; float p; };

void softmax_sorted(std::vector<Cand>& c, size_t& size, bool& sorted) {
    if (!sorted) {
        std::sort(c.begin(), c.begin() + size, [](const Cand& a, const Cand& b) { return a.logit > b.logit; });
        sorted = true;
    }
    const float max_l = c[0].logit;
    float cum = 0.0f;
    for (size_t i = 0; i < size; ++i) {
        const float p = expf(c[i].logit - max_l);
        c[i].p = p;
        cum += p;
    }
    for (size_t i = 0; i < size; ++i) c[i].p /= cum;
}
}  // namespace

int sample_token(const float* logits, int n_vocab, const int* last_tokens, int n_last, int top_k, float top_p,
                 float temperature, float repetition_penalty, int seed) {
    if (top_k == 1 && n_vocab > 0) {
        // Greedy fast path, same result as the chain below: after the penalty, partial_sort with k = 1 keeps the first
        // strict maximum in index order, and top-p / temperature / the draw over one candidate cannot change it.
        static thread_local std::vector<float> tmp;
        const float* lg = logits;
        if (n_last > 0 && repetition_penalty != 1.0f) {
            tmp.assign(logits, logits + n_vocab);
            std::vector<int> seen(last_tokens, last_tokens + n_last);
            std::sort(seen.begin(), seen.end());
            seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
            for (const int t : seen) {
                if (t < 0 || t >= n_vocab) continue;
                if (tmp[(size_t)t] <= 0) tmp[(size_t)t] *= repetition_penalty;
                else tmp[(size_t)t] /= repetition_penalty;
            }
            lg = tmp.data();
        }
        int best = 0;
        float bv = lg[0];
        for (int i = 1; i < n_vocab; ++i) {
            const float v = lg[i];
            if (v > bv) { bv = v; best = i; }
        }
        return best;
    }
    if (seed < 0) seed = (int)time(nullptr);
    std::mt19937 rng;
    rng.seed(seed);
    std::vector<Cand> c((size_t)n_vocab);
    for (int i = 0; i < n_vocab; ++i) c[i] = Cand{i, logits[i], 0.0f};
    size_t size = c.size();
    bool sorted = false;
    // repetition penalty
    if (n_last > 0 && repetition_penalty != 1.0f) {
        for (size_t i = 0; i < size; ++i) {
            if (std::find(last_tokens, last_tokens + n_last, c[i].id) == last_tokens + n_last) continue;
            if (c[i].logit <= 0) c[i].logit *= repetition_penalty;
            else c[i].logit /= repetition_penalty;
        }
        sorted = false;
    }
    // top-k (min_keep = 1)
    {
        int k = std::max(top_k, 1);
        k = std::min(k, (int)size);
        if (!sorted) {
            auto comp = [](const Cand& a, const Cand& b) { return a.logit > b.logit; };
            if (k == (int)size) std::sort(c.begin(), c.begin() + size, comp);
            else std::partial_sort(c.begin(), c.begin() + k, c.begin() + size, comp);
            sorted = true;
        }
        size = (size_t)k;
    }
    // top-p (min_keep = 1)
    if (top_p < 1.0f) {
        softmax_sorted(c, size, sorted);
        float cum = 0.0f;
        size_t last = size;
        for (size_t i = 0; i < size; ++i) {
            cum += c[i].p;
            if (cum >= top_p && i + 1 >= 1) { last = i + 1; break; }
        }
        size = last;
    }
    // temperature
    for (size_t i = 0; i < size; ++i) c[i].logit /= temperature;
    // draw
    softmax_sorted(c, size, sorted);
    std::vector<float> probs(size);
    for (size_t i = 0; i < size; ++i) probs[i] = c[i].p;
    std::discrete_distribution<> dist(probs.begin(), probs.end());
    return c[dist(rng)].id;
}

int sample_token_gpt(const float* logits, int n_vocab, const int* last_tokens, int n_last, int top_k, float top_p,
                     float temperature, float repetition_penalty, int seed) {
    if (seed < 0) seed = (int)time(nullptr);
    std::mt19937 rng((unsigned)seed);
    std::vector<std::pair<double, int>> lid;
    lid.reserve((size_t)n_vocab);
    const double temp = temperature;   // `double temp` parameter receives the float
    const double scale = 1.0 / temp;
    for (int i = 0; i < n_vocab; ++i) lid.push_back(std::make_pair(logits[i] * scale, i));
    const std::unordered_set<int> recent(last_tokens, last_tokens + n_last);
    for (const int token : recent) {
        if (token < 0 || token >= n_vocab) continue;
        double& logit = lid[(size_t)token].first;
        if (logit <= 0) logit *= repetition_penalty;
        else logit /= repetition_penalty;
    }
    if (top_k > n_vocab) top_k = n_vocab;
    if (top_k < 1) top_k = 1;
    std::partial_sort(lid.begin(), lid.begin() + top_k, lid.end(),
                      [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
    lid.resize((size_t)top_k);
    double maxl = -INFINITY;
    for (const auto& kv : lid) maxl = std::max(maxl, kv.first);
    std::vector<double> probs;
    probs.reserve(lid.size());
    double sum = 0.0;
    for (const auto& kv : lid) {
        const double p = exp(kv.first - maxl);
        probs.push_back(p);
        sum += p;
    }
    for (auto& p : probs) p /= sum;
    const double tp = top_p;
    if (tp < 1.0f) {
        double cumsum = 0.0f;
        for (int i = 0; i < top_k; i++) {
            cumsum += probs[(size_t)i];
            if (cumsum >= tp) {
                top_k = i + 1;
                probs.resize((size_t)top_k);
                lid.resize((size_t)top_k);
                break;
            }
        }
        cumsum = 1.0 / cumsum;
        for (size_t i = 0; i < probs.size(); i++) probs[i] *= cumsum;
    }
    std::discrete_distribution<> dist(probs.begin(), probs.end());
    return lid[(size_t)dist(rng)].second;
}

}  // namespace ctamd
