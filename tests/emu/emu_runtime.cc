// TEST INFRASTRUCTURE ONLY — see emu_runtime.h.
#include "emu_runtime.h"
#include <ucontext.h>
#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
namespace {
struct Rendezvous { int count = 0, expected = 0; unsigned gen = 0; };
struct Wave { Rendezvous r; uint64_t buf[2][64]; };
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    const unsigned* wait_gen = nullptr;
    unsigned wait_val = 0;
    dim3 tid;
    int lin = 0;
};
constexpr size_t kStack = 256 * 1024;
ucontext_t g_sched;
std::vector<Fiber> g_fibers;
std::vector<Wave> g_waves;
Rendezvous g_block;
int g_cur = -1;
const std::function<void()>* g_body = nullptr;

void release_if_full(Rendezvous& r) {
    if (r.count > 0 && r.count >= r.expected) { r.count = 0; r.gen++; }
}
void yield_to_sched() { swapcontext(&g_fibers[g_cur].ctx, &g_sched); }
void arrive(Rendezvous& r) {
    r.count++;
    if (r.count >= r.expected) { r.count = 0; r.gen++; return; }
    Fiber& f = g_fibers[g_cur];
    f.wait_gen = &r.gen;
    f.wait_val = r.gen;
    yield_to_sched();
    g_fibers[g_cur].wait_gen = nullptr;
}
void fiber_entry() {
    (*g_body)();
    Fiber& f = g_fibers[g_cur];
    f.done = true;
    Wave& w = g_waves[f.lin / 64];
    w.r.expected--;
    release_if_full(w.r);
    g_block.expected--;
    release_if_full(g_block);
    yield_to_sched();
}
}  // namespace

void syncthreads() { arrive(g_block); }

static unsigned long long g_spins = 0;
void spin_yield() {
    if (++g_spins > 2000000000ull) { fprintf(stderr, "emu: spin-wait never satisfied\n"); abort(); }
    yield_to_sched();
}
static std::vector<unsigned char> g_dyn;
void dyn_smem_reserve(size_t n) {
    if (n > 160 * 1024) { fprintf(stderr, "emu: %zu bytes of LDS requested (160 KB per CU)\n", n); abort(); }
    if (g_dyn.size() < n + 64) g_dyn.resize(n + 64);
}
unsigned char* dyn_smem() { return (unsigned char*)(((uintptr_t)g_dyn.data() + 63) & ~(uintptr_t)63); }

uint64_t wave_exchange(uint64_t v, int src_lane) {
    Fiber& f = g_fibers[g_cur];
    Wave& w = g_waves[f.lin / 64];
    const int slot = w.r.gen & 1;
    w.buf[slot][f.lin & 63] = v;
    arrive(w.r);
    return w.buf[slot][src_lane & 63];
}

void wave_sync() { arrive(g_waves[g_fibers[g_cur].lin / 64].r); }

uint64_t wave_ballot(int pred) {
    Fiber& f = g_fibers[g_cur];
    Wave& w = g_waves[f.lin / 64];
    const int slot = w.r.gen & 1;
    const int n_lanes = w.r.expected;  // live lanes (exited lanes never reach here)
    (void)n_lanes;
    w.buf[slot][f.lin & 63] = pred ? 1 : 0;
    // mark lanes beyond the wave's population as 0
    arrive(w.r);
    uint64_t m = 0;
    const int base = (f.lin / 64) * 64;
    for (int l = 0; l < 64; ++l) {
        const int lin = base + l;
        if (lin < (int)g_fibers.size() && !g_fibers[lin].done && w.buf[slot][l]) m |= (1ull << l);
    }
    return m;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int T = (int)(block.x * block.y * block.z);
    if (T <= 0 || T > 1024) { fprintf(stderr, "emu: bad block size %d\n", T); abort(); }
    if ((int)g_fibers.size() < T) {
        const size_t old = g_fibers.size();
        g_fibers.resize(T);
        for (size_t i = old; i < g_fibers.size(); ++i) g_fibers[i].stack = (char*)malloc(kStack);
    }
    const int n_waves = (T + 63) / 64;
    if ((int)g_waves.size() < n_waves) g_waves.resize(n_waves);
    ::blockDim = block;
    ::gridDim = grid;
    g_body = &body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                ::blockIdx = dim3(bx, by, bz);
                g_block = Rendezvous();
                g_block.expected = T;
                for (int w = 0; w < n_waves; ++w) {
                    g_waves[w].r = Rendezvous();
                    g_waves[w].r.expected = (w == n_waves - 1) ? T - 64 * w : 64;
                }
                for (int i = 0; i < T; ++i) {
                    Fiber& f = g_fibers[i];
                    f.done = false;
                    f.wait_gen = nullptr;
                    f.lin = i;
                    f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
                }
                g_spins = 0;
                int live = T;
                while (live > 0) {
                    bool progressed = false;
                    live = 0;
                    for (int i = 0; i < T; ++i) {
                        Fiber& f = g_fibers[i];
                        if (f.done) continue;
                        live++;
                        if (f.wait_gen && *f.wait_gen == f.wait_val) continue;  // still blocked
                        g_cur = i;
                        ::threadIdx = f.tid;
                        swapcontext(&g_sched, &f.ctx);
                        progressed = true;
                    }
                    if (live > 0 && !progressed) {
                        fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent collective or barrier\n", bx, by, bz);
                        abort();
                    }
                }
            }
    g_body = nullptr;
}
}  // namespace emu
