// Experiment (not part of the product): is ONE resident grid walking the phases of a token step (QKV, attention, Wo, gate+up, down per
// layer) with a FLAG-ARRAY grid barrier between phases faster than one launch per phase inside a hipGraph, for decode-sized weight
// streams and a 16 KB activation vector handed from phase to phase?  No block math: this is the bound of each structure.
//   L   one launch per phase (256 x 1024 threads, 4-record register ring per wave, ring requested BEFORE the activation vector is
//       loaded — what kernels_v9.h does), 5 x LAYERS launches chained in a hipGraph
//   P   one launch for everything: per phase  [request the phase's first 4 records]  ->  wait until all workgroups have published
//       the previous phase (every workgroup owns ONE word of a 1 KB flag array: one release store each, no read-modify-write; a
//       wave polls the whole array with one 16-byte load per lane)  ->  acquire, load the activation vector  ->  stream  ->  write
//       this workgroup's 16 outputs  ->  release, publish.  PRE=0: the ring is requested AFTER the wait (what the barrier alone costs).
//   Q   P with one 16-byte poll per lane and a pause between polls
//   R   no release / acquire fences at all (no L2 write-back, no L2 invalidate): outputs and flags are agent-scope stores, the stores
//       acknowledged (vmcnt(0)) before the flag goes out, the activation vector read with agent-scope loads; checksum against L
//   S   R, but every phase writes a vector at an address nobody has read in this launch, so consumers may read it with plain loads through
//       their XCD's L2 (one fill per XCD instead of 256 reads past L2 of the same 16 KB); only the 1 KB of flags is polled past L2
//   T   tagged data (below)
//   C   the same with the one-counter barrier of grid_barrier_probe.cpp (256 serialised read-modify-writes)
// Every poll loop has a cycle limit (an abort word ends all of them), so a scheduling surprise cannot hang the GPU.
//   hipcc -O3 --offload-arch=gfx950 -o persist_probe.bin persist_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <utility>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int REC = 1152, NPH = 5, NWG = 256;
struct Phases { int recs[NPH]; };   // records per wave and phase

__device__ __forceinline__ unsigned fold(u32x4 b, u32x4 h) { return (b[0] ^ b[1]) + (b[2] ^ b[3]) + (h[0] ^ h[1]) + (h[2] ^ h[3]); }

struct Ring {
    u32x4 body[4], hdr[4];
    const unsigned char* p;
    int left;
    __device__ __forceinline__ void issue(int k, int lane) {
        body[k] = __builtin_nontemporal_load((const u32x4*)(p + lane * 16));
        hdr[k] = __builtin_nontemporal_load((const u32x4*)(p + 1024 + (lane >> 3) * 16));
        if (left > 1) { --left; p += REC; }
    }
    __device__ __forceinline__ void begin(const unsigned char* base, int recs, int lane) {
        p = base; left = recs;
#pragma unroll
        for (int k = 0; k < 4; ++k) { issue(k, lane); __builtin_amdgcn_sched_barrier(0); }   // the loop's request order, so that hipcc's vmcnt bookkeeping agrees at the loop head
    }
    __device__ __forceinline__ unsigned drain(int recs, int lane) {   // the last round requests nothing (kernels_v9.h does the same)
        unsigned acc = 0;
        int st = 0;
        for (; st + 4 < recs; st += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned v = fold(body[k], hdr[k]);
                asm volatile("" : "+v"(v));
                issue(k, lane);
                acc += v;
                __builtin_amdgcn_sched_barrier(0);   // slot by slot: consume k, re-request k (left alone hipcc drains the whole ring first)
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned v = fold(body[k], hdr[k]);
            asm volatile("" : "+v"(v));
            if (st + k < recs) acc += v;
        }
        return acc;
    }
};

// the "prologue": every thread takes 16 bytes of the 16 KB activation vector, the workgroup reduces it through LDS (one barrier)
__device__ __forceinline__ u32x4 request_activation(const unsigned* act) {
    const u32x4 a = *(const u32x4*)(act + threadIdx.x * 4);
    __builtin_amdgcn_sched_barrier(0);
    return a;
}
__device__ __forceinline__ u32x4 request_activation_agent(const unsigned* act) {   // agent-scope loads: past this XCD's L2
    u32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = __hip_atomic_load(act + threadIdx.x * 4 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_sched_barrier(0);
    return a;
}
__device__ __forceinline__ unsigned take_activation(u32x4 a, unsigned* lds) {
    unsigned v = a[0] + a[1] + a[2] + a[3];
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += lds[i];
    return s;
}

__global__ void __launch_bounds__(1024) phase_kernel(const unsigned char* w, int recs, const unsigned* act_in, unsigned* act_out, unsigned* sink) {
    __shared__ unsigned lds[16];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Ring R;
    const u32x4 a = request_activation(act_in);   // first: loads return in order, the vector must not queue behind the ring
    R.begin(w + ((size_t)(blockIdx.x * 16 + wv) * recs) * REC, recs, lane);
    const unsigned s = take_activation(a, lds);
    const unsigned acc = R.drain(recs, lane) + s;
    if (lane < 1) act_out[blockIdx.x * 16 + wv] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, acc);
}

constexpr unsigned long long LIMIT = 2000000000ull;   // s_memtime ticks (about a second) — generous, only there to end a hang

template <int BAR> __device__ __forceinline__ void publish(unsigned* flags, unsigned phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (BAR < 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (BAR == 0 || BAR >= 2) __hip_atomic_store(flags + blockIdx.x, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(flags + 512, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// all workgroups have published `phase`
template <int BAR> __device__ __forceinline__ void wait_all(unsigned* flags, unsigned phase, unsigned* abort_word) {
    if (threadIdx.x < 64) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (;;) {
            bool ok;
            if (BAR == 0) {
                unsigned m = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const unsigned f = __hip_atomic_load(flags + threadIdx.x * 4 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); m = f < m ? f : m; }
                ok = __all(m >= phase);
            } else if (BAR >= 2) {   // one 16-byte agent-scope load per lane and a pause between polls
                u32x4 f;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(f) : "v"(flags + threadIdx.x * 4) : "memory");
                const unsigned m0 = f[0] < f[1] ? f[0] : f[1], m1 = f[2] < f[3] ? f[2] : f[3];
                ok = __all((m0 < m1 ? m0 : m1) >= phase);
                if (!ok) __builtin_amdgcn_s_sleep(8);
            } else {
                ok = __hip_atomic_load(flags + 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase * NWG;
            }
            if (ok) break;
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (__builtin_readcyclecounter() - t0 > LIMIT) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        if (BAR < 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int BAR, bool PRE>
__global__ void __launch_bounds__(1024) persistent_kernel(const unsigned char* w, Phases ph, int layers, unsigned* act, unsigned* flags, unsigned* abort_word, unsigned* sink) {
    __shared__ unsigned lds[16];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    size_t off = 0;
    unsigned n = 0, total = 0;
    for (int l = 0; l < layers; ++l)
        for (int q = 0; q < NPH; ++q, ++n) {
            const int recs = ph.recs[q];
            const unsigned char* base = w + off + ((size_t)(blockIdx.x * 16 + wv) * recs) * REC;
            off += (size_t)NWG * 16 * recs * REC;
            Ring R;
            if (PRE) R.begin(base, recs, lane);
            if (n) wait_all<BAR>(flags, n, abort_word);
            const u32x4 a = BAR == 3 ? request_activation_agent(act + (size_t)(n & 1) * 4096) : request_activation(act + (BAR == 4 ? (size_t)n : (size_t)(n & 1)) * 4096);
            if (!PRE) R.begin(base, recs, lane);
            const unsigned s = take_activation(a, lds);
            const unsigned acc = R.drain(recs, lane) + s;
            unsigned* out = act + (BAR == 4 ? (size_t)(n + 1) : (size_t)((n + 1) & 1)) * 4096 + blockIdx.x * 16 + wv;
            if (BAR >= 3) {   // no fences: the output goes out as an agent-scope store (past L2), acknowledged before the workgroup publishes
                if (lane < 1) __hip_atomic_store(out, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (lane < 1) *out = acc;
            total += acc;
            publish<BAR>(flags, n + 1);
        }
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, total);
}

// ---- T: tagged activations (value and phase tag in ONE 8-byte store: the consumer polls the data itself — no flags, no fences, no
// read-modify-write) and a ring that never stops at a phase boundary: a wave's records of all phases and layers are one static
// sequence, the four slots always hold the next four of them, whatever phase they belong to.  Four layers (100 record steps) are
// unrolled so that slot numbers and phase boundaries are compile-time facts.
constexpr int RECS[NPH] = {6, 1, 2, 11, 5};
constexpr int LSTEPS = RECS[0] + RECS[1] + RECS[2] + RECS[3] + RECS[4];
constexpr int phase_of(int st) { int q = 0; while (st >= RECS[q]) { st -= RECS[q]; ++q; } return q; }
constexpr int rec_of(int st) { int q = 0; while (st >= RECS[q]) { st -= RECS[q]; ++q; } return st; }
constexpr size_t phase_off(int q) { size_t o = 0; for (int i = 0; i < q; ++i) o += (size_t)NWG * 16 * RECS[i] * REC; return o; }
constexpr size_t LAYER_BYTES = phase_off(NPH);

struct TState {
    u32x4 body[4], hdr[4];
    const unsigned char* w;      // this layer group's base
    unsigned long long* act;     // 2 x 4096 {value, tag}
    unsigned* lds;
    unsigned wave_id, n, acc, total, lane;
    unsigned* abort_word;
};
template <int I> __device__ __forceinline__ void t_issue(TState& S) {   // request the record of step I (relative to the current 4-layer group) into slot I % 4
    constexpr int st = I % LSTEPS, q = phase_of(st), r = rec_of(st), k = I % 4;
    const size_t a = (size_t)(S.w + (size_t)(I / LSTEPS) * LAYER_BYTES + phase_off(q) + ((size_t)S.wave_id * RECS[q] + r) * REC);
    typedef const __attribute__((address_space(1))) u32x4* gp;   // built from integers the pointer must be told it is global, or the loads are FLAT
    const size_t u = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) | (size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);   // a scalar base
    S.body[k] = __builtin_nontemporal_load((gp)(u + S.lane * 16));
    S.hdr[k] = __builtin_nontemporal_load((gp)(u + 1024 + (S.lane >> 3) * 16));
    __builtin_amdgcn_sched_barrier(0);
}
template <int I> __device__ __forceinline__ void t_step(TState& S) {
    constexpr int st = I % LSTEPS, q = phase_of(st), r = rec_of(st), k = I % 4;
    if constexpr (r == 0) {   // first record of a phase: its input vector must have arrived — poll the data, 4 elements per thread
        const unsigned long long* in = S.act + (size_t)(S.n & 1) * 4096 + threadIdx.x * 4;
        unsigned v;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (;;) {
            unsigned long long e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
            v = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ok = ok && (unsigned)(e[i] >> 32) == S.n; v += (unsigned)e[i]; }
            if (__all(ok)) break;
            if (__hip_atomic_load(S.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (__builtin_readcyclecounter() - t0 > LIMIT) { __hip_atomic_store(S.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();   // the previous phase's readers of lds are done
        if (S.lane == 0) S.lds[threadIdx.x >> 6] = v;
        __syncthreads();
        unsigned sum = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += S.lds[i];
        S.acc = sum;
    }
    unsigned f = fold(S.body[k], S.hdr[k]);
    asm volatile("" : "+v"(f));
    S.acc += f;
    t_issue<I + 4>(S);
    if constexpr (r == RECS[q] - 1) {   // last record of a phase: publish this wave's output, value and tag in one store
        if (S.lane == 0)
            __hip_atomic_store(S.act + (size_t)((S.n + 1) & 1) * 4096 + S.wave_id, ((unsigned long long)(S.n + 1) << 32) | S.acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S.total += S.acc;
        ++S.n;
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int... I> __device__ __forceinline__ void t_steps(TState& S, std::integer_sequence<int, I...>) { (t_step<I>(S), ...); }

__global__ void __launch_bounds__(1024) tagged_kernel(const unsigned char* w, int layers, unsigned long long* act, unsigned* abort_word, unsigned* sink) {   // layers % 4 == 0
    __shared__ unsigned lds[16];
    TState S;
    S.lane = threadIdx.x & 63;
    S.wave_id = blockIdx.x * 16 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    S.w = w; S.act = act; S.lds = lds; S.abort_word = abort_word; S.n = 0; S.acc = 0; S.total = 0;
    t_issue<0>(S); t_issue<1>(S); t_issue<2>(S); t_issue<3>(S);
    for (int l = 0; l < layers; l += 4) {
        t_steps(S, std::make_integer_sequence<int, 4 * LSTEPS>{});
        S.w += 4 * LAYER_BYTES;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, S.total);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int layers = argc > 1 ? atoi(argv[1]) : 32;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    if (cus != NWG) { printf("this probe wants %d CUs, device has %d\n", NWG, cus); return 0; }
    const Phases ph = {{6, 1, 2, 11, 5}};   // qkv 28 MB, attention 4.7 MB (the K / V rows), wo 9.4 MB, gate+up 52 MB, down 24 MB
    size_t per_layer = 0;
    for (int q = 0; q < NPH; ++q) per_layer += (size_t)NWG * 16 * ph.recs[q] * REC;
    const size_t bytes = per_layer * layers;
    unsigned char* w; unsigned *act, *flags, *abort_word, *sink;
    CK(hipMalloc(&w, bytes + per_layer + 65536)); CK(hipMemset(w, 0x5a, bytes + per_layer + 65536));   // T's ring runs four records past the end
    const size_t act_bytes = (size_t)(layers * NPH + 2) * 4096 * 4;   // S: a fresh vector per phase
    CK(hipMalloc(&act, act_bytes)); CK(hipMemset(act, 1, act_bytes));
    CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&abort_word, 4)); CK(hipMalloc(&sink, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d layers x 5 phases, %.1f MB per layer (records per wave: %d %d %d %d %d), 256 x 1024 threads\n", layers, per_layer / 1e6, ph.recs[0], ph.recs[1],
           ph.recs[2], ph.recs[3], ph.recs[4]);

    auto reset = [&]() { CK(hipMemsetAsync(act, 1, act_bytes, s)); CK(hipMemsetAsync(sink, 0, 4, s)); };
    auto read_sink = [&]() { unsigned v = 0; CK(hipStreamSynchronize(s)); CK(hipMemcpy(&v, sink, 4, hipMemcpyDeviceToHost)); return v; };
    unsigned want = 0;
    // L: one launch per phase in a graph
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        size_t off = 0; unsigned n = 0;
        for (int l = 0; l < layers; ++l)
            for (int q = 0; q < NPH; ++q, ++n) {
                hipLaunchKernelGGL(phase_kernel, dim3(NWG), dim3(1024), 0, s, (const unsigned char*)w + off, ph.recs[q], (const unsigned*)act + (size_t)(n & 1) * 4096,
                                   act + (size_t)((n + 1) & 1) * 4096, sink);
                off += (size_t)NWG * 16 * ph.recs[q] * REC;
            }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        reset();
        CK(hipGraphLaunch(ge, s)); want = read_sink();
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("L  one launch per phase (hipGraph)          : %8.2f us per layer, %6.2f TB/s\n", ms * 1e3 / 3 / layers, per_layer / (ms * 1e-3 / 3 / layers) / 1e12);
    }
    // each phase alone: 64 launches of one size chained in a graph, walking through the buffer (HBM-cold)
    for (int q = 0; q < NPH; ++q) {
        hipGraph_t g; hipGraphExec_t ge;
        const size_t per = (size_t)NWG * 16 * ph.recs[q] * REC;
        const int nl = 64;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < nl; ++i)
            hipLaunchKernelGGL(phase_kernel, dim3(NWG), dim3(1024), 0, s, (const unsigned char*)w + ((size_t)i * per) % (bytes - per), ph.recs[q], (const unsigned*)act + (size_t)(i & 1) * 4096,
                               act + (size_t)((i + 1) & 1) * 4096, sink);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("L  phase %d alone (%2d records per wave, %5.1f MB): %6.2f us per launch, %5.2f TB/s\n", q, ph.recs[q], per / 1e6, ms * 1e3 / 3 / nl, per / (ms * 1e-3 / 3 / nl) / 1e12);
    }
    auto persistent = [&](auto kernel, const char* name) {
        float best = 1e30f; unsigned ab = 0, got = 0;
        for (int r = 0; r < 4; ++r) {
            CK(hipMemsetAsync(flags, 0, 4096, s)); CK(hipMemsetAsync(abort_word, 0, 4, s));
            if (r == 0) reset();
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(kernel, dim3(NWG), dim3(1024), 0, s, (const unsigned char*)w, ph, layers, act, flags, abort_word, sink);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
            if (r == 0) got = read_sink();
            unsigned a = 0; CK(hipMemcpy(&a, abort_word, 4, hipMemcpyDeviceToHost)); ab |= a;
        }
        printf("%s: %8.2f us per layer, %6.2f TB/s%s\n", name, best * 1e3 / layers, per_layer / (best * 1e-3 / layers) / 1e12, ab ? "   ABORTED (a wait hit its cycle limit)" : got == want ? "   (checksum equal to L)" : "   CHECKSUM DIFFERS FROM L: a stale value was read");
    };
    persistent(persistent_kernel<0, true>,  "P  persistent, flag array, ring before wait  ");
    persistent(persistent_kernel<0, false>, "P0 persistent, flag array, ring after wait   ");
    persistent(persistent_kernel<2, true>,  "Q  persistent, flag array x4 + pause, before  ");
    persistent(persistent_kernel<2, false>, "Q0 persistent, flag array x4 + pause, after   ");
    persistent(persistent_kernel<3, true>,  "R  persistent, flag array, NO fences, before  ");
    persistent(persistent_kernel<3, false>, "R0 persistent, flag array, NO fences, after   ");
    persistent(persistent_kernel<4, true>,  "S  R + a fresh vector per phase through L2, before");
    persistent(persistent_kernel<4, false>, "S0 R + a fresh vector per phase through L2, after ");
    if (layers % 4 == 0) {
        unsigned long long* act64; CK(hipMalloc(&act64, 2 * 4096 * 8));
        std::vector<unsigned long long> init(2 * 4096);
        for (int i = 0; i < 4096; ++i) { init[i] = 0x01010101ull; init[4096 + i] = 0xffffffff00000000ull; }   // phase 0 reads buffer 0 with tag 0
        float best = 1e30f; unsigned got = 0;
        for (int r = 0; r < 4; ++r) {
            CK(hipMemcpy(act64, init.data(), init.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemsetAsync(sink, 0, 4, s)); CK(hipMemsetAsync(abort_word, 0, 4, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(tagged_kernel, dim3(NWG), dim3(1024), 0, s, (const unsigned char*)w, layers, act64, abort_word, sink);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
            got = read_sink();
            unsigned a = 0; CK(hipMemcpy(&a, abort_word, 4, hipMemcpyDeviceToHost));
            if (a) { printf("T  ABORTED (a poll hit its cycle limit)\n"); break; }
        }
        printf("T  persistent, tagged data, continuous ring  : %8.2f us per layer, %6.2f TB/s%s\n", best * 1e3 / layers, per_layer / (best * 1e-3 / layers) / 1e12,
               got == want ? "   (checksum equal to L)" : "   CHECKSUM DIFFERS FROM L");
    }
    persistent(persistent_kernel<1, true>,  "C  persistent, one counter, ring before wait ");
    persistent(persistent_kernel<1, false>, "C0 persistent, one counter, ring after wait  ");
    return 0;
}
