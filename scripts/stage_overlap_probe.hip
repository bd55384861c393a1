// What does a chain of dependent stages cost when the stages are
//   (k)  one kernel per stage on one stream, replayed from a HIP graph (what the sampling loop does today),
//   (a0) the iterations of ONE persistent kernel separated by an XCD-hierarchical grid barrier,
//   (a)  the same, with every stage's dependency-free bytes (its weight slice) issued by LDS-DMA BEFORE the wait,
//   (b)  one kernel per stage on TWO streams (products on one, norms on the other) with no stream dependency between
//        them: a kernel issues its dependency-free bytes, then one lane polls a counter that the previous stage's
//        workgroups raise?
//
// The chain models `norm -> d x d split-K product -> norm -> ...` of the sampling loop at 128 rows, 256 workgroups:
//   P stage (product): 32 KB of weights per workgroup (dependency-free, LDS-DMA), 64 KB of the previous norm's output
//                      gathered in 1-KB pieces from 64 writers (dependent, LDS-DMA), a pass over both in LDS, a 32-KB fp32
//                      slab piece out;
//   N stage (norm):    no LDS; 32 KB of the previous product's slabs (dependent, register loads), 2 KB out.
// Every dependent word is checked (stage s writes the value s everywhere), so a stale read is counted, not hidden.
// Every spin is bounded: a workgroup that runs out of polls raises `failed` and every workgroup leaves at its next
// poll, so the grid always drains.
//
// build: hipcc --offload-arch=gfx950 -O3 scripts/stage_overlap_probe.hip -o scripts/stage_overlap_probe
// run:   scripts/stage_overlap_probe [pairs=600]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kWG = 256;                 // workgroups: one per CU
constexpr int kT = 256;                  // threads per workgroup
constexpr int kWeightKB = 32;            // P: dependency-free bytes per workgroup
constexpr int kActKB = 64;               // P: dependent bytes per workgroup
constexpr int kSlabKB = 32;              // P: output per workgroup = N: dependent bytes per workgroup
constexpr int kNormOutKB = 2;            // N: output per workgroup
constexpr int kRot = 16;                 // distinct weight slices (16 x 8 MB: more than the caches keep hot)
constexpr unsigned kSpin = 2000000u;

struct Sync {
    unsigned xcc_cnt[8][32];             // arrivals per XCD (each counter on a 128-byte line of its own)
    unsigned gen[8][32];                 // generation word per XCD
    unsigned top[32];                    // arrivals of XCD leaders
    unsigned census[8][32];              // workgroups per XCD (filled at kernel start)
    unsigned flat[32];                   // census barrier
    unsigned failed[32];
    unsigned errors[32];                 // dependent words that did not hold the expected value
    unsigned done[32];                   // variant (b): done[0] counts finished workgroups of all stages (monotonic)
};

struct Bufs {
    const char* weights;                 // kRot slices of kWG * kWeightKB KB
    char* act;                           // N output: kWG * kNormOutKB KB
    char* slab;                          // P output: kWG * kSlabKB KB
    Sync* sync;
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_relaxed(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_relaxed(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane polls until *p >= want; false when the spin limit ran out or another workgroup failed
__device__ __forceinline__ bool poll_ge(const unsigned* p, unsigned want, Sync* sy) {
    unsigned polls = 0;
    while (ld_relaxed(p) < want) {
        if (++polls > kSpin || ((polls & 1023u) == 0 && ld_relaxed(sy->failed) != 0)) {
            st_relaxed(sy->failed, 1u);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

// ---- stage bodies ------------------------------------------------------------------------------
// P, dependency-free part: the workgroup's weight slice -> LDS (32 instructions of 1 KB per wave: 8 per wave)
__device__ __forceinline__ void p_issue_weights(const Bufs& b, int stage, int wg, char* lds) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const char* src = b.weights + ((size_t)(stage % kRot) * kWG + wg) * (kWeightKB * 1024) + lane * 16;
#pragma unroll
    for (int j = 0; j < kWeightKB / 4; ++j) {
        const int piece = wid + 4 * j;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 1024), (lptr_t)(lds + piece * 1024), 16, 0, 2);
    }
}
// P, dependent part: 64 pieces of 1 KB from 64 different writers of the previous norm stage -> LDS
__device__ __forceinline__ void p_issue_act(const Bufs& b, int wg, char* lds) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < kActKB / 4; ++j) {
        const int piece = wid + 4 * j;
        const int writer = (wg * 5 + piece * 4 + 1) & (kWG - 1);
        const char* src = b.act + ((size_t)writer * kNormOutKB + (piece & (kNormOutKB - 1))) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + piece * 1024), 16, 0, 0);
    }
}
// P: pass over the staged bytes (weights summed, activation words checked), then the slab piece out
template <bool SC1>
__device__ __forceinline__ void p_consume_store(const Bufs& b, int stage, int wg, const char* lds_w, const char* lds_a, bool check) {
    const float want = (float)(stage - 1);
    float acc = 0.f;
    unsigned bad = 0;
#pragma unroll
    for (int j = 0; j < kWeightKB / 4; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds_w + (j * kT + threadIdx.x) * 16);
        acc += v.x + v.w;
    }
#pragma unroll
    for (int j = 0; j < kActKB / 4; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds_a + (j * kT + threadIdx.x) * 16);
        bad += (v.x != want) + (v.y != want) + (v.z != want) + (v.w != want);
    }
    if (check && bad) atomicAdd(b.sync->errors, bad);
    const float out = (float)stage + acc;                      // weights are zero: acc == 0
    char* dst = b.slab + (size_t)wg * (kSlabKB * 1024) + threadIdx.x * 16;
    const f32x4 o = {out, out, out, out};
#pragma unroll
    for (int j = 0; j < kSlabKB / 4; ++j) {
        if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + j * 4096), "v"(o) : "memory");
        else *reinterpret_cast<f32x4*>(dst + j * 4096) = o;
    }
}
// N: 32 KB of the previous product's slab pieces (8 pieces of 4 KB from 8 writers), checked; 2 KB out
template <bool SC1>
__device__ __forceinline__ void n_body(const Bufs& b, int stage, int wg, bool check) {
    const float want = (float)(stage - 1);
    f32x4 v[kSlabKB / 4];
#pragma unroll
    for (int j = 0; j < kSlabKB / 4; ++j) {
        const int writer = (wg * 3 + j * 32 + 7) & (kWG - 1);
        v[j] = *reinterpret_cast<const f32x4*>(b.slab + ((size_t)writer * kSlabKB + 4 * (j & 7)) * 1024 + threadIdx.x * 16);
    }
    unsigned bad = 0;
#pragma unroll
    for (int j = 0; j < kSlabKB / 4; ++j) bad += (v[j].x != want) + (v[j].y != want) + (v[j].z != want) + (v[j].w != want);
    if (check && bad) atomicAdd(b.sync->errors, bad);
    const float out = (float)stage;
    if (threadIdx.x < kNormOutKB * 64) {
        char* dst = b.act + (size_t)wg * (kNormOutKB * 1024) + threadIdx.x * 16;
        const f32x4 o = {out, out, out, out};
        if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(o) : "memory");
        else *reinterpret_cast<f32x4*>(dst) = o;
    }
}

// ---- (k) one kernel per stage ------------------------------------------------------------------
__global__ __launch_bounds__(kT) void k_p(Bufs b, int stage) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    p_issue_weights(b, stage, blockIdx.x, lds);
    p_issue_act(b, blockIdx.x, lds + kWeightKB * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    p_consume_store<false>(b, stage, blockIdx.x, lds, lds + kWeightKB * 1024, stage > 0);
}
__global__ __launch_bounds__(kT) void k_n(Bufs b, int stage) { n_body<false>(b, stage, blockIdx.x, stage > 0); }

// ---- (b) one kernel per stage, two streams, counters instead of stream order ---------------------
// wait: done[0] >= stage * kWG (all workgroups of all earlier stages); signal: done[0] += 1 per workgroup.
template <bool SC1>
__device__ __forceinline__ bool flag_wait(const Bufs& b, int stage) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = poll_ge(b.sync->done, (unsigned)stage * kWG, b.sync) ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok != 0;
}
template <bool SC1>
__device__ __forceinline__ void flag_signal(const Bufs& b) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if constexpr (!SC1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        add_relaxed(b.sync->done, 1u);
    }
}
template <bool SC1>
__global__ __launch_bounds__(kT) void k_p_flag(Bufs b, int stage) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    p_issue_weights(b, stage, blockIdx.x, lds);
    const bool ok = flag_wait<SC1>(b, stage);
    if (ok) {
        p_issue_act(b, blockIdx.x, lds + kWeightKB * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        p_consume_store<SC1>(b, stage, blockIdx.x, lds, lds + kWeightKB * 1024, stage > 0);
    }
    flag_signal<SC1>(b);                                       // also on failure: later stages must not wait for this one
}
template <bool SC1>
__global__ __launch_bounds__(kT) void k_n_flag(Bufs b, int stage) {
    const bool ok = flag_wait<SC1>(b, stage);
    if (ok) n_body<SC1>(b, stage, blockIdx.x, stage > 0);
    flag_signal<SC1>(b);
}

// ---- (a) persistent kernel with an XCD-hierarchical barrier -----------------------------------------
struct BarState { unsigned xcc, n_mine, n_xcc; };

template <bool SC1 = false>
__device__ __forceinline__ bool xcd_barrier(Sync* sy, const BarState& st, unsigned g) {
    __shared__ int ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's stores have reached the XCD's L2
    __syncthreads();
    if (threadIdx.x == 0) {
        bool good = true;
        const unsigned old = add_relaxed(&sy->xcc_cnt[st.xcc][0], 1u);
        if (old + 1 == st.n_mine * g) {
            // last arrival of this XCD: one write-back of the XCD's dirty lines for all its workgroups (write-through
            // stores have left the L2 already)
            if constexpr (!SC1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            add_relaxed(sy->top, 1u);
            good = poll_ge(sy->top, st.n_xcc * g, sy);
            st_relaxed(&sy->gen[st.xcc][0], g);                 // release the XCD first, then this CU's own invalidate
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            good = poll_ge(&sy->gen[st.xcc][0], g, sy);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ok = good ? 1 : 0;
    }
    __syncthreads();
    return ok != 0;
}

template <bool PREFETCH, bool SC1 = false, bool NOBAR = false>
__global__ __launch_bounds__(kT) void k_persistent(Bufs b, int pairs) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    Sync* sy = b.sync;
    __shared__ BarState st;
    __shared__ int ok0;
    // census: how many workgroups share this XCD, how many XCDs take part (placement is not contractual)
    if (threadIdx.x == 0) {
        st.xcc = xcc_id();
        add_relaxed(&sy->census[st.xcc][0], 1u);
        add_relaxed(sy->flat, 1u);
        ok0 = poll_ge(sy->flat, gridDim.x, sy) ? 1 : 0;
        unsigned n = 0;
        for (int x = 0; x < 8; ++x) n += ld_relaxed(&sy->census[x][0]) != 0;
        st.n_xcc = n;
        st.n_mine = ld_relaxed(&sy->census[st.xcc][0]);
    }
    __syncthreads();
    if (!ok0) return;
    const BarState my = st;
    const int wg = blockIdx.x;
    auto wbuf = [&](int i) -> char* { return lds + (i & 1) * (kWeightKB * 1024); };
    char* const abuf = lds + 2 * kWeightKB * 1024;
    unsigned g = 0;
    if (PREFETCH) p_issue_weights(b, 1, wg, wbuf(0));           // stage 1 is the first product
    for (int i = 0; i < pairs; ++i) {
        const int sn = 2 * i, sp = 2 * i + 1;
        n_body<SC1>(b, sn, wg, !NOBAR && sn > 0);
        if (NOBAR) __syncthreads();
        else if (!xcd_barrier<SC1>(sy, my, ++g)) return;
        if (!PREFETCH) p_issue_weights(b, sp, wg, wbuf(i));
        p_issue_act(b, wg, abuf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        p_consume_store<SC1>(b, sp, wg, wbuf(i), abuf, !NOBAR);
        // the next product's weights do not depend on anything: issue them before the wait
        if (PREFETCH && i + 1 < pairs) p_issue_weights(b, sp + 2, wg, wbuf(i + 1));
        if (NOBAR) __syncthreads();
        else if (!xcd_barrier<SC1>(sy, my, ++g)) return;
    }
}

// ---- (c) graph with a fork and a join per pair: the product beside the norm it depends on ----------------
// The norm of pair i raises done[0] by one per workgroup; the product of pair i issues its weights, then waits for
// done[0] >= (i + 1) * kWG.  Both kernels depend (graph edges) on the previous pair's product.
__global__ __launch_bounds__(kT) void k_n_sig(Bufs b, int stage) {
    n_body<false>(b, stage, blockIdx.x, stage > 0);
    flag_signal<false>(b);
}
__global__ __launch_bounds__(kT) void k_p_wait(Bufs b, int stage) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    p_issue_weights(b, stage, blockIdx.x, lds);
    if (!flag_wait<false>(b, (stage + 1) / 2)) return;
    p_issue_act(b, blockIdx.x, lds + kWeightKB * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    p_consume_store<false>(b, stage, blockIdx.x, lds, lds + kWeightKB * 1024, true);
}

// ---- host ------------------------------------------------------------------------------------------
static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const int pairs = argc > 1 ? atoi(argv[1]) : 600;
    if (pairs < 2 || pairs > 20000) { fprintf(stderr, "pairs must be 2..20000\n"); return 1; }
    const int stages = 2 * pairs;
    hipStream_t st, st2;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    Bufs b;
    char* w;
    const size_t wbytes = (size_t)kRot * kWG * kWeightKB * 1024;
    CHECK(hipMalloc(&w, wbytes));
    CHECK(hipMemset(w, 0, wbytes));
    b.weights = w;
    CHECK(hipMalloc(&b.act, (size_t)kWG * kNormOutKB * 1024));
    CHECK(hipMalloc(&b.slab, (size_t)kWG * kSlabKB * 1024));
    CHECK(hipMalloc(&b.sync, sizeof(Sync)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int ldsK = (kWeightKB + kActKB) * 1024, ldsP = (2 * kWeightKB + kActKB) * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_p), hipFuncAttributeMaxDynamicSharedMemorySize, ldsK));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_p_flag<false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsK));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_p_flag<true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsK));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_persistent<false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsP));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_persistent<true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsP));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_persistent<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsP));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_persistent<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsP));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_p_wait), hipFuncAttributeMaxDynamicSharedMemorySize, ldsK));

    auto reset = [&](hipStream_t s) {
        CHECK(hipMemsetAsync(b.sync, 0, sizeof(Sync), s));
        CHECK(hipMemsetAsync(b.act, 0, (size_t)kWG * kNormOutKB * 1024, s));
        CHECK(hipMemsetAsync(b.slab, 0, (size_t)kWG * kSlabKB * 1024, s));
    };
    auto report = [&](const char* name, float best_ms) -> int {
        Sync h;
        CHECK(hipMemcpy(&h, b.sync, sizeof(Sync), hipMemcpyDeviceToHost));
        printf("%-78s %7.2f us per (norm + product) pair   stale/wrong words %u%s\n", name, best_ms * 1e3f / pairs, h.errors[0],
               h.failed[0] ? "   SPIN LIMIT HIT" : "");
        fflush(stdout);
        return h.failed[0] ? 2 : 0;
    };

    // ---- (k) graph of one kernel per stage
    {
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < pairs; ++i) {
            hipLaunchKernelGGL(k_n, dim3(kWG), dim3(kT), 0, st, b, 2 * i);
            hipLaunchKernelGGL(k_p, dim3(kWG), dim3(kT), ldsK, st, b, 2 * i + 1);
        }
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            reset(st);
            CHECK(hipEventRecord(e0, st));
            CHECK(hipGraphLaunch(exec, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
        if (report("(k)  one kernel per stage, one stream, graph replay", best)) return 2;
    }
    // ---- (a0), (a) persistent kernel
    for (int pf = 0; pf < 4; ++pf) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            reset(st);
            CHECK(hipEventRecord(e0, st));
            if (pf == 3) hipLaunchKernelGGL((k_persistent<true, false, true>), dim3(kWG), dim3(kT), ldsP, st, b, pairs);
            else if (pf == 2) hipLaunchKernelGGL((k_persistent<true, true>), dim3(kWG), dim3(kT), ldsP, st, b, pairs);
            else if (pf) hipLaunchKernelGGL(k_persistent<true>, dim3(kWG), dim3(kT), ldsP, st, b, pairs);
            else hipLaunchKernelGGL(k_persistent<false>, dim3(kWG), dim3(kT), ldsP, st, b, pairs);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        if (report(pf == 3 ? "(a-) the bodies of (a) with NO grid barrier (timing only: results unchecked)"
                   : pf == 2 ? "(a') as (a), write-through (sc1) stores, no write-back at the barrier"
                   : pf  ? "(a)  persistent kernel, XCD-hierarchical barrier, weights issued before the wait"
                      : "(a0) persistent kernel, XCD-hierarchical barrier, weights issued after the wait",
                   best))
            return 2;
    }
    // ---- (c) graph: fork / join per pair, the product launched beside the norm it waits for
    {
        hipGraph_t graph;
        hipGraphExec_t exec;
        hipEvent_t ef, ej;
        CHECK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
        CHECK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < pairs; ++i) {
            CHECK(hipEventRecord(ef, st));
            CHECK(hipStreamWaitEvent(st2, ef, 0));
            hipLaunchKernelGGL(k_n_sig, dim3(kWG), dim3(kT), 0, st, b, 2 * i);
            hipLaunchKernelGGL(k_p_wait, dim3(kWG), dim3(kT), ldsK, st2, b, 2 * i + 1);
            CHECK(hipEventRecord(ej, st2));
            CHECK(hipStreamWaitEvent(st, ej, 0));
        }
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            reset(st);
            CHECK(hipEventRecord(e0, st));
            CHECK(hipGraphLaunch(exec, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
        if (report("(c)  graph, fork + join per pair: product launched beside the norm, polls its counter", best)) return 2;
    }
    // ---- (b) two streams, counters: host wall clock over the whole chain (the two streams share no event)
    for (int sc1 = 0; sc1 < 2; ++sc1) {
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            reset(st);
            CHECK(hipStreamSynchronize(st));
            CHECK(hipStreamSynchronize(st2));
            const double t0 = now_us();
            for (int i = 0; i < pairs; ++i) {
                if (sc1) {
                    hipLaunchKernelGGL(k_n_flag<true>, dim3(kWG), dim3(kT), 0, st, b, 2 * i);
                    hipLaunchKernelGGL(k_p_flag<true>, dim3(kWG), dim3(kT), ldsK, st2, b, 2 * i + 1);
                } else {
                    hipLaunchKernelGGL(k_n_flag<false>, dim3(kWG), dim3(kT), 0, st, b, 2 * i);
                    hipLaunchKernelGGL(k_p_flag<false>, dim3(kWG), dim3(kT), ldsK, st2, b, 2 * i + 1);
                }
            }
            const double t_launch = now_us();
            CHECK(hipStreamSynchronize(st));
            CHECK(hipStreamSynchronize(st2));
            const double t1 = now_us();
            if (rep == 1) printf("     (host: %d launches queued in %.0f us = %.2f us each)\n", stages, t_launch - t0, (t_launch - t0) / stages);
            if (rep > 0 && t1 - t0 < best) best = t1 - t0;
        }
        if (report(sc1 ? "(b') two streams, counters, write-through (sc1) stores instead of a release fence"
                       : "(b)  two streams, counters, plain stores + release fence per workgroup",
                   (float)(best * 1e-3)))
            return 2;
    }
    return 0;
}
