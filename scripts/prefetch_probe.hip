// Would the weight-streaming products of the sampling loop run faster if their weights were already in the 256 MiB
// Infinity Cache when they start -- pulled there by a small reader kernel on a forked branch of the graph while the
// previous product runs?  The probe replays, from one HIP graph, a chain of the library's own products with the shapes of
// a denoiser layer at 128 rows (three d x d, the SwiGLU up-projection, the down-projection; d = 2048, 5632 inner),
//   (a) over `layers` distinct layers of weights (13: 1.2 GB, nothing survives in the cache from one use to the next),
//   (b) over ONE layer's weights replayed (94 MB: resident -- the bound on what any prefetcher could give),
//   (c) as (a) with a reader kernel of `wg` workgroups forked before product i that reads the weights of product i + 1
//       and is joined before product i + 1 starts.
//
// build: hipcc --offload-arch=gfx950 -O3 scripts/prefetch_probe.hip -Iinclude -Lchimera-llm_amd/lcm -llcm_hip \
//              -Wl,-rpath,'$ORIGIN/../chimera-llm_amd/lcm' -o scripts/prefetch_probe
// run:   scripts/prefetch_probe [layers=13] [reps=20]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lcmhip.h"

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)
#define LCM(x)                                                                                     \
    do {                                                                                           \
        int r_ = (x);                                                                              \
        if (r_ != 0) {                                                                             \
            fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #x, r_, lcmhip_last_error()); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

// Reads `n_vec` 16-byte pieces and keeps nothing (the store can never happen: the weights are finite).
__global__ __launch_bounds__(256) void k_touch(const uint4* __restrict__ p, size_t n_vec, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc |= (v.x & v.y & v.z & v.w);
    }
    if (acc == 0xffffffffu) *sink = acc;
}

struct Product { const void *W, *W2; int N, K, epi; size_t bytes; };

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 13;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    if (layers < 1 || layers > 16 || reps < 2 || reps > 1000) { fprintf(stderr, "layers 1..16, reps 2..1000\n"); return 1; }
    const int M = 128, d = 2048, F = 5632;
    hipStream_t st, side;
    CHECK(hipStreamCreate(&st));
    CHECK(hipStreamCreate(&side));
    // weights: small finite bf16 values (0x3c00 = 2^-7)
    const size_t per_layer = (size_t)3 * d * d + (size_t)3 * F * d;
    uint16_t* w;
    CHECK(hipMalloc(&w, per_layer * layers * 2));
    {
        std::vector<uint16_t> h(per_layer, 0x3c00);
        for (int l = 0; l < layers; ++l) CHECK(hipMemcpy(w + per_layer * l, h.data(), per_layer * 2, hipMemcpyHostToDevice));
    }
    uint16_t *a_d, *a_f, *out;
    CHECK(hipMalloc(&a_d, (size_t)M * F * 2));
    CHECK(hipMemset(a_d, 0, (size_t)M * F * 2));
    a_f = a_d;
    CHECK(hipMalloc(&out, (size_t)M * F * 2));
    float* ws;
    const size_t ws_bytes = 64u << 20;
    CHECK(hipMalloc(&ws, ws_bytes));
    unsigned* sink;
    CHECK(hipMalloc(&sink, 256));

    auto products = [&](int n_layers_distinct) {
        std::vector<Product> v;
        for (int l = 0; l < layers; ++l) {
            const uint16_t* base = w + per_layer * (l % n_layers_distinct);
            for (int j = 0; j < 3; ++j) v.push_back({base + (size_t)j * d * d, nullptr, d, d, LCMHIP_EPI_STORE, (size_t)d * d * 2});
            const uint16_t* f = base + (size_t)3 * d * d;
            v.push_back({f, f + (size_t)F * d, F, d, LCMHIP_EPI_SWIGLU, (size_t)2 * F * d * 2});
            v.push_back({f + (size_t)2 * F * d, nullptr, d, F, LCMHIP_EPI_STORE, (size_t)F * d * 2});
        }
        return v;
    };
    auto launch = [&](const Product& p) {
        LCM(lcmhip_gemm(p.K == d ? a_d : a_f, p.K, p.W, p.W2, nullptr, nullptr, nullptr, out, p.N, M, p.N, p.K, p.epi,
                        LCMHIP_BF16, ws, ws_bytes, st));
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<hipEvent_t> ev(2 * 5 * layers + 2);
    for (auto& e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

    auto run = [&](const char* what, const std::vector<Product>& v, int touch_wg) {
        // eager once (module load, first-use attributes), then capture
        for (const auto& p : v) launch(p);
        CHECK(hipStreamSynchronize(st));
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        size_t n_ev = 0;
        bool pending_join = false;
        hipEvent_t join_ev{};
        for (size_t i = 0; i < v.size(); ++i) {
            if (touch_wg > 0 && i + 1 < v.size()) {
                // fork: the reader of product i + 1's weights starts when product i - 1 is done, beside product i
                hipEvent_t f = ev[n_ev++];
                CHECK(hipEventRecord(f, st));
                CHECK(hipStreamWaitEvent(side, f, 0));
                hipLaunchKernelGGL(k_touch, dim3(touch_wg), dim3(256), 0, side, (const uint4*)v[i + 1].W, v[i + 1].bytes / 16, sink);
                hipEvent_t j = ev[n_ev++];
                CHECK(hipEventRecord(j, side));
                launch(v[i]);
                if (pending_join) {}
                CHECK(hipStreamWaitEvent(st, j, 0));   // product i + 1 starts after its reader
                pending_join = true;
                join_ev = j;
            } else {
                launch(v[i]);
            }
        }
        (void)join_ev;
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float best = 1e30f, sum = 0.f;
        for (int r = 0; r < reps; ++r) {
            CHECK(hipEventRecord(e0, st));
            CHECK(hipGraphLaunch(exec, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
        size_t bytes = 0;
        for (const auto& p : v) bytes += p.bytes;
        printf("%-58s %7.1f us per replay (best %7.1f), %5.2f us per product, %.2f TB/s of weights\n", what,
               sum / (reps - 2) * 1e3f, best * 1e3f, sum / (reps - 2) * 1e3f / v.size(), bytes / (sum / (reps - 2) * 1e-3) / 1e12);
        fflush(stdout);
    };

    const auto distinct = products(layers);
    const auto resident = products(1);
    printf("%d layers x 5 products at %d rows, d = %d, inner %d\n", layers, M, d, F);
    run("(a) distinct weights per layer (HBM)", distinct, 0);
    run("(b) one layer's weights replayed (cache-resident bound)", resident, 0);
    for (int wg : {32, 64, 128, 256, 512}) {
        char name[96];
        snprintf(name, sizeof name, "(c) distinct + reader of the next weights, %3d workgroups", wg);
        run(name, distinct, wg);
    }
    run("(a) again", distinct, 0);
    return 0;
}
