// What does one dependent stage cost on this GPU when the stages are (a) separate kernels replayed from a HIP graph,
// (b) the iterations of ONE persistent kernel separated by a grid-wide barrier?  The sampling loop of the two-tower
// model is a chain of about 120 dependent small stages per diffusion step; this probe prices the two ways of
// expressing such a chain, with the same per-stage work: every workgroup reads `bytes_per_wg` of a large buffer
// (a weight slice), reduces it and writes one float that the next stage's workgroups read (the dependency).
//
// build: hipcc --offload-arch=gfx950 -O3 scripts/barrier_probe.hip -o scripts/barrier_probe
// run:   scripts/barrier_probe [workgroups=256] [stages=1200]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

// One stage of one workgroup: 256 threads x `vec_per_thread` float4 from its slice, plus the previous stage's outputs.
// Stage s reads slice s % kRot of the buffer (16 slices: 512 MB at 128 KB per workgroup, more than the caches hold).
constexpr int kRot = 16;
__device__ __forceinline__ float stage_work(const float4* __restrict__ w, const float* prev, int n_wg, int wg, int stage,
                                            int vec_per_thread, size_t slice_vec) {
    const float4* src = w + (size_t)(stage % kRot) * slice_vec + (size_t)wg * 256 * vec_per_thread + threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < vec_per_thread; ++i) {
        const float4 v = src[(size_t)i * 256];
        acc += v.x + v.y + v.z + v.w;
    }
    // the dependency: every workgroup reads what all workgroups of the previous stage wrote (fresh from L2)
    if (prev != nullptr) {
        for (int i = threadIdx.x; i < n_wg; i += 256) acc += __builtin_nontemporal_load(prev + i);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float r = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_stage(const float4* w, const float* prev, float* next, int stage, int vec_per_thread,
                                               size_t slice_vec) {
    const float r = stage_work(w, prev, gridDim.x, blockIdx.x, stage, vec_per_thread, slice_vec);
    if (threadIdx.x == 0) next[blockIdx.x] = r;
}

// All stages in one launch.  Grid barrier: a monotonic arrival counter; thread 0 of every workgroup adds 1 (release) and
// spins (acquire) until all workgroups of the stage have arrived.  The spin is bounded: after `spin_limit` polls the
// workgroup raises `*failed` and every workgroup leaves at its next poll, so the grid always drains.
__global__ __launch_bounds__(256) void k_persistent(const float4* w, float* bufs, unsigned* counter, unsigned* failed, int stages,
                                                    int vec_per_thread, size_t slice_vec, int with_work,
                                                    unsigned spin_limit) {
    const int n_wg = gridDim.x;
    __shared__ int stop;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        float* next = bufs + (size_t)(s & 1) * n_wg;
        const float* prev = s > 0 ? bufs + (size_t)((s - 1) & 1) * n_wg : nullptr;
        if (with_work) {
            const float r = stage_work(w, prev, n_wg, blockIdx.x, s, vec_per_thread, slice_vec);
            if (threadIdx.x == 0) next[blockIdx.x] = r;
        }
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(s + 1) * (unsigned)n_wg;
            unsigned polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++polls > spin_limit || __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    stop = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (stop) return;
    }
}

int main(int argc, char** argv) {
    const int n_wg = argc > 1 ? atoi(argv[1]) : 256;
    const int stages = argc > 2 ? atoi(argv[2]) : 1200;
    if (n_wg < 1 || n_wg > 256 || stages < 2 || stages > 100000) {
        fprintf(stderr, "workgroups must be 1..256 (one per CU: all co-resident), stages 2..100000\n");
        return 1;
    }
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const int kinds[3] = {0, 8, 32};          // KB read per workgroup and stage: 0 (no work), 32 KB, 128 KB
    const size_t slice_vec = (size_t)n_wg * 256 * 32;
    float4* w;
    CHECK(hipMalloc(&w, slice_vec * kRot * sizeof(float4)));
    CHECK(hipMemset(w, 0, slice_vec * kRot * sizeof(float4)));
    float* bufs;
    CHECK(hipMalloc(&bufs, 2 * 256 * sizeof(float)));
    CHECK(hipMemset(bufs, 0, 2 * 256 * sizeof(float)));
    unsigned *counter, *failed;
    CHECK(hipMalloc(&counter, 256));
    failed = counter + 32;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    for (int ki = 0; ki < 3; ++ki) {
        const int vpt = kinds[ki];
        // ---- (a) one kernel per stage, captured once and replayed
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int s = 0; s < stages; ++s) {
            float* next = bufs + (size_t)(s & 1) * 256;
            const float* prev = s > 0 ? bufs + (size_t)((s - 1) & 1) * 256 : nullptr;
            hipLaunchKernelGGL(k_stage, dim3(n_wg), dim3(256), 0, st, (const float4*)w, prev, next, s, vpt, slice_vec);
        }
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float best_graph = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0, st));
            CHECK(hipGraphLaunch(exec, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best_graph) best_graph = ms;
        }
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
        // ---- (a') the same kernels launched one by one into the stream (no graph): GPU time between two events; the
        // host queues them faster than they run only if a launch call costs less than a stage
        float best_eager = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, st));
            for (int s = 0; s < stages; ++s) {
                float* next = bufs + (size_t)(s & 1) * 256;
                const float* prev = s > 0 ? bufs + (size_t)((s - 1) & 1) * 256 : nullptr;
                hipLaunchKernelGGL(k_stage, dim3(n_wg), dim3(256), 0, st, (const float4*)w, prev, next, s, vpt, slice_vec);
            }
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best_eager) best_eager = ms;
        }
        // ---- (b) one persistent kernel with a grid barrier per stage (the stages read slice s % rot as above)
        float best_pers = 1e30f;
        unsigned h_failed = 0;
        for (int rep = 0; rep < 5 && !h_failed; ++rep) {
            CHECK(hipMemsetAsync(counter, 0, 256, st));
            CHECK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_persistent, dim3(n_wg), dim3(256), 0, st, (const float4*)w, bufs, counter, failed, stages, vpt,
                               slice_vec, vpt > 0 ? 1 : 0, 4000000u);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(&h_failed, failed, 4, hipMemcpyDeviceToHost));
            if (rep > 0 && ms < best_pers) best_pers = ms;
        }
        if (h_failed) {
            printf("%3d KB per workgroup and stage: persistent kernel gave up at a barrier (spin limit)\n", vpt * 4);
            return 2;
        }
        printf("%3d workgroups, %4d KB read per workgroup and stage, %d stages: kernel per stage: graph replay %.2f us/stage, "
               "stream launches %.2f us/stage; persistent kernel + grid barrier %.2f us/stage\n",
               n_wg, vpt * 4, stages, best_graph * 1e3f / stages, best_eager * 1e3f / stages, best_pers * 1e3f / stages);
        fflush(stdout);
    }
    return 0;
}
