// lcmhip_gemm_static: lcmhip_gemm for callers that must not allocate device memory (code under graph capture,
// tight serving loops, harnesses that forbid hipMalloc).  The split-K slab that lcmhip_gemm takes as a caller
// workspace lives in a file-scope __device__ array instead; every GPU the library runs on holds its own copy of
// that array (LCMHIP_STATIC_SCRATCH_BYTES) until the process ends.
#include <mutex>

#include "common.h"

namespace lcmhip {

// Split-K partial tiles [split][M][N] fp32.  Every element a launch reads was written by the same launch's
// producing kernel, so nothing here has to start at zero and stale contents of an earlier call are harmless.
static __device__ __align__(256) unsigned char scratch[LCMHIP_STATIC_SCRATCH_BYTES];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[kMaxDevices] = {};   // device address of `scratch` on each GPU (the array's name is only
                                                 // a host-side stand-in: never pass it to a kernel or a memset)

// Address of the current device's copy of `scratch`, resolved once per device.
static int scratch_for_current_device(void** out) {
    int dev = -1;
    LCM_HIP_CHECK(hipGetDevice(&dev));
    LCM_REQUIRE(dev >= 0 && dev < kMaxDevices, "gemm_static: device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        LCM_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        LCM_REQUIRE(p != nullptr && ((uintptr_t)p % 256) == 0, "gemm_static: bad scratch symbol address");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return LCMHIP_OK;
}

// Called by lcmhip_model_create / lcmhip_session_create (and lcmhip_static_init): the first resolution on a device
// loads this file's code object and places the array, which must not happen inside a stream capture.
int static_scratch_init() {
    void* p = nullptr;
    return scratch_for_current_device(&p);
}

}  // namespace lcmhip

using namespace lcmhip;

extern "C" int lcmhip_static_init(void) { return static_scratch_init(); }

extern "C" int lcmhip_gemm_static(const void* A, int32_t lda, const void* W, const void* W2, const void* bias,
                                  const void* bias2, const float* gate, void* out, int32_t ldo, int32_t M, int32_t N,
                                  int32_t K, int32_t epilogue, int32_t dtype, void* stream) {
    LCM_REQUIRE(A && W && out, "gemm_static: null argument");
    LCM_REQUIRE(M >= 1 && N >= 1 && K >= 1, "gemm_static: empty problem");
    // at least two split-K partial tiles must fit, otherwise the product is too large for the static scratch
    LCM_REQUIRE((size_t)M * (size_t)N * 4 * 2 <= (size_t)LCMHIP_STATIC_SCRATCH_BYTES,
                "gemm_static: M * N exceeds LCMHIP_STATIC_MAX_OUTPUTS (use lcmhip_gemm with a caller workspace)");
    void* ws = nullptr;
    LCM_TRY(scratch_for_current_device(&ws));
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = W; a.W2 = W2; a.bias = bias; a.bias2 = bias2; a.gate = gate;
    a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.epilogue = epilogue & 0xff; a.dtype = dtype;
    a.act = (epilogue >> 8) & 0xff;
    LCM_REQUIRE((epilogue >> 16) == 0 && a.act <= LCMHIP_ACT_HARDSIGMOID && (a.act == 0 || a.epilogue == LCMHIP_EPI_RELU),
                "gemm_static: an activation code goes with LCMHIP_EPI_RELU only");
    a.partial = static_cast<float*>(ws); a.partial_bytes = LCMHIP_STATIC_SCRATCH_BYTES; a.max_splits = kMaxSplits;
    return launch_gemm(a, static_cast<hipStream_t>(stream));
}
