// Attention of a few query rows per document against that document's cached keys/values
// (head_dim = 128).  Used for the denoiser's cross-attention (one query per document, keys = the
// zero-vector slot + every context position) and for the contextualizer's KV-cached causal
// self-attention (decode: one query; prefill: a short causal block).
//
// One wave per (query row, head); the workgroup holds 4 heads of one query row.  Pass 1: lane = key
// (each lane streams its key row in 16-byte pieces against the query broadcast from LDS);
// pass 2: softmax over the wave; pass 3: lane = 2 output dims, keys walked with the
// probabilities broadcast from LDS (256 contiguous bytes of V per wave-load).
#include "common.h"

namespace lcmhip {

constexpr int kHeadDim = 128;

template <typename T>
__global__ __launch_bounds__(256) void k_attention(const T* __restrict__ q, const T* __restrict__ kc,
                                                   const T* __restrict__ vc, const int32_t* __restrict__ slot_map,
                                                   T* __restrict__ out, int s, int n_heads, int cap, int base_keys,
                                                   int causal, int max_keys) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int heads_per_block = blockDim.x >> 6;
    const int hb = n_heads / heads_per_block;           // blocks per query row
    const int row = blockIdx.x / hb;                    // query row = doc * s + j
    const int head = (blockIdx.x % hb) * heads_per_block + wid;
    const int doc = row / s, j = row % s;
    const int slot = slot_map ? slot_map[doc] : doc;
    const int nkeys = base_keys + (causal ? j + 1 : 0);
    const int d = n_heads * kHeadDim;

    float* qs = smem + wid * (kHeadDim + max_keys);     // query, then scores/probabilities
    float* sc = qs + kHeadDim;

    // stage the query (fp32) for broadcast reads
    {
        const T* qp = q + (size_t)row * d + head * kHeadDim + lane * 2;
        qs[lane * 2] = Elem<T>::load(qp);
        qs[lane * 2 + 1] = Elem<T>::load(qp + 1);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): own LDS writes visible to the wave

    const T* kbase = kc + ((size_t)slot * cap) * d + head * kHeadDim;
    const T* vbase = vc + ((size_t)slot * cap) * d + head * kHeadDim;
    const float scale = 0.08838834764831845f;  // 128^-0.5

    float mx = -INFINITY;
    for (int key = lane; key < nkeys; key += 64) {
        const T* kr = kbase + (size_t)key * d;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < kHeadDim; c += 8) {
            float kv[8];
            load8<T>(kr + c, kv);
#pragma unroll
            for (int t = 0; t < 8; ++t) acc += kv[t] * qs[c + t];
        }
        acc *= scale;
        sc[key] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int key = lane; key < nkeys; key += 64) {
        const float p = __expf(sc[key] - mx);
        sc[key] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);

    float o0 = 0.f, o1 = 0.f;
    const T* vp = vbase + lane * 2;
    int key = 0;
    for (; key + 4 <= nkeys; key += 4) {
        float a0[4], a1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a0[t] = Elem<T>::load(vp + (size_t)(key + t) * d);
            a1[t] = Elem<T>::load(vp + (size_t)(key + t) * d + 1);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float p = sc[key + t];
            o0 += p * a0[t];
            o1 += p * a1[t];
        }
    }
    for (; key < nkeys; ++key) {
        const float p = sc[key];
        o0 += p * Elem<T>::load(vp + (size_t)key * d);
        o1 += p * Elem<T>::load(vp + (size_t)key * d + 1);
    }
    const float inv = 1.f / sum;
    T* op = out + (size_t)row * d + head * kHeadDim + lane * 2;
    Elem<T>::store(op, o0 * inv);
    Elem<T>::store(op + 1, o1 * inv);
}

int launch_attention(const void* q, const void* kc, const void* vc, const int32_t* slot_map, void* out, int n_docs,
                     int s, int n_heads, int cap, int base_keys, int causal, int dtype, hipStream_t st) {
    LCM_REQUIRE(n_heads % 4 == 0 || n_heads == 2, "attention: heads must be a multiple of 4 (or 2)");
    LCM_REQUIRE(n_docs >= 0 && s >= 1 && base_keys >= 0, "attention: bad shape");
    const int max_keys = base_keys + (causal ? s : 0);
    LCM_REQUIRE(max_keys >= 1 && max_keys <= cap, "attention: key range exceeds the cache");
    if (n_docs == 0) return LCMHIP_OK;
    const int waves = n_heads >= 4 ? 4 : n_heads;
    const size_t lds = (size_t)waves * (kHeadDim + max_keys) * sizeof(float);
    LCM_REQUIRE(lds <= 64 * 1024, "attention: more than ~3900 keys per query is not supported by this kernel");
    dim3 grid((unsigned)(n_docs * s * (n_heads / waves))), block(64 * waves);
    if (dtype == LCMHIP_BF16)
        hipLaunchKernelGGL((k_attention<bf16_t>), grid, block, lds, st, (const bf16_t*)q, (const bf16_t*)kc,
                           (const bf16_t*)vc, slot_map, (bf16_t*)out, s, n_heads, cap, base_keys, causal, max_keys);
    else if (dtype == LCMHIP_F32)
        hipLaunchKernelGGL((k_attention<float>), grid, block, lds, st, (const float*)q, (const float*)kc,
                           (const float*)vc, slot_map, (float*)out, s, n_heads, cap, base_keys, causal, max_keys);
    else
        return fail(LCMHIP_ERR_INVALID, "attention: unknown dtype");
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

}  // namespace lcmhip
