// Attention of a few query rows per document against that document's cached keys/values
// (head_dim = 128).  Used for the denoiser's cross-attention (one query per document, keys = the
// zero-vector slot + every context position) and for the contextualizer's KV-cached causal
// self-attention (decode: one query; prefill: a short causal block).
//
// One wave per (query row, head).  Lane = (key group kg = lane / 16, dim group dg = lane % 16): it
// owns 8 of the 128 dims and walks keys kg, kg + 4, kg + 8, ... with 16-byte loads (a key or value
// row of one head is 256 contiguous bytes = one wave-quarter), keeps an online softmax (running
// max / sum / 8 accumulators) and never touches LDS; the 4 key groups are merged with two
// shuffles at the end.  The query may come straight from a split-K slab (sum of partial
// products + bias), which saves the finishing pass of the q projection.
#include "common.h"

namespace lcmhip {

constexpr int kHeadDim = 128;

template <typename T>
__global__ __launch_bounds__(256) void k_attention(AttnQ qa, const T* __restrict__ kc, const T* __restrict__ vc,
                                                   const int32_t* __restrict__ slot_map, T* __restrict__ out, int s,
                                                   int n_heads, int cap, int base_keys, int causal) {
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int heads_per_block = blockDim.x >> 6;
    const int hb = n_heads / heads_per_block;           // blocks per query row
    const int row = blockIdx.x / hb;                    // query row = doc * s + j
    const int head = (blockIdx.x % hb) * heads_per_block + wid;
    const int doc = row / s, j = row % s;
    const int slot = slot_map ? slot_map[doc] : doc;
    const int nkeys = base_keys + (causal ? j + 1 : 0);
    const int d = n_heads * kHeadDim;
    const int kg = lane >> 4, dg = lane & 15;
    const int col = head * kHeadDim + dg * 8;

    float qv[8];
    if (qa.partial != nullptr) {
#pragma unroll
        for (int t = 0; t < 8; ++t) qv[t] = 0.f;
#pragma unroll 4
        for (int sp = 0; sp < qa.splits; ++sp) {
            float t8[8];
            load8<float>(qa.partial + ((size_t)sp * qa.rows + row) * d + col, t8);
#pragma unroll
            for (int t = 0; t < 8; ++t) qv[t] += t8[t];
        }
        if (qa.bias != nullptr) {
            float b8[8];
            load8<T>(reinterpret_cast<const T*>(qa.bias) + col, b8);
#pragma unroll
            for (int t = 0; t < 8; ++t) qv[t] += b8[t];
        }
        // the reference rounds q to the operand dtype before the scores
        if (sizeof(T) == 2) {
#pragma unroll
            for (int t = 0; t < 8; ++t) qv[t] = bf16_to_f32(f32_to_bf16(qv[t]));
        }
    } else {
        load8<T>(reinterpret_cast<const T*>(qa.q) + (size_t)row * d + col, qv);
    }
    const float scale = 0.08838834764831845f;  // 128^-0.5
#pragma unroll
    for (int t = 0; t < 8; ++t) qv[t] *= scale;

    const T* kbase = kc + ((size_t)slot * cap) * d + col;
    const T* vbase = vc + ((size_t)slot * cap) * d + col;

    float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.f;

    auto dot16 = [&](const float (&kv)[8]) {
        float p = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) p += kv[t] * qv[t];
        p += __shfl_xor(p, 1, 64);
        p += __shfl_xor(p, 2, 64);
        p += __shfl_xor(p, 4, 64);
        p += __shfl_xor(p, 8, 64);
        return p;   // identical in the 16 lanes of the key group
    };
    auto absorb = [&](float sc, const float (&vv)[8]) {
        const float mn = fmaxf(m, sc);
        const float alpha = __expf(m - mn);
        const float p = __expf(sc - mn);
        l = l * alpha + p;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = acc[t] * alpha + p * vv[t];
        m = mn;
    };

    int key = kg;
    // two keys per lane per iteration: four independent 16-byte loads in flight
    for (; key + 4 < nkeys; key += 8) {
        float k0[8], k1[8], v0[8], v1[8];
        load8<T>(kbase + (size_t)key * d, k0);
        load8<T>(kbase + (size_t)(key + 4) * d, k1);
        load8<T>(vbase + (size_t)key * d, v0);
        load8<T>(vbase + (size_t)(key + 4) * d, v1);
        const float s0 = dot16(k0), s1 = dot16(k1);
        absorb(s0, v0);
        absorb(s1, v1);
    }
    // tail: the shuffles inside dot16 need every lane of the 16-lane group, and all lanes of a group
    // share `key`, so a whole group is either in or out
    if (key < nkeys) {
        float k0[8], v0[8];
        load8<T>(kbase + (size_t)key * d, k0);
        load8<T>(vbase + (size_t)key * d, v0);
        const float s0 = dot16(k0);
        absorb(s0, v0);
    }
    // merge the four key groups (lanes differing in bits 4 and 5)
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const float mo = __shfl_xor(m, off, 64);
        const float lo = __shfl_xor(l, off, 64);
        const float mn = fmaxf(m, mo);
        const float a = (m == -INFINITY) ? 0.f : __expf(m - mn);
        const float b = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
        l = l * a + lo * b;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float ao = __shfl_xor(acc[t], off, 64);
            acc[t] = acc[t] * a + ao * b;
        }
        m = mn;
    }
    if (kg == 0) {
        const float inv = 1.f / l;
        float o[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = acc[t] * inv;
        store8<T>(out + (size_t)row * d + col, o);
    }
}

int launch_attention_q(const AttnQ& qa, const void* kc, const void* vc, const int32_t* slot_map, void* out, int n_docs,
                       int s, int n_heads, int cap, int base_keys, int causal, int dtype, hipStream_t st) {
    LCM_REQUIRE(n_heads % 4 == 0 || n_heads == 2, "attention: heads must be a multiple of 4 (or 2)");
    LCM_REQUIRE(n_docs >= 0 && s >= 1 && base_keys >= 0, "attention: bad shape");
    const int max_keys = base_keys + (causal ? s : 0);
    LCM_REQUIRE(max_keys >= 1 && max_keys <= cap, "attention: key range exceeds the cache");
    LCM_REQUIRE((qa.q != nullptr) != (qa.partial != nullptr), "attention: exactly one query source");
    if (n_docs == 0) return LCMHIP_OK;
    const int waves = n_heads >= 4 ? 4 : n_heads;
    dim3 grid((unsigned)(n_docs * s * (n_heads / waves))), block(64 * waves);
    if (dtype == LCMHIP_BF16)
        hipLaunchKernelGGL((k_attention<bf16_t>), grid, block, 0, st, qa, (const bf16_t*)kc, (const bf16_t*)vc, slot_map,
                           (bf16_t*)out, s, n_heads, cap, base_keys, causal);
    else if (dtype == LCMHIP_F32)
        hipLaunchKernelGGL((k_attention<float>), grid, block, 0, st, qa, (const float*)kc, (const float*)vc, slot_map,
                           (float*)out, s, n_heads, cap, base_keys, causal);
    else
        return fail(LCMHIP_ERR_INVALID, "attention: unknown dtype");
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

int launch_attention(const void* q, const void* kc, const void* vc, const int32_t* slot_map, void* out, int n_docs,
                     int s, int n_heads, int cap, int base_keys, int causal, int dtype, hipStream_t st) {
    AttnQ qa{};
    qa.q = q;
    return launch_attention_q(qa, kc, vc, slot_map, out, n_docs, s, n_heads, cap, base_keys, causal, dtype, st);
}

}  // namespace lcmhip
