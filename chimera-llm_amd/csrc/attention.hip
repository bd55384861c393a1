// Attention of a few query rows per document against that document's cached keys/values
// (head_dim = 128).  Used for the denoiser's cross-attention (one query per document, keys = the
// zero-vector slot + every context position) and for the contextualizer's KV-cached causal
// self-attention (decode: one query; prefill: a short causal block).
//
// Cache layout: [slot][head][position][128] -- the keys (values) of one (document, head) form one
// contiguous stream.  One wave per (query row, head), 4 heads per workgroup.  Lane = (key slot
// ks = lane / 4, dim group dg = lane % 4): the wave walks 16 keys at a time, the 4 lanes of a slot each
// own 32 of the 128 dims as four 16-byte pieces interleaved so that one load instruction reads 64
// contiguous bytes per key (4 KB contiguous per 4 instructions).  Every lane keeps an online softmax
// (running max / sum / 32 accumulators) over its keys ks, ks + 16, ... -- no LDS -- and the 16 slots
// are merged with shuffles at the end.  Two keys per lane are in flight per iteration (16 independent
// 16-byte loads).  The query may come straight from a split-K slab (sum of partial products + bias),
// which saves the finishing pass of the q projection.
#include <cstdlib>

#include "common.h"

namespace lcmhip {

constexpr int kHeadDim = 128;

// Eight consecutive elements as loaded (16 bytes of bf16, 32 bytes of f32) and their fp32 values.
template <typename T>
struct PieceT;
template <>
struct PieceT<bf16_t> { using type = __attribute__((ext_vector_type(4))) uint32_t; };
template <>
struct PieceT<float> { using type = __attribute__((ext_vector_type(8))) float; };   // native vectors stay in registers
template <typename T>
using Piece = typename PieceT<T>::type;

template <typename T>
__device__ __forceinline__ Piece<T> load_piece(const T* p);
template <>
__device__ __forceinline__ Piece<bf16_t> load_piece<bf16_t>(const bf16_t* p) {
    return *reinterpret_cast<const Piece<bf16_t>*>(p);
}
template <>
__device__ __forceinline__ Piece<float> load_piece<float>(const float* p) {
    return *reinterpret_cast<const Piece<float>*>(p);   // 32-byte aligned: pieces start at multiples of 8 floats
}
template <typename T>
__device__ __forceinline__ void unpack8(const Piece<T>& r, float (&v)[8]);
template <>
__device__ __forceinline__ void unpack8<bf16_t>(const Piece<bf16_t>& r, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __uint_as_float(r[j] << 16);
        v[2 * j + 1] = __uint_as_float(r[j] & 0xffff0000u);
    }
}
template <>
__device__ __forceinline__ void unpack8<float>(const Piece<float>& r, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = r[j];
}

// Sum of v[0..31] over the 16 key slots of a wave (lane bits 2..5) as a reduce-scatter: every round exchanges the half
// of the values this lane does NOT keep (30 shuffles instead of 128 for a butterfly over all 32 values).  Afterwards
// v[x], x < 2, holds the complete sum of original index 2 * (lane / 4) + x.
__device__ __forceinline__ void slot_reduce_scatter(float (&v)[32], int lane) {
#pragma unroll
    for (int half = 16, off = 32; half >= 2; half >>= 1, off >>= 1) {
        const bool up = (lane & off) != 0;
#pragma unroll
        for (int j = 0; j < half; ++j) {
            const float lo = v[j], hi = v[j + half];
            const float recv = __shfl_xor(up ? lo : hi, off, 64);
            v[j] = (up ? hi : lo) + recv;
        }
    }
}
// The inverse exchange: v[x], x < 2, of every slot -> all 32 values in every slot.
__device__ __forceinline__ void slot_all_gather(float (&v)[32], int lane) {
#pragma unroll
    for (int n = 2, off = 4; n <= 16; n <<= 1, off <<= 1) {
        const bool up = (lane & off) != 0;
#pragma unroll
        for (int j = 0; j < n; ++j) {
            const float a = v[j];
            const float r = __shfl_xor(a, off, 64);
            v[j] = up ? r : a;
            v[j + n] = up ? a : r;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_attention(AttnQ qa, const T* __restrict__ kc, const T* __restrict__ vc,
                                                   const int32_t* __restrict__ slot_map, T* __restrict__ out, int s,
                                                   int n_heads, int cap, int base_keys, int causal,
                                                   const int32_t* __restrict__ base_keys_dev, KeyWindow win) {
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int heads_per_block = blockDim.x >> 6;
    const int hb = n_heads / heads_per_block;           // blocks per query row
    const int row = blockIdx.x / hb;                    // query row = doc * s + j
    const int head = (blockIdx.x % hb) * heads_per_block + wid;
    const int doc = row / s, j = row % s;
    const int slot = slot_map ? slot_map[doc] : doc;
    // a launch recorded in a graph reads the key count from device memory: it changes with every appended position
    int nkeys = (base_keys_dev ? *base_keys_dev : base_keys) + (causal ? j + 1 : 0);
    if (win.max_keys > 0) {   // per-document window (shifted causal mask + key padding)
        if (win.shift) nkeys += win.shift[doc];
        nkeys = max(1, min(nkeys, win.limit ? min(win.limit[doc], win.max_keys) : win.max_keys));
    }
    const int d = n_heads * kHeadDim;
    const int ks = lane >> 2, dg = lane & 3;
    const int col = head * kHeadDim + dg * 8;           // piece i of this lane starts at col + 32 * i

    // Loads are requested in the order their data is needed (the memory counter retires in order): the query's own
    // operands, then the first round of keys / values, which so stays in flight while the query is assembled.
    constexpr int U = sizeof(T) == 2 ? 4 : 2;           // keys per slot and round (16 * U keys per wave)
    const size_t hbase = ((size_t)slot * n_heads + head) * cap * kHeadDim + dg * 8;
    const T* kbase = kc + hbase;
    const T* vbase = vc + hbase;
    float qv[4][8], b8[4][8];
    const bool from_slab = qa.partial != nullptr;
    const bool has_bias = from_slab && qa.bias != nullptr;
    if (from_slab) {
        // split-K slab of the q projection: key slot ks sums splits ks, ks + 16, ... for this lane's 32 dims (one
        // round of loads for up to 16 splits); the bias for every lane (slot 0 alone adds it), in the same round trip
#pragma unroll
        for (int i = 0; i < 4; ++i)
            load8<T>(reinterpret_cast<const T*>(has_bias ? qa.bias : (const void*)kc) + col + 32 * i, b8[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 8; ++t) qv[i][t] = 0.f;
        if (ks < qa.splits) {
            const float* src = qa.partial + ((size_t)ks * qa.rows + row) * d + col;
#pragma unroll
            for (int i = 0; i < 4; ++i) load8<float>(src + 32 * i, qv[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) load8<T>(reinterpret_cast<const T*>(qa.q) + (size_t)row * d + col + 32 * i, qv[i]);
    }
    Piece<T> kp0[U][4], vp0[U][4];
    const bool first_full = ks + 16 * (U - 1) < nkeys;  // this slot takes part in a first full round
    if (first_full) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) kp0[u][i] = load_piece<T>(kbase + (size_t)(ks + 16 * u) * kHeadDim + 32 * i);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) vp0[u][i] = load_piece<T>(vbase + (size_t)(ks + 16 * u) * kHeadDim + 32 * i);
    }
    if (from_slab) {
        for (int sp = ks + 16; sp < qa.splits; sp += 16) {   // more than 16 splits (the callers cap them well below)
            const float* src = qa.partial + ((size_t)sp * qa.rows + row) * d + col;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t8[8];
                load8<float>(src + 32 * i, t8);
#pragma unroll
                for (int t = 0; t < 8; ++t) qv[i][t] += t8[t];
            }
        }
        {
            const float on = (has_bias && ks == 0) ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 8; ++t) qv[i][t] = fmaf(b8[i][t], on, qv[i][t]);
        }
        // the 16 slots are added (reduce-scatter + all-gather), which leaves the same bits in every slot
        {
            float flat[32];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 8; ++t) flat[8 * i + t] = qv[i][t];
            slot_reduce_scatter(flat, lane);
            slot_all_gather(flat, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 8; ++t) qv[i][t] = flat[8 * i + t];
        }
        // the reference rounds q to the operand dtype before the scores
        if (sizeof(T) == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 8; ++t) qv[i][t] = bf16_to_f32(f32_to_bf16(qv[i][t]));
        }
    }
    const float scale = 0.08838834764831845f;  // 128^-0.5
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t) qv[i][t] *= scale;

    float m = -INFINITY, l = 0.f, acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[i][t] = 0.f;

    // score of one key: partial dot over this lane's 32 dims, summed over the slot's 4 lanes
    auto score = [&](const Piece<T> (&kp)[4]) {
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float kv[8];
            unpack8<T>(kp[i], kv);
#pragma unroll
            for (int t = 0; t < 8; ++t) p += kv[t] * qv[i][t];
        }
        p += __shfl_xor(p, 1, 64);
        p += __shfl_xor(p, 2, 64);
        return p;
    };
    auto absorb = [&](float sc, const Piece<T> (&vp)[4]) {
        const float mn = fmaxf(m, sc);
        const float alpha = __expf(m - mn);
        const float p = __expf(sc - mn);
        l = l * alpha + p;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float vv[8];
            unpack8<T>(vp[i], vv);
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[i][t] = acc[i][t] * alpha + p * vv[t];
        }
        m = mn;
    };

    // U keys per slot and iteration (16 * U keys per wave), kept packed in registers until they are used
    int key = ks;
    if (first_full) {   // the round requested above
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) sc[u] = score(kp0[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) absorb(sc[u], vp0[u]);
        key += 16 * U;
    }
    for (; key + 16 * (U - 1) < nkeys; key += 16 * U) {
        Piece<T> kp[U][4], vp[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) kp[u][i] = load_piece<T>(kbase + (size_t)(key + 16 * u) * kHeadDim + 32 * i);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) vp[u][i] = load_piece<T>(vbase + (size_t)(key + 16 * u) * kHeadDim + 32 * i);
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) sc[u] = score(kp[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) absorb(sc[u], vp[u]);
    }
    // tail: one key per slot and iteration; all 4 lanes of a slot share `key`, so the shuffles inside
    // score() stay among active lanes
    for (; key < nkeys; key += 16) {
        Piece<T> kp[4], vp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) kp[i] = load_piece<T>(kbase + (size_t)key * kHeadDim + 32 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) vp[i] = load_piece<T>(vbase + (size_t)key * kHeadDim + 32 * i);
        const float s0 = score(kp);
        absorb(s0, vp);
    }
    // merge the 16 key slots (lane bits 2..5): common maximum, rescale once, then sums
    float M = m;
#pragma unroll
    for (int off = 4; off <= 32; off <<= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
    const float f = (m == -INFINITY) ? 0.f : __expf(m - M);      // slot 0 always holds key 0: M is finite
    float lsum = l * f;
#pragma unroll
    for (int off = 4; off <= 32; off <<= 1) lsum += __shfl_xor(lsum, off, 64);
    float flat[32];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t) flat[8 * i + t] = acc[i][t] * f;
    slot_reduce_scatter(flat, lane);
    // this lane now holds flat index 2 ks + x = piece (ks / 4), element 2 (ks % 4) + x of its dim group
    {
        const float inv = 1.f / lsum;
        T* dst = out + (size_t)row * d + col + 32 * (ks >> 2) + 2 * (ks & 3);
        if (sizeof(T) == 4) {
            *reinterpret_cast<float2*>(dst) = make_float2(flat[0] * inv, flat[1] * inv);
        } else {
            *reinterpret_cast<uint32_t*>(dst) = (uint32_t)f32_to_bf16(flat[0] * inv) | ((uint32_t)f32_to_bf16(flat[1] * inv) << 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Causal prefill with many queries per document (bf16): a matrix-core kernel.  One workgroup = 64
// consecutive queries of one (document, head); wave w owns queries 16w .. 16w + 15.  Keys / values are
// staged 64 at a time, both tiles row-major in LDS (16-byte stores), and consumed in groups of 32; the next
// tile's rows are requested from HBM before the current tile's products start.  Both products run on
// v_mfma_f32_16x16x32_bf16 without any cross-lane exchange in between:
//   S^T[key][query]   = K (A operand, 16 keys x 32 dims)      x Q^T (B operand, from registers)
//   O^T[dim][query]  += V^T (A operand, 16 dims x 32 key slots) x P^T (B operand = the lane's own scores)
// After the first product lane (query q = lane % 16, quad = lane / 16) holds the scores of keys
// 4 quad .. 4 quad + 3 of both 16-key halves -- exactly the 8 "k slots" the second product wants from this
// lane if key slot 8 g + e is defined as key 4 g + e (e < 4) or 16 + 4 g + e - 4 (e >= 4).  The V^T operand in
// that slot order is two transposed LDS reads (ds_read_b64_tr_b16: the 16 lanes of quad g read the block of keys
// 4 g .. 4 g + 3 x 16 dims and each receives one dim's column) of the row-major tile: keys 4 g + .. and
// 16 + 4 g + ...  Rows of 288 bytes put the 8 key rows that a 32-lane half touches on 8 disjoint bank groups.
// Softmax statistics are per query = per lane (replicated over the 4 quads).
// ------------------------------------------------------------------------------------------------
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
using u32x4_t = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4_t = __attribute__((ext_vector_type(4))) float;

constexpr int kPfKeys = 32;                        // keys of one pair of products (one softmax update)
constexpr int kPfStage = 64;                       // keys staged per barrier pair: two such groups
constexpr int kPfKRow = 256 + 16;                  // padded K row (bytes)
constexpr int kPfVRow = 256 + 32;                  // padded V row (bytes)
using s16x4_t = __attribute__((ext_vector_type(4))) short;
using s16x8_t = __attribute__((ext_vector_type(8))) short;

__global__ __launch_bounds__(256, 4) void k_attention_prefill(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                           const bf16_t* __restrict__ vc, const int32_t* __restrict__ slot_map,
                                                           bf16_t* __restrict__ out, int s, int n_heads, int cap, int base_keys,
                                                           KeyWindow win) {
    __shared__ __attribute__((aligned(16))) char smem[kPfStage * kPfKRow + kPfStage * kPfVRow];
    char* ks = smem;
    char* vs = smem + kPfStage * kPfKRow;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int qtiles = (s + 63) / 64;
    // Workgroups are dealt to the 8 XCDs round-robin; the query tiles of one (document, head) walk the same keys, so
    // consecutive tile ids are mapped onto ONE XCD (its L2 then serves the second walk).
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
    const int qt = bid % qtiles;
    const int head = (bid / qtiles) % n_heads;
    const int doc = bid / (qtiles * n_heads);
    const int slot = slot_map ? slot_map[doc] : doc;
    const int d = n_heads * kHeadDim;
    const int qi = lane & 15, quad = lane >> 4;
    const int j = qt * 64 + wid * 16 + qi;             // query index inside the document's block of s
    const int jc = j < s ? j : s - 1;                   // rows past the end repeat the last query; never stored
    int my_keys = base_keys + jc + 1;                   // keys this query sees
    int block_keys = base_keys + min(s, qt * 64 + 64);  // keys the last query of the workgroup sees
    int wave_keys = base_keys + min(s, qt * 64 + wid * 16 + 16);
    if (win.max_keys > 0) {   // per-document window (shifted causal mask + key padding); uniform per workgroup
        const int sh = win.shift ? win.shift[doc] : 0;
        const int lim = win.limit ? min(win.limit[doc], win.max_keys) : win.max_keys;
        my_keys = max(1, min(my_keys + sh, lim));
        block_keys = max(1, min(block_keys + sh, lim));
        wave_keys = max(1, min(wave_keys + sh, lim));
    }

    // Q^T fragments: dims 32 * st + 8 * quad .. + 7 of query qi, pre-scaled by 1 / sqrt(128) in fp32 like the reference
    bf16x8_t qf[4];
    {
        const bf16_t* qrow = q + ((size_t)doc * s + jc) * d + head * kHeadDim;
#pragma unroll
        for (int st = 0; st < 4; ++st) qf[st] = *reinterpret_cast<const bf16x8_t*>(qrow + 32 * st + 8 * quad);
    }
    const float scale = 0.08838834764831845f;

    f32x4_t o[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) o[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;

    const size_t hbase = ((size_t)slot * n_heads + head) * cap * kHeadDim;
    // staging: thread -> (key = tid / 16 (+16), 16-byte piece = tid % 16) of the 32 x 256-byte tiles
    const int skey = tid >> 4, spiece = tid & 15;

    constexpr int NH = kPfStage / 16;
    u32x4_t kreg[NH], vreg[NH];
    auto request_tile = [&](int k0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            int key = k0 + skey + 16 * h;
            key = key < block_keys ? key : block_keys - 1;
            kreg[h] = *reinterpret_cast<const u32x4_t*>(kc + hbase + (size_t)key * kHeadDim + spiece * 8);
            vreg[h] = *reinterpret_cast<const u32x4_t*>(vc + hbase + (size_t)key * kHeadDim + spiece * 8);
        }
    };
    request_tile(0);
    // transposed reads of the V tile: lane 4 q + p of quad g supplies the address of key row 4 g + q, dims 4 p .. 4 p + 3
    // of the 16-dim block (+ 32 t bytes for block t, + 16 rows for the second half of the key slots)
    const char* vtr0 = vs + (4 * quad + (qi >> 2)) * kPfVRow + 8 * (qi & 3);

    for (int kt = 0; kt < block_keys; kt += kPfStage) {
        // Raw barriers: __syncthreads() would also wait for the rows requested ahead.  The previous tile is consumed:
        // the wait retires this wave's LDS reads of it (only LDS operations count here, the global rows requested
        // ahead stay in flight), so no other wave's stores of the next tile can overtake a read still pending ...
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int kk = skey + 16 * h;                                       // key inside the tile
            *reinterpret_cast<u32x4_t*>(ks + kk * kPfKRow + spiece * 16) = kreg[h];
            *reinterpret_cast<u32x4_t*>(vs + kk * kPfVRow + spiece * 16) = vreg[h];
        }
        if (kt + kPfStage < block_keys) request_tile(kt + kPfStage);   // in flight during this tile's products
        // ... and this tile is visible once every wave's LDS stores have completed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int sub = 0; sub < kPfStage / kPfKeys; ++sub) {
        const int k0 = kt + sub * kPfKeys;
        if (k0 >= wave_keys) break;      // wave-uniform: every further key of the tile is masked for this wave's queries
        const char* ksub = ks + sub * kPfKeys * kPfKRow;
        const char* vtr = vtr0 + sub * kPfKeys * kPfVRow;

        // ---- scores of the tile's two 16-key halves
        f32x4_t sc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            sc[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ksub + (16 * h + qi) * kPfKRow + 64 * st + 16 * quad);
                sc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[st], sc[h], 0, 0, 0);
            }
        }
        float p[8];
        float tmax = -INFINITY;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = k0 + 16 * h + 4 * quad + e;
                const float v = key < my_keys ? sc[h][e] * scale : -INFINITY;
                p[4 * h + e] = v;
                tmax = fmaxf(tmax, v);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);                 // finite: key 0 is visible to every query
        const float alpha = __expf(m - mn);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p[e] = __expf(p[e] - mn);
            psum += p[e];
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l = l * alpha + psum;
        m = mn;
        bf16x8_t pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (__bf16)p[e];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            o[t] *= alpha;
            typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4;
            const s16x4_t v_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vtr + 32 * t));
            const s16x4_t v_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vtr + 16 * kPfVRow + 32 * t));
            const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(v_lo, v_hi, 0, 1, 2, 3, 4, 5, 6, 7));
            o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[t], 0, 0, 0);
        }
        }
    }
    if (j < s) {
        const float inv = 1.f / l;
        bf16_t* orow = out + ((size_t)doc * s + j) * d + head * kHeadDim;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(o[t][0] * inv) | ((uint32_t)f32_to_bf16(o[t][1] * inv) << 16);
            pk.y = (uint32_t)f32_to_bf16(o[t][2] * inv) | ((uint32_t)f32_to_bf16(o[t][3] * inv) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(orow) + 16 * t + 4 * quad) = pk;
        }
    }
}

int launch_attention_q(const AttnQ& qa, const void* kc, const void* vc, const int32_t* slot_map, void* out, int n_docs,
                       int s, int n_heads, int cap, int base_keys, int causal, int dtype, hipStream_t st,
                       const int32_t* base_keys_dev, const KeyWindow* win) {
    LCM_REQUIRE(n_heads >= 1, "attention: needs at least one head");
    LCM_REQUIRE(n_docs >= 0 && s >= 1 && base_keys >= 0, "attention: bad shape");
    KeyWindow kw;
    if (win) kw = *win;
    LCM_REQUIRE(kw.max_keys >= 0 && kw.max_keys <= cap, "attention: key window exceeds the cache");
    LCM_REQUIRE(kw.max_keys > 0 || (kw.shift == nullptr && kw.limit == nullptr), "attention: a key window needs max_keys");
    const int max_keys = kw.max_keys > 0 ? kw.max_keys : base_keys + (causal ? s : 0);
    LCM_REQUIRE(max_keys >= 1 && max_keys <= cap, "attention: key range exceeds the cache");
    LCM_REQUIRE((qa.q != nullptr) != (qa.partial != nullptr), "attention: exactly one query source");
    if (n_docs == 0) return LCMHIP_OK;
    constexpr int prefill_min = 4;   // shortest causal query block that takes the matrix-core kernel
    if (causal && dtype == LCMHIP_BF16 && qa.q != nullptr && s >= prefill_min && base_keys_dev == nullptr) {
        const int qtiles = (s + 63) / 64;
        hipLaunchKernelGGL(k_attention_prefill, dim3((unsigned)(n_docs * n_heads * qtiles)), dim3(256), 0, st,
                           (const bf16_t*)qa.q, (const bf16_t*)kc, (const bf16_t*)vc, slot_map, (bf16_t*)out, s, n_heads, cap,
                           base_keys, kw);
        LCM_HIP_CHECK(hipGetLastError());
        return LCMHIP_OK;
    }
    const int waves = n_heads % 4 == 0 ? 4 : (n_heads % 2 == 0 ? 2 : 1);   // heads of one query row per workgroup, a wave each
    dim3 grid((unsigned)(n_docs * s * (n_heads / waves))), block(64 * waves);
    if (dtype == LCMHIP_BF16)
        hipLaunchKernelGGL((k_attention<bf16_t>), grid, block, 0, st, qa, (const bf16_t*)kc, (const bf16_t*)vc, slot_map,
                           (bf16_t*)out, s, n_heads, cap, base_keys, causal, base_keys_dev, kw);
    else if (dtype == LCMHIP_F32)
        hipLaunchKernelGGL((k_attention<float>), grid, block, 0, st, qa, (const float*)kc, (const float*)vc, slot_map,
                           (float*)out, s, n_heads, cap, base_keys, causal, base_keys_dev, kw);
    else
        return fail(LCMHIP_ERR_INVALID, "attention: unknown dtype");
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

int launch_attention(const void* q, const void* kc, const void* vc, const int32_t* slot_map, void* out, int n_docs,
                     int s, int n_heads, int cap, int base_keys, int causal, int dtype, hipStream_t st) {
    AttnQ qa{};
    qa.q = q;
    return launch_attention_q(qa, kc, vc, slot_map, out, n_docs, s, n_heads, cap, base_keys, causal, dtype, st, nullptr, nullptr);
}

}  // namespace lcmhip
