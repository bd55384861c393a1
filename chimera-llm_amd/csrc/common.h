// Shared helpers for the lcmhip kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/lcmhip.h"

namespace lcmhip {

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define LCM_HIP_CHECK(expr)                                                                    \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::lcmhip::fail(LCMHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define LCM_REQUIRE(cond, msg)                                                                 \
    do {                                                                                       \
        if (!(cond)) return ::lcmhip::fail(LCMHIP_ERR_INVALID, std::string(msg) + " [" #cond "]"); \
    } while (0)

#define LCM_TRY(expr)                                                                          \
    do {                                                                                       \
        int _r = (expr);                                                                       \
        if (_r != LCMHIP_OK) return _r;                                                        \
    } while (0)

// ---- dtypes -------------------------------------------------------------------------------------
struct bf16_t {
    uint16_t bits;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// Round-to-nearest-even through the compiler's cast (v_cvt_pk_bf16_f32 on gfx950; keeps NaN a NaN).
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int kDtype = LCMHIP_F32;
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
    static constexpr int kDtype = LCMHIP_BF16;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(p->bits); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { p->bits = f32_to_bf16(v); }
};

inline size_t dtype_size(int dtype) { return dtype == LCMHIP_BF16 ? 2 : 4; }

// 8 consecutive elements of T as floats (16-byte loads for bf16, two for f32).
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

// Non-temporal forms for streams far larger than the caches (read once / written once): the lines do not displace
// anything and stores go out without waiting in L2.  Measured on the copy calibration kernel at >= 256 MiB: +7 %.
using nv_f4 = __attribute__((ext_vector_type(4))) float;
using nv_u4 = __attribute__((ext_vector_type(4))) uint32_t;
template <typename T, bool NT>
__device__ __forceinline__ void load8x(const T* p, float (&v)[8]) {
    if constexpr (!NT) {
        load8<T>(p, v);
    } else if constexpr (sizeof(T) == 4) {
        const nv_f4 a = __builtin_nontemporal_load(reinterpret_cast<const nv_f4*>(p));
        const nv_f4 b = __builtin_nontemporal_load(reinterpret_cast<const nv_f4*>(p) + 1);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
        const nv_u4 u = __builtin_nontemporal_load(reinterpret_cast<const nv_u4*>(p));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(u[i] << 16);
            v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
        }
    }
}
template <typename T, bool NT>
__device__ __forceinline__ void store8x(T* p, const float (&v)[8]) {
    if constexpr (!NT) {
        store8<T>(p, v);
    } else if constexpr (sizeof(T) == 4) {
        __builtin_nontemporal_store(nv_f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<nv_f4*>(p));
        __builtin_nontemporal_store(nv_f4{v[4], v[5], v[6], v[7]}, reinterpret_cast<nv_f4*>(p) + 1);
    } else {
        nv_u4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
        __builtin_nontemporal_store(w, reinterpret_cast<nv_u4*>(p));
    }
}

// ---- wave / block reductions (wave = 64 lanes) --------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Sum over a block of NW waves; `red` is NW floats of LDS. Every thread gets the total.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6;
    __syncthreads();  // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

// ---- profiling hook ---------------------------------------------------------------------------
// A Profiler brackets launches with event pairs when enabled; see session.cpp.
struct Profiler;
struct ProfScope {
    Profiler* p;
    int slot;
    hipStream_t st;
    ProfScope(Profiler* p, const char* name, double bytes, double flops, hipStream_t st);
    ~ProfScope();
};

// Most K splits any split-K plan uses (bounds the fp32 partial-tile slabs).
constexpr int kMaxSplits = 12;

// ---- kernel launchers (implemented in the .hip files) ---------------------------------------------
// Inner activation of the two-matrix FFN (LCMHIP_ACT_*): torch.nn.ReLU / Tanh / ELU / LeakyReLU / SELU / GELU (erf
// form) / SiLU / Softsign / Sigmoid / Hardsigmoid with their default arguments, evaluated in fp32.  `kind` is
// uniform over a launch.
// `slope`: negative-side factor of LeakyReLU (0.01) / PReLU (learned), see act_slope().
__device__ __forceinline__ float apply_act(float v, int kind, float slope = 0.01f) {
    switch (kind) {
        case LCMHIP_ACT_TANH: return tanhf(v);
        case LCMHIP_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case LCMHIP_ACT_LEAKY_RELU: return v > 0.f ? v : slope * v;   // and PReLU: see act_kind()
        case LCMHIP_ACT_SELU: return 1.0507009873554804934193349852946f * (v > 0.f ? v : 1.6732632423543772848170429916717f * expm1f(v));
        case LCMHIP_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        case LCMHIP_ACT_SILU: return v / (1.f + __expf(-v));
        case LCMHIP_ACT_SOFTSIGN: return v / (1.f + fabsf(v));
        case LCMHIP_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        case LCMHIP_ACT_HARDSIGMOID: return fminf(fmaxf(v * (1.f / 6.f) + 0.5f, 0.f), 1.f);
        default: return fmaxf(v, 0.f);
    }
}

// Write-through stores (sc1: the line goes to memory at device scope as it is written instead of staying dirty in the
// XCD's L2 until the write-back at the kernel's end).  For the short kernels of the sampling loop, whose outputs the
// next kernel reads from all 8 XCDs anyway, that shortens every kernel boundary behind a split-K product: with the
// slab stores of the K < 4096 products and both stores of the pending-product norm written through, config
// two_tower_diffusion 1.6B, 64 documents: 1389 -> 1424 sentence-embeddings/s (self_ov 10.35 -> 9.69 us, cross_o 7.66 ->
// 7.21, cross_q 7.34 -> 7.00 per launch); 8 / 16 / 32 / 128 / 256 documents +0.4 / 1.3 / 2.0 / 1.9 / 2.3 %.  Not for
// the K = 5632 slabs (12.8 -> 13.6 us at 64 documents), the SwiGLU / attention / q outputs (no change) or with the
// non-temporal bit (nt, alone or with sc1: slower norm).
typedef float st_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_through(void* dst, st_f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void store8_through(void* dst, uint64_t v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
}

struct GemmArgs {
    const void* A; int lda;
    const void* W; const void* W2;
    const void* bias; const void* bias2;
    const float* gate;          // (N,) fp32 or null
    // RESID with a gate that differs per row: row m reads gate + clamp(gate_rows[m], 0, gate_steps - 1) * gate_stride
    const int32_t* gate_rows; size_t gate_stride; int gate_steps;
    void* out; int ldo;
    const int32_t* out_rows;    // optional row map for `out` (length M) or null
    // With out_rows: > 0 selects the KV-cache layout [slot][head][position][128] -- a mapped row index
    // r = slot * out_head_cap + position and column n go to ((slot * N/128 + n/128) * out_head_cap + position) * 128 + n%128
    int out_head_cap;
    int M, N, K;
    int epilogue;
    int act;                    // LCMHIP_ACT_* of the LCMHIP_EPI_RELU epilogue (0 = ReLU)
    const void* act_slope;      // LCMHIP_ACT_PRELU: the learned slope (one element of the operand dtype)
    int dtype;
    // RESID only: rows [M, M + extra_rows) of `out` get gate[n] * extra_const[n] added
    int extra_rows; const float* extra_const;
    // split-K: fp32 slab [splits][M][N] (optional; without it the K range is never split)
    float* partial; size_t partial_bytes; int max_splits;
    int tiles_x, tiles_y;       // filled by the launcher of the 256 x 256 kernel (tile grid behind its 1-D launch)
    // Stacked products (stack = 2 or 3; 0 / 1 = off): `stack` weight matrices of N rows each are multiplied by the same
    // A in one launch, matrix g >= 1 with its own weight, bias and output (matrix 0 uses W / bias / out).  Linear
    // epilogues without a K split only; every output shares ldo, out_rows and out_head_cap.
    int stack;
    const void* stack_W[2]; const void* stack_bias[2]; void* stack_out[2];
    // Per-matrix epilogue switches of a stacked product (bit g = matrix g; many-row kernels, operand-dtype STORE only):
    //  scatter_mask: matrices whose rows go through out_rows / out_head_cap (0: all of them, the plain meaning of out_rows);
    //  rope_mask:    matrices whose outputs are rotated after the store rounding -- consecutive pairs (2i, 2i + 1) of every
    //                128-wide head by the angle row rope_pos[m] of the cos / sin tables ([position][64]), as k_rope_store does.
    int scatter_mask, rope_mask;
    const int32_t* rope_pos; const float* rope_cos; const float* rope_sin;
    int stack_sel;              // set by the kernels: the matrix of the workgroup's tile
    int slab_through;           // set by the launcher: split-K slab stores are written through (store16_through)
};
// PReLU runs as LeakyReLU with its learned slope: act_kind() is the `kind` and act_slope() the `slope` of apply_act().
__device__ __forceinline__ int act_kind(const GemmArgs& p) { return p.act == LCMHIP_ACT_PRELU ? LCMHIP_ACT_LEAKY_RELU : p.act; }
template <typename T>
__device__ __forceinline__ float act_slope(const GemmArgs& p) {
    return p.act == LCMHIP_ACT_PRELU ? Elem<T>::load(reinterpret_cast<const T*>(p.act_slope)) : 0.01f;
}
// True when launch_gemm would run this stacked product as ONE launch of a many-row kernel (whose epilogue knows the
// per-matrix switches above); otherwise it runs one product per matrix and the switches must not be set.
bool gemm_stack_is_fused(const GemmArgs& a);
// Many-row kernels: does launch_gemm take them for this (unsplit) product, and what does the launch cost in waves of tiles?
bool gemm_big_applicable(const GemmArgs& a, bool may_split);
double gemm_big_wave_cost(const GemmArgs& a, int epi, int m_rows);
// C = A W^T with the epilogue applied (split-K + finishing pass when a.partial is given and it pays).
int launch_gemm(const GemmArgs& a, hipStream_t st);
// Resolves the current device's copy of the static scratch array of lcmhip_gemm_static (noalloc.hip), once per device.
int static_scratch_init();
// Raw partial products only: slab [*splits][M][N] in a.partial, to be summed by the consumer.
int launch_gemm_partial(const GemmArgs& a, int* splits, hipStream_t st);

// A product still sitting in a split-K slab, to be folded into the residual stream by the next
// norm kernel:  x[r] = (init ? 0 : x[r]) + gate * (sum_s slab[s][src(r)] + bias)   for r <  M
//               x[r] = x[r] + gate * extra_const                                    for r >= M (dup == 0)
// with src(r) = r, or r - M for the duplicated rows when dup == 1.
struct Pending {
    const float* partial = nullptr;
    int splits = 0;
    int M = 0;
    const void* bias = nullptr;       // operand dtype, nullable
    const float* gate = nullptr;      // fp32, nullable (= 1)
    const float* extra_const = nullptr;
    int dup = 0;
    int init = 0;
};

// Per-row modulation: row r takes shift / scale (and a pending product's gate) at offset row_step[r] * step_stride
// floats from the pointers given (one modulation table entry per diffusion step); indices are clamped to
// [0, n_steps).  row_step == nullptr: every row reads the pointers as they are.
struct RowSteps {
    const int32_t* row_step = nullptr;
    size_t step_stride = 0;
    int n_steps = 1;
};

// Per-document key window of a causal attention launch: query j of document i sees
//   clamp(base_keys + shift[i] + j + 1, 1, min(limit[i], max_keys)) keys.  Null arrays mean 0 / max_keys.
struct KeyWindow {
    const int32_t* shift = nullptr;
    const int32_t* limit = nullptr;
    int max_keys = 0;
};

int launch_normalize(const void* x, const void* center, const void* scale, const void* cmin, const void* cmax,
                     void* y, int64_t rows, int dim, int dtype, bool inverse, hipStream_t st);
// fp32 in -> T out, used by the frontend: y = (x - center) / scale with T-typed center/scale.
// Packed real FFT of every row (dim a power of two, 64 .. 4096) and its inverse; see lcmhip_rfft_pack.
int launch_rfft_pack(const float* x, float* y, int64_t rows, int dim, int inverse, hipStream_t st);
int launch_normalize_f32_in(const float* x, const void* center, const void* scale, void* y, int64_t rows, int dim,
                            int dtype, hipStream_t st);
// Rows [0, rows_norm) are normalised (+ modulated) into y; when `pend` is given, rows [0, rows_total)
// of x first absorb the pending product (x is updated in place).
int launch_norm_modulate(float* x, const void* w, const void* b, const float* shift, const float* scale,
                         void* y, int rows_total, int rows_norm, int dim, int norm_kind, float eps, int dtype,
                         const Pending* pend, hipStream_t st, const RowSteps* steps = nullptr);
int launch_attention(const void* q, const void* kc, const void* vc, const int32_t* slot_map, void* out, int n_docs,
                     int s, int n_heads, int cap, int base_keys, int causal, int dtype, hipStream_t st);
// Query source of the attention kernel: a finished (rows, d) tensor, or the split-K slab of the q projection.
struct AttnQ {
    const void* q;
    const float* partial;
    int splits;
    int rows;
    const void* bias;
};
int launch_attention_q(const AttnQ& qa, const void* kc, const void* vc, const int32_t* slot_map, void* out, int n_docs,
                       int s, int n_heads, int cap, int base_keys, int causal, int dtype, hipStream_t st,
                       const int32_t* base_keys_dev = nullptr, const KeyWindow* win = nullptr);

struct DdimScalars {
    float sqrt_a_t, sqrt_b_t;        // sqrt(alpha_prod_t), sqrt(1 - alpha_prod_t)
    float sqrt_a_prev, dir_coef;     // sqrt(alpha_prod_t_prev), sqrt(1 - alpha_prod_t_prev - sigma^2)
    float guidance_scale, guidance_rescale;
    float inv_eps_scaling;           // 1 / epsilon_scaling or 1
    float clip_noise;
    float clip_sample_range;         // < 0: off
    int prediction_type;
    int mode;                        // 0: plain step on out_cond with prediction_type; 1: CFG (eps space)
    float sigma;                     // eta * sqrt(variance_t): weight of the variance noise (0: deterministic)
    float thr_ratio;                 // dynamic thresholding quantile; < 0: off
    float thr_max;                   // sample_max_value
};
// Optional inputs / outputs of the fused step: the model output still sitting in the split-K slab of the final
// projection ([splits][rows_total][dim]; conditional rows first, unconditional rows `rows` further down) with its
// bias, and a copy of prev_sample in the operand dtype for the next step's initial projection.
struct DdimIO {
    const float* partial = nullptr;
    int splits = 0;
    int rows_total = 0;
    const void* bias = nullptr;
    void* prev_cast = nullptr;
    int bf16 = 0;                    // dtype of bias and prev_cast
    const float* variance_noise = nullptr;   // (rows, dim) N(0, 1) draws, required when sigma > 0
};
int launch_ddim_step(const float* sample, const float* out_c, const float* out_u, float* prev, float* x0, int rows,
                     int dim, const DdimScalars& sc, hipStream_t st, const DdimIO* io = nullptr);
// Final: y = x0 * scale + center (T-typed center/scale) -> fp32
int launch_denormalize_f32_out(const float* x0, const void* center, const void* scale, float* y, int64_t rows,
                               int dim, int dtype, hipStream_t st, const void* cmin = nullptr, const void* cmax = nullptr);
// cast != nullptr: also a copy of the latents in `dtype` (the first step's initial projection reads it)
int launch_init_latents(const float* noise, float* lat, int64_t n, float sigma, float clip, hipStream_t st,
                        void* cast = nullptr, int dtype = LCMHIP_F32);
// q/k/v (rows, d) T: rotate q in place, write rotated k and v into the cache rows given by cache_rows.
// In-place LayerNorm / RMSNorm (norm_kind: LCMHIP_NORM_RMS or anything else = LayerNorm with optional bias) of `rows`
// rows of x0 and, when x1 is given, of x1, both in the operand dtype.
int launch_row_norm(void* x0, const void* w0, const void* b0, void* x1, const void* w1, const void* b1, int rows, int dim,
                    int norm_kind, float eps, int dtype, hipStream_t st);
int launch_rope_store(void* q, const void* k, const void* v, void* kc, void* vc, const int32_t* cache_rows,
                      const int32_t* positions, const float* cos_tab, const float* sin_tab, int rows, int d,
                      int cap, int use_rope, int dtype, hipStream_t st);
int launch_add_pos(float* x, const void* table, const int32_t* positions, int rows, int d, int dtype, hipStream_t st);
// x[r][:] += the sinusoidal encoding of positions[r] (first half sines, second half cosines; LCMHIP_POS_SINE)
int launch_add_sine_pos(float* x, const int32_t* positions, int rows, int d, hipStream_t st);
// Helpers of the post-norm / normformer layer orders and of scale_residual (n = rows * d elements, d % 8 == 0):
int launch_cast_rows(const float* x, void* y, int64_t n, int dtype, hipStream_t st);                 // y = T(x)
int launch_take_rows(float* x, const void* y, int64_t n, int dtype, int accumulate, hipStream_t st);  // x = float(y) or x += float(y)
int launch_scale_cols(float* x, const void* scale, int rows, int d, int dtype, hipStream_t st);      // x[r][c] *= scale[c]
int launch_fill_rows(void* cache, const void* vec, const int32_t* rows_idx, int n_rows, int d, int cap, int dtype,
                     hipStream_t st);
// Element offset of column n of mapped row r (= slot * cap + position) in a [slot][head][position][128] cache.
__host__ __device__ inline size_t kv_cache_offset(int r, int n, int cap, int d) {
    const int slot = r / cap, pos = r % cap;
    return (((size_t)slot * (d >> 7) + (n >> 7)) * cap + pos) * 128 + (n & 127);
}
// Rotation of one pair (x0, x1) by the angle whose cosine / sine are cs / sn.  Every product and sum is rounded on its
// own (no contraction into fused multiply-adds), so that every kernel that rotates gives the same bits.
__device__ __forceinline__ void rope_rotate(float& x0, float& x1, float cs, float sn) {
#pragma clang fp contract(off)
    const float a = x0 * cs, b = x1 * sn, c = x0 * sn, d = x1 * cs;
    x0 = a - b;
    x1 = c + d;
}
int launch_finalize_step(const float* new_emb, void* seqs, int dtype, const int32_t* doc_rows, const int32_t* prompt_lens,
                         int n_active, int step, int total_len, int dim, const float* eos, float* cos_out, hipStream_t st);
int launch_stream_copy(const void* src, void* dst, int64_t bytes, int nontemporal, hipStream_t st);
int launch_silu_cast(const float* x, void* y, int64_t n, int dtype, hipStream_t st);
int launch_cast(const float* x, void* y, int64_t n, int dtype, hipStream_t st);
int launch_broadcast_rows(const float* src, float* dst, int rows, int d, hipStream_t st);
int launch_copy_rows_to_f32(const void* src, float* dst, int64_t n, int dtype, hipStream_t st);

}  // namespace lcmhip
