// Bandwidth-bound kernels of the generation path: SONAR normalizer, Layer/RMS norm fused with the
// adaLN modulation, the fused DDIM + classifier-free-guidance update, RoPE + KV-cache scatter.
// All of them read/write 16 bytes per lane, keep statistics in fp32 and reduce with wave shuffles.
#include <algorithm>

#include "common.h"

namespace lcmhip {

using vf4 = __attribute__((ext_vector_type(4))) float;   // native vector: what the non-temporal builtins take


// ------------------------------------------------------------------------------------------------
// SONAR normalizer: y = (x - center) / scale  [clamp]   |   y = [clamp](x) * scale + center
// ------------------------------------------------------------------------------------------------
// A workgroup walks whole tiles of 256 * U consecutive 8-element vectors (8 or 16 KB contiguous per tile: its requests
// stay within a few DRAM pages -- a grid-wide stride between a lane's pieces costs a quarter of the rate).  When the
// row length divides 256 vectors a lane stays on one column group for the whole launch: center / scale (and
// 1 / scale) are loaded once.  NT: non-temporal loads and stores, for streams far larger than the caches.
template <typename T, bool NT>
__device__ __forceinline__ void load4x(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load4x<float, false>(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <>
__device__ __forceinline__ void load4x<float, true>(const float* p, float (&v)[8]) {
    const nv_f4 a = __builtin_nontemporal_load(reinterpret_cast<const nv_f4*>(p));
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}
template <bool NT>
__device__ __forceinline__ void store4x(float* p, const float (&v)[8]) {
    if constexpr (NT) __builtin_nontemporal_store(nv_f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<nv_f4*>(p));
    else *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

// VEC: elements per lane piece -- 8, or 4 when every operand is fp32 (a lane then moves 16 bytes per access: with 32
// bytes per lane one wave instruction covers only half of every line it touches).
template <typename TI, typename TO, typename TP, bool INVERSE, bool NT, int VEC = 8>
__global__ __launch_bounds__(256) void k_normalizer(const TI* __restrict__ x, const TP* __restrict__ center,
                                                    const TP* __restrict__ scale, const TP* __restrict__ cmin,
                                                    const TP* __restrict__ cmax, TO* __restrict__ y, int64_t n_vec,
                                                    int vec_per_row) {
    constexpr int U = 4;
    const bool fixed_col = (256 % vec_per_row) == 0;
    const bool clip = cmin != nullptr;
    float c[8], s[8], lo[8], hi[8];
    auto params = [&](int col) {
        if constexpr (VEC == 4) {
            load4x<float, false>(reinterpret_cast<const float*>(center) + col, c);
            load4x<float, false>(reinterpret_cast<const float*>(scale) + col, s);
#pragma unroll
            for (int j = 4; j < 8; ++j) { c[j] = 0.f; s[j] = 1.f; }
        } else {
            load8<TP>(center + col, c);
            load8<TP>(scale + col, s);
        }
        if (!INVERSE) {
            // one exact division per column group, then a multiply per element (differs from a per-element division by
            // at most one fp32 ulp; the division sequence would otherwise make the kernel VALU- rather than HBM-bound)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = 1.f / s[j];
        }
        if (clip) {
            if constexpr (VEC == 4) {
                load4x<float, false>(reinterpret_cast<const float*>(cmin) + col, lo);
                load4x<float, false>(reinterpret_cast<const float*>(cmax) + col, hi);
            } else {
                load8<TP>(cmin + col, lo);
                load8<TP>(cmax + col, hi);
            }
        }
    };
    if (fixed_col) params((int)(threadIdx.x % vec_per_row) * VEC);
    const int64_t tiles = (n_vec + 256 * U - 1) / (256 * U);
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t i0 = t * 256 * U + threadIdx.x;
        float v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * 256 < n_vec) {
                if constexpr (VEC == 4) load4x<float, NT>(reinterpret_cast<const float*>(x) + (i0 + u * 256) * 4, v[u]);
                else load8x<TI, NT>(x + (i0 + u * 256) * 8, v[u]);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * 256;
            if (i >= n_vec) break;
            if (!fixed_col) params((int)(i % vec_per_row) * VEC);
            if (INVERSE) {
                if (clip) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[u][j] = fminf(fmaxf(v[u][j], lo[j]), hi[j]);
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[u][j] = v[u][j] * s[j] + c[j];
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[u][j] = (v[u][j] - c[j]) * s[j];
                if (clip) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[u][j] = fminf(fmaxf(v[u][j], lo[j]), hi[j]);
                }
            }
            if constexpr (VEC == 4) store4x<NT>(reinterpret_cast<float*>(y) + i * 4, v[u]);
            else store8x<TO, NT>(y + i * 8, v[u]);
        }
    }
}

static int grid_for(int64_t n_items, int block) {
    int64_t g = (n_items + block - 1) / block;
    if (g > 2048) g = 2048;  // 256 CUs x 8 workgroups, grid-stride over the rest
    if (g < 1) g = 1;
    return (int)g;
}

// Grid of the normalizer: one workgroup per 1024-vector tile, at most 8 per CU.
static int normalizer_grid(int64_t n_vec, int vec_per_row) {
    (void)vec_per_row;
    const int64_t tiles = (n_vec + 1023) / 1024;
    return (int)std::max<int64_t>(1, std::min<int64_t>(tiles, 4096));
}

// Streams this large gain from the non-temporal forms (they no longer fit the caches anyway).
static bool use_nontemporal(int64_t bytes) { return bytes >= ((int64_t)128 << 20); }

int launch_normalize(const void* x, const void* center, const void* scale, const void* cmin, const void* cmax,
                     void* y, int64_t rows, int dim, int dtype, bool inverse, hipStream_t st) {
    LCM_REQUIRE(dim % 8 == 0 && dim <= 8192 && rows >= 0, "normalizer: dim must be a multiple of 8, at most 8192");
    LCM_REQUIRE((cmin == nullptr) == (cmax == nullptr), "normalizer: clip_min and clip_max go together");
    if (rows == 0) return LCMHIP_OK;
    const int vec = dtype == LCMHIP_F32 ? 4 : 8;          // 16 bytes per lane and access
    const int64_t n_vec = rows * (dim / vec);
    const int g = normalizer_grid(n_vec, dim / vec);
    const bool nt = use_nontemporal(2 * rows * (int64_t)dim * (int64_t)dtype_size(dtype));
#define LCM_NORMALIZER(T, INV, V)                                                                                   \
    if (nt) hipLaunchKernelGGL((k_normalizer<T, T, T, INV, true, V>), dim3(g), dim3(256), 0, st, (const T*)x, (const T*)center, \
                               (const T*)scale, (const T*)cmin, (const T*)cmax, (T*)y, n_vec, dim / V);             \
    else hipLaunchKernelGGL((k_normalizer<T, T, T, INV, false, V>), dim3(g), dim3(256), 0, st, (const T*)x, (const T*)center,   \
                            (const T*)scale, (const T*)cmin, (const T*)cmax, (T*)y, n_vec, dim / V)
    if (dtype == LCMHIP_BF16) {
        if (inverse) { LCM_NORMALIZER(bf16_t, true, 8); } else { LCM_NORMALIZER(bf16_t, false, 8); }
    } else if (dtype == LCMHIP_F32) {
        if (inverse) { LCM_NORMALIZER(float, true, 4); } else { LCM_NORMALIZER(float, false, 4); }
    } else {
        return fail(LCMHIP_ERR_INVALID, "normalizer: unknown dtype");
    }
#undef LCM_NORMALIZER
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

int launch_normalize_f32_in(const float* x, const void* center, const void* scale, void* y, int64_t rows, int dim,
                            int dtype, hipStream_t st) {
    LCM_REQUIRE(dim % 8 == 0, "normalizer: dim must be a multiple of 8");
    if (rows == 0) return LCMHIP_OK;
    const int64_t n_vec = rows * (dim / 8);
    const int g = normalizer_grid(n_vec, dim / 8);
    if (dtype == LCMHIP_BF16)
        hipLaunchKernelGGL((k_normalizer<float, bf16_t, bf16_t, false, false>), dim3(g), dim3(256), 0, st, x,
                           (const bf16_t*)center, (const bf16_t*)scale, (const bf16_t*)nullptr,
                           (const bf16_t*)nullptr, (bf16_t*)y, n_vec, dim / 8);
    else
        hipLaunchKernelGGL((k_normalizer<float, float, float, false, false>), dim3(g), dim3(256), 0, st, x,
                           (const float*)center, (const float*)scale, (const float*)nullptr, (const float*)nullptr,
                           (float*)y, n_vec, dim / 8);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

int launch_denormalize_f32_out(const float* x0, const void* center, const void* scale, float* y, int64_t rows,
                               int dim, int dtype, hipStream_t st, const void* cmin, const void* cmax) {
    LCM_REQUIRE(dim % 8 == 0, "normalizer: dim must be a multiple of 8");
    LCM_REQUIRE((cmin == nullptr) == (cmax == nullptr), "normalizer: clip_min and clip_max go together");
    if (rows == 0) return LCMHIP_OK;
    const int64_t n_vec = rows * (dim / 8);
    const int g = normalizer_grid(n_vec, dim / 8);
    if (dtype == LCMHIP_BF16)
        hipLaunchKernelGGL((k_normalizer<float, float, bf16_t, true, false>), dim3(g), dim3(256), 0, st, x0,
                           (const bf16_t*)center, (const bf16_t*)scale, (const bf16_t*)cmin,
                           (const bf16_t*)cmax, y, n_vec, dim / 8);
    else
        hipLaunchKernelGGL((k_normalizer<float, float, float, true, false>), dim3(g), dim3(256), 0, st, x0,
                           (const float*)center, (const float*)scale, (const float*)cmin, (const float*)cmax,
                           y, n_vec, dim / 8);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Layer / RMS norm + adaLN modulation.  One workgroup (256 lanes) per row; the row stays in
// registers between the statistics pass and the output pass (dim <= 8192).
// ------------------------------------------------------------------------------------------------
template <typename T, int KIND, int NT>
__global__ __launch_bounds__(NT) void k_norm_modulate(float* __restrict__ x, const T* __restrict__ w,
                                                       const T* __restrict__ b, const float* __restrict__ shift,
                                                       const float* __restrict__ scale, T* __restrict__ y, int dim,
                                                       int rows_norm, float eps, Pending pd, RowSteps rs) {
    constexpr int MAXV = 2048 / NT;  // float4 chunks per lane (dim <= 8192)
    __shared__ float red[NT / 64];
    const int row = blockIdx.x;
    if (rs.row_step != nullptr) {   // this row's own entry of the modulation table
        const size_t moff = (size_t)min(max(rs.row_step[row], 0), rs.n_steps - 1) * rs.step_stride;
        if (shift != nullptr) { shift += moff; scale += moff; }
        if (pd.gate != nullptr) pd.gate += moff;
    }
    float* xr = x + (size_t)row * dim;
    const int nv = dim / 4;
    float4 v[MAXV];
    // ---- absorb the pending split-K product into the residual stream
    const bool has_pending = pd.partial != nullptr;
    const int src_row = row < pd.M ? row : (pd.dup ? row - pd.M : -1);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = threadIdx.x + i * NT;
        if (c < nv) {
            if (has_pending && pd.init) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            else v[i] = *reinterpret_cast<const float4*>(xr + c * 4);
            if (has_pending) {
                float4 a;
                if (src_row >= 0) {
                    a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
                    for (int s = 0; s < pd.splits; ++s) {
                        const float4 t = *reinterpret_cast<const float4*>(pd.partial + ((size_t)s * pd.M + src_row) * dim + c * 4);
                        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
                    }
                    if (pd.bias != nullptr) {
                        const T* bp = reinterpret_cast<const T*>(pd.bias) + c * 4;
                        a.x += Elem<T>::load(bp); a.y += Elem<T>::load(bp + 1);
                        a.z += Elem<T>::load(bp + 2); a.w += Elem<T>::load(bp + 3);
                    }
                } else {
                    a = *reinterpret_cast<const float4*>(pd.extra_const + c * 4);
                }
                if (pd.gate != nullptr) {
                    const float4 g = *reinterpret_cast<const float4*>(pd.gate + c * 4);
                    a.x *= g.x; a.y *= g.y; a.z *= g.z; a.w *= g.w;
                }
                v[i].x += a.x; v[i].y += a.y; v[i].z += a.z; v[i].w += a.w;
                *reinterpret_cast<float4*>(xr + c * 4) = v[i];
            }
        }
    }
    if (row >= rows_norm) return;   // uniform per workgroup
    // norm weight / bias / modulation for this lane's columns: requested now, consumed after the reduction
    float pw[MAXV][4], pb[MAXV][4], psc[MAXV][4], psh[MAXV][4];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = threadIdx.x + i * NT;
        if (c < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pw[i][j] = Elem<T>::load(w + c * 4 + j);
                pb[i][j] = (KIND == LCMHIP_NORM_LAYERNORM && b != nullptr) ? Elem<T>::load(b + c * 4 + j) : 0.f;
            }
            if (scale != nullptr) {
                const float4 a = *reinterpret_cast<const float4*>(scale + c * 4);
                const float4 h = *reinterpret_cast<const float4*>(shift + c * 4);
                psc[i][0] = a.x; psc[i][1] = a.y; psc[i][2] = a.z; psc[i][3] = a.w;
                psh[i][0] = h.x; psh[i][1] = h.y; psh[i][2] = h.z; psh[i][3] = h.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { psc[i][j] = 0.f; psh[i][j] = 0.f; }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = threadIdx.x + i * NT;
        if (c < nv) {
            if (KIND == LCMHIP_NORM_RMS)
                s += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
            else
                s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    }
    s = block_sum<NT / 64>(s, red);
    float mean = 0.f, rstd;
    if (KIND == LCMHIP_NORM_RMS) {
        rstd = rsqrtf(s / dim + eps);
    } else {
        mean = s / dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = threadIdx.x + i * NT;
            if (c < nv) {
                const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
                q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
            }
        }
        q = block_sum<NT / 64>(q, red);
        rstd = rsqrtf(q / dim + eps);
    }
    T* yr = y + (size_t)row * dim;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = threadIdx.x + i * NT;
        if (c < nv) {
            const int col = c * 4;
            float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = (o[j] - mean) * rstd * pw[i][j] + pb[i][j];
                o[j] = t * (1.f + psc[i][j]) + psh[i][j];
            }
            if (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(yr) + col) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
                pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(yr) + col) = pk;
            }
        }
    }
}

// The same kernel for the shapes of the sampling loop (a pending split-K product, dim == 4 * NT: one float4 per
// lane), written without a single data-dependent branch around a load: every operand -- residual, up to MAXS slabs,
// bias, gate, constant row, norm weight / bias, modulation -- is requested back to back (absent ones are read from a
// valid dummy address and masked), so the kernel costs ONE memory round trip instead of one per operand group.
// Splits past pd.splits re-read the last slab (MAXS - splits <= 3 redundant, cache-resident loads).  Same arithmetic,
// in the same order, as k_norm_modulate.
template <typename T>
__device__ __forceinline__ float4 load4(const T* p);
template <>
__device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

template <typename T, int KIND, int NT, int MAXS, bool INIT>
__global__ __launch_bounds__(NT) void k_norm_pending(float* __restrict__ x, const T* __restrict__ w, const T* __restrict__ b,
                                                      const float* __restrict__ shift, const float* __restrict__ scale,
                                                      T* __restrict__ y, int rows_norm, float eps, Pending pd, RowSteps rs) {
    constexpr int dim = 4 * NT;
    __shared__ float red[NT / 64];
    const int row = blockIdx.x, c4 = threadIdx.x * 4;
    size_t moff = 0;
    if (rs.row_step != nullptr) moff = (size_t)min(max(rs.row_step[row], 0), rs.n_steps - 1) * rs.step_stride;
    float* xr = x + (size_t)row * dim + c4;
    const int src_row = row < pd.M ? row : (pd.dup ? row - pd.M : -1);
    const bool real = src_row >= 0;                       // uniform per workgroup
    const int sr = real ? src_row : 0;
    const float* dummy = pd.partial + c4;                 // valid fp32 memory for absent operands
    // ---- every load
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);          // INIT: the product starts the residual stream
    if constexpr (!INIT) xv = *reinterpret_cast<const float4*>(xr);
    float4 t[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
        t[s] = *reinterpret_cast<const float4*>(pd.partial + ((size_t)min(s, pd.splits - 1) * pd.M + sr) * dim + c4);
    const float4 bv = load4<T>((pd.bias != nullptr ? reinterpret_cast<const T*>(pd.bias) : w) + c4);
    const float4 gv = *reinterpret_cast<const float4*>(pd.gate != nullptr ? pd.gate + moff + c4 : dummy);
    const float4 ev = *reinterpret_cast<const float4*>(pd.extra_const != nullptr ? pd.extra_const + c4 : dummy);
    const float4 wv = load4<T>(w + c4);
    const float4 nb = load4<T>((KIND == LCMHIP_NORM_LAYERNORM && b != nullptr ? b : w) + c4);
    const float4 scv = *reinterpret_cast<const float4*>(scale != nullptr ? scale + moff + c4 : dummy);
    const float4 shv = *reinterpret_cast<const float4*>(shift != nullptr ? shift + moff + c4 : dummy);
    // ---- residual update: x = (init ? 0 : x) + gate * (sum of slabs + bias | constant row)
    // (slabs past the split count are weighted by 0 instead of skipped: a skipped use lets the compiler sink the
    // load into the branch and wait for it there, one extra round trip)
    float a[4] = {t[0].x, t[0].y, t[0].z, t[0].w};
#pragma unroll
    for (int s = 1; s < MAXS; ++s) {
        const float on = s < pd.splits ? 1.f : 0.f;
        a[0] = fmaf(t[s].x, on, a[0]); a[1] = fmaf(t[s].y, on, a[1]); a[2] = fmaf(t[s].z, on, a[2]); a[3] = fmaf(t[s].w, on, a[3]);
    }
    {
        const float on = pd.bias != nullptr ? 1.f : 0.f;
        a[0] = fmaf(bv.x, on, a[0]); a[1] = fmaf(bv.y, on, a[1]); a[2] = fmaf(bv.z, on, a[2]); a[3] = fmaf(bv.w, on, a[3]);
    }
    {   // rows past the product take the constant row instead (a blend, not a branch: see above)
        const float keep = real ? 1.f : 0.f, take = real ? 0.f : 1.f;
        a[0] = fmaf(ev.x, take, a[0] * keep); a[1] = fmaf(ev.y, take, a[1] * keep);
        a[2] = fmaf(ev.z, take, a[2] * keep); a[3] = fmaf(ev.w, take, a[3] * keep);
    }
    if (pd.gate != nullptr) { a[0] *= gv.x; a[1] *= gv.y; a[2] *= gv.z; a[3] *= gv.w; }
    float v[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += a[j];
    store16_through(xr, st_f32x4{v[0], v[1], v[2], v[3]});   // both stores written through: see store16_through
    if (row >= rows_norm) return;   // uniform per workgroup
    // ---- statistics
    float mean = 0.f, rstd;
    if (KIND == LCMHIP_NORM_RMS) {
        const float ss = block_sum<NT / 64>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3], red);
        rstd = rsqrtf(ss / dim + eps);
    } else {
        mean = block_sum<NT / 64>(v[0] + v[1] + v[2] + v[3], red) / dim;
        const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
        const float q = block_sum<NT / 64>(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3, red);
        rstd = rsqrtf(q / dim + eps);
    }
    const float pw[4] = {wv.x, wv.y, wv.z, wv.w};
    const float pb[4] = {nb.x, nb.y, nb.z, nb.w};
    const float psc[4] = {scv.x, scv.y, scv.z, scv.w}, psh[4] = {shv.x, shv.y, shv.z, shv.w};
    const bool has_b = KIND == LCMHIP_NORM_LAYERNORM && b != nullptr, has_mod = scale != nullptr;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float tt = (v[j] - mean) * rstd * pw[j] + (has_b ? pb[j] : 0.f);
        o[j] = tt * (1.f + (has_mod ? psc[j] : 0.f)) + (has_mod ? psh[j] : 0.f);
    }
    T* yr = y + (size_t)row * dim + c4;
    if (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(yr)) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        uint2 pk;
        pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
        pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
        store8_through(yr, (uint64_t)pk.x | ((uint64_t)pk.y << 32));
    }
}

template <typename T, int KIND, int NT>
static void launch_norm_pending(float* x, const void* w, const void* b, const float* shift, const float* scale, void* y,
                                int rows, int rows_norm, float eps, const Pending& pd, const RowSteps& rs, hipStream_t st) {
#define LCM_NP(MAXS)                                                                                                   \
    {                                                                                                                  \
        if (pd.init)                                                                                                   \
            hipLaunchKernelGGL((k_norm_pending<T, KIND, NT, MAXS, true>), dim3(rows), dim3(NT), 0, st, x, (const T*)w, \
                               (const T*)b, shift, scale, (T*)y, rows_norm, eps, pd, rs);                              \
        else                                                                                                           \
            hipLaunchKernelGGL((k_norm_pending<T, KIND, NT, MAXS, false>), dim3(rows), dim3(NT), 0, st, x, (const T*)w, \
                               (const T*)b, shift, scale, (T*)y, rows_norm, eps, pd, rs);                              \
    }
    if (pd.splits == 1) LCM_NP(1)
    else if (pd.splits <= 4) LCM_NP(4)
    else if (pd.splits <= 8) LCM_NP(8)
    else LCM_NP(12)
#undef LCM_NP
}

template <typename T, int KIND>
static bool try_norm_pending(float* x, const void* w, const void* b, const float* shift, const float* scale, void* y,
                             int rows, int rows_norm, int dim, float eps, const Pending& pd, const RowSteps& rs, hipStream_t st) {
    if (pd.splits > 12) return false;
    switch (dim) {
        case 1024: launch_norm_pending<T, KIND, 256>(x, w, b, shift, scale, y, rows, rows_norm, eps, pd, rs, st); return true;
        case 2048: launch_norm_pending<T, KIND, 512>(x, w, b, shift, scale, y, rows, rows_norm, eps, pd, rs, st); return true;
        case 4096: launch_norm_pending<T, KIND, 1024>(x, w, b, shift, scale, y, rows, rows_norm, eps, pd, rs, st); return true;
        default: return false;
    }
}

// Many rows, nothing pending: one wave per row (4 rows per workgroup), the row in registers, statistics by wave
// shuffles only -- no LDS, no barrier.  dim <= 4096, dim % 256 == 0.
template <typename T, int KIND, int NV, bool NT>
__global__ __launch_bounds__(256) void k_norm_rows(const float* __restrict__ x, const T* __restrict__ w,
                                                   const T* __restrict__ b, const float* __restrict__ shift,
                                                   const float* __restrict__ scale, T* __restrict__ y, int rows, float eps,
                                                   RowSteps rs) {
    constexpr int dim = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;   // wave-uniform
    if (rs.row_step != nullptr && scale != nullptr) {   // this row's own entry of the modulation table
        const size_t moff = (size_t)min(max(rs.row_step[row], 0), rs.n_steps - 1) * rs.step_stride;
        shift += moff;
        scale += moff;
    }
    const float* xr = x + (size_t)row * dim;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (NT) {
            const nv_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nv_f4*>(xr + (lane + 64 * i) * 4));
            v[i] = make_float4(t[0], t[1], t[2], t[3]);
        } else {
            v[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        s += KIND == LCMHIP_NORM_RMS ? v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w
                                     : v[i].x + v[i].y + v[i].z + v[i].w;
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (KIND == LCMHIP_NORM_RMS) {
        rstd = rsqrtf(s / dim + eps);
    } else {
        mean = s / dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / dim + eps);
    }
    T* yr = y + (size_t)row * dim;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
        float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        float sc[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (scale != nullptr) {
            const float4 a = *reinterpret_cast<const float4*>(scale + col);
            const float4 h = *reinterpret_cast<const float4*>(shift + col);
            sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w;
            sh[0] = h.x; sh[1] = h.y; sh[2] = h.z; sh[3] = h.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pw = Elem<T>::load(w + col + j);
            const float pb = (KIND == LCMHIP_NORM_LAYERNORM && b != nullptr) ? Elem<T>::load(b + col + j) : 0.f;
            const float t = (o[j] - mean) * rstd * pw + pb;
            o[j] = t * (1.f + sc[j]) + sh[j];
        }
        if (sizeof(T) == 4) {
            if constexpr (NT) __builtin_nontemporal_store(nv_f4{o[0], o[1], o[2], o[3]}, reinterpret_cast<nv_f4*>(reinterpret_cast<float*>(yr) + col));
            else *reinterpret_cast<float4*>(reinterpret_cast<float*>(yr) + col) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
            pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
            if constexpr (NT) __builtin_nontemporal_store(((uint64_t)pk.y << 32) | pk.x, reinterpret_cast<uint64_t*>(reinterpret_cast<uint16_t*>(yr) + col));
            else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(yr) + col) = pk;
        }
    }
}

// The same rows when every row shares one modulation (or has none): a workgroup owns a CONTIGUOUS range of rows and
// walks it four rows at a time (one per wave), so that (i) the per-column operands -- norm weight / bias, 1 + scale,
// shift: up to 20 KB that every row of the kernel above fetches again through the CU's vector L1, more than the 12 KB
// the row itself moves -- are loaded once per wave and kept in registers, (ii) the next row of a wave is requested
// before the current one is reduced, and (iii) the grid is a few workgroups per CU instead of one per four rows.
// The arithmetic per element is that of k_norm_rows in the same order: the two kernels give the same bits.
template <typename T, int KIND, int NV, bool MOD, bool NT>
__global__ __launch_bounds__(256) void k_norm_rows_walk(const float* __restrict__ x, const T* __restrict__ w,
                                                        const T* __restrict__ b, const float* __restrict__ shift,
                                                        const float* __restrict__ scale, T* __restrict__ y, int rows,
                                                        int rows_per_wg, float eps) {
    constexpr int dim = NV * 256;
    constexpr bool BIAS = KIND == LCMHIP_NORM_LAYERNORM;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row_end = min(rows, ((int)blockIdx.x + 1) * rows_per_wg);
    int row = blockIdx.x * rows_per_wg + wid;
    if (row >= row_end) return;   // wave-uniform
    float pw[NV][4], pb[BIAS ? NV : 1][4], sc1[MOD ? NV : 1][4], sh[MOD ? NV : 1][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pw[i][j] = Elem<T>::load(w + col + j);
            if constexpr (BIAS) pb[i][j] = b != nullptr ? Elem<T>::load(b + col + j) : 0.f;
        }
        if constexpr (MOD) {
            const float4 a = *reinterpret_cast<const float4*>(scale + col);
            const float4 h = *reinterpret_cast<const float4*>(shift + col);
            sc1[i][0] = 1.f + a.x; sc1[i][1] = 1.f + a.y; sc1[i][2] = 1.f + a.z; sc1[i][3] = 1.f + a.w;
            sh[i][0] = h.x; sh[i][1] = h.y; sh[i][2] = h.z; sh[i][3] = h.w;
        }
    }
    auto load_row = [&](int r, float4 (&v)[NV]) {
        const float* xr = x + (size_t)r * dim;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (NT) {
                const nv_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nv_f4*>(xr + (lane + 64 * i) * 4));
                v[i] = make_float4(t[0], t[1], t[2], t[3]);
            } else {
                v[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
            }
        }
    };
    float4 v[NV], nxt[NV];
    load_row(row, v);
    for (; row < row_end; row += 4) {
        const bool more = row + 4 < row_end;
        if (more) load_row(row + 4, nxt);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            s += KIND == LCMHIP_NORM_RMS ? v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w
                                         : v[i].x + v[i].y + v[i].z + v[i].w;
        s = wave_sum(s);
        float mean = 0.f, rstd;
        if (KIND == LCMHIP_NORM_RMS) {
            rstd = rsqrtf(s / dim + eps);
        } else {
            mean = s / dim;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
                q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
            }
            q = wave_sum(q);
            rstd = rsqrtf(q / dim + eps);
        }
        T* yr = y + (size_t)row * dim;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = (o[j] - mean) * rstd * pw[i][j] + (BIAS ? pb[BIAS ? i : 0][j] : 0.f);
                o[j] = MOD ? t * sc1[MOD ? i : 0][j] + sh[MOD ? i : 0][j] : t * (1.f + 0.f) + 0.f;
            }
            if (sizeof(T) == 4) {
                if constexpr (NT) __builtin_nontemporal_store(nv_f4{o[0], o[1], o[2], o[3]}, reinterpret_cast<nv_f4*>(reinterpret_cast<float*>(yr) + col));
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(yr) + col) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
                pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
                if constexpr (NT) __builtin_nontemporal_store(((uint64_t)pk.y << 32) | pk.x, reinterpret_cast<uint64_t*>(reinterpret_cast<uint16_t*>(yr) + col));
                else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(yr) + col) = pk;
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = nxt[i];
        }
    }
}

template <typename T, int KIND>
static bool launch_norm_rows(const float* x, const void* w, const void* b, const float* shift, const float* scale, void* y,
                             int rows, int dim, float eps, hipStream_t st, const RowSteps& rs) {
    const dim3 grid((rows + 3) / 4), block(256);
    const bool nt = use_nontemporal((int64_t)rows * dim * (4 + (int64_t)sizeof(T)));
    if (rs.row_step == nullptr && dim <= 2048) {
        // one modulation for every row: the walking kernel with at least 8 rows (two per wave) and at most 32 workgroups
        // per CU (131 k rows of 1024 / 65 k rows of 2048: 2 / 4 / 8 / 16 / 24 / 32 / 64 per CU measured 0.71 / 0.68 / 0.67 /
        // 0.73 / 0.74 / 0.79 / 0.79 and 0.69 / 0.71 / 0.71 / 0.75 / 0.77 / 0.78 / 0.77 of the 8 TB/s peak; the kernel
        // above 0.70 / 0.63)
        constexpr int wg_per_cu = 32;
        {
            int rows_per_wg = std::max(8, (rows + 256 * wg_per_cu - 1) / (256 * wg_per_cu));
            rows_per_wg = (rows_per_wg + 3) / 4 * 4;
            const dim3 wgrid((rows + rows_per_wg - 1) / rows_per_wg);
#define LCM_WALK2(NV, MOD)                                                                                              \
    if (nt) hipLaunchKernelGGL((k_norm_rows_walk<T, KIND, NV, MOD, true>), wgrid, block, 0, st, x, (const T*)w, (const T*)b, \
                               shift, scale, (T*)y, rows, rows_per_wg, eps);                                            \
    else hipLaunchKernelGGL((k_norm_rows_walk<T, KIND, NV, MOD, false>), wgrid, block, 0, st, x, (const T*)w, (const T*)b,   \
                            shift, scale, (T*)y, rows, rows_per_wg, eps)
#define LCM_WALK(NV) { if (scale != nullptr) { LCM_WALK2(NV, true); } else { LCM_WALK2(NV, false); } }
            switch (dim) {
                case 256: LCM_WALK(1); return true;
                case 1024: LCM_WALK(4); return true;
                case 2048: LCM_WALK(8); return true;
                default: break;
            }
#undef LCM_WALK
#undef LCM_WALK2
        }
    }
#define LCM_ROWS(NV)                                                                                                  \
    if (nt) hipLaunchKernelGGL((k_norm_rows<T, KIND, NV, true>), grid, block, 0, st, x, (const T*)w, (const T*)b, shift, scale, \
                               (T*)y, rows, eps, rs);                                                                  \
    else hipLaunchKernelGGL((k_norm_rows<T, KIND, NV, false>), grid, block, 0, st, x, (const T*)w, (const T*)b, shift, scale,   \
                            (T*)y, rows, eps, rs)
    switch (dim) {
        case 256: { LCM_ROWS(1); } return true;
        case 1024: { LCM_ROWS(4); } return true;
        case 2048: { LCM_ROWS(8); } return true;
        case 4096: { LCM_ROWS(16); } return true;
        default: return false;
    }
#undef LCM_ROWS
}

int launch_norm_modulate(float* x, const void* w, const void* b, const float* shift, const float* scale,
                         void* y, int rows_total, int rows_norm, int dim, int norm_kind, float eps, int dtype,
                         const Pending* pend, hipStream_t st, const RowSteps* steps) {
    LCM_REQUIRE(dim % 4 == 0 && dim <= 8192, "norm: dim must be a multiple of 4 and <= 8192");
    LCM_REQUIRE((shift == nullptr) == (scale == nullptr), "norm: shift and scale go together");
    LCM_REQUIRE(w != nullptr, "norm: weight required");
    LCM_REQUIRE(rows_norm <= rows_total, "norm: rows_norm exceeds rows_total");
    if (rows_total <= 0) return LCMHIP_OK;
    Pending pd;
    if (pend) {
        pd = *pend;
        LCM_REQUIRE(pd.partial && pd.splits >= 1 && pd.M >= 1, "norm: bad pending product");
        LCM_REQUIRE(pd.dup ? rows_total <= 2 * pd.M : (rows_total <= pd.M || pd.extra_const != nullptr),
                    "norm: rows beyond the pending product need a constant");
    }
    const int rows = pend ? rows_total : rows_norm;
    if (rows <= 0) return LCMHIP_OK;
    RowSteps rs;
    if (steps) {
        rs = *steps;
        LCM_REQUIRE(rs.row_step == nullptr || rs.n_steps >= 1, "norm: per-row steps need a table size");
    }
    if (!pend && rows >= 1024) {
        bool done = false;
        if (dtype == LCMHIP_BF16)
            done = norm_kind == LCMHIP_NORM_RMS ? launch_norm_rows<bf16_t, LCMHIP_NORM_RMS>(x, w, b, shift, scale, y, rows, dim, eps, st, rs)
                                                : launch_norm_rows<bf16_t, LCMHIP_NORM_LAYERNORM>(x, w, b, shift, scale, y, rows, dim, eps, st, rs);
        else if (dtype == LCMHIP_F32)
            done = norm_kind == LCMHIP_NORM_RMS ? launch_norm_rows<float, LCMHIP_NORM_RMS>(x, w, b, shift, scale, y, rows, dim, eps, st, rs)
                                                : launch_norm_rows<float, LCMHIP_NORM_LAYERNORM>(x, w, b, shift, scale, y, rows, dim, eps, st, rs);
        if (done) {
            LCM_HIP_CHECK(hipGetLastError());
            return LCMHIP_OK;
        }
    }
    if (pend) {
        bool done = false;
        if (dtype == LCMHIP_BF16)
            done = norm_kind == LCMHIP_NORM_RMS ? try_norm_pending<bf16_t, LCMHIP_NORM_RMS>(x, w, b, shift, scale, y, rows, rows_norm, dim, eps, pd, rs, st)
                                                : try_norm_pending<bf16_t, LCMHIP_NORM_LAYERNORM>(x, w, b, shift, scale, y, rows, rows_norm, dim, eps, pd, rs, st);
        else if (dtype == LCMHIP_F32)
            done = norm_kind == LCMHIP_NORM_RMS ? try_norm_pending<float, LCMHIP_NORM_RMS>(x, w, b, shift, scale, y, rows, rows_norm, dim, eps, pd, rs, st)
                                                : try_norm_pending<float, LCMHIP_NORM_LAYERNORM>(x, w, b, shift, scale, y, rows, rows_norm, dim, eps, pd, rs, st);
        if (done) {
            LCM_HIP_CHECK(hipGetLastError());
            return LCMHIP_OK;
        }
    }
    // wide rows: 512 lanes per row keep twice as many loads in flight per CU
#define LCM_NORM(T, KIND)                                                                                      \
    if (dim >= 2048)                                                                                           \
        hipLaunchKernelGGL((k_norm_modulate<T, KIND, 512>), dim3(rows), dim3(512), 0, st, x, (const T*)w,      \
                           (const T*)b, shift, scale, (T*)y, dim, rows_norm, eps, pd, rs);                     \
    else                                                                                                       \
        hipLaunchKernelGGL((k_norm_modulate<T, KIND, 256>), dim3(rows), dim3(256), 0, st, x, (const T*)w,      \
                           (const T*)b, shift, scale, (T*)y, dim, rows_norm, eps, pd, rs)
    if (dtype == LCMHIP_BF16) {
        if (norm_kind == LCMHIP_NORM_RMS) { LCM_NORM(bf16_t, LCMHIP_NORM_RMS); } else { LCM_NORM(bf16_t, LCMHIP_NORM_LAYERNORM); }
    } else if (dtype == LCMHIP_F32) {
        if (norm_kind == LCMHIP_NORM_RMS) { LCM_NORM(float, LCMHIP_NORM_RMS); } else { LCM_NORM(float, LCMHIP_NORM_LAYERNORM); }
    } else {
        return fail(LCMHIP_ERR_INVALID, "norm: unknown dtype");
    }
#undef LCM_NORM
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused DDIM step (+ classifier-free guidance with rescale).  One workgroup per row; dim <= 4096.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float eps_from_output(float out, float x, const DdimScalars& sc) {
    if (sc.prediction_type == LCMHIP_PRED_EPSILON) return out;
    if (sc.prediction_type == LCMHIP_PRED_SAMPLE) return (x - sc.sqrt_a_t * out) / sc.sqrt_b_t;
    return sc.sqrt_a_t * out + sc.sqrt_b_t * x;  // v-prediction
}

__global__ __launch_bounds__(256) void k_ddim_step(const float* __restrict__ sample, const float* __restrict__ out_c,
                                                   const float* __restrict__ out_u, float* __restrict__ prev,
                                                   float* __restrict__ x0_out, int dim, DdimScalars sc, DdimIO io) {
    constexpr int MAXV = 4;
    __shared__ float red[4];
    __shared__ float thr_abs[4096];     // |x0| of the row (dynamic thresholding only)
    __shared__ float thr_pick[2];
    const size_t base = (size_t)blockIdx.x * dim;
    const int nv = dim / 4;
    // model output of row `r` (conditional rows first, then the unconditional ones): a finished tensor, or the sum of
    // the split-K slabs of the final projection plus its bias (same order as the finishing pass would use)
    auto model_out = [&](const float* fin, int r, int ch) {
        if (io.partial == nullptr) return *reinterpret_cast<const float4*>(fin + base + ch * 4);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        // eight slabs requested at once (a loop of dependent load-add steps costs a memory round trip per slab); slabs
        // past the split count repeat the last one and weigh 0; same summation order
        for (int sp0 = 0; sp0 < io.splits; sp0 += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = *reinterpret_cast<const float4*>(io.partial + ((size_t)min(sp0 + u, io.splits - 1) * io.rows_total + r) * dim + ch * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float on = sp0 + u < io.splits ? 1.f : 0.f;
                a.x = fmaf(t[u].x, on, a.x); a.y = fmaf(t[u].y, on, a.y); a.z = fmaf(t[u].z, on, a.z); a.w = fmaf(t[u].w, on, a.w);
            }
        }
        if (io.bias != nullptr) {
            if (io.bf16) {
                const bf16_t* bp = reinterpret_cast<const bf16_t*>(io.bias) + ch * 4;
                a.x += Elem<bf16_t>::load(bp); a.y += Elem<bf16_t>::load(bp + 1); a.z += Elem<bf16_t>::load(bp + 2); a.w += Elem<bf16_t>::load(bp + 3);
            } else {
                const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(io.bias) + ch * 4);
                a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            }
        }
        return a;
    };
    const int row_c = blockIdx.x, row_u = blockIdx.x + gridDim.x;
    float x[MAXV][4], e[MAXV][4], x0[MAXV][4];
    if (sc.mode == 1) {
        // guidance in epsilon space
        float c[MAXV][4];
        float sum_c = 0.f, sum_g = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int ch = threadIdx.x + i * 256;
            if (ch < nv) {
                const float4 xv = *reinterpret_cast<const float4*>(sample + base + ch * 4);
                const float4 cv = model_out(out_c, row_c, ch);
                const float4 uv = model_out(out_u, row_u, ch);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, cs[4] = {cv.x, cv.y, cv.z, cv.w},
                            us[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[i][j] = xs[j];
                    const float ec = eps_from_output(cs[j], xs[j], sc);
                    const float eu = eps_from_output(us[j], xs[j], sc);
                    c[i][j] = ec;
                    e[i][j] = eu + sc.guidance_scale * (ec - eu);
                    sum_c += ec;
                    sum_g += e[i][j];
                }
            }
        }
        const float mean_c = block_sum<4>(sum_c, red) / dim;
        const float mean_g = block_sum<4>(sum_g, red) / dim;
        float q_c = 0.f, q_g = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int ch = threadIdx.x + i * 256;
            if (ch < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dc = c[i][j] - mean_c, dg = e[i][j] - mean_g;
                    q_c += dc * dc;
                    q_g += dg * dg;
                }
            }
        }
        // torch.std default: unbiased (n - 1)
        const float std_pos = sqrtf(block_sum<4>(q_c, red) / (dim - 1));
        const float std_cfg = sqrtf(block_sum<4>(q_g, red) / (dim - 1));
        const float factor = sc.guidance_rescale * (std_pos / std_cfg) + (1.f - sc.guidance_rescale);
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[i][j] *= factor;
                x0[i][j] = (x[i][j] - sc.sqrt_b_t * e[i][j]) / sc.sqrt_a_t;  // step(prediction_type="epsilon")
            }
    } else {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int ch = threadIdx.x + i * 256;
            if (ch < nv) {
                const float4 xv = *reinterpret_cast<const float4*>(sample + base + ch * 4);
                const float4 cv = model_out(out_c, row_c, ch);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, cs[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[i][j] = xs[j];
                    if (sc.prediction_type == LCMHIP_PRED_EPSILON) {
                        e[i][j] = cs[j];
                        x0[i][j] = (xs[j] - sc.sqrt_b_t * cs[j]) / sc.sqrt_a_t;
                    } else if (sc.prediction_type == LCMHIP_PRED_SAMPLE) {
                        x0[i][j] = cs[j];
                        e[i][j] = (xs[j] - sc.sqrt_a_t * cs[j]) / sc.sqrt_b_t;
                    } else {
                        x0[i][j] = sc.sqrt_a_t * xs[j] - sc.sqrt_b_t * cs[j];
                        e[i][j] = sc.sqrt_a_t * cs[j] + sc.sqrt_b_t * xs[j];
                    }
                }
            }
        }
    }
    // Dynamic thresholding (ddim.py:242-280): s = clamp(quantile(|x0|, ratio), 1, max); x0 = clamp(x0, -s, s) / s.
    // The quantile interpolates linearly between the two order statistics around ratio * (dim - 1), like
    // torch.quantile; order statistics by counting (rank = smaller values + equal values at lower indices).
    float thr_s = 0.f;
    if (sc.thr_ratio >= 0.f) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int ch = threadIdx.x + i * 256;
            if (ch < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) thr_abs[ch * 4 + j] = fabsf(x0[i][j]);
            }
        }
        __syncthreads();
        const float pos = sc.thr_ratio * (float)(dim - 1);
        const int lo = (int)floorf(pos), hi = (int)ceilf(pos);
        const float frac = pos - (float)lo;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int ch = threadIdx.x + i * 256;
            if (ch < nv) {
                for (int j = 0; j < 4; ++j) {
                    const int idx = ch * 4 + j;
                    const float a = thr_abs[idx];
                    int rank = 0;
                    for (int k = 0; k < dim; ++k) {
                        const float b = thr_abs[k];
                        rank += (b < a || (b == a && k < idx)) ? 1 : 0;
                    }
                    if (rank == lo) thr_pick[0] = a;
                    if (rank == hi) thr_pick[1] = a;
                }
            }
        }
        __syncthreads();
        thr_s = thr_pick[0] + frac * (thr_pick[1] - thr_pick[0]);
        thr_s = fminf(fmaxf(thr_s, 1.f), sc.thr_max);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int ch = threadIdx.x + i * 256;
        if (ch < nv) {
            float p[4], z[4];
            float4 vn = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sc.sigma > 0.f) vn = *reinterpret_cast<const float4*>(io.variance_noise + base + ch * 4);
            const float vns[4] = {vn.x, vn.y, vn.z, vn.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ee = e[i][j] * sc.inv_eps_scaling;
                float z0 = x0[i][j];
                if (sc.thr_ratio >= 0.f) z0 = fminf(fmaxf(z0, -thr_s), thr_s) / thr_s;
                else if (sc.clip_sample_range >= 0.f) z0 = fminf(fmaxf(z0, -sc.clip_sample_range), sc.clip_sample_range);
                float pr = sc.sqrt_a_prev * z0 + sc.dir_coef * ee;
                if (sc.sigma > 0.f) pr += sc.sigma * vns[j];
                p[j] = fminf(fmaxf(pr, -sc.clip_noise), sc.clip_noise);
                z[j] = z0;
            }
            *reinterpret_cast<float4*>(prev + base + ch * 4) = make_float4(p[0], p[1], p[2], p[3]);
            if (x0_out != nullptr) *reinterpret_cast<float4*>(x0_out + base + ch * 4) = make_float4(z[0], z[1], z[2], z[3]);
            if (io.prev_cast != nullptr) {   // the next step's initial projection reads the latents in the operand dtype
                if (io.bf16) {
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(p[0]) | ((uint32_t)f32_to_bf16(p[1]) << 16);
                    pk.y = (uint32_t)f32_to_bf16(p[2]) | ((uint32_t)f32_to_bf16(p[3]) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(io.prev_cast) + base + ch * 4) = pk;
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(io.prev_cast) + base + ch * 4) = make_float4(p[0], p[1], p[2], p[3]);
                }
            }
        }
    }
}


// The same update for many rows of finished model outputs (no split-K slab, no thresholding): one wave per row,
// 4 rows per workgroup, every operand of a row requested up front (3 * NV float4 per lane), statistics by wave
// shuffles only -- no LDS, no barrier.  dim = 256 * NV.
template <int NV>
__global__ __launch_bounds__(256) void k_ddim_rows(const float* __restrict__ sample, const float* __restrict__ out_c,
                                                   const float* __restrict__ out_u, float* __restrict__ prev,
                                                   float* __restrict__ x0_out, int rows, DdimScalars sc,
                                                   const float* __restrict__ variance_noise) {
    constexpr int dim = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;   // wave-uniform
    const size_t base = (size_t)row * dim;
    float4 xv[NV], cv[NV], uv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) xv[i] = *reinterpret_cast<const float4*>(sample + base + (lane + 64 * i) * 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) cv[i] = *reinterpret_cast<const float4*>(out_c + base + (lane + 64 * i) * 4);
    if (sc.mode == 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) uv[i] = *reinterpret_cast<const float4*>(out_u + base + (lane + 64 * i) * 4);
    }
    float x[NV][4], e[NV][4], x0[NV][4];
    if (sc.mode == 1) {
        float c[NV][4];
        float sum_c = 0.f, sum_g = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float xs[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}, cs[4] = {cv[i].x, cv[i].y, cv[i].z, cv[i].w},
                        us[4] = {uv[i].x, uv[i].y, uv[i].z, uv[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[i][j] = xs[j];
                const float ec = eps_from_output(cs[j], xs[j], sc);
                const float eu = eps_from_output(us[j], xs[j], sc);
                c[i][j] = ec;
                e[i][j] = eu + sc.guidance_scale * (ec - eu);
                sum_c += ec;
                sum_g += e[i][j];
            }
        }
        const float mean_c = wave_sum(sum_c) / dim, mean_g = wave_sum(sum_g) / dim;
        float q_c = 0.f, q_g = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dc = c[i][j] - mean_c, dg = e[i][j] - mean_g;
                q_c += dc * dc;
                q_g += dg * dg;
            }
        const float std_pos = sqrtf(wave_sum(q_c) / (dim - 1));     // torch.std default: unbiased (n - 1)
        const float std_cfg = sqrtf(wave_sum(q_g) / (dim - 1));
        const float factor = sc.guidance_rescale * (std_pos / std_cfg) + (1.f - sc.guidance_rescale);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[i][j] *= factor;
                x0[i][j] = (x[i][j] - sc.sqrt_b_t * e[i][j]) / sc.sqrt_a_t;
            }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float xs[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}, cs[4] = {cv[i].x, cv[i].y, cv[i].z, cv[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[i][j] = xs[j];
                if (sc.prediction_type == LCMHIP_PRED_EPSILON) {
                    e[i][j] = cs[j];
                    x0[i][j] = (xs[j] - sc.sqrt_b_t * cs[j]) / sc.sqrt_a_t;
                } else if (sc.prediction_type == LCMHIP_PRED_SAMPLE) {
                    x0[i][j] = cs[j];
                    e[i][j] = (xs[j] - sc.sqrt_a_t * cs[j]) / sc.sqrt_b_t;
                } else {
                    x0[i][j] = sc.sqrt_a_t * xs[j] - sc.sqrt_b_t * cs[j];
                    e[i][j] = sc.sqrt_a_t * cs[j] + sc.sqrt_b_t * xs[j];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const size_t off = base + (lane + 64 * i) * 4;
        float4 vn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sc.sigma > 0.f) vn = *reinterpret_cast<const float4*>(variance_noise + off);
        const float vns[4] = {vn.x, vn.y, vn.z, vn.w};
        float p[4], z[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ee = e[i][j] * sc.inv_eps_scaling;
            float z0 = x0[i][j];
            if (sc.clip_sample_range >= 0.f) z0 = fminf(fmaxf(z0, -sc.clip_sample_range), sc.clip_sample_range);
            float pr = sc.sqrt_a_prev * z0 + sc.dir_coef * ee;
            if (sc.sigma > 0.f) pr += sc.sigma * vns[j];
            p[j] = fminf(fmaxf(pr, -sc.clip_noise), sc.clip_noise);
            z[j] = z0;
        }
        __builtin_nontemporal_store(vf4{p[0], p[1], p[2], p[3]}, reinterpret_cast<vf4*>(prev + off));
        if (x0_out != nullptr) __builtin_nontemporal_store(vf4{z[0], z[1], z[2], z[3]}, reinterpret_cast<vf4*>(x0_out + off));
    }
}

int launch_ddim_step(const float* sample, const float* out_c, const float* out_u, float* prev, float* x0, int rows,
                     int dim, const DdimScalars& sc, hipStream_t st, const DdimIO* io_in) {
    LCM_REQUIRE(dim % 4 == 0 && dim <= 4096 && dim > 1, "ddim: dim must be a multiple of 4 and <= 4096");
    DdimIO io;
    if (io_in) io = *io_in;
    if (io.partial != nullptr) {
        LCM_REQUIRE(io.splits >= 1 && io.rows_total >= (sc.mode == 1 ? 2 * rows : rows), "ddim: bad split-K model output");
    } else {
        LCM_REQUIRE(out_c != nullptr && (sc.mode == 0 || out_u != nullptr), "ddim: guidance needs the unconditional output");
    }
    LCM_REQUIRE(sc.sigma <= 0.f || io.variance_noise != nullptr, "ddim: eta > 0 needs variance noise");
    LCM_REQUIRE(sc.thr_ratio < 0.f || (sc.thr_ratio <= 1.f && sc.thr_max >= 1.f), "ddim: bad thresholding parameters");
    if (rows <= 0) return LCMHIP_OK;
    // many rows of finished outputs: the one-wave-per-row form (no LDS, no barriers, all loads up front)
    if (rows >= 1024 && io.partial == nullptr && io.prev_cast == nullptr && sc.thr_ratio < 0.f && (dim == 1024 || dim == 2048)) {
        const dim3 grid((rows + 3) / 4);
        if (dim == 1024) hipLaunchKernelGGL((k_ddim_rows<4>), grid, dim3(256), 0, st, sample, out_c, out_u, prev, x0, rows, sc, io.variance_noise);
        else hipLaunchKernelGGL((k_ddim_rows<8>), grid, dim3(256), 0, st, sample, out_c, out_u, prev, x0, rows, sc, io.variance_noise);
        LCM_HIP_CHECK(hipGetLastError());
        return LCMHIP_OK;
    }
    hipLaunchKernelGGL(k_ddim_step, dim3(rows), dim3(256), 0, st, sample, out_c, out_u, prev, x0, dim, sc, io);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_latents(const float* __restrict__ noise, float* __restrict__ lat,
                                                      int64_t n4, float sigma, float clip, void* cast, int bf16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(noise)[i];
        v.x = fminf(fmaxf(v.x * sigma, -clip), clip);
        v.y = fminf(fmaxf(v.y * sigma, -clip), clip);
        v.z = fminf(fmaxf(v.z * sigma, -clip), clip);
        v.w = fminf(fmaxf(v.w * sigma, -clip), clip);
        reinterpret_cast<float4*>(lat)[i] = v;
        if (cast != nullptr) {
            if (bf16) {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
                reinterpret_cast<uint2*>(cast)[i] = pk;
            } else {
                reinterpret_cast<float4*>(cast)[i] = v;
            }
        }
    }
}

int launch_init_latents(const float* noise, float* lat, int64_t n, float sigma, float clip, hipStream_t st, void* cast,
                        int dtype) {
    LCM_REQUIRE(n % 4 == 0, "latents: size must be a multiple of 4");
    if (n == 0) return LCMHIP_OK;
    hipLaunchKernelGGL(k_init_latents, dim3(grid_for(n / 4, 256)), dim3(256), 0, st, noise, lat, n / 4, sigma, clip, cast,
                       dtype == LCMHIP_BF16 ? 1 : 0);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// End of a generation step (finalize_step of the generators): active row r belongs to document doc = doc_rows[r].
// A document still inside its prompt (prompt_lens[doc] > step) keeps the prompt embedding already stored at
// seqs[doc][step]; every other row stores its new embedding there (cast to the buffer's dtype).  Both cosines of
// the stopping rules are taken on the fp32 value: against the EOS vector and against the document's previous
// sentence seqs[doc][step - 1].  cos_out: (n_active, 2).  One workgroup per row, dim <= 4096.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_finalize_step(const float* __restrict__ new_emb, T* __restrict__ seqs,
                                                       const int32_t* __restrict__ doc_rows,
                                                       const int32_t* __restrict__ prompt_lens, int step, int total_len,
                                                       int dim, const float* __restrict__ eos, float* __restrict__ cos_out) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const int doc = doc_rows ? doc_rows[r] : r;
    const bool in_prompt = prompt_lens != nullptr && prompt_lens[doc] > step;      // uniform per workgroup
    T* cur = seqs + ((size_t)doc * total_len + step) * dim;
    const T* prev = step > 0 ? cur - dim : nullptr;
    float dot_e = 0.f, dot_p = 0.f, nv = 0.f, ne = 0.f, np = 0.f;
    for (int c = threadIdx.x * 8; c < dim; c += 256 * 8) {
        float v[8];
        if (in_prompt) {
            load8<T>(cur + c, v);
        } else {
            load8<float>(new_emb + (size_t)r * dim + c, v);
            store8<T>(cur + c, v);
        }
        float e[8], p[8];
        if (eos != nullptr) load8<float>(eos + c, e);
        if (prev != nullptr) load8<T>(prev + c, p);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            nv += v[j] * v[j];
            if (eos != nullptr) { dot_e += e[j] * v[j]; ne += e[j] * e[j]; }
            if (prev != nullptr) { dot_p += p[j] * v[j]; np += p[j] * p[j]; }
        }
    }
    nv = block_sum<4>(nv, red);
    dot_e = block_sum<4>(dot_e, red); ne = block_sum<4>(ne, red);
    dot_p = block_sum<4>(dot_p, red); np = block_sum<4>(np, red);
    if (threadIdx.x == 0) {
        const float eps = 1e-8f;     // torch.nn.functional.cosine_similarity: each norm is clamped from below
        const float nvs = fmaxf(sqrtf(nv), eps);
        cos_out[2 * r] = eos != nullptr ? dot_e / (nvs * fmaxf(sqrtf(ne), eps)) : 0.f;
        cos_out[2 * r + 1] = prev != nullptr ? dot_p / (nvs * fmaxf(sqrtf(np), eps)) : 0.f;
    }
}

int launch_finalize_step(const float* new_emb, void* seqs, int dtype, const int32_t* doc_rows, const int32_t* prompt_lens,
                         int n_active, int step, int total_len, int dim, const float* eos, float* cos_out, hipStream_t st) {
    LCM_REQUIRE(dim % 8 == 0 && dim <= 4096, "finalize_step: dim must be a multiple of 8, at most 4096");
    LCM_REQUIRE(step >= 0 && step < total_len, "finalize_step: step outside the sequence buffer");
    if (n_active <= 0) return LCMHIP_OK;
    if (dtype == LCMHIP_BF16)
        hipLaunchKernelGGL((k_finalize_step<bf16_t>), dim3(n_active), dim3(256), 0, st, new_emb, (bf16_t*)seqs, doc_rows,
                           prompt_lens, step, total_len, dim, eos, cos_out);
    else if (dtype == LCMHIP_F32)
        hipLaunchKernelGGL((k_finalize_step<float>), dim3(n_active), dim3(256), 0, st, new_emb, (float*)seqs, doc_rows,
                           prompt_lens, step, total_len, dim, eos, cos_out);
    else
        return fail(LCMHIP_ERR_INVALID, "finalize_step: unknown dtype");
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// LayerNorm / RMSNorm of whole rows held in the operand dtype, in place: the q / k LayerNorm of
// lcm/nn/transformer/attention.py:143-199 (over the full model dimension, before the head split).  Blocks [0, rows)
// take the rows of x0, blocks [rows, 2 rows) those of x1 (its own parameters) when x1 is given.  dim <= 8192.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void k_row_norm(T* __restrict__ x0, const T* __restrict__ w0, const T* __restrict__ b0,
                                                  T* __restrict__ x1, const T* __restrict__ w1, const T* __restrict__ b1,
                                                  int rows, int dim, float eps) {
    __shared__ float red[4];
    int r = blockIdx.x;
    T* x = x0;
    const T* w = w0;
    const T* b = b0;
    if (r >= rows) {
        r -= rows; x = x1; w = w1; b = b1;
    }
    x += (size_t)r * dim;
    float v[4][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (threadIdx.x + 256 * i) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) v[i][t] = 0.f;
        if (c < dim) {
            load8<T>(x + c, v[i]);
#pragma unroll
            for (int t = 0; t < 8; ++t) s += KIND == LCMHIP_NORM_RMS ? v[i][t] * v[i][t] : v[i][t];
        }
    }
    s = block_sum<4>(s, red);
    float mean = 0.f, rstd;
    if (KIND == LCMHIP_NORM_RMS) {
        rstd = rsqrtf(s / dim + eps);
    } else {
        mean = s / dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (threadIdx.x + 256 * i) * 8;
            if (c < dim) {
#pragma unroll
                for (int t = 0; t < 8; ++t) q += (v[i][t] - mean) * (v[i][t] - mean);
            }
        }
        q = block_sum<4>(q, red);
        rstd = rsqrtf(q / dim + eps);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (threadIdx.x + 256 * i) * 8;
        if (c < dim) {
            float pw[8], pb[8];
            load8<T>(w + c, pw);
#pragma unroll
            for (int t = 0; t < 8; ++t) pb[t] = 0.f;
            if (KIND == LCMHIP_NORM_LAYERNORM && b != nullptr) load8<T>(b + c, pb);
#pragma unroll
            for (int t = 0; t < 8; ++t) v[i][t] = (v[i][t] - mean) * rstd * pw[t] + pb[t];
            store8<T>(x + c, v[i]);
        }
    }
}

int launch_row_norm(void* x0, const void* w0, const void* b0, void* x1, const void* w1, const void* b1, int rows, int dim,
                    int norm_kind, float eps, int dtype, hipStream_t st) {
    LCM_REQUIRE(dim % 8 == 0 && dim <= 8192, "row norm: dim must be a multiple of 8 and <= 8192");
    LCM_REQUIRE(x0 && w0 && (x1 == nullptr || w1 != nullptr), "row norm: rows and weights required");
    if (rows <= 0) return LCMHIP_OK;
    const dim3 grid(x1 ? 2 * rows : rows), block(256);
#define LCM_RN(T, K) hipLaunchKernelGGL((k_row_norm<T, K>), grid, block, 0, st, (T*)x0, (const T*)w0, (const T*)b0, (T*)x1, \
                                        (const T*)w1, (const T*)b1, rows, dim, eps)
    if (dtype == LCMHIP_BF16) {
        if (norm_kind == LCMHIP_NORM_RMS) LCM_RN(bf16_t, LCMHIP_NORM_RMS); else LCM_RN(bf16_t, LCMHIP_NORM_LAYERNORM);
    } else if (dtype == LCMHIP_F32) {
        if (norm_kind == LCMHIP_NORM_RMS) LCM_RN(float, LCMHIP_NORM_RMS); else LCM_RN(float, LCMHIP_NORM_LAYERNORM);
    } else {
        return fail(LCMHIP_ERR_INVALID, "row norm: unknown dtype");
    }
#undef LCM_RN
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// q/k/v: (rows, d).  Rotates consecutive pairs (2i, 2i+1) of every 128-wide head of q (in place) and
// k by the angle table row `positions[r]`, and scatters k, v to cache row `cache_rows[r]`.
template <typename T>
__global__ __launch_bounds__(256) void k_rope_store(T* __restrict__ q, const T* __restrict__ k,
                                                    const T* __restrict__ v, T* __restrict__ kc, T* __restrict__ vc,
                                                    const int32_t* __restrict__ cache_rows,
                                                    const int32_t* __restrict__ positions,
                                                    const float* __restrict__ cos_tab,
                                                    const float* __restrict__ sin_tab, int d, int cap, int use_rope) {
    const int r = blockIdx.x;
    const size_t src = (size_t)r * d;
    const int crow = cache_rows[r];
    const int pos = positions[r];
    for (int c = threadIdx.x * 8; c < d; c += blockDim.x * 8) {
        float qv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, kv[8], vv[8];
        if (q != nullptr) load8<T>(q + src + c, qv);   // without q: a plain scatter of k / v rows into the cache
        load8<T>(k + src + c, kv);
        load8<T>(v + src + c, vv);
        if (use_rope) {
            const int pair0 = (c & 127) >> 1;  // pair index inside the head
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float cs = cos_tab[pos * 64 + pair0 + j], sn = sin_tab[pos * 64 + pair0 + j];
                rope_rotate(qv[2 * j], qv[2 * j + 1], cs, sn);
                rope_rotate(kv[2 * j], kv[2 * j + 1], cs, sn);
            }
            if (q != nullptr) store8<T>(q + src + c, qv);
        }
        const size_t dst = kv_cache_offset(crow, c, cap, d);   // 8 consecutive dims never straddle a head
        store8<T>(kc + dst, kv);
        store8<T>(vc + dst, vv);
    }
}

int launch_rope_store(void* q, const void* k, const void* v, void* kc, void* vc, const int32_t* cache_rows,
                      const int32_t* positions, const float* cos_tab, const float* sin_tab, int rows, int d,
                      int cap, int use_rope, int dtype, hipStream_t st) {
    LCM_REQUIRE(d % 128 == 0, "rope: model dim must be a multiple of the 128-wide head");
    if (rows <= 0) return LCMHIP_OK;
    if (dtype == LCMHIP_BF16)
        hipLaunchKernelGGL((k_rope_store<bf16_t>), dim3(rows), dim3(256), 0, st, (bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, (bf16_t*)kc, (bf16_t*)vc, cache_rows, positions, cos_tab, sin_tab, d, cap,
                           use_rope);
    else
        hipLaunchKernelGGL((k_rope_store<float>), dim3(rows), dim3(256), 0, st, (float*)q, (const float*)k,
                           (const float*)v, (float*)kc, (float*)vc, cache_rows, positions, cos_tab, sin_tab, d, cap,
                           use_rope);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void k_add_pos(float* __restrict__ x, const T* __restrict__ table,
                                                 const int32_t* __restrict__ positions, int d) {
    const int r = blockIdx.x;
    const T* row = table + (size_t)positions[r] * d;
    for (int c = threadIdx.x * 8; c < d; c += blockDim.x * 8) {
        float a[8], b[8];
        load8<float>(x + (size_t)r * d + c, a);
        load8<T>(row + c, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
        store8<float>(x + (size_t)r * d + c, a);
    }
}

int launch_add_pos(float* x, const void* table, const int32_t* positions, int rows, int d, int dtype, hipStream_t st) {
    LCM_REQUIRE(d % 8 == 0, "pos: dim must be a multiple of 8");
    if (rows <= 0) return LCMHIP_OK;
    if (dtype == LCMHIP_BF16)
        hipLaunchKernelGGL((k_add_pos<bf16_t>), dim3(rows), dim3(256), 0, st, x, (const bf16_t*)table, positions, d);
    else
        hipLaunchKernelGGL((k_add_pos<float>), dim3(rows), dim3(256), 0, st, x, (const float*)table, positions, d);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// fairseq2 SinusoidalPositionEncoder as the reference's frontend uses it (lcm/nn/transformer/factory.py:199-204): the
// table is built in fp32 -- outer(position, exp(-i ln(10000) / (h - 1))), sin of it in the first h = d / 2 columns, cos
// in the last h -- and added to the sequence in fp32.  Computed on the fly here (one exp and one sin / cos per element).
__global__ __launch_bounds__(256) void k_add_sine_pos(float* __restrict__ x, const int32_t* __restrict__ positions, int d) {
    const int r = blockIdx.x, h = d / 2;
    const float pos = (float)positions[r];
    const float step = -9.210340371976184f / (float)(h - 1);   // -ln(10000) / (h - 1)
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const int i = c < h ? c : c - h;
        const float a = pos * expf((float)i * step);
        x[(size_t)r * d + c] += c < h ? sinf(a) : cosf(a);
    }
}

int launch_add_sine_pos(float* x, const int32_t* positions, int rows, int d, hipStream_t st) {
    LCM_REQUIRE(d % 2 == 0 && d >= 4, "pos: the sinusoidal encoding needs an even dim");
    if (rows <= 0) return LCMHIP_OK;
    hipLaunchKernelGGL(k_add_sine_pos, dim3(rows), dim3(256), 0, st, x, positions, d);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void k_cast_rows(const float* __restrict__ x, T* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8];
        load8<float>(x + i * 8, a);
        store8<T>(y + i * 8, a);
    }
}
template <typename T, bool ACC>
__global__ __launch_bounds__(256) void k_take_rows(float* __restrict__ x, const T* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8], b[8];
        load8<T>(y + i * 8, b);
        if (ACC) {
            load8<float>(x + i * 8, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] += a[j];
        }
        store8<float>(x + i * 8, b);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_scale_cols(float* __restrict__ x, const T* __restrict__ scale, int d) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x * 8; c < d; c += blockDim.x * 8) {
        float a[8], b[8];
        load8<float>(x + (size_t)r * d + c, a);
        load8<T>(scale + c, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] *= b[j];
        store8<float>(x + (size_t)r * d + c, a);
    }
}

static inline unsigned flat_grid(int64_t n8) { return (unsigned)std::min<int64_t>(8192, (n8 + 255) / 256); }

int launch_cast_rows(const float* x, void* y, int64_t n, int dtype, hipStream_t st) {
    LCM_REQUIRE(n % 8 == 0, "cast: element count must be a multiple of 8");
    if (n <= 0) return LCMHIP_OK;
    if (dtype == LCMHIP_BF16) hipLaunchKernelGGL((k_cast_rows<bf16_t>), dim3(flat_grid(n / 8)), dim3(256), 0, st, x, (bf16_t*)y, n / 8);
    else hipLaunchKernelGGL((k_cast_rows<float>), dim3(flat_grid(n / 8)), dim3(256), 0, st, x, (float*)y, n / 8);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

int launch_take_rows(float* x, const void* y, int64_t n, int dtype, int accumulate, hipStream_t st) {
    LCM_REQUIRE(n % 8 == 0, "take: element count must be a multiple of 8");
    if (n <= 0) return LCMHIP_OK;
    const dim3 grid(flat_grid(n / 8)), block(256);
    if (dtype == LCMHIP_BF16) {
        if (accumulate) hipLaunchKernelGGL((k_take_rows<bf16_t, true>), grid, block, 0, st, x, (const bf16_t*)y, n / 8);
        else hipLaunchKernelGGL((k_take_rows<bf16_t, false>), grid, block, 0, st, x, (const bf16_t*)y, n / 8);
    } else {
        if (accumulate) hipLaunchKernelGGL((k_take_rows<float, true>), grid, block, 0, st, x, (const float*)y, n / 8);
        else hipLaunchKernelGGL((k_take_rows<float, false>), grid, block, 0, st, x, (const float*)y, n / 8);
    }
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

int launch_scale_cols(float* x, const void* scale, int rows, int d, int dtype, hipStream_t st) {
    LCM_REQUIRE(d % 8 == 0, "scale: dim must be a multiple of 8");
    if (rows <= 0) return LCMHIP_OK;
    if (dtype == LCMHIP_BF16) hipLaunchKernelGGL((k_scale_cols<bf16_t>), dim3(rows), dim3(256), 0, st, x, (const bf16_t*)scale, d);
    else hipLaunchKernelGGL((k_scale_cols<float>), dim3(rows), dim3(256), 0, st, x, (const float*)scale, d);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// Real FFT of one row per workgroup, packed like the reference's FFTInterface (sonar_normalizer/builder.py:58-84):
// forward = torch.fft.rfft(x, norm="backward") as [Re X_0 .. Re X_{n/2}, Im X_1 .. Im X_{n/2-1}]; inverse = torch.fft.irfft
// of that packing with Im X_0 = Im X_{n/2} = 0.  A radix-2 decimation-in-time transform of n complex points in LDS
// (bit-reversed load, log2 n stages of n / 2 butterflies, twiddles e^{-+2 pi i k / n} from sincospif into LDS once per
// workgroup); the inverse runs the conjugate transform on the Hermitian extension and keeps the real parts / n.  A
// normalizer transform of a few rows per sentence: written for exactness, not speed.
template <bool INV>
__global__ __launch_bounds__(256) void k_rfft_pack(const float* x, float* y, int n, int logn) {   // y may be x
    extern __shared__ float2 fft_lds[];
    float2* a = fft_lds;
    float2* tw = fft_lds + n;
    const int tid = threadIdx.x, h = n / 2;
    const float* xr = x + (size_t)blockIdx.x * n;
    float* yr = y + (size_t)blockIdx.x * n;
    for (int k = tid; k < h; k += 256) {
        float sn, cs;
        sincospif(2.f * (float)k / (float)n, &sn, &cs);
        tw[k] = make_float2(cs, INV ? sn : -sn);
    }
    for (int i = tid; i < n; i += 256) {
        float2 v;
        if (!INV) {
            v = make_float2(xr[i], 0.f);
        } else {   // X_i for i <= n/2, conj(X_{n-i}) above
            const int k = i <= h ? i : n - i;
            const float im = (k == 0 || k == h) ? 0.f : xr[h + k];
            v = make_float2(xr[k], i <= h ? im : -im);
        }
        a[__brev((unsigned)i) >> (32 - logn)] = v;
    }
    __syncthreads();
    for (int s = 1; s <= logn; ++s) {
        const int half = 1 << (s - 1);
        for (int j = tid; j < h; j += 256) {
            const int pos = j & (half - 1);
            const int i0 = ((j >> (s - 1)) << s) + pos, i1 = i0 + half;
            const float2 w = tw[pos << (logn - s)], b = a[i1], u = a[i0];
            const float2 t = make_float2(w.x * b.x - w.y * b.y, w.x * b.y + w.y * b.x);
            a[i0] = make_float2(u.x + t.x, u.y + t.y);
            a[i1] = make_float2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
    const float inv_n = 1.f / (float)n;
    for (int i = tid; i < n; i += 256) {
        if (!INV) yr[i] = i <= h ? a[i].x : a[i - h].y;
        else yr[i] = a[i].x * inv_n;
    }
}

int launch_rfft_pack(const float* x, float* y, int64_t rows, int dim, int inverse, hipStream_t st) {
    LCM_REQUIRE(dim >= 64 && dim <= 4096 && (dim & (dim - 1)) == 0, "rfft: dim must be a power of two in [64, 4096]");
    LCM_REQUIRE(rows >= 0 && rows <= 0x7fffffff, "rfft: bad row count");
    if (rows == 0) return LCMHIP_OK;
    int logn = 0;
    while ((1 << logn) < dim) ++logn;
    const size_t lds = (size_t)dim * 12;   // n points + n / 2 twiddles, 8 bytes each
    if (inverse) hipLaunchKernelGGL((k_rfft_pack<true>), dim3((unsigned)rows), dim3(256), lds, st, x, y, dim, logn);
    else hipLaunchKernelGGL((k_rfft_pack<false>), dim3((unsigned)rows), dim3(256), lds, st, x, y, dim, logn);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// cache[rows_idx[i]][:] = vec[:]   (vec null: zeros)
template <typename T>
__global__ __launch_bounds__(256) void k_fill_rows(T* __restrict__ cache, const T* __restrict__ vec,
                                                   const int32_t* __restrict__ rows_idx, int d, int cap) {
    const int crow = rows_idx[blockIdx.x];
    for (int c = threadIdx.x * 8; c < d; c += blockDim.x * 8) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (vec != nullptr) load8<T>(vec + c, a);
        store8<T>(cache + kv_cache_offset(crow, c, cap, d), a);
    }
}

int launch_fill_rows(void* cache, const void* vec, const int32_t* rows_idx, int n_rows, int d, int cap, int dtype,
                     hipStream_t st) {
    LCM_REQUIRE(d % 128 == 0, "fill: dim must be a multiple of the 128-wide head");
    if (n_rows <= 0) return LCMHIP_OK;
    if (dtype == LCMHIP_BF16)
        hipLaunchKernelGGL((k_fill_rows<bf16_t>), dim3(n_rows), dim3(256), 0, st, (bf16_t*)cache, (const bf16_t*)vec,
                           rows_idx, d, cap);
    else
        hipLaunchKernelGGL((k_fill_rows<float>), dim3(n_rows), dim3(256), 0, st, (float*)cache, (const float*)vec,
                           rows_idx, d, cap);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

template <typename T, bool SILU>
__global__ __launch_bounds__(256) void k_cast(const float* __restrict__ x, T* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8];
        load8<float>(x + i * 8, a);
        if (SILU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = a[j] / (1.f + __expf(-a[j]));
        }
        store8<T>(y + i * 8, a);
    }
}

static int cast_impl(const float* x, void* y, int64_t n, int dtype, bool silu, hipStream_t st) {
    LCM_REQUIRE(n % 8 == 0, "cast: size must be a multiple of 8");
    if (n == 0) return LCMHIP_OK;
    const int g = grid_for(n / 8, 256);
    if (dtype == LCMHIP_BF16) {
        if (silu) hipLaunchKernelGGL((k_cast<bf16_t, true>), dim3(g), dim3(256), 0, st, x, (bf16_t*)y, n / 8);
        else hipLaunchKernelGGL((k_cast<bf16_t, false>), dim3(g), dim3(256), 0, st, x, (bf16_t*)y, n / 8);
    } else {
        if (silu) hipLaunchKernelGGL((k_cast<float, true>), dim3(g), dim3(256), 0, st, x, (float*)y, n / 8);
        else hipLaunchKernelGGL((k_cast<float, false>), dim3(g), dim3(256), 0, st, x, (float*)y, n / 8);
    }
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}
int launch_silu_cast(const float* x, void* y, int64_t n, int dtype, hipStream_t st) { return cast_impl(x, y, n, dtype, true, st); }
int launch_cast(const float* x, void* y, int64_t n, int dtype, hipStream_t st) { return cast_impl(x, y, n, dtype, false, st); }

// dst[r][:] = src[r % src_rows][:]  (duplicates the conditional rows for the unconditional half)
__global__ __launch_bounds__(256) void k_broadcast_rows(const float* __restrict__ src, float* __restrict__ dst,
                                                        int src_rows, int d4) {
    const int r = blockIdx.x;
    const float4* s = reinterpret_cast<const float4*>(src) + (size_t)(r % src_rows) * d4;
    float4* o = reinterpret_cast<float4*>(dst) + (size_t)r * d4;
    for (int c = threadIdx.x; c < d4; c += blockDim.x) o[c] = s[c];
}

int launch_broadcast_rows(const float* src, float* dst, int rows, int d, hipStream_t st) {
    // rows = number of destination rows; sources are the first rows / 2
    LCM_REQUIRE(d % 4 == 0 && rows % 2 == 0, "broadcast: bad shape");
    if (rows <= 0) return LCMHIP_OK;
    hipLaunchKernelGGL(k_broadcast_rows, dim3(rows), dim3(256), 0, st, src, dst, rows / 2, d / 4);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void k_to_f32(const T* __restrict__ x, float* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8];
        load8<T>(x + i * 8, a);
        store8<float>(y + i * 8, a);
    }
}

// Stream-copy calibration: what the part delivers for a plain 16-byte-per-lane copy (the yardstick of the fused
// elementwise kernels).  Grid sized to the CUs (8 workgroups per CU), four 16-byte pieces in flight per lane,
// optionally non-temporal loads and stores.
template <bool NT, bool TILED>
__global__ __launch_bounds__(256) void k_stream_copy(const vf4* __restrict__ src, vf4* __restrict__ dst, int64_t n16) {
    constexpr int U = 4;
    if constexpr (TILED) {
        // a workgroup walks whole 16-KB tiles (U contiguous 4-KB rows of pieces): its requests stay within few DRAM pages
        const int64_t tiles = (n16 + 256 * U - 1) / (256 * U);
        for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
            const int64_t i0 = t * 256 * U + threadIdx.x;
            vf4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (i0 + u * 256 < n16) v[u] = NT ? __builtin_nontemporal_load(src + i0 + u * 256) : src[i0 + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (i0 + u * 256 < n16) {
                    if (NT) __builtin_nontemporal_store(v[u], dst + i0 + u * 256);
                    else dst[i0 + u * 256] = v[u];
                }
        }
    } else {
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += U * stride) {
            vf4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (i + u * stride < n16) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (i + u * stride < n16) {
                    if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
                    else dst[i + u * stride] = v[u];
                }
        }
    }
}

// mode bit 0: non-temporal loads / stores; bit 1: tile-contiguous walk instead of a grid-wide stride;
// bits 8..: workgroups per CU (0 = 8)
int launch_stream_copy(const void* src, void* dst, int64_t bytes, int mode, hipStream_t st) {
    LCM_REQUIRE(bytes >= 0 && bytes % 16 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                "stream_copy: 16-byte aligned pointers and size");
    if (bytes == 0) return LCMHIP_OK;
    const int64_t n16 = bytes / 16;
    const int per_cu = (mode >> 8) > 0 ? (mode >> 8) : 8;
    const int g = (int)std::min<int64_t>(256 * per_cu, (n16 + 255) / 256);
    const vf4* a = (const vf4*)src;
    vf4* b = (vf4*)dst;
    switch (mode & 3) {
        case 0: hipLaunchKernelGGL((k_stream_copy<false, false>), dim3(g), dim3(256), 0, st, a, b, n16); break;
        case 1: hipLaunchKernelGGL((k_stream_copy<true, false>), dim3(g), dim3(256), 0, st, a, b, n16); break;
        case 2: hipLaunchKernelGGL((k_stream_copy<false, true>), dim3(g), dim3(256), 0, st, a, b, n16); break;
        default: hipLaunchKernelGGL((k_stream_copy<true, true>), dim3(g), dim3(256), 0, st, a, b, n16); break;
    }
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

int launch_copy_rows_to_f32(const void* src, float* dst, int64_t n, int dtype, hipStream_t st) {
    LCM_REQUIRE(n % 8 == 0, "copy: size must be a multiple of 8");
    if (n == 0) return LCMHIP_OK;
    const int g = grid_for(n / 8, 256);
    if (dtype == LCMHIP_BF16) hipLaunchKernelGGL((k_to_f32<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, dst, n / 8);
    else hipLaunchKernelGGL((k_to_f32<float>), dim3(g), dim3(256), 0, st, (const float*)src, dst, n / 8);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

}  // namespace lcmhip
