// Epilogue shared by the GEMM kernels.  Accumulator convention (weight rows are the matrix-core A operand,
// activation rows the B operand): for every 16 x 16 tile (tn, tm) of a wave, lane l holds the 4 consecutive
// output columns n = 16 * tn + 4 * (l / 16) .. + 3 of output row m = 16 * tm + l % 16.
#pragma once
#include "common.h"

namespace lcmhip {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;   // native vector: stays in registers (HIP's uint4 is a struct)

constexpr int EPI_PARTIAL = 100;   // internal: raw fp32 partial tile to the split-K slab

// m_wave0: first output row of the wave; n_wave0: first output column of the wave's tile 0 (for SWIGLU in the
// space of output columns: accumulator tiles [0, TN/2) hold the gate rows, [TN/2, TN) the inner rows of the
// same columns); split: index of the K split (EPI_PARTIAL only).
template <typename T, int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, const f32x4 (&acc)[TN][TM], int m_wave0, int n_wave0,
                                              int split, int lane) {
    constexpr bool DUAL = (EPI == LCMHIP_EPI_SWIGLU);
    const int quad = 4 * (lane >> 4);
    if constexpr (EPI == EPI_PARTIAL) {
        float* slab = p.partial + (size_t)split * p.M * p.N;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n_wave0 + tn * 16 + quad;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int m = m_wave0 + tm * 16 + (lane & 15);
                if (m >= p.M) continue;
                float* dst = slab + (size_t)m * p.N + n;
                if (p.slab_through) store16_through(dst, acc[tn][tm]);
                else *reinterpret_cast<f32x4*>(dst) = acc[tn][tm];
            }
        }
    } else {
        const T* bias = reinterpret_cast<const T*>(p.bias);
        const T* bias2 = reinterpret_cast<const T*>(p.bias2);
        const float slope = EPI == LCMHIP_EPI_RELU ? act_slope<T>(p) : 0.f;
        const int kind = act_kind(p);
        constexpr int TNO = DUAL ? TN / 2 : TN;   // output column tiles per wave
#pragma unroll
        for (int tn = 0; tn < TNO; ++tn) {
            const int n = n_wave0 + tn * 16 + quad;
            float bv[4] = {0.f, 0.f, 0.f, 0.f}, bv2[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {1.f, 1.f, 1.f, 1.f};
            if (bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = Elem<T>::load(bias + n + r);
            }
            if constexpr (DUAL) {
                if (bias2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv2[r] = Elem<T>::load(bias2 + n + r);
                }
            }
            if constexpr (EPI == LCMHIP_EPI_RESID) {
                if (p.gate) {
                    const float4 g = *reinterpret_cast<const float4*>(p.gate + n);
                    gv[0] = g.x; gv[1] = g.y; gv[2] = g.z; gv[3] = g.w;
                }
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int m = m_wave0 + tm * 16 + (lane & 15);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[tn][tm][r] + bv[r];
                if constexpr (EPI == LCMHIP_EPI_RESID) {
                    // rows [M, M + extra_rows) take a constant instead of the product
                    const bool real = m < p.M;
                    const bool extra = !real && m < p.M + p.extra_rows;
                    if (extra) {
                        const float4 c = *reinterpret_cast<const float4*>(p.extra_const + n);
                        v[0] = c.x; v[1] = c.y; v[2] = c.z; v[3] = c.w;
                    }
                    if (p.gate_rows != nullptr && real) {
                        const float4 g = *reinterpret_cast<const float4*>(
                            p.gate + (size_t)min(max(p.gate_rows[m], 0), p.gate_steps - 1) * p.gate_stride + n);
                        gv[0] = g.x; gv[1] = g.y; gv[2] = g.z; gv[3] = g.w;
                    }
                    if (real || extra) {
                        float4* dstp = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n);
                        float4 o = *dstp;
                        o.x += gv[0] * v[0]; o.y += gv[1] * v[1]; o.z += gv[2] * v[2]; o.w += gv[3] * v[3];
                        *dstp = o;
                    }
                } else {
                    if (m >= p.M) continue;
                    if constexpr (EPI == LCMHIP_EPI_RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], kind, slope);
                    }
                    if constexpr (DUAL) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float g = v[r];
                            const float in = acc[tn + TN / 2][tm][r] + bv2[r];
                            v[r] = in * (g / (1.f + __expf(-g)));
                        }
                    }
                    size_t oidx;
                    if (p.out_rows == nullptr) oidx = (size_t)m * p.ldo + n;
                    else if (p.out_head_cap > 0) oidx = kv_cache_offset(p.out_rows[m], n, p.out_head_cap, p.N);
                    else oidx = (size_t)p.out_rows[m] * p.ldo + n;
                    if constexpr (EPI == LCMHIP_EPI_STORE_F32 || sizeof(T) == 4) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + oidx) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        uint2 o;
                        o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                        o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + oidx) = o;
                    }
                }
            }
        }
    }
}

}  // namespace lcmhip
