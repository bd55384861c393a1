// C = A * W^T with fused epilogues, for the row counts of sentence-level generation
// (M = 1 .. a few hundred rows, weights streamed once from HBM).
//
// Tiling: one workgroup = 4 waves = one BM x BN output tile, K walked in 128-byte slices.  Both
// operands go through LDS in full 128-byte lines (rows padded to 144 B: conflict-free for the
// 16-byte fragment reads) so that every global load instruction covers whole cache lines; the
// next slice is held in registers while the current one feeds the MFMAs (one barrier per slice).
// MFMA orientation: the WEIGHT rows are the matrix-core A operand and the ACTIVATION rows the B
// operand, so each lane ends up with 4 consecutive output columns of one output row -- one
// vector load of bias/gate and one vector store per accumulator.
//   bf16 operands: v_mfma_f32_16x16x32_bf16;  f32 operands: v_mfma_f32_16x16x4_f32 (exact fp32).
#include "common.h"

namespace lcmhip {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
    static constexpr int kLaneBytes = 16;  // bytes of K one lane feeds per MFMA
    using Frag = bf16x8;
    __device__ static __forceinline__ Frag ld(const char* p) { return *reinterpret_cast<const Frag*>(p); }
    __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mma<float> {
    static constexpr int kLaneBytes = 4;
    using Frag = float;
    __device__ static __forceinline__ Frag ld(const char* p) { return *reinterpret_cast<const float*>(p); }
    __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

constexpr int kSliceBytes = 128;  // bytes of K per tile row per slice
constexpr int kRowBytes = 144;    // padded LDS row

template <typename T, int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs p) {
    constexpr bool DUAL = (EPI == LCMHIP_EPI_SWIGLU);
    constexpr int R = BM + BN * (DUAL ? 2 : 1);  // staged rows per slice
    constexpr int NCH = R / 32;                  // 16-byte chunks per thread per slice
    constexpr int WN = BN / 32;                  // waves along N
    constexpr int KSTEP_BYTES = 4 * Mma<T>::kLaneBytes;
    constexpr int KSTEPS = kSliceBytes / KSTEP_BYTES;
    static_assert((BM / 32) * WN == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) char smem[2 * R * kRowBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wn = wid % WN, wm = wid / WN;
    const int n0 = blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;
    const size_t kbytes = (size_t)p.K * sizeof(T);

    // per-thread source pointers of its chunks
    const char* src[NCH];
    int dst[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int q = tid + i * 256;
        const int row = q >> 3, c = q & 7;
        const char* base;
        if (row < BM) {
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            base = reinterpret_cast<const char*>(p.A) + (size_t)m * p.lda * sizeof(T);
        } else if (row < BM + BN) {
            base = reinterpret_cast<const char*>(p.W) + (size_t)(n0 + row - BM) * kbytes;
        } else {
            base = reinterpret_cast<const char*>(p.W2) + (size_t)(n0 + row - BM - BN) * kbytes;
        }
        src[i] = base + c * 16;
        dst[i] = row * kRowBytes + c * 16;
    }

    f32x4 acc[2][2], acc2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    const int KT = (int)(kbytes / kSliceBytes);
    uint4 regs[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) regs[i] = *reinterpret_cast<const uint4*>(src[i]);
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<uint4*>(smem + dst[i]) = regs[i];
    __syncthreads();

    const int frow = lane & 15;
    const int fk = (lane >> 4) * Mma<T>::kLaneBytes;
    for (int kt = 0; kt < KT; ++kt) {
        const char* buf = smem + (kt & 1) * R * kRowBytes;
        if (kt + 1 < KT) {
            const size_t off = (size_t)(kt + 1) * kSliceBytes;
#pragma unroll
            for (int i = 0; i < NCH; ++i) regs[i] = *reinterpret_cast<const uint4*>(src[i] + off);
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int kb = ks * KSTEP_BYTES + fk;
            typename Mma<T>::Frag w[2], w2[2], x[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                w[t] = Mma<T>::ld(buf + (BM + wn * 32 + t * 16 + frow) * kRowBytes + kb);
                if constexpr (DUAL) w2[t] = Mma<T>::ld(buf + (BM + BN + wn * 32 + t * 16 + frow) * kRowBytes + kb);
                x[t] = Mma<T>::ld(buf + (wm * 32 + t * 16 + frow) * kRowBytes + kb);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    acc[tn][tm] = Mma<T>::mma(w[tn], x[tm], acc[tn][tm]);
                    if constexpr (DUAL) acc2[tn][tm] = Mma<T>::mma(w2[tn], x[tm], acc2[tn][tm]);
                }
        }
        if (kt + 1 < KT) {
            char* nbuf = smem + ((kt + 1) & 1) * R * kRowBytes;
#pragma unroll
            for (int i = 0; i < NCH; ++i) *reinterpret_cast<uint4*>(nbuf + dst[i]) = regs[i];
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds columns n..n+3 of row m for each (tn, tm)
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* bias2 = reinterpret_cast<const T*>(p.bias2);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 32 + tn * 16 + 4 * (lane >> 4);
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
        for (int tm = 0; tm < 2; ++tm) {
            const int m = m0 + wm * 32 + tm * 16 + (lane & 15);
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
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if constexpr (DUAL) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float g = v[r];
                        const float in = acc2[tn][tm][r] + bv2[r];
                        v[r] = in * (g / (1.f + __expf(-g)));
                    }
                }
                const size_t orow = p.out_rows ? (size_t)p.out_rows[m] : (size_t)m;
                if constexpr (EPI == LCMHIP_EPI_STORE_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) =
                        make_float4(v[0], v[1], v[2], v[3]);
                } else if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) =
                        make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 o;
                    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + orow * p.ldo + n) = o;
                }
            }
        }
    }
}

template <typename T, int BM, int BN>
static int dispatch_epi(const GemmArgs& a, hipStream_t st) {
    const int rows = a.M + (a.epilogue == LCMHIP_EPI_RESID ? a.extra_rows : 0);
    dim3 grid(a.N / BN, (rows + BM - 1) / BM), block(256);
    switch (a.epilogue) {
        case LCMHIP_EPI_STORE: hipLaunchKernelGGL((k_gemm<T, BM, BN, LCMHIP_EPI_STORE>), grid, block, 0, st, a); break;
        case LCMHIP_EPI_STORE_F32: hipLaunchKernelGGL((k_gemm<T, BM, BN, LCMHIP_EPI_STORE_F32>), grid, block, 0, st, a); break;
        case LCMHIP_EPI_RELU: hipLaunchKernelGGL((k_gemm<T, BM, BN, LCMHIP_EPI_RELU>), grid, block, 0, st, a); break;
        case LCMHIP_EPI_SWIGLU: hipLaunchKernelGGL((k_gemm<T, BM, BN, LCMHIP_EPI_SWIGLU>), grid, block, 0, st, a); break;
        case LCMHIP_EPI_RESID: hipLaunchKernelGGL((k_gemm<T, BM, BN, LCMHIP_EPI_RESID>), grid, block, 0, st, a); break;
        default: return fail(LCMHIP_ERR_INVALID, "gemm: unknown epilogue");
    }
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// Tile choice by row count: few rows -> narrow-M tiles keep the grid large; otherwise 128 x 32
// (all rows of a sentence step share one pass over the weight tile).
static void pick_tile(const GemmArgs& a, int& bm, int& bn) {
    const int rows = a.M + (a.epilogue == LCMHIP_EPI_RESID ? a.extra_rows : 0);
    const bool dual = a.epilogue == LCMHIP_EPI_SWIGLU;  // two weight tiles: keep LDS under 64 KiB
    if (rows <= 32 && !dual && a.N % 128 == 0) { bm = 32; bn = 128; }
    else if (rows <= 64 && a.N % 64 == 0) { bm = 64; bn = 64; }
    else { bm = 128; bn = 32; }
}

template <typename T>
static int dispatch_tile(const GemmArgs& a, hipStream_t st) {
    int bm, bn;
    pick_tile(a, bm, bn);
    if (bm == 32) return dispatch_epi<T, 32, 128>(a, st);
    if (bm == 64) return dispatch_epi<T, 64, 64>(a, st);
    return dispatch_epi<T, 128, 32>(a, st);
}

int launch_gemm(const GemmArgs& a, hipStream_t st) {
    LCM_REQUIRE(a.M >= 1 && a.N >= 1 && a.K >= 1, "gemm: empty problem");
    const size_t es = dtype_size(a.dtype);
    LCM_REQUIRE((a.K * es) % kSliceBytes == 0, "gemm: K must cover whole 128-byte slices");
    LCM_REQUIRE(a.N % 32 == 0, "gemm: N must be a multiple of 32");
    LCM_REQUIRE(((size_t)a.lda * es) % 16 == 0 && ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0,
                "gemm: operands must be 16-byte aligned");
    LCM_REQUIRE(a.epilogue != LCMHIP_EPI_SWIGLU || a.W2 != nullptr, "gemm: SWIGLU needs W2");
    LCM_REQUIRE(a.ldo % 4 == 0 && ((uintptr_t)a.out % 16) == 0, "gemm: output must be 16-byte aligned");
    LCM_REQUIRE(a.extra_rows == 0 || (a.epilogue == LCMHIP_EPI_RESID && a.extra_const), "gemm: extra rows need RESID");
    if (a.dtype == LCMHIP_BF16) return dispatch_tile<bf16_t>(a, st);
    if (a.dtype == LCMHIP_F32) return dispatch_tile<float>(a, st);
    return fail(LCMHIP_ERR_INVALID, "gemm: unknown dtype");
}

}  // namespace lcmhip
