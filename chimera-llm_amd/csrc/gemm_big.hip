// C = A * W^T for bf16 operands and MANY rows (M >= 256: context prefill, large document batches, schedule
// tables): a matrix-core-bound GEMM, unlike the weight-streaming kernel of gemm.hip.
//
// Tile 128 (rows) x 128 (weight rows) x 64 (K) per workgroup of 4 waves (2 x 2); each wave owns 64 x 64 =
// 4 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16 (weight rows = A operand, activation rows = B operand,
// same accumulator convention as gemm.hip, so the epilogue is shared).  Both operands are K-contiguous, so a
// tile row is 128 contiguous bytes.  Staging is global -> LDS by LDS-DMA (global_load_lds, 16 bytes per lane):
// one wave instruction fills 8 rows x 128 B of the stage, lane l -> (row l / 8, 16-byte slot l % 8).  The LDS
// image stays lane-linear; bank conflicts of the fragment reads (16 rows x one 16-byte slot) are removed by
// swizzling on the SOURCE side: slot s of row r holds the K piece s ^ ((r / 2) % 8).  Two stages: the DMA of K tile
// t + 1 runs while the matrix cores consume tile t; one barrier per K tile.  (A three-stage variant with one
// workgroup per CU and a counted vmcnt across a bare s_barrier measured 20-25 % slower than two stages with two
// workgroups per CU.)
#include <cstdlib>

#include "common.h"
#include "gemm_epilogue.h"

namespace lcmhip {

namespace {
constexpr int kBM = 128, kBN = 128;
constexpr int kRowB = 128;                      // bytes of K per tile row and stage (64 bf16)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
}  // namespace

// Epilogue of the 128 x 128 kernel.  In the accumulators a lane holds 4 columns of 16 different rows, so storing
// from there touches 16 rows x 32..64 bytes per instruction.  Each wave instead passes its 64 x 64 tile through its
// own piece of the (now idle) staging memory, 32 rows at a time, and stores row-contiguous 128..256-byte segments;
// bias and activations are applied on the way in, the residual update (read-modify-write) and row maps on the way out.
template <typename T, int EPI>
__device__ __forceinline__ void big_epilogue(const GemmArgs& p, const f32x4 (&acc)[4][4], char* smem, int wid,
                                             int m_wave0, int n_wave0, int split, int lane) {
    constexpr bool DUAL = (EPI == LCMHIP_EPI_SWIGLU);
    constexpr int TNO = DUAL ? 2 : 4;            // output column tiles of the wave
    constexpr int NC = 16 * TNO;                 // output columns of the wave
    constexpr int LROW = NC + 4;                 // padded row (floats): 16 rows x one 16-byte piece hit 16 bank groups
    constexpr bool F32OUT = (EPI == EPI_PARTIAL || EPI == LCMHIP_EPI_STORE_F32 || EPI == LCMHIP_EPI_RESID || sizeof(T) == 4);
    float* tile = reinterpret_cast<float*>(smem) + wid * (32 * 68);
    const int quad = 4 * (lane >> 4), frow = lane & 15;
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* bias2 = reinterpret_cast<const T*>(p.bias2);
    float bv[TNO][4], bv2[TNO][4];
#pragma unroll
    for (int tn = 0; tn < TNO; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n_wave0 + tn * 16 + quad + r;
            bv[tn][r] = (EPI != EPI_PARTIAL && bias) ? Elem<T>::load(bias + n) : 0.f;
            bv2[tn][r] = (DUAL && bias2) ? Elem<T>::load(bias2 + n) : 0.f;
        }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();   // every wave is done with the staging buffers / with the previous half's tile
#pragma unroll
        for (int tn = 0; tn < TNO; ++tn)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tm = 2 * half + t;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[tn][tm][r] + bv[tn][r];
                    if constexpr (EPI == LCMHIP_EPI_RELU) v[r] = fmaxf(v[r], 0.f);
                    if constexpr (DUAL) {
                        const float g = v[r];
                        v[r] = (acc[tn + 2][tm][r] + bv2[tn][r]) * (g / (1.f + __expf(-g)));
                    }
                }
                *reinterpret_cast<float4*>(tile + (16 * t + frow) * LROW + 16 * tn + quad) = make_float4(v[0], v[1], v[2], v[3]);
            }
        __syncthreads();
        if constexpr (F32OUT) {
            constexpr int LPR = NC / 4, RPP = 64 / LPR;     // lanes per row, rows per pass
            const int c = (lane % LPR) * 4, n = n_wave0 + c;
            float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (EPI == LCMHIP_EPI_RESID) {
                if (p.gate && !p.gate_rows) g = *reinterpret_cast<const float4*>(p.gate + n);
            }
#pragma unroll
            for (int ps = 0; ps < 32 / RPP; ++ps) {
                const int r = ps * RPP + lane / LPR;
                const int m = m_wave0 + 32 * half + r;
                float4 v = *reinterpret_cast<const float4*>(tile + r * LROW + c);
                if constexpr (EPI == EPI_PARTIAL) {
                    if (m < p.M) *reinterpret_cast<float4*>(p.partial + ((size_t)split * p.M + m) * p.N + n) = v;
                } else if constexpr (EPI == LCMHIP_EPI_RESID) {
                    const bool real = m < p.M;
                    const bool extra = !real && m < p.M + p.extra_rows;
                    if (extra) v = *reinterpret_cast<const float4*>(p.extra_const + n);
                    if (p.gate_rows != nullptr && real)
                        g = *reinterpret_cast<const float4*>(p.gate + (size_t)min(max(p.gate_rows[m], 0), p.gate_steps - 1) * p.gate_stride + n);
                    if (real || extra) {
                        float4* dstp = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n);
                        float4 o = *dstp;
                        o.x += g.x * v.x; o.y += g.y * v.y; o.z += g.z * v.z; o.w += g.w * v.w;
                        *dstp = o;
                    }
                } else {
                    if (m < p.M) {
                        size_t oidx;
                        if (p.out_rows == nullptr) oidx = (size_t)m * p.ldo + n;
                        else if (p.out_head_cap > 0) oidx = kv_cache_offset(p.out_rows[m], n, p.out_head_cap, p.N);
                        else oidx = (size_t)p.out_rows[m] * p.ldo + n;
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + oidx) = v;
                    }
                }
            }
        } else {
            constexpr int LPR = NC / 8, RPP = 64 / LPR;
            const int c = (lane % LPR) * 8, n = n_wave0 + c;
#pragma unroll
            for (int ps = 0; ps < 32 / RPP; ++ps) {
                const int r = ps * RPP + lane / LPR;
                const int m = m_wave0 + 32 * half + r;
                const float4 a = *reinterpret_cast<const float4*>(tile + r * LROW + c);
                const float4 b = *reinterpret_cast<const float4*>(tile + r * LROW + c + 4);
                if (m < p.M) {
                    size_t oidx;
                    if (p.out_rows == nullptr) oidx = (size_t)m * p.ldo + n;
                    else if (p.out_head_cap > 0) oidx = kv_cache_offset(p.out_rows[m], n, p.out_head_cap, p.N);
                    else oidx = (size_t)p.out_rows[m] * p.ldo + n;
                    uint4 o;
                    o.x = (uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16);
                    o.y = (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16);
                    o.z = (uint32_t)f32_to_bf16(b.x) | ((uint32_t)f32_to_bf16(b.y) << 16);
                    o.w = (uint32_t)f32_to_bf16(b.z) | ((uint32_t)f32_to_bf16(b.w) << 16);
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.out) + oidx) = o;
                }
            }
        }
    }
}

// WMV wave rows: the tile is (64 * WMV) rows x 128 weight rows, 2 * WMV waves of 64 x 64 each.
template <int EPI, int WMV>
__global__ __launch_bounds__(128 * WMV) void k_gemm_big(GemmArgs p) {
    constexpr bool DUAL = (EPI == LCMHIP_EPI_SWIGLU);
    constexpr int BM = 64 * WMV, NWAVES = 2 * WMV;
    constexpr int STAGE = (BM + kBN) * kRowB;
    constexpr int NGRP = (BM + kBN) / 8 / NWAVES;   // 8-row groups each wave stages per K tile
    constexpr int AGRP = BM / 8 / NWAVES;           // ... of which activation rows
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int bx = blockIdx.x, bz = blockIdx.z;
    const int m0 = bz * BM;
    const size_t kbytes = (size_t)p.K * 2;
    const int nk = (int)(kbytes / kRowB);
    const int kps = (nk + (int)gridDim.y - 1) / (int)gridDim.y;
    const int k_begin = blockIdx.y * kps;
    const int k_end = min(nk, k_begin + kps);

    // ---- staging plan: wave w fills row groups w, w + 4, ... (8 rows each); groups 0..15 = activation rows,
    // 16..31 = weight rows.  This lane's row inside a group is lane / 8, its slot lane % 8 receives K piece
    // (lane % 8) ^ (lane / 8).
    const int srow = lane >> 3;
    // swizzle: slot s of tile row r holds K piece s ^ ((r / 2) % 8).  Two consecutive rows share one 256-byte bank
    // line, so the 16 rows x one piece of a fragment read land on 16 distinct 16-byte bank groups.  For the rows
    // this lane stages, r = 8 * (wid + NWAVES j) + srow, hence (r / 2) % 8 = 4 * (wid % 2) + srow / 2 for every j.
    const size_t piece_off = (size_t)(((lane & 7) ^ (4 * (wid & 1) + (srow >> 1))) * 16);
    const char* src[NGRP];
#pragma unroll
    for (int j = 0; j < NGRP; ++j) {
        const int grp = wid + NWAVES * j;       // j < AGRP: activations, else weights (wave-uniform)
        if (j < AGRP) {
            int m = m0 + grp * 8 + srow;
            m = m < p.M ? m : p.M - 1;          // rows past the end repeat the last row; never stored
            src[j] = reinterpret_cast<const char*>(p.A) + (size_t)m * p.lda * 2 + piece_off;
        } else {
            const int t = (grp - BM / 8) * 8 + srow;   // weight row inside the tile, 0..127
            if constexpr (DUAL) {
                // wave column wn' = t / 64 owns 32 gate rows followed by 32 inner rows of the same 32 outputs
                const int wcol = t >> 6, rr = t & 63;
                const size_t row = (size_t)bx * (kBN / 2) + wcol * 32 + (rr & 31);
                src[j] = reinterpret_cast<const char*>(rr < 32 ? p.W : p.W2) + row * kbytes + piece_off;
            } else {
                src[j] = reinterpret_cast<const char*>(p.W) + ((size_t)bx * kBN + t) * kbytes + piece_off;
            }
        }
    }
    auto stage = [&](int kt, int buf) {
        const size_t koff = (size_t)kt * kRowB;
        char* dst = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < NGRP; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + koff), (lptr_t)(dst + (wid + NWAVES * j) * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: row = tile row base + lane % 16, K piece of k-step ks = 4 * ks + lane / 16
    const int frow = lane & 15;
    const int fsw = (frow >> 1) & 7;   // tile row bases are multiples of 16
    const int x_row0 = wm * 64 + frow;
    const int w_row0 = BM + wn * 64 + frow;

    stage(k_begin, 0);
    for (int kt = k_begin; kt < k_end; ++kt) {
        const int buf = (kt - k_begin) & 1;
        // tile kt has landed for every wave, and every wave is done reading the other stage
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* base = smem + buf * STAGE;
        // all 16 fragments of the K tile first, then the 32 matrix-core instructions
        bf16x8 w[2][4], x[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = ((4 * ks + (lane >> 4)) ^ fsw) * 16;
#pragma unroll
            for (int t = 0; t < 4; ++t) w[ks][t] = *reinterpret_cast<const bf16x8*>(base + (w_row0 + 16 * t) * kRowB + slot);
#pragma unroll
            for (int t = 0; t < 4; ++t) x[ks][t] = *reinterpret_cast<const bf16x8*>(base + (x_row0 + 16 * t) * kRowB + slot);
        }
        // the DMA of the next tile is issued behind this tile's fragment reads, not ahead of them
        if (kt + 1 < k_end) stage(kt + 1, buf ^ 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks][tn], x[ks][tm], acc[tn][tm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }

    static_assert(2 * STAGE >= NWAVES * 32 * 68 * 4, "the epilogue tiles must fit the staging buffers");
    big_epilogue<bf16_t, EPI>(p, acc, smem, wid, m0 + wm * 64, DUAL ? bx * (kBN / 2) + wn * 32 : bx * kBN + wn * 64,
                              blockIdx.y, lane);
}

// K splits for the split-K (slab) form: enough workgroups for the chip, at least 8 K tiles per split, and the
// splits divide the K tiles
int gemm_big_splits(const GemmArgs& a, int max_splits) {
    const int tiles = (a.N / kBN) * ((a.M + kBM - 1) / kBM);   // counted in 128-row tiles
    const int nk = a.K / 64;
    int s = 1;
    while (tiles * s < 256 && nk % (s * 2) == 0 && nk / (s * 2) >= 8 && s * 2 <= (max_splits > 0 ? max_splits : 64)) s *= 2;
    return s;
}

// The 128 x 128 kernel pays once its workgroups (tiles x K splits when a slab is available) cover most of the
// chip; below that the 64 x 64 weight-streaming kernel is faster (measured: M = 256 x N = 2048 14 vs 25 us).
bool gemm_big_applicable(const GemmArgs& a, bool may_split) {
    static const int min_rows = [] { const char* e = getenv("LCMHIP_BIG_MIN_ROWS"); return e ? atoi(e) : 256; }();
    const int wrows = a.epilogue == LCMHIP_EPI_SWIGLU ? 2 * a.N : a.N;
    if (!(a.dtype == LCMHIP_BF16 && a.M >= min_rows && a.K % 64 == 0 && wrows % kBN == 0 && a.lda % 8 == 0)) return false;
    const int tiles = (wrows / kBN) * ((a.M + kBM - 1) / kBM);
    const int s = may_split ? gemm_big_splits(a, a.max_splits) : 1;
    return tiles * s >= 192;
}

// epi: LCMHIP_EPI_* or EPI_PARTIAL; splits >= 1 (only with EPI_PARTIAL); m_rows: rows the row tiles must cover
int launch_gemm_big(const GemmArgs& a, int epi, int splits, int m_rows, hipStream_t st) {
    const int wrows = epi == LCMHIP_EPI_SWIGLU ? 2 * a.N : a.N;
    // (256 x 128 tiles with 8 waves and one workgroup per CU measured 10-20 % slower than two 128 x 128 workgroups.)
    dim3 grid(wrows / kBN, splits, (m_rows + kBM - 1) / kBM), block(256);
#define LCM_BIG(E) hipLaunchKernelGGL((k_gemm_big<E, 2>), grid, block, 0, st, a)
    switch (epi) {
        case LCMHIP_EPI_STORE: LCM_BIG(LCMHIP_EPI_STORE); break;
        case LCMHIP_EPI_STORE_F32: LCM_BIG(LCMHIP_EPI_STORE_F32); break;
        case LCMHIP_EPI_RELU: LCM_BIG(LCMHIP_EPI_RELU); break;
        case LCMHIP_EPI_SWIGLU: LCM_BIG(LCMHIP_EPI_SWIGLU); break;
        case LCMHIP_EPI_RESID: LCM_BIG(LCMHIP_EPI_RESID); break;
        case EPI_PARTIAL: LCM_BIG(EPI_PARTIAL); break;
        default: return fail(LCMHIP_ERR_INVALID, "gemm: unknown epilogue");
    }
#undef LCM_BIG
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

}  // namespace lcmhip
