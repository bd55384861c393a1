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
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace lcmhip {

namespace {
constexpr int kBM = 128, kBN = 128;
constexpr int kRowB = 128;                      // bytes of K per tile row and stage (64 bf16)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
}  // namespace

// Stacked products: the weight-row tile index counts through all matrices.  Selects the tile's matrix (weight, bias,
// output) in the workgroup's copy of the arguments and returns the tile index inside that matrix.
__device__ __forceinline__ int big_select_matrix(GemmArgs& p, int bx, int tiles_per_matrix) {
    p.stack_sel = 0;
    if (p.stack > 1) {
        const int g = bx / tiles_per_matrix;
        bx -= g * tiles_per_matrix;
        p.stack_sel = g;
        // constant indices only: a run-time index into the argument struct would move all of it to scratch memory
        if (g == 1) {
            p.W = p.stack_W[0]; p.bias = p.stack_bias[0]; p.out = p.stack_out[0];
        } else if (g >= 2) {
            p.W = p.stack_W[1]; p.bias = p.stack_bias[1]; p.out = p.stack_out[1];
        }
    }
    return bx;
}

// Residual values of one 32-row round of a wave's output tile, in the lane order in which big_epilogue stores it
// (RESID only): requested ahead of the round -- round 0 by the kernel itself before the tile is staged (requesting
// it earlier, with the DMA of the last K tile, measured the same: the read-modify-write is bound by its bytes).
// NC: output columns of a wave (64, or 32 for the SwiGLU pairing and for the 32-weight-row wave tile).
template <int EPI, int NC>
__host__ __device__ constexpr int big_resid_rounds() {
    return EPI == LCMHIP_EPI_RESID ? 32 / (64 / (NC / 4)) : 1;
}
template <int EPI, int NC>
__device__ __forceinline__ void big_load_resid(const GemmArgs& p, int m_wave0, int n_wave0, int half, int lane,
                                               float4 (&o)[big_resid_rounds<EPI, NC>()]) {
    if constexpr (EPI == LCMHIP_EPI_RESID) {
        constexpr int LPR = NC / 4, RPP = 64 / LPR;
        const int n = n_wave0 + (lane % LPR) * 4;
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int m = m_wave0 + 32 * half + ps * RPP + lane / LPR;
            o[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < p.M + p.extra_rows)
                o[ps] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.out) + (size_t)m * p.ldo + n);
        }
    }
}

// Epilogue of the many-row kernels.  In the accumulators a lane holds 4 columns of 16 different rows, so storing
// from there touches 16 rows x 32..64 bytes per instruction.  Each wave instead passes its (16 TM) x 64 tile through
// its own piece of the (now idle) staging memory, 32 rows at a time, and stores row-contiguous 128..256-byte segments;
// bias and activations are applied on the way in, the residual update (read-modify-write) and row maps on the way out.
template <int EPI, int TNW>
__host__ __device__ constexpr int big_wave_cols() { return 16 * (EPI == LCMHIP_EPI_SWIGLU ? TNW / 2 : TNW); }

// TNW: weight-row tiles of the wave (4, or 2 for the 256 x 128 tile); TM: its activation-row tiles.
template <typename T, int EPI, int TM, int TNW = 4>
__device__ __forceinline__ void big_epilogue(const GemmArgs& p, const f32x4 (&acc)[TNW][TM], char* smem, int wid,
                                             int m_wave0, int n_wave0, int split, int lane,
                                             float4 (&res_cur)[big_resid_rounds<EPI, big_wave_cols<EPI, TNW>()>()]) {
    constexpr bool DUAL = (EPI == LCMHIP_EPI_SWIGLU);
    constexpr int TNO = DUAL ? TNW / 2 : TNW;    // output column tiles of the wave
    constexpr int NC = 16 * TNO;                 // output columns of the wave
    constexpr int LROW = NC + 4;                 // padded row (floats): 16 rows x one 16-byte piece hit 16 bank groups
    constexpr bool F32OUT = (EPI == EPI_PARTIAL || EPI == LCMHIP_EPI_STORE_F32 || EPI == LCMHIP_EPI_RESID || sizeof(T) == 4);
    float* tile = reinterpret_cast<float*>(smem) + wid * (32 * 68);
    const int quad = 4 * (lane >> 4), frow = lane & 15;
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* bias2 = reinterpret_cast<const T*>(p.bias2);
    const float slope = EPI == LCMHIP_EPI_RELU ? act_slope<T>(p) : 0.f;
    const int kind = act_kind(p);
    float bv[TNO][4], bv2[TNO][4];
#pragma unroll
    for (int tn = 0; tn < TNO; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n_wave0 + tn * 16 + quad + r;
            bv[tn][r] = (EPI != EPI_PARTIAL && bias) ? Elem<T>::load(bias + n) : 0.f;
            bv2[tn][r] = (DUAL && bias2) ? Elem<T>::load(bias2 + n) : 0.f;
        }
    // RESID: the residual values of a 32-row round are requested one round ahead (round 0 by the caller, see
    // big_resid_rounds / big_load_resid), so that the read-modify-write does not wait for a memory round trip per round
    constexpr int RPASS = big_resid_rounds<EPI, NC>();
    float4 res_next[RPASS];
    auto load_resid = [&](int half, float4 (&o)[RPASS]) { big_load_resid<EPI, NC>(p, m_wave0, n_wave0, half, lane, o); };
#pragma unroll
    for (int half = 0; half < TM / 2; ++half) {
        __syncthreads();   // every wave is done with the staging buffers / with the previous round's tile
#pragma unroll
        for (int tn = 0; tn < TNO; ++tn)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tm = 2 * half + t;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[tn][tm][r] + bv[tn][r];
                    if constexpr (EPI == LCMHIP_EPI_RELU) v[r] = apply_act(v[r], kind, slope);
                    if constexpr (DUAL) {
                        const float g = v[r];
                        v[r] = (acc[tn + TNW / 2][tm][r] + bv2[tn][r]) * (g / (1.f + __expf(-g)));
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
            if (half + 1 < TM / 2) load_resid(half + 1, res_next);
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
                        float4 o = res_cur[ps < RPASS ? ps : 0];
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
#pragma unroll
            for (int ps = 0; ps < RPASS; ++ps) res_cur[ps] = res_next[ps];
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
                    const bool scatter = p.out_rows != nullptr && (p.scatter_mask == 0 || ((p.scatter_mask >> p.stack_sel) & 1));
                    size_t oidx;
                    if (!scatter) oidx = (size_t)m * p.ldo + n;
                    else if (p.out_head_cap > 0) oidx = kv_cache_offset(p.out_rows[m], n, p.out_head_cap, p.N);
                    else oidx = (size_t)p.out_rows[m] * p.ldo + n;
                    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                    if constexpr (EPI == LCMHIP_EPI_STORE) {
                        if ((p.rope_mask >> p.stack_sel) & 1) {
                            // the projection's output is rounded to the operand dtype first (as a separate rotation pass
                            // would read it), rotated in fp32, and rounded again by the store
                            const int pos = p.rope_pos[m], pair0 = (n & 127) >> 1;      // n is a multiple of 8
                            const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cos + pos * 64 + pair0);
                            const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sin + pos * 64 + pair0);
                            const float cs[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                v[2 * j] = bf16_to_f32(f32_to_bf16(v[2 * j]));
                                v[2 * j + 1] = bf16_to_f32(f32_to_bf16(v[2 * j + 1]));
                                rope_rotate(v[2 * j], v[2 * j + 1], cs[j], sn[j]);
                            }
                        }
                    }
                    uint4 o;
                    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                    o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
                    o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
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
    const int bx = big_select_matrix(p, blockIdx.x, p.N / kBN), bz = blockIdx.z;
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

    const int m_wave0 = m0 + wm * 64, n_wave0 = DUAL ? bx * (kBN / 2) + wn * 32 : bx * kBN + wn * 64;
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
    float4 res0[big_resid_rounds<EPI, big_wave_cols<EPI, 4>()>()];
    big_load_resid<EPI, big_wave_cols<EPI, 4>()>(p, m_wave0, n_wave0, 0, lane, res0);
    big_epilogue<bf16_t, EPI, 4>(p, acc, smem, wid, m_wave0, n_wave0, blockIdx.y, lane, res0);
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves (2 along the activation rows x 4 along the weight rows; a wave owns 128 x 64 outputs =
// 8 x 4 accumulator tiles), K step 64, one workgroup per CU, for problems with at least about one tile per CU.
// The 128 KB of LDS hold two K tiles, each as four 16-KB half-tiles (activation rows 0-127 / 128-255, weight
// rows 0-127 / 128-255; same swizzled lane-linear image as above).  A K tile is consumed in four phases of 16
// matrix-core instructions: phase i multiplies the wave's activation row tiles 2 i, 2 i + 1 by all four of its
// weight row tiles.  The weight fragments of a K tile stay in registers for its four phases.  A phase issues half of
// its products, then the LDS reads of the NEXT phase's activation fragments and the DMA of one half-tile, then the
// other half of its products (the reads have eight products to land behind):
//   phase 0: read X(1)              | DMA X-half 1 of tile t+1
//   phase 1: barrier A, read X(2)   | DMA W-half 0 of tile t+2 (into the buffer being consumed)
//   phase 2: read X(3)              | DMA W-half 1 of tile t+2
//   phase 3: wait: at most those two W half-tiles pending and no LDS read pending; barrier B;
//            DMA X-half 0 of tile t+2; read X(0) and the weight fragments of tile t+1 (weight tiles 2-3 in place
//            behind their last products, tiles 0-1 through spare registers).
// Hazards: the weight halves of a buffer are refilled only after barrier A, which a wave reaches with its weight
// fragments of that K tile consumed by phase 0's products; the activation halves only after barrier B, which every
// wave passes with no LDS read pending; a K tile is read only after the counted wait of phase 3 plus barrier B.
// The DMA of a half-tile is in flight for three to six phases, across the barriers.
// Measured on the SwiGLU projection 4096 x (2 x 5632) x 2048: 1.07 PFLOP/s against 0.91 for the 128 x 128 kernel;
// variants that read all fragments in two phases (0.96), that keep four barriers per K tile (1.00), an XCD-banded
// versus a plain tile order (equal) and DMA issue staggered between the two waves of a SIMD (+1 %) were measured too.
// ------------------------------------------------------------------------------------------------
constexpr int kHT = 128 * kRowB;          // one half-tile: 128 rows x 128 bytes
constexpr int kBig256Lds = 2 * 4 * kHT;   // 128 KB

template <int EPI>
__global__ __launch_bounds__(512) void k_gemm_big256(GemmArgs p) {
    constexpr bool DUAL = (EPI == LCMHIP_EPI_SWIGLU);
    constexpr int WT = 4;                     // weight-row tiles of 16 per wave
    constexpr int BNW = 64 * WT;              // weight rows of the tile
    extern __shared__ __attribute__((aligned(1024))) char smem256[];
    char* const smem = smem256;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3;
    // Tile order.  Workgroup ids go round-robin over the 8 XCDs (each with its own L2), so XCD x runs ids x, x + 8, ...
    // in order.  Each XCD gets one contiguous piece of a band-major tile order (bands of 2 activation-row tiles, weight
    // tiles inside): the 32 tiles it runs at a time share 2 activation tiles and 16 weight tiles through its L2.
    int bx, by;
    {
        const int nx = p.tiles_x, ny = p.tiles_y, T = nx * ny;
        const int id = blockIdx.x, xcd = id & 7, q = T >> 3, r = T & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        const int band = L / (2 * nx), rem = L - band * 2 * nx;
        const int rows = min(2, ny - 2 * band);
        bx = big_select_matrix(p, rem / rows, p.N / BNW);
        by = 2 * band + rem % rows;
    }
    const int m0 = by * 256;
    const uint32_t kbytes = (uint32_t)p.K * 2;
    const int nk = p.K / 64;

    // staging: a half-tile takes two DMA instructions per lane; in round j wave w fills rows 8 (8 j + w) .. + 7,
    // lane -> (row lane / 8, slot lane % 8); slot s of row r holds K piece s ^ ((r / 2) % 8), and for these rows
    // (r / 2) % 8 = 4 (w % 2) + (lane / 8) / 2 in both rounds.
    const int srow = lane >> 3;
    const uint32_t piece_off = (uint32_t)(((lane & 7) ^ ((4 * (wid & 1) + (srow >> 1)) & 7)) * 16);
    // activation rows: [half][round] byte offsets from p.A (rows past the end are clamped, so they are not a uniform
    // step apart); weight rows: one offset for (half 0, round 0), the others a uniform step further
    uint32_t offX[2][2], offW0;
    bool w2 = false;           // SwiGLU: this lane stages rows of the second matrix (the same in every half and round)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (8 * j + wid) * 8 + srow;      // row inside the half-tile
            int m = m0 + 128 * h + r;
            m = m < p.M ? m : p.M - 1;                   // rows past the end repeat the last row; never stored
            offX[h][j] = (uint32_t)m * (uint32_t)p.lda * 2u + piece_off;
        }
    uint32_t w_step_j, w_step_h;                         // byte steps to round 1 / to half 1 (wave-uniform)
    {
        const int r = wid * 8 + srow;                    // weight row inside the tile: t = 128 h + 64 j + r
        if constexpr (DUAL) {
            // wave column t / 64 owns 32 gate rows followed by 32 inner rows of the same 32 outputs
            offW0 = (uint32_t)(bx * 128 + (r & 31)) * kbytes + piece_off;
            w2 = r >= 32;
            w_step_j = 32u * kbytes;
            w_step_h = 64u * kbytes;
        } else {
            offW0 = (uint32_t)(bx * BNW + r) * kbytes + piece_off;
            w_step_j = 64u * kbytes;
            w_step_h = 128u * kbytes;
        }
    }
    const char* const Ab = reinterpret_cast<const char*>(p.A);
    const char* const Wsel = reinterpret_cast<const char*>((DUAL && w2) ? p.W2 : p.W);
    auto stage_x = [&](int h, int kt, int buf) {
        char* dst = smem + buf * (4 * kHT) + h * kHT + wid * 1024;
        const uint32_t koff = (uint32_t)kt * kRowB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ab + offX[h][j] + koff), (lptr_t)(dst + j * 8192), 16, 0, 0);
    };
    auto stage_w = [&](int h, int kt, int buf) {
        char* dst = smem + buf * (4 * kHT) + (2 + h) * kHT + wid * 1024;
        const uint32_t koff = (uint32_t)kt * kRowB + (uint32_t)h * w_step_h;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(Wsel + offW0 + (koff + (uint32_t)j * w_step_j)), (lptr_t)(dst + j * 8192), 16, 0, 0);
    };

    f32x4 acc[WT][8];
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses inside a K tile's 64 KB: row = base + lane % 16, K piece of k-step ks = 4 ks + lane / 16
    const int frow = lane & 15;
    const int fsw = (frow >> 1) & 7;
    const int x_off = wr * kHT + frow * kRowB;                               // + 16 tm rows
    const int w_off = (2 + (wc >> 1)) * kHT + ((wc & 1) * 64 + frow) * kRowB;   // + 16 tn rows
    const int slot0 = (((lane >> 4)) ^ fsw) * 16, slot1 = ((4 + (lane >> 4)) ^ fsw) * 16;

    // prologue: K tile 0 whole; of K tile 1 the weight halves and activation half 0 (nk >= 4: K is a multiple of 256)
    stage_w(0, 0, 0); stage_w(1, 0, 0); stage_x(0, 0, 0); stage_x(1, 0, 0);
    stage_w(0, 1, 1); stage_w(1, 1, 1); stage_x(0, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragments: w[ks][tn] = the wave's 64 weight rows; x?[ks][b] = activation row tiles 2 i + b of phase i
#define LCM_RD_W2(BASE, TN0)                                                                               \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
        w[0][TN0 + a] = *reinterpret_cast<const bf16x8*>((BASE) + w_off + (TN0 + a) * 16 * kRowB + slot0); \
        w[1][TN0 + a] = *reinterpret_cast<const bf16x8*>((BASE) + w_off + (TN0 + a) * 16 * kRowB + slot1); \
    }
#define LCM_RD_X2(DST, BASE, I)                                                                            \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                        \
        DST[0][b] = *reinterpret_cast<const bf16x8*>((BASE) + x_off + (2 * (I) + b) * 16 * kRowB + slot0); \
        DST[1][b] = *reinterpret_cast<const bf16x8*>((BASE) + x_off + (2 * (I) + b) * 16 * kRowB + slot1); \
    }
    // 8 products: weight tiles TN0, TN0 + 1 x the phase's two activation tiles x both k-steps
#define LCM_MMA8(TN0, XF, I)                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                       \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                      \
            _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                  \
                acc[TN0 + a][2 * (I) + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks][TN0 + a], XF[ks][b], acc[TN0 + a][2 * (I) + b], 0, 0, 0);

    bf16x8 w[2][WT], xe[2][2], xo[2][2];
    LCM_RD_X2(xe, smem, 0)
    LCM_RD_W2(smem, 0)
    LCM_RD_W2(smem, 2)
    // one K tile; MORE1 / MORE2: K tiles kt + 1 / kt + 2 exist (compile-time: the loop body has no branches).
    // hipcc waits for EVERY outstanding LDS read whenever a product needs one, so each phase first issues the half
    // of its products whose operands were read a phase ago, and only then the reads for the next phase.
#define LCM_PIN() __builtin_amdgcn_sched_barrier(0)
    auto tile = [&](int kt, auto more1_t, auto more2_t) {
        constexpr bool more1 = decltype(more1_t)::value, more2 = decltype(more2_t)::value;
        const int buf = kt & 1;
        const char* base = smem + buf * (4 * kHT);
        const char* nbase = smem + (buf ^ 1) * (4 * kHT);
        {
            // ---- phase 0
            __builtin_amdgcn_s_setprio(1);
            LCM_MMA8(0, xe, 0)
            __builtin_amdgcn_s_setprio(0);
            LCM_PIN();
            LCM_RD_X2(xo, base, 1)
            if constexpr (more1) stage_x(1, kt + 1, buf ^ 1);
            LCM_PIN();
            __builtin_amdgcn_s_setprio(1);
            LCM_MMA8(2, xe, 0)
            __builtin_amdgcn_s_setprio(0);
            // ---- phase 1: every wave has consumed its weight fragments of this K tile -> barrier A frees the weight halves
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            LCM_MMA8(0, xo, 1)
            __builtin_amdgcn_s_setprio(0);
            LCM_PIN();
            LCM_RD_X2(xe, base, 2)
            if constexpr (more2) stage_w(0, kt + 2, buf);
            LCM_PIN();
            __builtin_amdgcn_s_setprio(1);
            LCM_MMA8(2, xo, 1)
            __builtin_amdgcn_s_setprio(0);
            // ---- phase 2
            __builtin_amdgcn_s_setprio(1);
            LCM_MMA8(0, xe, 2)
            __builtin_amdgcn_s_setprio(0);
            LCM_PIN();
            LCM_RD_X2(xo, base, 3)
            if constexpr (more2) stage_w(1, kt + 2, buf);
            LCM_PIN();
            __builtin_amdgcn_s_setprio(1);
            LCM_MMA8(2, xe, 2)
            __builtin_amdgcn_s_setprio(0);
            // ---- phase 3: K tile kt + 1 has landed (at most the two weight half-tiles of kt + 2 pending), all reads of
            // this buffer's activation halves are complete -> barrier B; then the next tile's fragments: weight tiles
            // 2-3 in place behind their last products, weight tiles 0-1 into spare registers
            if constexpr (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (more2) stage_x(0, kt + 2, buf);
            __builtin_amdgcn_s_setprio(1);
            LCM_MMA8(2, xo, 3)
            __builtin_amdgcn_s_setprio(0);
            LCM_PIN();
            bf16x8 wn[2][2];
            if constexpr (more1) {
                LCM_RD_X2(xe, nbase, 0)
                LCM_RD_W2(nbase, 2)
    #pragma unroll
                for (int a = 0; a < 2; ++a) {
                    wn[0][a] = *reinterpret_cast<const bf16x8*>(nbase + w_off + a * 16 * kRowB + slot0);
                    wn[1][a] = *reinterpret_cast<const bf16x8*>(nbase + w_off + a * 16 * kRowB + slot1);
                }
            }
            LCM_PIN();
            __builtin_amdgcn_s_setprio(1);
            LCM_MMA8(0, xo, 3)
            __builtin_amdgcn_s_setprio(0);
            if constexpr (more1) {
    #pragma unroll
                for (int a = 0; a < 2; ++a) { w[0][a] = wn[0][a]; w[1][a] = wn[1][a]; }
            }
        }
    };
    using yes = std::true_type;
    using no = std::false_type;
    for (int kt = 0; kt < nk - 2; ++kt) tile(kt, yes{}, yes{});
    tile(nk - 2, yes{}, no{});
    tile(nk - 1, no{}, no{});
#undef LCM_PIN
#undef LCM_RD_W2
#undef LCM_RD_X2
#undef LCM_MMA8

    constexpr int NC = big_wave_cols<EPI, 4>();
    const int n_wave0 = DUAL ? bx * 128 + wc * 32 : bx * BNW + wc * 16 * WT;
    float4 res0[big_resid_rounds<EPI, NC>()];
    big_load_resid<EPI, NC>(p, m0 + wr * 128, n_wave0, 0, lane, res0);
    big_epilogue<bf16_t, EPI, 8, 4>(p, acc, smem, wid, m0 + wr * 128, n_wave0, 0, lane, res0);
}

// ------------------------------------------------------------------------------------------------
// 256 (activation rows) x 128 (weight rows) tile, 8 waves as 4 (activation rows) x 2 (weight rows): a wave owns 64 x 64
// outputs = 4 x 4 accumulator tiles, so a K tile costs it 8 + 8 fragment reads for 32 products (a 128 x 32 wave tile
// would need 16 + 4).  Staging, swizzle, buffers, barriers and hazards are those of k_gemm_big256:
// two K tiles of [activation half 0][activation half 1][weight half] in LDS, four phases of 8 products (activation tile
// i of the wave x its 4 weight tiles x both k-steps), barrier A before phase 1 frees the weight half, the counted wait
// + barrier B in phase 3 publish K tile kt + 1 and free the activation halves.  The weight fragments of tile kt + 1
// are read during phases 1-2 of tile kt (its weight half is published at barrier A by a counted vmcnt), its first
// activation tile in phase 3 behind barrier B.  Measured on 4096 x 2048 x 2048 / x 5632 against the 128 x 128 kernel
// (45-46 / 95 us on the slower device class): compiler-issued fragment reads 42.4 / 88.9 us, hand-issued reads with
// counted waits 40.8-42.3 / 84.6-84.9 us; a ring of three K tiles in LDS with the DMA two tiles ahead 42.1-43.4 /
// 87.3-87.5 us (slower: the loop is not waiting for memory); the 2 x 4 wave layout (wave tile 128 x 32) equal to this one
// with compiler-issued reads.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512) void k_gemm_big256x128(GemmArgs p) {
    static_assert(EPI != LCMHIP_EPI_SWIGLU, "linear epilogues only");
    extern __shared__ __attribute__((aligned(1024))) char smem256[];
    char* const smem = smem256;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    int bx, by;
    {
        const int nx = p.tiles_x, ny = p.tiles_y, T = nx * ny;
        const int id = blockIdx.x, xcd = id & 7, q = T >> 3, r = T & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        const int band = L / (2 * nx), rem = L - band * 2 * nx;
        const int rows = min(2, ny - 2 * band);
        bx = big_select_matrix(p, rem / rows, p.N / 128);
        by = 2 * band + rem % rows;
    }
    const int m0 = by * 256;
    const uint32_t kbytes = (uint32_t)p.K * 2;
    const int nk = p.K / 64;

    const int srow = lane >> 3;
    const uint32_t piece_off = (uint32_t)(((lane & 7) ^ ((4 * (wid & 1) + (srow >> 1)) & 7)) * 16);
    uint32_t offX[2][2], offW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (8 * j + wid) * 8 + srow;      // row inside a half-tile
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0 + 128 * h + r;
            m = m < p.M ? m : p.M - 1;               // rows past the end repeat the last row; never stored
            offX[h][j] = (uint32_t)m * (uint32_t)p.lda * 2u + piece_off;
        }
        offW[j] = (uint32_t)(bx * 128 + r) * kbytes + piece_off;
    }
    const char* const Ab = reinterpret_cast<const char*>(p.A);
    const char* const Wb = reinterpret_cast<const char*>(p.W);
    auto stage_x = [&](int h, int kt, int buf) {
        char* dst = smem + buf * (4 * kHT) + h * kHT + wid * 1024;
        const uint32_t koff = (uint32_t)kt * kRowB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ab + offX[h][j] + koff), (lptr_t)(dst + j * 8192), 16, 0, 0);
    };
    auto stage_w = [&](int kt, int buf) {
        char* dst = smem + buf * (4 * kHT) + 2 * kHT + wid * 1024;
        const uint32_t koff = (uint32_t)kt * kRowB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(Wb + offW[j] + koff), (lptr_t)(dst + j * 8192), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fsw = (frow >> 1) & 7;
    const int x_off = (wr >> 1) * kHT + ((wr & 1) * 64 + frow) * kRowB;     // + 16 tm rows
    const int w_off = 2 * kHT + (wc * 64 + frow) * kRowB;                   // + 16 tn rows
    const int slot0 = (((lane >> 4)) ^ fsw) * 16, slot1 = ((4 + (lane >> 4)) ^ fsw) * 16;

#ifdef LCM_DMA_STAMPS
    // timeline probe (see k_gemm_dma): start, first K tile landed, K loop done, epilogue done, per workgroup
    uint64_t* const stamps = p.partial != nullptr ? reinterpret_cast<uint64_t*>(p.partial) + 4 * (size_t)blockIdx.x : nullptr;
    if (stamps && tid == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();
#endif
    // prologue: K tile 0 whole; of K tile 1 the weight half and activation half 0 (nk >= 4)
    stage_w(0, 0); stage_x(0, 0, 0); stage_x(1, 0, 0);
    stage_w(1, 1); stage_x(0, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef LCM_DMA_STAMPS
    if (stamps && tid == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
#endif

    // Fragment reads are issued by hand (ds_read_b128 through inline asm) and retired with COUNTED lgkmcnt waits: the
    // compiler's own waits drain every pending LDS read before a product, which leaves a read only the few products
    // issued behind it to land.  Reads retire in order, so "at most n younger reads pending" retires a fragment.  The
    // compiler does not know that an asm result is still in flight: every wait is followed by a scheduling fence so that
    // no product moves above it, nothing copies a fragment (the weight fragments of consecutive K tiles alternate
    // between two register sets instead), and the last tile ends with lgkmcnt(0).
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t xa[2] = {lds0 + (uint32_t)(x_off + slot0), lds0 + (uint32_t)(x_off + slot1)};
    const uint32_t wa[2] = {lds0 + (uint32_t)(w_off + slot0), lds0 + (uint32_t)(w_off + slot1)};
#define LCM_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF))
#define LCM_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory")
#define LCM_PIN() __builtin_amdgcn_sched_barrier(0)
    // activation tile TM of K-tile buffer BUF (compile-time): both k-steps
#define LCM_RD_X1(DST, BUF, TM)                                      \
    {                                                                \
        LCM_DSR(DST[0], xb[BUF][0], (TM) * 16 * kRowB);              \
        LCM_DSR(DST[1], xb[BUF][1], (TM) * 16 * kRowB);              \
    }
    // weight tiles TN0, TN0 + 1 of K-tile buffer BUF: both k-steps
#define LCM_RD_W2(DST, BUF, TN0)                                             \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                          \
        LCM_DSR(DST[0][TN0 + a], wb[BUF][0], ((TN0) + a) * 16 * kRowB);      \
        LCM_DSR(DST[1][TN0 + a], wb[BUF][1], ((TN0) + a) * 16 * kRowB);      \
    }
    // 8 products: the wave's 4 weight tiles x activation tile TM x both k-steps
#define LCM_MMA8W(WF, XF, TM)                                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                           \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                          \
            acc[a][TM] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[ks][a], XF[ks], acc[a][TM], 0, 0, 0);      \
    __builtin_amdgcn_s_setprio(0);

    // the two K-tile buffers are 64 KB apart: beyond the 16-bit offset field, so each has its own base registers
    const uint32_t xb[2][2] = {{xa[0], xa[1]}, {xa[0] + 4u * kHT, xa[1] + 4u * kHT}};
    const uint32_t wb[2][2] = {{wa[0], wa[1]}, {wa[0] + 4u * kHT, wa[1] + 4u * kHT}};

    bf16x8 wA[2][4], wB[2][4], xe[2], xo[2];
    LCM_RD_X1(xe, 0, 0)
    LCM_RD_W2(wA, 0, 0)
    LCM_RD_W2(wA, 0, 2)
    // One K tile in buffer BUF; WC = its weight fragments (read during the previous tile), WN = the register set that
    // receives the next tile's.  LDS reads per phase: 2 / 6 / 6 / 2.
    //   phase 0: read X1                     | DMA X-half 1 of tile kt+1         | retire X0 (+ all of WC) | products on X0
    //   phase 1: wait: W(kt+1) landed; barrier A (frees this buffer's weight half, publishes W(kt+1))
    //            read X2, WN tiles 0-1       | DMA W of tile kt+2 (this buffer)  | retire X1               | products on X1
    //   phase 2: read X3, WN tiles 2-3                                            | retire X2               | products on X2
    //   phase 3: wait: X(kt+1) landed, no LDS read pending; barrier B (frees this buffer's activation halves)
    //            read X0 of tile kt+1        | DMA X-half 0 of tile kt+2                                    | products on X3
#define LCM_TILE(BUF, KT, WC, WN, MORE1, MORE2)                                                        \
    {                                                                                                  \
        LCM_RD_X1(xo, BUF, 1)                                                                          \
        if (MORE1) stage_x(1, (KT) + 1, (BUF) ^ 1);                                                    \
        LCM_LGKM(2);                                                                                   \
        LCM_PIN();                                                                                     \
        LCM_MMA8W(WC, xe, 0)                                                                           \
        LCM_PIN();                                                                                     \
        if (MORE1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                    \
        __builtin_amdgcn_s_barrier();                                                                  \
        LCM_RD_X1(xe, BUF, 2)                                                                          \
        if (MORE1) { LCM_RD_W2(WN, (BUF) ^ 1, 0) }                                                     \
        if (MORE2) stage_w((KT) + 2, BUF);                                                             \
        if (MORE1) { LCM_LGKM(6); } else { LCM_LGKM(2); }                                              \
        LCM_PIN();                                                                                     \
        LCM_MMA8W(WC, xo, 1)                                                                           \
        LCM_PIN();                                                                                     \
        LCM_RD_X1(xo, BUF, 3)                                                                          \
        if (MORE1) { LCM_RD_W2(WN, (BUF) ^ 1, 2) }                                                     \
        if (MORE1) { LCM_LGKM(10); } else { LCM_LGKM(2); }                                             \
        LCM_PIN();                                                                                     \
        LCM_MMA8W(WC, xe, 2)                                                                           \
        LCM_PIN();                                                                                     \
        if (MORE2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                    \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
        LCM_LGKM(0);                                                                                   \
        __builtin_amdgcn_s_barrier();                                                                  \
        if (MORE2) stage_x(0, (KT) + 2, BUF);                                                          \
        if (MORE1) { LCM_RD_X1(xe, (BUF) ^ 1, 0) }                                                     \
        LCM_PIN();                                                                                     \
        LCM_MMA8W(WC, xo, 3)                                                                           \
        LCM_PIN();                                                                                     \
    }
    for (int kt = 0; kt < nk - 2; kt += 2) {
        LCM_TILE(0, kt, wA, wB, true, true)
        LCM_TILE(1, kt + 1, wB, wA, true, true)
    }
    LCM_TILE(0, nk - 2, wA, wB, true, false)
    LCM_TILE(1, nk - 1, wB, wA, false, false)
#undef LCM_TILE
#undef LCM_MMA8W
#undef LCM_RD_W2
#undef LCM_RD_X1
#undef LCM_PIN
#undef LCM_LGKM
#undef LCM_DSR

    constexpr int NC = big_wave_cols<EPI, 4>();
    const int m_wave0 = m0 + wr * 64, n_wave0 = bx * 128 + wc * 64;
    float4 res0[big_resid_rounds<EPI, NC>()];
    big_load_resid<EPI, NC>(p, m_wave0, n_wave0, 0, lane, res0);
#ifdef LCM_DMA_STAMPS
    if (stamps && tid == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
#endif
    big_epilogue<bf16_t, EPI, 4>(p, acc, smem, wid, m_wave0, n_wave0, 0, lane, res0);
#ifdef LCM_DMA_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    if (stamps && tid == 0) stamps[3] = __builtin_amdgcn_s_memrealtime();
#endif
}

// K splits for the split-K (slab) form: two workgroups per CU (512; with one per CU a workgroup's four waves have
// nothing to overlap their barriers with: at 1024 rows ffn_out 46.8 -> 31.6 us, self_ov 22.4 -> 18.3 us, end to end
// +8 % at 512 documents; 768 and 1024 measured slower again), at least 8 K tiles per split (4: slower), and the splits
// divide the K tiles.
int gemm_big_splits(const GemmArgs& a, int max_splits) {
    const int tiles = (a.N / kBN) * ((a.M + kBM - 1) / kBM);   // counted in 128-row tiles
    const int nk = a.K / 64;
    constexpr int target = 512, min_kt = 8;
    int s = 1;
    while (tiles * s < target && nk % (s * 2) == 0 && nk / (s * 2) >= min_kt && s * 2 <= (max_splits > 0 ? max_splits : 64)) s *= 2;
    return s;
}

// The 128 x 128 kernel pays once its workgroups (tiles x K splits when a slab is available) cover most of the
// chip and there are at least 384 rows; below that the weight-streaming LDS-DMA tiles of gemm.hip are faster (at 256
// rows the denoiser's ffn_out takes 16.5 against 19.3 us, the sampling loop at 128 documents runs 3.7 % faster; at
// 384 rows the 128 x 128 kernel wins by 7 %).
bool gemm_big_applicable(const GemmArgs& a, bool may_split) {
    constexpr int min_rows = 384;
    const int wrows = (a.epilogue == LCMHIP_EPI_SWIGLU ? 2 * a.N : a.N) * std::max(1, a.stack);
    if (a.stack > 1 && a.N % kBN != 0) return false;
    if (!(a.dtype == LCMHIP_BF16 && a.M >= min_rows && a.K % 64 == 0 && wrows % kBN == 0 && a.lda % 8 == 0)) return false;
    const int tiles = (wrows / kBN) * ((a.M + kBM - 1) / kBM);
    const int s = may_split ? gemm_big_splits(a, a.max_splits) : 1;
    // a product that needs a K split to fill the chip must reach 320 workgroups (a d x d product at 512 rows has 256:
    // 12.8 us on the LDS-DMA tile against 15.6 us here), an unsplit one 192
    return tiles * s >= (s > 1 ? 320 : 192);
}

// epi: LCMHIP_EPI_* or EPI_PARTIAL; splits >= 1 (only with EPI_PARTIAL); m_rows: rows the row tiles must cover
// The 256 x 256 kernel (one workgroup per CU) against the 128 x 128 kernel (two per CU): whichever needs less time
// for its waves of tiles.  A full wave of 256 x 256 tiles takes about 1.6 x a full wave of 128 x 128 tiles (measured
// on the SwiGLU projection: 50-56 us against 32-34 us; four times the work at 1.15-1.2 x the rate), and a partly
// filled last wave takes as long as a full one: at 1024 rows 176 tiles in one wave take 56 us against 67 us for 704
// tiles of 128 x 128 in two.  The ratio differs between devices of the pool (see below).  No K split, 32-bit operand
// offsets.  LCMHIP_BIG256=0 (read at every launch: tests compare the kernels inside one process) turns the kernel off.
static bool big256_applicable(const GemmArgs& a, int epi, int splits, int m_rows) {
    const char* e = getenv("LCMHIP_BIG256");
    if ((e && atoi(e) == 0) || splits != 1 || epi == EPI_PARTIAL) return false;
    if (a.N % 256 != 0 && a.stack > 1) return false;
    const int wrows = (epi == LCMHIP_EPI_SWIGLU ? 2 * a.N : a.N) * std::max(1, a.stack);
    if (wrows % 256 != 0 || a.K % 64 != 0 || a.K < 128) return false;
    if ((size_t)a.M * a.lda * 2 >= (1ull << 31) || (size_t)wrows * a.K * 2 >= (1ull << 31)) return false;
    const int tiles = (wrows / 256) * ((m_rows + 255) / 256);
    const int tiles128 = (wrows / kBN) * ((m_rows + kBM - 1) / kBM);
    const int waves256 = (tiles + 255) / 256, waves128 = (tiles128 + 511) / 512;
    // wave-time ratio 1.6 (the stacked q / k / v product of 4096 rows, 384 tiles in 2 waves against 1536 in 3, measured
    // 2 % faster at 1.3 on a slower device of the pool: too little to move the routing the batch sweep was tuned with)
    return 16 * waves256 < 10 * waves128;
}

// The 256 x 128 tile for linear products without a K split that the 256 x 256 kernel does not take: it replaces the
// 128 x 128 kernel there (the same wave quantisation: one 256 x 128 tile per CU is two 128 x 128 tiles per CU).  In-process
// A/B against the 128 x 128 kernel at 4096 rows, slower / faster device class of the pool: 2048 x 2048 store +9..13 % /
// +1 %, 2048 x 5632 store +12 % / +13 %, residual update 2048 x 2048 +9..10 % / +16 %, 2048 x 5632 +12 % / +16 %,
// stacked q / k / v (6144 x 2048) +5 % / +8 %; 2048^3 +6 % / +7 %.  Results are bit-identical (same K order per output).
// LCMHIP_BIG256_N128 = 0 turns it off (read at every launch: the A/B script switches it inside one process).
static bool big256n128_applicable(const GemmArgs& a, int epi, int splits, int m_rows) {
    const char* e = getenv("LCMHIP_BIG256_N128");
    const int mode = e ? atoi(e) : 1;
    if (mode == 0 || splits != 1 || epi == EPI_PARTIAL || epi == LCMHIP_EPI_SWIGLU) return false;
    if (a.N % 128 != 0 || a.K % 128 != 0 || a.K < 256) return false;   // the K loop runs in pairs of 64-wide tiles
    if ((size_t)a.M * a.lda * 2 >= (1ull << 31) || (size_t)a.N * a.K * 2 >= (1ull << 31)) return false;   // 32-bit offsets
    (void)m_rows;
    return true;
}

// wt = 4: the 256 x 256 tile kernel; wt = 2: the 256 x 128 tile kernel (linear epilogues only)
static int launch_gemm_big256(const GemmArgs& a_in, int epi, int m_rows, int wt, hipStream_t st) {
    const int wrows = (epi == LCMHIP_EPI_SWIGLU ? 2 * a_in.N : a_in.N) * std::max(1, a_in.stack);
    GemmArgs a = a_in;
    a.tiles_x = wrows / (64 * wt);
    a.tiles_y = (m_rows + 255) / 256;
    dim3 grid(a.tiles_x * a.tiles_y), block(512);
#define LCM_BIG256(K, E)                                                                                           \
    {                                                                                                              \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&K<E>),                   \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, kBig256Lds); \
        LCM_HIP_CHECK(attr);                                                                                       \
        hipLaunchKernelGGL((K<E>), grid, block, kBig256Lds, st, a);                                                \
    }
    if (wt == 4) {
        switch (epi) {
            case LCMHIP_EPI_STORE: LCM_BIG256(k_gemm_big256, LCMHIP_EPI_STORE); break;
            case LCMHIP_EPI_STORE_F32: LCM_BIG256(k_gemm_big256, LCMHIP_EPI_STORE_F32); break;
            case LCMHIP_EPI_RELU: LCM_BIG256(k_gemm_big256, LCMHIP_EPI_RELU); break;
            case LCMHIP_EPI_SWIGLU: LCM_BIG256(k_gemm_big256, LCMHIP_EPI_SWIGLU); break;
            case LCMHIP_EPI_RESID: LCM_BIG256(k_gemm_big256, LCMHIP_EPI_RESID); break;
            default: return fail(LCMHIP_ERR_INVALID, "gemm: unknown epilogue");
        }
    } else {
        switch (epi) {
            case LCMHIP_EPI_STORE: LCM_BIG256(k_gemm_big256x128, LCMHIP_EPI_STORE); break;
            case LCMHIP_EPI_STORE_F32: LCM_BIG256(k_gemm_big256x128, LCMHIP_EPI_STORE_F32); break;
            case LCMHIP_EPI_RELU: LCM_BIG256(k_gemm_big256x128, LCMHIP_EPI_RELU); break;
            case LCMHIP_EPI_RESID: LCM_BIG256(k_gemm_big256x128, LCMHIP_EPI_RESID); break;
            default: return fail(LCMHIP_ERR_INVALID, "gemm: the 256 x 128 tile takes linear epilogues only");
        }
    }
#undef LCM_BIG256
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

// Time of an unsplit many-row launch in units of one full wave of 128 x 128 tiles (512 workgroups), from the wave
// quantisation and the measured wave-time ratio of the routing above.  For callers that can cut one product into two
// launches (the stacked q / k / v product: q on its own + k, v stacked).
double gemm_big_wave_cost(const GemmArgs& a, int epi, int m_rows) {
    const int wrows = (epi == LCMHIP_EPI_SWIGLU ? 2 * a.N : a.N) * std::max(1, a.stack);
    if (big256_applicable(a, epi, 1, m_rows)) return 1.6 * (((wrows / 256) * ((m_rows + 255) / 256) + 255) / 256);
    return (double)(((wrows / kBN) * ((m_rows + kBM - 1) / kBM) + 511) / 512);
}

int launch_gemm_big(const GemmArgs& a, int epi, int splits, int m_rows, hipStream_t st) {
    if (a.stack > 1)
        LCM_REQUIRE(a.stack <= 3 && splits == 1 && a.N % kBN == 0 && (epi == LCMHIP_EPI_STORE || epi == LCMHIP_EPI_STORE_F32 || epi == LCMHIP_EPI_RELU),
                    "gemm: stacked products take up to 3 matrices, a linear epilogue and no K split");
    if (big256_applicable(a, epi, splits, m_rows)) return launch_gemm_big256(a, epi, m_rows, 4, st);
    if (big256n128_applicable(a, epi, splits, m_rows)) return launch_gemm_big256(a, epi, m_rows, 2, st);
    const int wrows = (epi == LCMHIP_EPI_SWIGLU ? 2 * a.N : a.N) * std::max(1, a.stack);
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
