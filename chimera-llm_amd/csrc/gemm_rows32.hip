// Unsplit products of the sampling loop on 32 x 32 output tiles (bf16, at most a few hundred rows).
//
// The split-K products of the loop leave fp32 slabs that a norm kernel of its own has to sum, add to the residual
// stream and normalize before the next product can start: two dependent launches per product, and a launch boundary
// costs as much as the work of either.  This kernel removes the norm launch on both sides of a product:
//   * operand side (NORM): the activation operand is read from the fp32 residual stream and RMS-normalized + modulated
//     on the way into LDS -- y = ((x * rstd) * w) * (1 + scale) + shift, rounded to bf16 exactly where the norm kernel
//     rounds it -- with rstd from per-row partial sums of squares that the PREVIOUS product left behind;
//   * result side (RESID): the tile is complete (no K split), so the epilogue applies the gated residual update,
//     resid = resid_in + gate * (acc + bias) -- into ANOTHER buffer when the operand side reads the stream, whose rows
//     other workgroups are still reading -- and leaves the partial sums of squares of the UPDATED rows (one per row and
//     16 columns, a fixed summation order: deterministic) for the next product's operand side.
// Without a K split the grid has to come from the tile count: 32 rows x 32 weight rows per workgroup of 4 waves (one
// 16 x 16 matrix-core tile each) gives 256 workgroups at 128 x 2048.  A workgroup then moves 32 weight rows + 32
// operand rows of the whole K: 256-byte K chunks, everything global -> LDS by LDS-DMA (global_load_lds), S stages, one
// raw s_barrier per chunk with counted vmcnt waits; the fp32 rows of a NORM operand are staged as they are and
// converted LDS -> LDS one chunk ahead of the matrix cores.  The bf16 LDS images are lane-linear, bank conflicts of the
// fragment reads are removed on the source side (slot s of tile row r holds K piece s ^ (r % 16)), as in k_gemm_dma.
#include <algorithm>

#include "gemm_epilogue.h"

namespace lcmhip {

constexpr int kR32Chunk = 256;                 // bytes of K per bf16 tile row and stage (128 k)
constexpr int kR32Tile = 32 * kR32Chunk;       // 8 KB: 32 bf16 rows of one chunk
constexpr int kR32TileF = 32 * 2 * kR32Chunk;  // 16 KB: 32 fp32 rows of one chunk
typedef const __attribute__((address_space(1))) void* r32_gptr_t;
typedef __attribute__((address_space(3))) void* r32_lptr_t;
typedef uint32_t r32_u32x2 __attribute__((ext_vector_type(2)));

constexpr int r32_stage_bytes(bool norm) { return norm ? kR32TileF + kR32Tile : 2 * kR32Tile; }
// NORM: stages, two converted operand tiles, 32 rstd values, the per-column operands of the norm (weight bf16, scale and
// shift fp32) for the whole K
inline size_t r32_lds_bytes(bool norm, int stages, int K) {
    return (size_t)stages * r32_stage_bytes(norm) + (norm ? 2 * kR32Tile + 256 + 10 * (size_t)K : 0);
}

// Waits until at most n of this wave's vector-memory instructions are in flight (n uniform, a multiple of 2 up to 24).
__device__ __forceinline__ void r32_wait_vm(int n) {
#define LCM_R32_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        LCM_R32_W(2) LCM_R32_W(4) LCM_R32_W(6) LCM_R32_W(8) LCM_R32_W(10) LCM_R32_W(12) LCM_R32_W(14) LCM_R32_W(16)
        LCM_R32_W(18) LCM_R32_W(20) LCM_R32_W(22) LCM_R32_W(24)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef LCM_R32_W
}

template <bool NORM, bool RESID, int S>
__global__ __launch_bounds__(256) void k_gemm_rows32(Rows32Args p) {
    extern __shared__ __attribute__((aligned(1024))) char smem_r32[];
    char* const smem = smem_r32;
    constexpr int kStage = r32_stage_bytes(NORM);
    constexpr int kWOff = NORM ? kR32TileF : kR32Tile;        // the weight rows inside a stage
    constexpr int kPer = NORM ? 6 : 4;                        // DMA instructions per wave and chunk
    char* const abuf = smem + S * kStage;                     // NORM: the two converted operand tiles
    float* const rstd = reinterpret_cast<float*>(abuf + 2 * kR32Tile);
    char* const opw = reinterpret_cast<char*>(rstd) + 256;
    float* const opsc = reinterpret_cast<float*>(opw + 2 * (size_t)p.K);
    float* const opsh = opsc + p.K;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid & 1, wn = wid >> 1;                    // 2 x 2 waves, one 16 x 16 tile each
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const uint32_t kbytes = (uint32_t)p.K * 2;
    const int nc = p.K / 128;

    // ---- the loads everything else waits for are REQUESTED first (the memory counter retires in order) and used after
    // the first stages have been requested too
    u32x4 pre_w[2];
    f32x4 pre_sc[2][2], pre_sh[2][2];
    f32x4 pre_st[8];
    float pre_tail[3] = {0.f, 0.f, 0.f};
    if constexpr (NORM) {
        const bool mod = p.scale != nullptr;
        const float* sc_src = mod ? p.scale : reinterpret_cast<const float*>(p.norm_w);   // any valid address when absent
        const float* sh_src = mod ? p.shift : reinterpret_cast<const float*>(p.norm_w);
#pragma unroll
        for (int t = 0; t < 2; ++t) {                      // K <= 4096: at most two rounds of 256 x 8 columns
            const int i0 = (t * 256 + tid) * 8;
            const int i = i0 < p.K ? i0 : 0;               // a round past K loads (and later drops) column group 0
            pre_w[t] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.norm_w) + 2 * i);
            const int im = mod ? i : 0;
            pre_sc[t][0] = *reinterpret_cast<const f32x4*>(sc_src + im);
            pre_sc[t][1] = *reinterpret_cast<const f32x4*>(sc_src + im + 4);
            pre_sh[t][0] = *reinterpret_cast<const f32x4*>(sh_src + im);
            pre_sh[t][1] = *reinterpret_cast<const f32x4*>(sh_src + im + 4);
        }
        // partial sums of squares of the tile's 32 rows: 8 lanes per row, each a contiguous eighth of the row's values,
        // all requested at once (at most 32 per lane: 8 vectors, or up to 3 single values when fewer than 4)
        const int r = tid >> 3, q = tid & 7;
        const int per = p.stats_parts >> 3;
        const float* row0 = p.stats_in + (size_t)min(m0 + r, p.M - 1) * p.stats_parts;
        const float* sp = row0 + q * per;
        const int nv = per >> 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) pre_st[j] = *reinterpret_cast<const f32x4*>(j < nv ? sp + 4 * j : row0);   // unused: the row's start
#pragma unroll
        for (int j = 0; j < 3; ++j) pre_tail[j] = sp[4 * nv + j < per ? 4 * nv + j : 0];
    }

    // ---- staging.  bf16 rows (weights; the operand when it is bf16): instruction g fills tile rows 4 g .. 4 g + 3
    // (lane -> row lane / 16, slot lane % 16), wave w issues g = w and w + 4.  fp32 operand rows: instruction i fills
    // rows 2 i, 2 i + 1 (lane -> row lane / 32, 16-byte piece lane % 32), wave w issues i = w, w + 4, w + 8, w + 12.
    const int srow = lane >> 4;
    const uint32_t piece_off = (uint32_t)(((lane & 15) ^ (4 * wid + srow)) * 16);     // tile row % 16 = 4 w + lane / 16
    const char* wsrc[2];
    const char* asrc[NORM ? 4 : 2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 4 * (wid + 4 * j) + srow;
        wsrc[j] = reinterpret_cast<const char*>(p.W) + (size_t)(n0 + r) * kbytes + piece_off;
        if constexpr (!NORM) {
            const int m = min(m0 + r, p.M - 1);          // rows past the end repeat the last row; never stored
            asrc[j] = reinterpret_cast<const char*>(p.A) + (size_t)m * kbytes + piece_off;
        }
    }
    if constexpr (NORM) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = min(m0 + 2 * (wid + 4 * j) + (lane >> 5), p.M - 1);
            asrc[j] = reinterpret_cast<const char*>(p.x32 + (size_t)m * p.K) + (lane & 31) * 16;
        }
    }
    // A workgroup walks K from a chunk of its own and wraps around: the tiles that share operand rows (same row tile, the
    // column tiles x, x + 8, ... of one XCD) or weight rows (same column tile, all row tiles: one XCD too) otherwise ask
    // their L2 for the same lines at the same moment, and every one of them waits out the miss.  Staggered by two / four
    // chunks, one of them takes the miss and the others hit.  (The summation order of an output depends on its tile.)
    const int rot = (int)((2u * (blockIdx.x >> 3) + 4u * blockIdx.y) % (unsigned)nc);
    auto chunk_of = [&](int c) { const int r = c + rot; return r >= nc ? r - nc : r; };
    auto issue = [&](int c) {
        char* base = smem + (c % S) * kStage;
        const size_t kc = (size_t)chunk_of(c);
        if constexpr (NORM) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((r32_gptr_t)(asrc[j] + kc * 512), (r32_lptr_t)(base + (wid + 4 * j) * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds((r32_gptr_t)(asrc[j] + kc * kR32Chunk), (r32_lptr_t)(base + (wid + 4 * j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((r32_gptr_t)(wsrc[j] + kc * kR32Chunk), (r32_lptr_t)(base + kWOff + (wid + 4 * j) * 1024), 16, 0, 0);
    };
    // fp32 rows of chunk c -> the bf16 operand tile (c & 1): thread -> 4 consecutive k (kg) of rows rg, rg + 8, ...
    auto convert = [&](int c) {
        const char* xs = smem + (c % S) * kStage;
        char* ab = abuf + (c & 1) * kR32Tile;
        const int kg = tid & 31, rg = tid >> 5;
        const int k = chunk_of(c) * 128 + 4 * kg;
        // (native vector types on every LDS access of the loop: an aggregate copy hides the address from the
        // compiler's wait-count pass, which then drains all outstanding LDS-DMA before it)
        const r32_u32x2 wp = *reinterpret_cast<const r32_u32x2*>(opw + 2 * k);
        const float pw[4] = {__uint_as_float(wp[0] << 16), __uint_as_float(wp[0] & 0xffff0000u), __uint_as_float(wp[1] << 16),
                             __uint_as_float(wp[1] & 0xffff0000u)};
        const f32x4 sc = *reinterpret_cast<const f32x4*>(opsc + k), sh = *reinterpret_cast<const f32x4*>(opsh + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = rg + 8 * j;
            const f32x4 v = *reinterpret_cast<const f32x4*>(xs + r * 512 + kg * 16);
            const float rs = rstd[r];
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float tt = v[i] * rs * pw[i];          // the arithmetic of the norm kernels, in their order
                o[i] = tt * (1.f + sc[i]) + sh[i];
            }
            r32_u32x2 pk;
            pk[0] = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
            pk[1] = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
            *reinterpret_cast<r32_u32x2*>(ab + r * kR32Chunk + (((kg >> 1) ^ (r & 15)) * 16) + (kg & 1) * 8) = pk;
        }
    };
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, kq = lane >> 4;
    auto mma = [&](int c) {
        const char* base = smem + (c % S) * kStage;
        const char* ax = NORM ? abuf + (c & 1) * kR32Tile : base;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = ((4 * ks + kq) ^ frow) * 16;
            const bf16x8 w = *reinterpret_cast<const bf16x8*>(base + kWOff + (16 * wn + frow) * kR32Chunk + slot);
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(ax + (16 * wm + frow) * kR32Chunk + slot);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);
        }
    };

    const int pre = min(S - 1, nc);
    for (int c = 0; c < pre; ++c) issue(c);
    if constexpr (NORM) {
        const bool mod = p.scale != nullptr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = (t * 256 + tid) * 8;
            if (i < p.K) {
                const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<u32x4*>(opw + 2 * i) = pre_w[t];
                *reinterpret_cast<f32x4*>(opsc + i) = mod ? pre_sc[t][0] : z; *reinterpret_cast<f32x4*>(opsc + i + 4) = mod ? pre_sc[t][1] : z;
                *reinterpret_cast<f32x4*>(opsh + i) = mod ? pre_sh[t][0] : z; *reinterpret_cast<f32x4*>(opsh + i + 4) = mod ? pre_sh[t][1] : z;
            }
        }
        {
            const int per = p.stats_parts >> 3, nv = per >> 2;
            float ss = 0.f;                              // in the order of the row's values
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nv) { ss += pre_st[j][0]; ss += pre_st[j][1]; ss += pre_st[j][2]; ss += pre_st[j][3]; }
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (4 * nv + j < per) ss += pre_tail[j];
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            ss += __shfl_xor(ss, 4, 64);
            if ((tid & 7) == 0) rstd[tid >> 3] = rsqrtf(ss / (float)p.K + p.eps);
        }
        // chunk 0 landed, the norm operands written: convert chunk 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        r32_wait_vm((pre - 1) * kPer);
        __builtin_amdgcn_s_barrier();
        convert(0);
        for (int c = 0; c < nc; ++c) {
            // chunk c + 1 landed (to be converted now); chunk c converted by every wave
            const int issued = min(c + S - 2, nc - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            r32_wait_vm(max(0, issued - (c + 1)) * kPer);
            __builtin_amdgcn_s_barrier();
            if (c + S - 1 < nc) issue(c + S - 1);
            if (c + 1 < nc) convert(c + 1);
            mma(c);
        }
    } else {
        for (int c = 0; c < nc; ++c) {
            const int issued = min(c + S - 2, nc - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            r32_wait_vm((issued - c) * kPer);
            __builtin_amdgcn_s_barrier();
            if (c + S - 1 < nc) issue(c + S - 1);
            mma(c);
        }
    }

    // ---- epilogue: lane l holds output row m = 16 wm + l % 16, columns n .. n + 3 with n = 16 wn + 4 (l / 16)
    const int m = m0 + 16 * wm + frow, n = n0 + 16 * wn + 4 * kq;
    if constexpr (RESID) {
        float ss = 0.f;
        if (m < p.M) {
            const bf16_t* bias = reinterpret_cast<const bf16_t*>(p.bias);
            float g[4] = {1.f, 1.f, 1.f, 1.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.gate != nullptr) {
                const float4 g4 = *reinterpret_cast<const float4*>(p.gate + n);
                g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
            }
            if (bias != nullptr) {
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = Elem<bf16_t>::load(bias + n + i);
            }
            float* dst = p.resid + (size_t)m * p.ldr + n;
            const float4 x4 = *reinterpret_cast<const float4*>((p.resid_in != nullptr ? p.resid_in : p.resid) + (size_t)m * p.ldr + n);
            float x[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[i] += g[i] * (acc[i] + b[i]);
                ss += x[i] * x[i];
            }
            store16_through(dst, st_f32x4{x[0], x[1], x[2], x[3]});    // read next by every XCD: see store16_through
        }
        if (p.stats_out != nullptr) {
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            if (kq == 0 && m < p.M) {
                float* sp = p.stats_out + (size_t)m * (p.N / 16) + (blockIdx.x * 2 + wn);
                asm volatile("global_store_dword %0, %1, off sc1" ::"v"(sp), "v"(ss) : "memory");
            }
        }
    } else {
        if (m < p.M) store16_through(p.out32 + (size_t)m * p.N + n, st_f32x4{acc[0], acc[1], acc[2], acc[3]});
    }
}

bool gemm_rows32_applicable(const Rows32Args& a) {
    if (a.M < 1 || a.K % 128 != 0 || a.N % 32 != 0 || a.W == nullptr) return false;
    const bool norm = a.x32 != nullptr;
    if (norm) return a.K <= 4096 && a.stats_in != nullptr && a.stats_parts >= 8 && a.stats_parts % 8 == 0 && a.stats_parts <= 256 && a.norm_w != nullptr;
    return a.A != nullptr;
}

int launch_gemm_rows32(const Rows32Args& a, hipStream_t st) {
    LCM_REQUIRE(gemm_rows32_applicable(a), "rows32: unsupported shape or missing operand");
    LCM_REQUIRE((a.x32 != nullptr) != (a.A != nullptr), "rows32: exactly one operand source");
    LCM_REQUIRE((a.resid != nullptr) != (a.out32 != nullptr), "rows32: exactly one result");
    LCM_REQUIRE((a.shift == nullptr) == (a.scale == nullptr), "rows32: shift and scale go together");
    LCM_REQUIRE(a.resid == nullptr || (a.ldr >= a.N && a.ldr % 4 == 0), "rows32: bad residual stride");
    // every workgroup reads whole rows of x32 while the others store their tiles of resid
    LCM_REQUIRE(a.x32 == nullptr || a.resid == nullptr || a.x32 != a.resid, "rows32: the normalized operand and the updated rows must be different buffers");
    const bool norm = a.x32 != nullptr, resid = a.resid != nullptr;
    const dim3 grid(a.N / 32, (a.M + 31) / 32), block(256);
#define LCM_R32(NORM, RESID, S)                                                                                        \
    {                                                                                                                  \
        const size_t lds = r32_lds_bytes(NORM, S, a.K);                                                                \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_rows32<NORM, RESID, S>), \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
        LCM_HIP_CHECK(attr);                                                                                           \
        hipLaunchKernelGGL((k_gemm_rows32<NORM, RESID, S>), grid, block, lds, st, a);                                  \
    }
    if (norm) {
        const bool five = r32_lds_bytes(true, 5, a.K) <= 160 * 1024;
        if (resid) { if (five) LCM_R32(true, true, 5) else LCM_R32(true, true, 4) }
        else { if (five) LCM_R32(true, false, 5) else LCM_R32(true, false, 4) }
    } else {
        if (resid) LCM_R32(false, true, 8) else LCM_R32(false, false, 8)
    }
#undef LCM_R32
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

}  // namespace lcmhip
