// C = A * W^T for the row counts of sentence-level generation (M = 1 .. a few hundred rows): the
// weights are streamed once from HBM and the job is to keep every CU's memory queue full.
//
// Decomposition: grid = (weight-row tiles, K splits, row tiles).  With split-K the fp32 partial tiles go to a slab
// [split][row][col]; they are summed by the consumer (the fused residual+norm kernel, the attention kernel, or
// k_splitk_finish) -- no atomics, deterministic.
// MFMA orientation: WEIGHT rows are the matrix-core A operand, ACTIVATION rows the B operand, so
// a lane holds 4 consecutive output columns of one row (vector bias/gate loads, vector stores).
//   bf16 operands: v_mfma_f32_16x16x32_bf16;  f32 operands: v_mfma_f32_16x16x4_f32 (exact fp32).
// Two kernels:
//   k_gemm_dma  bf16, row tiles of 64 (also for fewer rows) or 128 rows: operands staged global -> LDS by LDS-DMA,
//               three stages, see its own comment;
//   k_gemm      fp32 (and bf16 when LCMHIP_GEMM_DMA=0): a workgroup owns a
//               (32*WM rows) x (16*TN*WN weight rows) tile and a range of 512-byte K chunks; a whole chunk of both
//               operands -- 24..36 16-byte loads per lane, all issued back to back -- is pulled into registers,
//               written to LDS in full 512-byte rows (+16 B pad: conflict-free 16-byte fragment reads) and consumed
//               by the matrix cores while the next chunk's loads are in flight.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace lcmhip {


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

// Compile-time loop: f(std::integral_constant<int, I>) for I in [0, N).  Array indices are then
// constants in the source itself, so register arrays never fall back to scratch memory.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int kChunkBytes = 512;             // bytes of K per tile row per chunk
constexpr int kRowBytes = kChunkBytes + 16;  // padded LDS row

// WM x WN waves; each wave owns 32 activation rows x (16 * TN) weight rows.
template <typename T, int WM, int WN, int TN, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm(GemmArgs p) {
    constexpr bool DUAL = (EPI == LCMHIP_EPI_SWIGLU);
    constexpr int WR = 16 * TN;                  // weight rows per wave
    constexpr int BM = 32 * WM, BNR = WR * WN;   // activation rows / weight rows per tile
    constexpr int R = BM + BNR;
    constexpr int KSTEP_BYTES = 4 * Mma<T>::kLaneBytes;
    constexpr int KSTEPS = kChunkBytes / KSTEP_BYTES;
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
    constexpr int RP = 2 * WM * WN;              // tile rows one pass of the staging loads covers (32 lanes per row)
    static_assert(!DUAL || (WR / 2) % RP == 0, "a staging pass must not straddle the gate / inner halves");

    __shared__ __attribute__((aligned(16))) char smem[R * kRowBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wn = wid % WN, wm = wid / WN;
    const int m0 = blockIdx.z * BM;
    const size_t kbytes = (size_t)p.K * sizeof(T);
    const int nchunks = (int)(kbytes / kChunkBytes);
    const int cps = (nchunks + (int)gridDim.y - 1) / (int)gridDim.y;   // chunks per split
    const int c_begin = blockIdx.y * cps;
    const int c_end = min(nchunks, c_begin + cps);

    // Lane (r0 = tid / 32, c = tid % 32) moves the 16-byte piece c of tile rows r0, r0 + RP, r0 + 2 RP, ...:
    // 32 consecutive lanes cover one 512-byte row segment.  Row addresses are recomputed from two
    // scalars per operand (no per-piece pointer arrays).
    constexpr int NA = BM / RP, NW = BNR / RP;
    const int r0 = tid >> 5;
    const size_t col_off = (size_t)(tid & 31) * 16 + (size_t)c_begin * kChunkBytes;
    const char* a_base = reinterpret_cast<const char*>(p.A) + col_off;
    const size_t a_stride = (size_t)p.lda * sizeof(T);
    const char* w_base;
    const char* w2_base = nullptr;
    if constexpr (DUAL) {
        w_base = reinterpret_cast<const char*>(p.W) + (size_t)(blockIdx.x * (BNR / 2) + r0) * kbytes + col_off;
        w2_base = reinterpret_cast<const char*>(p.W2) + (size_t)(blockIdx.x * (BNR / 2) + r0) * kbytes + col_off;
    } else {
        w_base = reinterpret_cast<const char*>(p.W) + (size_t)(blockIdx.x * BNR + r0) * kbytes + col_off;
    }
    char* lds_dst = smem + r0 * kRowBytes + (tid & 31) * 16;

    // One register set: while a chunk is consumed from LDS the next chunk's loads are in flight.  (A second set,
    // i.e. two chunks in flight, measured slower: the tile is bound by LDS and L1 intake, not by load latency;
    // so did non-temporal weight loads, because both row tiles read every weight row.)
    u32x4 ra[NA], rw[NW];
#define LCM_LOAD_CHUNK(OFF)                                                                              \
    {                                                                                                    \
        const size_t off_ = (OFF);                                                                       \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                 \
            int m = m0 + r0 + RP * i;                                                                    \
            m = m < p.M ? m : p.M - 1;                                                                   \
            ra[i] = *reinterpret_cast<const u32x4*>(a_base + (size_t)m * a_stride + off_);               \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) {                                                 \
            if constexpr (DUAL) {                                                                        \
                /* tile row RP i + r0 belongs to wave (RP i) / WR; a wave's first WR/2 rows = gate, rest = inner */ \
                const int grp = (RP * i) / WR, rr = (RP * i) % WR;                                       \
                const char* b = (rr < WR / 2 ? w_base : w2_base) + (size_t)(grp * (WR / 2) + (rr % (WR / 2))) * kbytes; \
                rw[i] = *reinterpret_cast<const u32x4*>(b + off_);                                       \
            } else {                                                                                     \
                rw[i] = *reinterpret_cast<const u32x4*>(w_base + (size_t)(RP * i) * kbytes + off_);      \
            }                                                                                            \
        }                                                                                                \
    }
#define LCM_STORE_CHUNK()                                                                                \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                   \
            *reinterpret_cast<u32x4*>(lds_dst + (RP * i) * kRowBytes) = ra[i];                           \
        _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                   \
            *reinterpret_cast<u32x4*>(lds_dst + (BM + RP * i) * kRowBytes) = rw[i];                      \
    }

    f32x4 acc[TN][2];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fk = (lane >> 4) * Mma<T>::kLaneBytes;
    const char* w_frag = smem + (BM + wn * WR + frow) * kRowBytes + fk;
    const char* x_frag = smem + (wm * 32 + frow) * kRowBytes + fk;

    LCM_LOAD_CHUNK(0);   // every split owns at least one chunk (see make_plan)
    for (int ch = c_begin; ch < c_end; ++ch) {
        if (ch != c_begin) __syncthreads();   // previous chunk fully consumed before it is overwritten
        LCM_STORE_CHUNK();
        __syncthreads();
        if (ch + 1 < c_end) LCM_LOAD_CHUNK((size_t)(ch + 1 - c_begin) * kChunkBytes);
#pragma unroll 2
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int kb = ks * KSTEP_BYTES;
            typename Mma<T>::Frag w[TN], x[2];
#pragma unroll
            for (int t = 0; t < TN; ++t) w[t] = Mma<T>::ld(w_frag + t * 16 * kRowBytes + kb);
#pragma unroll
            for (int t = 0; t < 2; ++t) x[t] = Mma<T>::ld(x_frag + t * 16 * kRowBytes + kb);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = Mma<T>::mma(w[tn], x[tm], acc[tn][tm]);
        }
    }
#undef LCM_LOAD_CHUNK
#undef LCM_STORE_CHUNK

    // ---- epilogue: lane holds 4 consecutive columns of row m for each (tn, tm)
    gemm_epilogue<T, EPI, 2, TN>(p, acc, m0 + wm * 32, DUAL ? blockIdx.x * (BNR / 2) + wn * (WR / 2) : blockIdx.x * BNR + wn * WR,
                                 blockIdx.y, lane);
}

// Sums the split-K slabs and applies a (non-dual) epilogue.  One lane = 4 consecutive columns.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void k_splitk_finish(GemmArgs p, int splits) {
    const int n4 = p.N / 4;
    const int rows = p.M + (EPI == LCMHIP_EPI_RESID ? p.extra_rows : 0);
    const int64_t total = (int64_t)rows * n4;
    const T* bias = reinterpret_cast<const T*>(p.bias);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i % n4) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < p.M) {
#pragma unroll 4
            for (int s = 0; s < splits; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(p.partial + ((size_t)s * p.M + m) * p.N + n);
                v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
            }
            if (bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += Elem<T>::load(bias + n + r);
            }
        } else {
            const float4 c = *reinterpret_cast<const float4*>(p.extra_const + n);
            v[0] = c.x; v[1] = c.y; v[2] = c.z; v[3] = c.w;
        }
        if constexpr (EPI == LCMHIP_EPI_RESID) {
            float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
            if (p.gate) {
                const size_t goff = (p.gate_rows != nullptr && m < p.M)
                                        ? (size_t)min(max(p.gate_rows[m], 0), p.gate_steps - 1) * p.gate_stride : 0;
                g = *reinterpret_cast<const float4*>(p.gate + goff + n);
            }
            float4* dstp = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n);
            float4 o = *dstp;
            o.x += g.x * v[0]; o.y += g.y * v[1]; o.z += g.z * v[2]; o.w += g.w * v[3];
            *dstp = o;
        } else {
            if constexpr (EPI == LCMHIP_EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], act_kind(p), act_slope<T>(p));
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

// ---- host-side dispatch -------------------------------------------------------------------------------
bool gemm_big_applicable(const GemmArgs& a, bool may_split);   // gemm_big.hip: many-row bf16 problems
int gemm_big_splits(const GemmArgs& a, int max_splits);
int launch_gemm_big(const GemmArgs& a, int epi, int splits, int m_rows, hipStream_t st);

constexpr double kDmaChunk256 = 256.0;   // bytes of K per tile row and pipeline stage of the LDS-DMA tiles

struct Plan {
    int wm, wn, tn;    // wave grid and weight tiles per wave
    int n_tiles, m_tiles, splits;
};

static Plan make_plan(const GemmArgs& a, bool may_split) {
    const int rows = a.M;   // rows that go through the matrix cores
    const bool dual = a.epilogue == LCMHIP_EPI_SWIGLU;
    const int wrows = dual ? 2 * a.N : a.N;   // weight rows streamed
    Plan pl{};
    // bf16 with at most 64 rows: the 64-row LDS-DMA tile also for fewer than 33 rows (the unused rows cost matrix-core
    // work only; measured on the SwiGLU up-projection at 16 rows: 38.7 us register-staged 32 x 256 -> 17.5 us)
    const bool dma64 = a.dtype == LCMHIP_BF16;      // bf16 always takes the LDS-DMA tiles
    if (!dma64 && rows <= 32 && wrows % 256 == 0) { pl.wm = 1; pl.wn = 4; pl.tn = 4; }         // 32 x 256
    else if (!dma64 && rows <= 64 && wrows % 128 == 0 && wrows >= 128 * 192) { pl.wm = 2; pl.wn = 2; pl.tn = 4; }   // 64 x 128
    else if (rows > 64) { pl.wm = 4; pl.wn = 2; pl.tn = 2; }      // 128 x 64, 8 waves: every weight row read by one workgroup
    else { pl.wm = 2; pl.wn = 2; pl.tn = 2; }                                        // 64 x 64, two per CU
    const int bm = 32 * pl.wm, bnr = 16 * pl.tn * pl.wn;
    pl.n_tiles = wrows / bnr;
    pl.m_tiles = (rows + bm - 1) / bm;
    pl.splits = 1;
    if (may_split) {
        const int nchunks = (int)((size_t)a.K * dtype_size(a.dtype) / kChunkBytes);
        const int tiles = pl.n_tiles * pl.m_tiles;
        // enough workgroups to put two on every CU, every split at least one chunk
        const int target = pl.wm == 4 ? 256 : 512;           // the tall tile fits once per CU (128 / 192 / 320 / 512
                                                             // and 256 / 384 / 768 measured end to end: no better)
        int s = (target + tiles - 1) / tiles;
        s = std::min(s, nchunks);
        // the LDS-DMA tall tile wants at least two 512-byte chunks per split (four stages of its pipeline):
        // measured 10.6 -> 10.1 us on 128 x 2048 x 2048; the 64-row tile is faster with one (7.1 vs 7.7 us)
        // (re-measured with the DMA spread over the k-steps: one chunk per split makes self_ov itself 9 % faster, 6.2
        // against 6.9 us, and the whole loop 1.5 % slower, 1313 against 1333 sentence-embeddings/s: twice the slabs for
        // the next norm kernel)
        if (pl.wm == 4 && a.dtype == LCMHIP_BF16) s = std::min(s, std::max(1, nchunks / 2));
        s = std::min(s, a.max_splits > 0 ? a.max_splits : 64);
        pl.splits = std::max(1, s);
        // re-balance: every split gets the same number of chunks where possible
        const int cps = (nchunks + pl.splits - 1) / pl.splits;
        pl.splits = (nchunks + cps - 1) / cps;
    }
    return pl;
}

// Does a K split followed by a finishing pass (lcmhip_gemm's form) beat the unsplit launch?  Estimate in microseconds
// from what the LDS-DMA tiles were measured at: a workgroup advances one 256-byte K chunk per 0.7 us (128-row tile, one
// workgroup per CU) or 0.55 us (64-row tile, two per CU), workgroups beyond one round of the chip queue behind the
// first, and the finishing pass costs a launch (4 us) plus its slab traffic at 4 TB/s.  Checked against bf16 shapes of
// 64 to 1024 rows, N = 2048, K = 2048 / 5632: it splits 64 x 2048 x 2048 (9 against 12 us), 256 and 512 x 2048 x 5632
// (16.0 against 27.5 us, 24.5 against 28.5 us) and leaves 384 and 768 x 2048 x 2048 (12.2 against 17.0 us, 13.0 against
// 20.4 us) unsplit.
static bool split_pays_with_finish(const GemmArgs& a, const Plan& pl) {
    const bool tall = pl.wm == 4;
    const double per_chunk = tall ? 0.7 : 0.55;
    const int round = 256 * (tall ? 1 : 2);
    const double nch = (double)a.K * 2.0 / kDmaChunk256;
    const int wgs = pl.n_tiles * pl.m_tiles;
    auto rounds = [&](int n) { return (double)((n + round - 1) / round); };
    const double unsplit = rounds(wgs) * nch * per_chunk;
    const double split = rounds(wgs * pl.splits) * (nch / pl.splits) * per_chunk + 4.0 + (pl.splits + 1.0) * (double)a.M * a.N * 4.0 / 4.0e6;
    return split < unsplit;
}

// ------------------------------------------------------------------------------------------------
// The 128 x 64 tile (8 waves) for bf16 with LDS-DMA staging: 256-byte K chunks go global -> LDS directly
// (global_load_lds, 16 bytes per lane; one instruction fills 4 tile rows), three stages, the DMA two chunks ahead of
// the matrix cores across one raw barrier per chunk -- no staging registers and no LDS store instructions.
// The LDS image is lane-linear, so bank conflicts of the fragment reads (16 rows x one 16-byte piece) are removed on
// the SOURCE side: slot s of tile row r holds K piece s ^ (r % 16).
// ------------------------------------------------------------------------------------------------
constexpr int kDmaChunk = 256;                       // bytes of K per tile row and stage
constexpr int dma_stage_bytes(int wm) { return (32 * wm + 64) * kDmaChunk; }   // activation rows + 64 weight rows
constexpr int dma_lds_bytes(int wm) { return 3 * dma_stage_bytes(wm); }         // 144 KB (128 rows) / 96 KB (64 rows)
typedef const __attribute__((address_space(1))) void* dma_gptr_t;
typedef __attribute__((address_space(3))) void* dma_lptr_t;

// WM wave rows: (32 WM) activation rows x 64 weight rows per workgroup of 2 WM waves.
// NT: non-temporal weight DMA (a single row tile reads every weight row exactly once).
template <int EPI, int WM, bool NT>
__global__ __launch_bounds__(128 * WM) void k_gemm_dma(GemmArgs p) {
    constexpr bool DUAL = (EPI == LCMHIP_EPI_SWIGLU);
    constexpr int NW = 2 * WM;                       // waves
    constexpr int BM = 32 * WM;                      // activation rows
    constexpr int XG = BM / 4;                       // row groups (4 rows) of the activation part
    constexpr int NG = (BM + 64) / 4 / NW;           // row groups per wave and stage
    constexpr int XJ = XG / NW;                      // ... of which activation groups
    constexpr int kDmaStage = dma_stage_bytes(WM);
    extern __shared__ __attribute__((aligned(1024))) char smem_dma[];
    char* const smem = smem_dma;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid & 1, wm = wid >> 1;           // 4 x 2 waves, 32 activation rows x 32 weight rows each
    const int m0 = blockIdx.z * BM;
    const uint32_t kbytes = (uint32_t)p.K * 2;
#ifdef LCM_DMA_STAMPS
    // timeline probe (scripts/gemm_dma_timeline.py): 100 MHz wall-clock stamps of workgroup (x, y) into the otherwise
    // unused split-K workspace of an unsplit launch: start, first chunk landed, K loop done, epilogue done
    uint64_t* const stamps = (EPI != EPI_PARTIAL && p.partial != nullptr)
                                 ? reinterpret_cast<uint64_t*>(p.partial) + 4 * (size_t)(blockIdx.x + gridDim.x * blockIdx.z) : nullptr;
    if (stamps && tid == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();
#endif
    const int nch512 = (int)(kbytes / kChunkBytes);
    const int cps = (nch512 + (int)gridDim.y - 1) / (int)gridDim.y;      // 512-byte chunks per split, as planned
    const int c_begin = 2 * blockIdx.y * cps, c_end = 2 * min(nch512, (int)(blockIdx.y + 1) * cps);   // in 256-byte chunks
    const int nc = c_end - c_begin;

    // staging: row group g = 4 tile rows; wave w fills groups w, w + NW, ...: first the activation rows, then the
    // weight rows.  Lane -> (row lane / 16 of the group, slot lane % 16); for every group of this wave the tile row r
    // has r % 16 = 4 (w % 4) + lane / 16 (NW is a multiple of 4).
    const int srow = lane >> 4;
    const uint32_t piece_off = (uint32_t)(((lane & 15) ^ (4 * (wid & 3) + srow)) * 16) + (uint32_t)c_begin * kDmaChunk;
    const char* src[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int g = wid + NW * j;
        if (j < XJ) {
            int m = m0 + 4 * g + srow;
            m = m < p.M ? m : p.M - 1;               // rows past the end repeat the last row; never stored
            src[j] = reinterpret_cast<const char*>(p.A) + (size_t)m * p.lda * 2 + piece_off;
        } else {
            const int t = 4 * (g - XG) + srow;       // weight row inside the tile, 0..63
            if constexpr (DUAL) {
                // wave column t / 32 owns 16 gate rows followed by 16 inner rows of the same 16 outputs
                const int wcol = t >> 5, rr = t & 31;
                const size_t row = (size_t)blockIdx.x * 32 + wcol * 16 + (rr & 15);
                src[j] = reinterpret_cast<const char*>(rr < 16 ? p.W : p.W2) + row * kbytes + piece_off;
            } else {
                src[j] = reinterpret_cast<const char*>(p.W) + ((size_t)blockIdx.x * 64 + t) * kbytes + piece_off;
            }
        }
    }
    // A workgroup walks its K range from a chunk of its own and wraps around: the workgroups of an XCD (ids i, i + 8,
    // ...) all read the SAME activation chunk from their L2 at about the same time otherwise (every one of them needs the
    // whole row tile), and those reads queue on the few L2 channels that hold it.
    const int rot = nc > 1 ? (int)(((blockIdx.x >> 3) * 3u + (blockIdx.x & 7u)) % (unsigned)nc) : 0;
    auto chunk_of = [&](int c) { const int r = c + rot; return r >= nc ? r - nc : r; };
    auto stage = [&](int c, int buf) {
        char* dst = smem + buf * kDmaStage + wid * 1024;
        const uint32_t koff = (uint32_t)chunk_of(c) * kDmaChunk;
#pragma unroll
        for (int j = 0; j < XJ; ++j)
            __builtin_amdgcn_global_load_lds((dma_gptr_t)(src[j] + koff), (dma_lptr_t)(dst + j * NW * 1024), 16, 0, 0);
#pragma unroll
        for (int j = XJ; j < NG; ++j) {
            if constexpr (NT) __builtin_amdgcn_global_load_lds((dma_gptr_t)(src[j] + koff), (dma_lptr_t)(dst + j * NW * 1024), 16, 0, 2);
            else __builtin_amdgcn_global_load_lds((dma_gptr_t)(src[j] + koff), (dma_lptr_t)(dst + j * NW * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, kq = lane >> 4;
    const int x_off = (wm * 32 + frow) * kDmaChunk;
    const int w_off = (BM + wn * 32 + frow) * kDmaChunk;

    stage(0, 0);
    if (nc > 1) stage(1, 1);
    // one DMA instruction of chunk c + 2 (piece j of this wave)
    auto stage_piece = [&](int c, int buf, int j) {
        char* dst = smem + buf * kDmaStage + wid * 1024 + j * NW * 1024;
        const uint32_t koff = (uint32_t)chunk_of(c) * kDmaChunk;
        if (NT && j >= XJ) __builtin_amdgcn_global_load_lds((dma_gptr_t)(src[j] + koff), (dma_lptr_t)dst, 16, 0, 2);
        else __builtin_amdgcn_global_load_lds((dma_gptr_t)(src[j] + koff), (dma_lptr_t)dst, 16, 0, 0);
    };
    for (int c = 0; c < nc; ++c) {
        if (c + 1 < nc) {
            if constexpr (NG == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
#ifdef LCM_DMA_STAMPS
        if (stamps && tid == 0 && c == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
#endif
        const char* base = smem + (c % 3) * kDmaStage;
        const bool more = c + 2 < nc;
        const int nbuf = (c + 2) % 3;
        // the DMA of chunk c + 2 is spread over the four k-steps (behind each step's fragment reads) instead of one burst
        // behind the barrier, where every wave of the CU queues its instructions at the same moment
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 w[2], x[2];
            const int slot = ((4 * ks + kq) ^ frow) * 16;
#pragma unroll
            for (int t = 0; t < 2; ++t) w[t] = *reinterpret_cast<const bf16x8*>(base + w_off + t * 16 * kDmaChunk + slot);
#pragma unroll
            for (int t = 0; t < 2; ++t) x[t] = *reinterpret_cast<const bf16x8*>(base + x_off + t * 16 * kDmaChunk + slot);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                constexpr int per = NG / 4, extra = NG % 4;     // NG = 6: 2, 2, 1, 1 pieces; NG = 8: 2 each
                const int first = ks * per + (ks < extra ? ks : extra);
                const int count = per + (ks < extra ? 1 : 0);
#pragma unroll
                for (int j = 0; j < count; ++j) stage_piece(c + 2, nbuf, first + j);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[tn], x[tm], acc[tn][tm], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifdef LCM_DMA_STAMPS
    if (stamps && tid == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
#endif
    gemm_epilogue<bf16_t, EPI, 2, 2>(p, acc, m0 + wm * 32, DUAL ? blockIdx.x * 32 + wn * 16 : blockIdx.x * 64 + wn * 32,
                                     blockIdx.y, lane);
#ifdef LCM_DMA_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    if (stamps && tid == 0) stamps[3] = __builtin_amdgcn_s_memrealtime();
#endif
}


template <int WM>
static int launch_dma(const GemmArgs& a, const Plan& pl, int epi, hipStream_t st) {
    dim3 grid(pl.n_tiles, pl.splits, pl.m_tiles), block(128 * WM);
    constexpr int kDmaLds = dma_lds_bytes(WM);
    // Non-temporal weight DMA when a single row tile reads every weight row once (the weights of a model do not
    // fit the caches between launches): +1 % end to end.
    const bool nt = pl.m_tiles == 1;
#define LCM_DMA1(E, N)                                                                                              \
    {                                                                                                               \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_dma<E, WM, N>),    \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, kDmaLds);    \
        LCM_HIP_CHECK(attr);                                                                                        \
        hipLaunchKernelGGL((k_gemm_dma<E, WM, N>), grid, block, kDmaLds, st, a);                                    \
    }
#define LCM_DMA(E) { if (nt) LCM_DMA1(E, true) else LCM_DMA1(E, false) }
    switch (epi) {
        case LCMHIP_EPI_STORE: LCM_DMA(LCMHIP_EPI_STORE); break;
        case LCMHIP_EPI_STORE_F32: LCM_DMA(LCMHIP_EPI_STORE_F32); break;
        case LCMHIP_EPI_RELU: LCM_DMA(LCMHIP_EPI_RELU); break;
        case LCMHIP_EPI_SWIGLU: LCM_DMA(LCMHIP_EPI_SWIGLU); break;
        case LCMHIP_EPI_RESID: LCM_DMA(LCMHIP_EPI_RESID); break;
        case EPI_PARTIAL: LCM_DMA(EPI_PARTIAL); break;
        default: return fail(LCMHIP_ERR_INVALID, "gemm: unknown epilogue");
    }
#undef LCM_DMA
#undef LCM_DMA1
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

template <typename T, int WM, int WN, int TN>
static int launch_cfg(const GemmArgs& a, const Plan& pl, int epi, hipStream_t st) {
    dim3 grid(pl.n_tiles, pl.splits, pl.m_tiles), block(64 * WM * WN);
    switch (epi) {
        case LCMHIP_EPI_STORE: hipLaunchKernelGGL((k_gemm<T, WM, WN, TN, LCMHIP_EPI_STORE>), grid, block, 0, st, a); break;
        case LCMHIP_EPI_STORE_F32: hipLaunchKernelGGL((k_gemm<T, WM, WN, TN, LCMHIP_EPI_STORE_F32>), grid, block, 0, st, a); break;
        case LCMHIP_EPI_RELU: hipLaunchKernelGGL((k_gemm<T, WM, WN, TN, LCMHIP_EPI_RELU>), grid, block, 0, st, a); break;
        case LCMHIP_EPI_SWIGLU: hipLaunchKernelGGL((k_gemm<T, WM, WN, TN, LCMHIP_EPI_SWIGLU>), grid, block, 0, st, a); break;
        case LCMHIP_EPI_RESID: hipLaunchKernelGGL((k_gemm<T, WM, WN, TN, LCMHIP_EPI_RESID>), grid, block, 0, st, a); break;
        case EPI_PARTIAL: hipLaunchKernelGGL((k_gemm<T, WM, WN, TN, EPI_PARTIAL>), grid, block, 0, st, a); break;
        default: return fail(LCMHIP_ERR_INVALID, "gemm: unknown epilogue");
    }
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

template <typename T>
static int launch_plan(const GemmArgs& a0, const Plan& pl, int epi, hipStream_t st) {
    GemmArgs a = a0;
    a.slab_through = epi == EPI_PARTIAL && a.K < 4096;   // see store16_through
    if (pl.wm == 1) return launch_cfg<T, 1, 4, 4>(a, pl, epi, st);
    if (pl.wm == 4 && pl.wn == 2) {
        if (std::is_same<T, bf16_t>::value) return launch_dma<4>(a, pl, epi, st);
        return launch_cfg<T, 4, 2, 2>(a, pl, epi, st);
    }
    if (pl.tn == 4) return launch_cfg<T, 2, 2, 4>(a, pl, epi, st);
    if (std::is_same<T, bf16_t>::value) return launch_dma<2>(a, pl, epi, st);
    return launch_cfg<T, 2, 2, 2>(a, pl, epi, st);
}

template <typename T>
static int launch_finish(const GemmArgs& a, int splits, hipStream_t st) {
    const int rows = a.M + (a.epilogue == LCMHIP_EPI_RESID ? a.extra_rows : 0);
    const int64_t total = (int64_t)rows * (a.N / 4);
    const int grid = (int)std::min<int64_t>(2048, (total + 255) / 256);
    switch (a.epilogue) {
        case LCMHIP_EPI_STORE: hipLaunchKernelGGL((k_splitk_finish<T, LCMHIP_EPI_STORE>), dim3(grid), dim3(256), 0, st, a, splits); break;
        case LCMHIP_EPI_STORE_F32: hipLaunchKernelGGL((k_splitk_finish<T, LCMHIP_EPI_STORE_F32>), dim3(grid), dim3(256), 0, st, a, splits); break;
        case LCMHIP_EPI_RELU: hipLaunchKernelGGL((k_splitk_finish<T, LCMHIP_EPI_RELU>), dim3(grid), dim3(256), 0, st, a, splits); break;
        case LCMHIP_EPI_RESID: hipLaunchKernelGGL((k_splitk_finish<T, LCMHIP_EPI_RESID>), dim3(grid), dim3(256), 0, st, a, splits); break;
        default: return fail(LCMHIP_ERR_INVALID, "gemm: epilogue cannot be applied after split-K");
    }
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

static int check_args(const GemmArgs& a) {
    LCM_REQUIRE(a.M >= 1 && a.N >= 1 && a.K >= 1, "gemm: empty problem");
    const size_t es = dtype_size(a.dtype);
    LCM_REQUIRE(a.dtype == LCMHIP_BF16 || a.dtype == LCMHIP_F32, "gemm: unknown dtype");
    LCM_REQUIRE((a.K * es) % kChunkBytes == 0, "gemm: K must cover whole 512-byte chunks (256 bf16 / 128 f32 elements)");
    LCM_REQUIRE(a.N % 64 == 0, "gemm: N must be a multiple of 64");
    LCM_REQUIRE(a.epilogue != LCMHIP_EPI_SWIGLU || (a.W2 != nullptr && a.N % 32 == 0), "gemm: SWIGLU needs W2");
    LCM_REQUIRE(((size_t)a.lda * es) % 16 == 0 && ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0,
                "gemm: operands must be 16-byte aligned");
    LCM_REQUIRE(a.extra_rows == 0 || (a.epilogue == LCMHIP_EPI_RESID && a.extra_const), "gemm: extra rows need RESID");
    LCM_REQUIRE(a.gate_rows == nullptr || (a.epilogue == LCMHIP_EPI_RESID && a.gate && a.gate_steps >= 1 && a.extra_rows == 0),
                "gemm: per-row gates need RESID, a gate table and no extra rows");
    LCM_REQUIRE(a.out_head_cap == 0 || (a.out_rows != nullptr && a.N % 128 == 0), "gemm: the KV-cache layout needs a row map and 128-wide heads");
    return LCMHIP_OK;
}

int gemm_partial_splits(const GemmArgs& a) {
    if (check_args(a) != LCMHIP_OK) return -1;
    if (gemm_big_applicable(a, true)) return gemm_big_splits(a, a.max_splits);
    return make_plan(a, true).splits;
}

// Writes raw fp32 partial tiles to a.partial ([splits][M][N]); returns the split count through *splits.
int launch_gemm_partial(const GemmArgs& a, int* splits, hipStream_t st) {
    LCM_TRY(check_args(a));
    LCM_REQUIRE(a.partial != nullptr && a.epilogue != LCMHIP_EPI_SWIGLU, "gemm: partial mode needs a slab and a linear epilogue");
    if (gemm_big_applicable(a, true)) {
        const int s = gemm_big_splits(a, a.max_splits);
        LCM_REQUIRE((size_t)s * a.M * a.N * 4 <= a.partial_bytes, "gemm: split-K slab too small");
        *splits = s;
        return launch_gemm_big(a, EPI_PARTIAL, s, a.M, st);
    }
    const Plan pl = make_plan(a, true);
    LCM_REQUIRE((size_t)pl.splits * a.M * a.N * 4 <= a.partial_bytes, "gemm: split-K slab too small");
    *splits = pl.splits;
    return a.dtype == LCMHIP_BF16 ? launch_plan<bf16_t>(a, pl, EPI_PARTIAL, st) : launch_plan<float>(a, pl, EPI_PARTIAL, st);
}

bool gemm_stack_is_fused(const GemmArgs& a) {
    return a.stack > 1 && a.stack <= 3 && check_args(a) == LCMHIP_OK && gemm_big_applicable(a, false);
}

int launch_gemm(const GemmArgs& a, hipStream_t st) {
    LCM_TRY(check_args(a));
    LCM_REQUIRE(a.out != nullptr && a.ldo % 4 == 0 && ((uintptr_t)a.out % 16) == 0, "gemm: output must be 16-byte aligned");
    if (a.stack > 1) {
        // stacked matrices over one A: a single launch of a many-row kernel when the stack as a whole fills the chip
        // without a K split, otherwise one product per matrix
        LCM_REQUIRE(a.stack <= 3 && (a.epilogue == LCMHIP_EPI_STORE || a.epilogue == LCMHIP_EPI_STORE_F32 || a.epilogue == LCMHIP_EPI_RELU),
                    "gemm: stacked products take up to 3 matrices and a linear epilogue");
        if (gemm_big_applicable(a, false)) return launch_gemm_big(a, a.epilogue, 1, a.M, st);
        LCM_REQUIRE(a.rope_mask == 0 && a.scatter_mask == 0, "gemm: per-matrix epilogue switches need the single-launch form (gemm_stack_is_fused)");
        for (int g = 0; g < a.stack; ++g) {
            GemmArgs one = a;
            one.stack = 0;
            if (g > 0) { one.W = a.stack_W[g - 1]; one.bias = a.stack_bias[g - 1]; one.out = a.stack_out[g - 1]; }
            LCM_TRY(launch_gemm(one, st));
        }
        return LCMHIP_OK;
    }
    const bool can_split = a.partial != nullptr && a.epilogue != LCMHIP_EPI_SWIGLU;
    // A K split here is followed by a finishing pass over the slabs (unlike the pending form, whose consumer sums
    // them), which a many-row product never earns back: 1024 x 2048 x 2048 took 24.4 us split four ways on the
    // matrix-core kernel against 13.7 us unsplit, the residual update 26.0 against 15.9 us, 1024 x 2048 x 5632 36.6
    // against 30.5 us.  So the matrix-core kernels run unsplit in this form ...
    if (gemm_big_applicable(a, false)) {
        // (one exception, measured: at most one 128 x 128 tile per CU and a long K loop -- 2048 x 2048 x 5632: 54.7 us
        // split in two against 59.5-65 us; with K = 2048 the split loses, 30.7 against 25.4-27.2 us)
        const int tiles = (a.N / 128) * ((a.M + 127) / 128);
        int s = (can_split && a.stack <= 1 && tiles <= 256 && a.K >= 4096) ? gemm_big_splits(a, a.max_splits) : 1;
        if ((size_t)s * a.M * a.N * 4 > a.partial_bytes) s = 1;
        if (s > 1) {
            LCM_TRY(launch_gemm_big(a, EPI_PARTIAL, s, a.M, st));
            return launch_finish<bf16_t>(a, s, st);
        }
        return launch_gemm_big(a, a.epilogue, 1, a.M + (a.epilogue == LCMHIP_EPI_RESID ? a.extra_rows : 0), st);
    }
    Plan pl = make_plan(a, can_split);
    if (can_split && (size_t)pl.splits * a.M * a.N * 4 > a.partial_bytes) pl.splits = 1;
    // ... and the weight-streaming tiles split only where the estimate says the finishing pass is paid for
    if (pl.splits > 1 && a.dtype == LCMHIP_BF16 && !split_pays_with_finish(a, pl)) pl.splits = 1;
    if (pl.splits > 1) {
        LCM_TRY(a.dtype == LCMHIP_BF16 ? launch_plan<bf16_t>(a, pl, EPI_PARTIAL, st) : launch_plan<float>(a, pl, EPI_PARTIAL, st));
        return a.dtype == LCMHIP_BF16 ? launch_finish<bf16_t>(a, pl.splits, st) : launch_finish<float>(a, pl.splits, st);
    }
    if (a.epilogue == LCMHIP_EPI_RESID && a.extra_rows > 0) {
        // direct RESID with extra constant rows: the row tiles must cover them
        Plan p2 = pl;
        const int bm = 32 * p2.wm;
        p2.m_tiles = (a.M + a.extra_rows + bm - 1) / bm;
        return a.dtype == LCMHIP_BF16 ? launch_plan<bf16_t>(a, p2, a.epilogue, st) : launch_plan<float>(a, p2, a.epilogue, st);
    }
    return a.dtype == LCMHIP_BF16 ? launch_plan<bf16_t>(a, pl, a.epilogue, st) : launch_plan<float>(a, pl, a.epilogue, st);
}

}  // namespace lcmhip
