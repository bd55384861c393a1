// Host side of liblcm_hip.so: weight table, per-batch session (caches + scratch carved from a
// caller-provided workspace), the contextualizer append, the denoising loop and the C ABI.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace lcmhip {

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

// ------------------------------------------------------------------------------------------------ profiler
struct Profiler {
    struct Pair { hipEvent_t a, b; int row; };
    struct Row { std::string name; int64_t launches = 0; double ms = 0, bytes = 0, flops = 0; };
    bool enabled = false;
    std::vector<Pair> pairs;      // pairs recorded since the last collect
    std::vector<Pair> free_pairs;
    // At most this many pairs are outstanding: begin() drains the stream and collects when the limit is reached.
    // Unbounded, a profiled pass of a long run holds tens of thousands of live HIP events; rocprofv3 in counter
    // collection mode (--pmc) died with a host-side SIGSEGV inside its dispatch interception at about 4000 profiled
    // launches (512 documents x 496 prompt positions; 128 documents x 100 steps alike), whatever kernel came next.
    static constexpr size_t kMaxPairs = 1024;
    std::vector<Row> rows;
    std::map<std::string, int> index;

    int begin(const char* name, double bytes, double flops, hipStream_t st) {
        if (!enabled) return -1;
        int r;
        auto it = index.find(name);
        if (it == index.end()) {
            r = (int)rows.size();
            index[name] = r;
            rows.push_back(Row{name});
        } else {
            r = it->second;
        }
        rows[r].launches += 1;
        rows[r].bytes += bytes;
        rows[r].flops += flops;
        if (pairs.size() >= kMaxPairs && hipStreamSynchronize(st) == hipSuccess) collect();
        Pair p;
        if (!free_pairs.empty()) {
            p = free_pairs.back();
            free_pairs.pop_back();
        } else {
            if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return -1;
        }
        p.row = r;
        (void)hipEventRecord(p.a, st);
        pairs.push_back(p);
        return (int)pairs.size() - 1;
    }
    void end(int slot, hipStream_t st) {
        if (slot >= 0) (void)hipEventRecord(pairs[slot].b, st);
    }
    void collect() {
        for (auto& p : pairs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) rows[p.row].ms += ms;
            free_pairs.push_back(p);
        }
        pairs.clear();
    }
    void clear_rows() {
        rows.clear();
        index.clear();
    }
    ~Profiler() {
        for (auto& p : pairs) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        for (auto& p : free_pairs) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    }
};

ProfScope::ProfScope(Profiler* p_, const char* name, double bytes, double flops, hipStream_t st_)
    : p(p_), slot(-1), st(st_) {
    if (p && p->enabled) slot = p->begin(name, bytes, flops, st);
}
ProfScope::~ProfScope() {
    if (p && slot >= 0) p->end(slot, st);
}

// ------------------------------------------------------------------------------------------------ model
struct Tensor {
    const void* ptr = nullptr;
    std::vector<int64_t> shape;
};
struct Lin { const void* w = nullptr; const void* b = nullptr; int out = 0, in = 0; };
struct NormW { const void* w = nullptr; const void* b = nullptr; };
struct AttnW { Lin q, k, v, o; NormW qn, kn; };   // qn / kn: optional q / k LayerNorm (attention.py:143-199)
struct FfnW { Lin gate, inner, out; const void* act_slope = nullptr; };   // act_slope: PReLU's learned slope
struct CtxLayer {
    NormW sa_norm, ffn_norm;
    AttnW sa;
    FfnW ffn;
    NormW attn_norm, inner_norm;             // normformer order only: after the attention block / on the FFN's inner activations
    const void* residual_scale = nullptr;    // scale_residual only
};
struct DenLayer { NormW sa_norm, ca_norm, ffn_norm; AttnW sa, ca; FfnW ffn; Lin mod; Lin ov; };   // ov: optional folded output_proj * v_proj

}  // namespace lcmhip

struct lcmhip_model {
    lcmhip_model_desc desc;
    std::map<std::string, lcmhip::Tensor> tensors;
    bool finalized = false;
    const void* center = nullptr;
    const void* scale = nullptr;
    const void* clip_min = nullptr;     // optional (clipping normalizer, sonar_normalizer/builder.py:126-132)
    const void* clip_max = nullptr;
    lcmhip::Lin pre_linear, t_fc1, t_fc2, init_proj, final_proj, postnet;
    bool base = false;          // base LCM: frontend -> causal decoder -> postnet, no denoiser tower
    const void* pos_table = nullptr;
    const void* den_pos_table = nullptr;   // learned positions of the denoiser's latents (lcm_denoiser.py:211-212), optional
    std::vector<lcmhip::CtxLayer> ctx;
    std::vector<lcmhip::DenLayer> den;
    lcmhip::NormW ctx_final, den_final;
};

namespace lcmhip {

static int get_lin(lcmhip_model* m, const std::string& prefix, int out, int in, bool bias, Lin* l) {
    auto w = m->tensors.find(prefix + ".weight");
    if (w == m->tensors.end()) return fail(LCMHIP_ERR_STATE, "missing weight: " + prefix + ".weight");
    if (w->second.shape.size() != 2 || w->second.shape[0] != out || w->second.shape[1] != in)
        return fail(LCMHIP_ERR_INVALID, "bad shape for " + prefix + ".weight");
    l->w = w->second.ptr;
    l->out = out;
    l->in = in;
    l->b = nullptr;
    if (bias) {
        auto b = m->tensors.find(prefix + ".bias");
        if (b == m->tensors.end()) return fail(LCMHIP_ERR_STATE, "missing weight: " + prefix + ".bias");
        if (b->second.shape.size() != 1 || b->second.shape[0] != out)
            return fail(LCMHIP_ERR_INVALID, "bad shape for " + prefix + ".bias");
        l->b = b->second.ptr;
    }
    return LCMHIP_OK;
}

static bool has_tensor(const lcmhip_model* m, const std::string& name) { return m->tensors.count(name) != 0; }

static int get_vec(lcmhip_model* m, const std::string& name, int n, const void** p) {
    auto t = m->tensors.find(name);
    if (t == m->tensors.end()) return fail(LCMHIP_ERR_STATE, "missing weight: " + name);
    if (t->second.shape.size() != 1 || t->second.shape[0] != n) return fail(LCMHIP_ERR_INVALID, "bad shape for " + name);
    *p = t->second.ptr;
    return LCMHIP_OK;
}

// "standard": weight + bias; "rms" and "fp32": weight only; "unit": no parameters (the session substitutes ones)
static int get_norm(lcmhip_model* m, const std::string& prefix, int d, int kind, NormW* n) {
    n->w = n->b = nullptr;
    if (kind != LCMHIP_NORM_UNIT) LCM_TRY(get_vec(m, prefix + ".weight", d, &n->w));
    if (kind == LCMHIP_NORM_LAYERNORM) LCM_TRY(get_vec(m, prefix + ".bias", d, &n->b));
    return LCMHIP_OK;
}

// Kernel family of a norm style: everything but RMS is a LayerNorm whose absent parameters are 1 / 0.
static inline int kernel_norm_kind(int kind) { return kind == LCMHIP_NORM_RMS ? LCMHIP_NORM_RMS : LCMHIP_NORM_LAYERNORM; }

static int get_attn(lcmhip_model* m, const std::string& p, int d, const lcmhip_tower_desc& t, AttnW* a) {
    const bool out_bias = t.out_proj_bias != 0;
    a->qn = a->kn = NormW{};
    if (t.qk_layernorm) {
        LCM_TRY(get_norm(m, p + ".q_layer_norm", d, t.norm_kind, &a->qn));
        LCM_TRY(get_norm(m, p + ".k_layer_norm", d, t.norm_kind, &a->kn));
    }
    LCM_TRY(get_lin(m, p + ".q_proj", d, d, true, &a->q));
    LCM_TRY(get_lin(m, p + ".k_proj", d, d, true, &a->k));
    LCM_TRY(get_lin(m, p + ".v_proj", d, d, true, &a->v));
    LCM_TRY(get_lin(m, p + ".output_proj", d, d, out_bias, &a->o));
    return LCMHIP_OK;
}

static int get_ffn(lcmhip_model* m, const std::string& p, int d, const lcmhip_tower_desc& t, FfnW* f) {
    if (t.ffn_kind == LCMHIP_FFN_SWIGLU) LCM_TRY(get_lin(m, p + ".gate_proj", t.ffn_dim, d, true, &f->gate));
    LCM_TRY(get_lin(m, p + ".inner_proj", t.ffn_dim, d, true, &f->inner));
    LCM_TRY(get_lin(m, p + ".output_proj", d, t.ffn_dim, true, &f->out));
    f->act_slope = nullptr;   // torch.nn.PReLU() as the inner activation module: one learned slope
    if (t.ffn_kind == LCMHIP_FFN_RELU && t.ffn_activation == LCMHIP_ACT_PRELU) LCM_TRY(get_vec(m, p + ".inner_activation.weight", 1, &f->act_slope));
    return LCMHIP_OK;
}

static int check_tower(const lcmhip_tower_desc& t, int d, int dtype, const char* which) {
    LCM_REQUIRE(t.num_layers >= 1, std::string(which) + ": needs at least one layer");
    LCM_REQUIRE(t.num_heads >= 2 && d == t.num_heads * 128, std::string(which) + ": head_dim must be 128");
    // the down-projection streams K = ffn_dim operands in whole 512-byte chunks
    LCM_REQUIRE(t.ffn_dim % (dtype == LCMHIP_BF16 ? 256 : 128) == 0,
                std::string(which) + ": ffn_dim must be a multiple of 128 (fp32) / 256 (bf16)");
    LCM_REQUIRE(t.norm_kind >= LCMHIP_NORM_LAYERNORM && t.norm_kind <= LCMHIP_NORM_UNIT, "unknown norm kind");
    LCM_REQUIRE(t.ffn_kind == LCMHIP_FFN_RELU || t.ffn_kind == LCMHIP_FFN_SWIGLU, "unknown ffn kind");
    LCM_REQUIRE(t.ffn_activation >= LCMHIP_ACT_RELU && t.ffn_activation <= LCMHIP_ACT_PRELU, "unknown ffn activation");
    LCM_REQUIRE(t.pos_kind >= LCMHIP_POS_NONE && t.pos_kind <= LCMHIP_POS_SINE, "unknown position kind");
    LCM_REQUIRE(t.norm_order >= LCMHIP_ORDER_PRE && t.norm_order <= LCMHIP_ORDER_NORMFORMER, "unknown norm order");
    LCM_REQUIRE((t.no_final_norm | 1) == 1 && (t.scale_residual | 1) == 1, std::string(which) + ": no_final_norm / scale_residual are 0 or 1");
    return LCMHIP_OK;
}

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace lcmhip

// ------------------------------------------------------------------------------------------------ session
struct lcmhip_session {
    lcmhip_model* model = nullptr;
    int max_batch = 0, cap = 0, max_steps = 0, rows_cap = 0, chunk_rows = 0;
    bool keep_context = false;
    char* init_bias_pos = nullptr;  // initial_proj bias + position 0 of the denoiser's learned positions (operand dtype)
    bool stack_products = true;     // q / k / v (and cross k / v) of many rows in one launch; off with LCMHIP_SESSION_SEPARATE_QKV
    bool fuse_rope_store = true;    // ... with the rotation and the cache scatter in its epilogue; off with LCMHIP_SESSION_SEPARATE_ROPE
    bool use_graphs = true;         // the sampling loop is replayed from a HIP graph; off with LCMHIP_SESSION_NO_GRAPH
    // carved device buffers
    int32_t *slot_map = nullptr, *pos_rows = nullptr, *self_rows = nullptr, *cross_rows = nullptr, *slot0_rows = nullptr;
    char *self_k = nullptr, *self_v = nullptr, *cross_k = nullptr, *cross_v = nullptr;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    float *mod_table = nullptr, *uconst = nullptr;
    float *xres = nullptr, *outp = nullptr, *lat = nullptr, *lat2 = nullptr, *x0 = nullptr, *f32tmp = nullptr;
    char *act_a = nullptr, *act_q = nullptr, *act_k = nullptr, *act_v = nullptr, *act_o = nullptr, *act_h = nullptr,
         *act_in = nullptr, *act_lat = nullptr;
    float* kept_ctx = nullptr;
    char* ones = nullptr;                          // model_dim ones in the operand dtype: weight of the "unit" norm style
    float *slab_p = nullptr, *slab_q = nullptr;   // split-K slabs: pending products / internal finishing
    int32_t* d_cross_keys = nullptr;              // device copy of ctx_len + 1 (read by graph-recorded attention launches)
    float *noise_buf = nullptr, *out_buf = nullptr;   // fixed addresses for graph-recorded denoising
    std::map<int, hipGraphExec_t> graphs;         // one recorded denoising loop per number of active documents
    hipStream_t graph_stream = nullptr;           // the stream the last recorded loop was launched on
    bool graph_launched = false;
    size_t slab_bytes = 0;
    // host state
    std::vector<int32_t> h_slot;
    int n_active = 0, ctx_len = 0;
    bool have_schedule = false;
    int n_steps = 0;
    std::vector<int32_t> timesteps;
    std::vector<lcmhip::DdimScalars> scalars;
    std::vector<int> step_mode_rows;  // 1: guidance step (2B rows), 0: plain
    lcmhip_denoise_params params{};
    bool do_cfg = false;
    lcmhip::Profiler prof;
};

namespace lcmhip {

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(char* b) : base(b) {}
    template <typename T>
    T* take(size_t bytes) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += align_up(bytes);
        return p;
    }
};

// Lays out the workspace; with base == nullptr only measures it.
static size_t carve(lcmhip_session* s, char* base) {
    const lcmhip_model_desc& d = s->model->desc;
    const size_t es = dtype_size(d.dtype);
    const size_t B = s->max_batch, cap = s->cap, dm = d.model_dim, D = d.sonar_dim;
    const size_t R = s->rows_cap;
    const size_t F = std::max(d.context.ffn_dim, d.denoiser.ffn_dim);
    Carver c(base);
    s->slot_map = c.take<int32_t>(B * 4);
    s->slot0_rows = c.take<int32_t>(B * 4);
    s->pos_rows = c.take<int32_t>(R * 4);
    s->self_rows = c.take<int32_t>(R * 4);
    s->cross_rows = c.take<int32_t>(R * 4);
    s->self_k = c.take<char>((size_t)d.context.num_layers * B * cap * dm * es);
    s->self_v = c.take<char>((size_t)d.context.num_layers * B * cap * dm * es);
    s->cross_k = c.take<char>((size_t)d.denoiser.num_layers * B * (cap + 1) * dm * es);
    s->cross_v = c.take<char>((size_t)d.denoiser.num_layers * B * (cap + 1) * dm * es);
    s->rope_cos = c.take<float>(cap * 64 * 4);
    s->rope_sin = c.take<float>(cap * 64 * 4);
    s->mod_table = c.take<float>((size_t)s->max_steps * d.denoiser.num_layers * 9 * dm * 4);
    s->uconst = c.take<float>((size_t)d.denoiser.num_layers * dm * 4);
    s->xres = c.take<float>(R * dm * 4);
    s->f32tmp = c.take<float>(R * std::max(std::max(dm, D), (size_t)d.timestep_freq_dim) * 4);
    s->outp = c.take<float>(2 * B * D * 4);
    s->lat = c.take<float>(B * D * 4);
    s->lat2 = c.take<float>(B * D * 4);
    s->x0 = c.take<float>(B * D * 4);
    s->act_in = c.take<char>(R * std::max(D, (size_t)d.timestep_freq_dim) * es);
    s->act_lat = c.take<char>(B * D * es);
    s->act_a = c.take<char>(R * dm * es);
    s->act_q = c.take<char>(R * dm * es);
    s->act_k = c.take<char>(R * dm * es);
    s->act_v = c.take<char>(R * dm * es);
    s->act_o = c.take<char>(R * dm * es);
    s->act_h = c.take<char>(R * F * es);
    s->kept_ctx = s->keep_context ? c.take<float>(B * cap * dm * 4) : nullptr;
    // split-K slabs: every split of a denoiser-sized problem (2 rows per document), or two slabs of a full chunk
    s->slab_bytes = std::max((size_t)kMaxSplits * std::max<size_t>(2 * B, 512), 2 * R) * std::max(dm, D) * 4;
    s->slab_p = c.take<float>(s->slab_bytes);
    s->slab_q = c.take<float>(s->slab_bytes);
    s->d_cross_keys = c.take<int32_t>(256);
    s->ones = c.take<char>(dm * es);
    s->init_bias_pos = c.take<char>(dm * es);
    s->noise_buf = c.take<float>(B * D * 4);
    s->out_buf = c.take<float>(B * D * 4);
    return c.off;
}

// ---- tiny index kernels ---------------------------------------------------------------------------
// Row r = (doc i = r / sc, token jj = r % sc) of an append chunk:
//   pos[r] = pos0 + jj; self_rows[r] = slot[i] * cap + pos[r]; cross_rows[r] = slot[i] * (cap + 1) + pos[r] + 1
__global__ void k_make_rows(const int32_t* __restrict__ slot_map, int32_t* pos, int32_t* self_rows,
                            int32_t* cross_rows, int rows, int sc, int pos0, int cap) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int i = r / sc, jj = r % sc;
    const int p = pos0 + jj;
    const int slot = slot_map[i];
    pos[r] = p;
    self_rows[r] = slot * cap + p;
    cross_rows[r] = slot * (cap + 1) + p + 1;
}

__global__ void k_set_int(int32_t* p, int32_t v) { *p = v; }

// pos[r] = j0 + r % sc: the position of row r inside its document (rows = documents x sc positions)
__global__ void k_row_positions(int32_t* __restrict__ pos, int rows, int sc, int j0) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) pos[r] = j0 + r % sc;
}

// out[i] = a[i] + b[i] in the operand dtype (a may be null = 0)
template <typename T>
__global__ void k_vec_add(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Elem<T>::store(out + i, (a ? Elem<T>::load(a + i) : 0.f) + Elem<T>::load(b + i));
}

__global__ void k_add_int(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int n, int add) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + add;
}

// Row r = (doc i = r / sc, position j0 + r % sc) of a pass over (n, n_pos) items: dst[r] = src[i * n_pos + j0 + r % sc]
__global__ void k_gather_rows_int(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int rows, int sc, int n_pos,
                                  int j0) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    dst[r] = src[(size_t)(r / sc) * n_pos + j0 + r % sc];
}

__global__ void k_slot0_rows(int32_t* slot0_rows, int32_t* slot_map, int n, int cap1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    slot_map[i] = i;
    slot0_rows[i] = i * cap1;
}

// y[r][:] = (x[(i * s_total + j0 + jj)][:] - center) / scale, r = i * sc + jj   (fp32 in, T out)
template <typename T>
__global__ __launch_bounds__(128) void k_normalize_gather(const float* __restrict__ x, const T* __restrict__ center,
                                                          const T* __restrict__ scale, T* __restrict__ y, int D,
                                                          int sc, int s_total, int j0, int apply, float escale,
                                                          const T* __restrict__ cmin, const T* __restrict__ cmax) {
    const int r = blockIdx.x;
    const int i = r / sc, jj = r % sc;
    const float* src = x + ((size_t)i * s_total + j0 + jj) * D;
    T* dst = y + (size_t)r * D;
    for (int c = threadIdx.x * 8; c < D; c += blockDim.x * 8) {
        float v[8];
        load8<float>(src + c, v);
        if (apply) {
            float ce[8], sv[8];
            load8<T>(center + c, ce);
            load8<T>(scale + c, sv);
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = (v[t] - ce[t]) / sv[t];
            if (cmin != nullptr) {   // clipping normalizer: clamp after centring and scaling
                float lo[8], hi[8];
                load8<T>(cmin + c, lo);
                load8<T>(cmax + c, hi);
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = fminf(fmaxf(v[t], lo[t]), hi[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] *= escale;     // frontend embedding scale (1 unless scale_embeddings)
        store8<T>(dst + c, v);
    }
}

// dst[rows_idx-like mapping]: kept_ctx[(slot[i] * cap + pos[r])][:] = src[r][:] as fp32
template <typename T>
__global__ __launch_bounds__(256) void k_keep_rows(const T* __restrict__ src, float* __restrict__ dst,
                                                   const int32_t* __restrict__ rows_idx, int d) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x * 8; c < d; c += blockDim.x * 8) {
        float v[8];
        load8<T>(src + (size_t)r * d + c, v);
        store8<float>(dst + (size_t)rows_idx[r] * d + c, v);
    }
}

// Output request of a base-LCM forward: postnet + optional sampling noise + denormalize.
struct BaseOut {
    float* out = nullptr;          // (n_active, n_out, D) fp32; n_out = n_tokens if all else 1
    const float* noise = nullptr;  // like out, or null
    float temperature = 0.f;
    int all = 0;
};

// out[dst] = denormalize(act(y[r]) + temperature * noise[dst]) for the rows of one append chunk
template <typename T>
__global__ __launch_bounds__(128) void k_base_out(const float* __restrict__ y, const float* __restrict__ noise,
                                                  const T* __restrict__ center, const T* __restrict__ scale,
                                                  float* __restrict__ out, int D, int sc, int n_tokens, int j0, int all,
                                                  float temperature, const T* __restrict__ cmin, const T* __restrict__ cmax,
                                                  int act) {
    const int r = blockIdx.x;
    const int i = r / sc, j = j0 + r % sc;
    if (!all && j != n_tokens - 1) return;
    const size_t dst = all ? (size_t)i * n_tokens + j : (size_t)i;
    for (int c = threadIdx.x * 8; c < D; c += blockDim.x * 8) {
        float v[8];
        load8<float>(y + (size_t)r * D + c, v);
        if (act > 0) {   // the postnet's own activation (projection.py:83-84), before noise and denormalization
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = apply_act(v[t], act - 1);
        }
        if (noise != nullptr) {
            float z[8];
            load8<float>(noise + dst * D + c, z);
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] += temperature * z[t];
        }
        if (center != nullptr) {
            float ce[8], sv[8];
            load8<T>(center + c, ce);
            load8<T>(scale + c, sv);
            if (cmin != nullptr) {   // clipping normalizer: clamp before scaling back
                float lo[8], hi[8];
                load8<T>(cmin + c, lo);
                load8<T>(cmax + c, hi);
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = fminf(fmaxf(v[t], lo[t]), hi[t]);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = v[t] * sv[t] + ce[t];
        }
        store8<float>(out + dst * D + c, v);
    }
}

// ---- GEMM helpers with profiling ---------------------------------------------------------------------
static GemmArgs gemm_args(lcmhip_session* s, const void* A, int lda, const Lin& W, const Lin* W2, int M, int epi) {
    GemmArgs a{};
    a.A = A; a.lda = lda;
    a.W = W.w; a.bias = W.b;
    a.W2 = W2 ? W2->w : nullptr; a.bias2 = W2 ? W2->b : nullptr;
    a.M = M; a.N = W.out; a.K = W.in;
    a.epilogue = epi; a.dtype = s->model->desc.dtype;
    a.max_splits = kMaxSplits;
    return a;
}

static void gemm_cost(const GemmArgs& a, double out_es, double* bytes, double* flops) {
    const double es = (double)dtype_size(a.dtype);
    const double nw = a.W2 ? 2.0 : 1.0;
    *bytes = nw * a.N * (double)a.K * es + (double)a.M * a.K * es + (double)(a.M + a.extra_rows) * a.N * out_es;
    *flops = 2.0 * nw * a.M * (double)a.N * a.K;
}

// Product with its epilogue applied (split-K through slab_q + a finishing pass when that pays).
static int gemm(lcmhip_session* s, const char* tag, const void* A, int lda, const Lin& W, const Lin* W2,
                const float* gate, void* out, int ldo, int M, int epi, hipStream_t st,
                const int32_t* out_rows = nullptr, int extra_rows = 0, const float* extra_const = nullptr,
                int out_head_cap = 0, const RowSteps* gate_steps = nullptr, int act = LCMHIP_ACT_RELU,
                const void* act_slope = nullptr) {
    GemmArgs a = gemm_args(s, A, lda, W, W2, M, epi);
    a.gate = gate;
    a.act = act;
    a.act_slope = act_slope;
    if (gate_steps && gate_steps->row_step) {
        a.gate_rows = gate_steps->row_step; a.gate_stride = gate_steps->step_stride; a.gate_steps = gate_steps->n_steps;
    }
    a.out = out; a.ldo = ldo; a.out_rows = out_rows; a.out_head_cap = out_head_cap;
    a.extra_rows = extra_rows; a.extra_const = extra_const;
    a.partial = s->slab_q; a.partial_bytes = s->slab_bytes;
    double bytes, flops;
    const double es = (double)dtype_size(a.dtype);
    gemm_cost(a, epi == LCMHIP_EPI_STORE_F32 ? 4.0 : (epi == LCMHIP_EPI_RESID ? 8.0 : es), &bytes, &flops);
    ProfScope ps(&s->prof, tag, bytes, flops, st);
    return launch_gemm(a, st);
}

// Two or three Linears over the same rows (q / k / v, or a layer's cross-attention k / v): one launch of a many-row
// kernel when the stacked weight rows fill the chip, otherwise one product each (see launch_gemm).
static int gemm_stack(lcmhip_session* s, const char* tag, const void* A, int lda, int n, const Lin* const* W,
                      void* const* outs, int ldo, int M, hipStream_t st, const int32_t* out_rows = nullptr,
                      int out_head_cap = 0) {
    GemmArgs a = gemm_args(s, A, lda, *W[0], nullptr, M, LCMHIP_EPI_STORE);
    a.out = outs[0]; a.ldo = ldo; a.out_rows = out_rows; a.out_head_cap = out_head_cap;
    a.partial = s->slab_q; a.partial_bytes = s->slab_bytes;
    bool same = s->stack_products;
    for (int g = 1; g < n; ++g) {
        same = same && W[g]->out == W[0]->out && W[g]->in == W[0]->in && (W[g]->b != nullptr) == (W[0]->b != nullptr);
        a.stack_W[g - 1] = W[g]->w; a.stack_bias[g - 1] = W[g]->b; a.stack_out[g - 1] = outs[g];
    }
    double bytes, flops;
    gemm_cost(a, (double)dtype_size(a.dtype), &bytes, &flops);
    if (same) {
        a.stack = n;
        ProfScope ps(&s->prof, tag, bytes * n, flops * n, st);
        return launch_gemm(a, st);
    }
    for (int g = 0; g < n; ++g) {
        GemmArgs one = gemm_args(s, A, lda, *W[g], nullptr, M, LCMHIP_EPI_STORE);
        one.out = outs[g]; one.ldo = ldo; one.out_rows = out_rows; one.out_head_cap = out_head_cap;
        one.partial = s->slab_q; one.partial_bytes = s->slab_bytes;
        ProfScope ps(&s->prof, tag, bytes, flops, st);
        LCM_TRY(launch_gemm(one, st));
    }
    return LCMHIP_OK;
}

// Self-attention q / k / v of many rows with the rotation and the cache scatter in the product's epilogue: q lands
// rotated in `q_out`, k (rotated) and v directly in the layer's cache rows.  Returns false (nothing launched) when the
// product would not run as one launch of a many-row kernel; the caller then takes the separate passes.
static bool qkv_rope_store_fused(lcmhip_session* s, const char* tag, const void* A, int lda, const AttnW& w, void* q_out,
                                 void* k_cache, void* v_cache, int M, int cap, bool rope, hipStream_t st, int* rc) {
    const Lin* const W[3] = {&w.q, &w.k, &w.v};
    GemmArgs a = gemm_args(s, A, lda, *W[0], nullptr, M, LCMHIP_EPI_STORE);
    if (!s->fuse_rope_store || !s->stack_products || a.dtype != LCMHIP_BF16) return false;
    a.out = q_out; a.ldo = a.N; a.out_rows = s->self_rows; a.out_head_cap = cap;
    a.partial = s->slab_q; a.partial_bytes = s->slab_bytes;
    void* const outs[3] = {q_out, k_cache, v_cache};
    for (int g = 1; g < 3; ++g) {
        if (W[g]->out != W[0]->out || W[g]->in != W[0]->in || (W[g]->b != nullptr) != (W[0]->b != nullptr)) return false;
        a.stack_W[g - 1] = W[g]->w; a.stack_bias[g - 1] = W[g]->b; a.stack_out[g - 1] = outs[g];
    }
    a.stack = 3;
    a.scatter_mask = 0b110;
    a.rope_mask = rope ? 0b011 : 0;
    a.rope_pos = s->pos_rows; a.rope_cos = s->rope_cos; a.rope_sin = s->rope_sin;
    if (a.N % 128 != 0 || !gemm_stack_is_fused(a)) return false;
    double bytes, flops;
    gemm_cost(a, (double)dtype_size(a.dtype), &bytes, &flops);
    ProfScope ps(&s->prof, tag, bytes * 3, flops * 3, st);
    // Wave quantisation: 4096 rows x 3 x 2048 outputs are 768 tiles of 256 x 128 = three full waves of the chip, while q
    // on its own (256 such tiles, one wave) + k and v stacked (256 tiles of 256 x 256, one wave at 1.5 x the time) are
    // 2.5: two launches when the estimate says so (same kernels, same K order per output: same bits).  Base-LCM forward
    // 32 x 128, alternating in one process: 14.19 -> 14.02 ms.
    GemmArgs aq = a;
    aq.stack = 0; aq.out_rows = nullptr; aq.out_head_cap = 0; aq.scatter_mask = 0; aq.rope_mask = rope ? 1 : 0;
    GemmArgs akv = a;
    akv.W = W[1]->w; akv.bias = W[1]->b; akv.out = k_cache;
    akv.stack = 2; akv.stack_W[0] = W[2]->w; akv.stack_bias[0] = W[2]->b; akv.stack_out[0] = v_cache;
    akv.stack_W[1] = nullptr; akv.stack_bias[1] = nullptr; akv.stack_out[1] = nullptr;
    akv.scatter_mask = 0b11; akv.rope_mask = rope ? 0b01 : 0;
    if (gemm_big_applicable(aq, false) && gemm_stack_is_fused(akv) &&
        gemm_big_wave_cost(aq, LCMHIP_EPI_STORE, M) + gemm_big_wave_cost(akv, LCMHIP_EPI_STORE, M) + 0.15 <
            gemm_big_wave_cost(a, LCMHIP_EPI_STORE, M)) {
        *rc = launch_gemm(aq, st);
        if (*rc == LCMHIP_OK) *rc = launch_gemm(akv, st);
        return true;
    }
    *rc = launch_gemm(a, st);
    return true;
}

// Product left in slab_p for the next norm kernel to fold into the residual stream.
static int gemm_pending(lcmhip_session* s, const char* tag, const void* A, int lda, const Lin& W, const float* gate,
                        int M, Pending* pend, hipStream_t st, float* slab = nullptr) {
    GemmArgs a = gemm_args(s, A, lda, W, nullptr, M, LCMHIP_EPI_STORE_F32);
    a.bias = nullptr;
    if (slab == nullptr) slab = s->slab_p;
    a.partial = slab; a.partial_bytes = s->slab_bytes;
    double bytes, flops;
    gemm_cost(a, 0.0, &bytes, &flops);
    int splits = 0;
    {
        ProfScope ps(&s->prof, tag, bytes, flops, st);
        LCM_TRY(launch_gemm_partial(a, &splits, st));
    }
    *pend = Pending{};
    pend->partial = slab;
    pend->splits = splits;
    pend->M = M;
    pend->bias = W.b;
    pend->gate = gate;
    return LCMHIP_OK;
}

static int norm(lcmhip_session* s, const char* tag, float* x, const NormW& w, const float* shift,
                const float* scale, void* y, int rows_total, int rows_norm, int kind, const Pending* pend,
                hipStream_t st, const RowSteps* steps = nullptr) {
    const lcmhip_model_desc& d = s->model->desc;
    const double es = (double)dtype_size(d.dtype);
    double bytes = rows_norm * (double)d.model_dim * (4.0 + es);
    if (pend) bytes += (double)rows_total * d.model_dim * 4.0 * (1.0 + pend->splits);
    ProfScope ps(&s->prof, tag, bytes, 0.0, st);
    return launch_norm_modulate(x, w.w != nullptr ? w.w : s->ones, w.b, shift, scale, y, rows_total, rows_norm, d.model_dim,
                                kernel_norm_kind(kind), d.norm_eps, d.dtype, pend, st, steps);
}

// q / k LayerNorm of an attention block, in place on projected rows (either pointer may be null).
static int qk_norm(lcmhip_session* s, const char* tag, const lcmhip_tower_desc& t, const AttnW& a, void* q, void* k, int rows,
                   hipStream_t st) {
    const lcmhip_model_desc& d = s->model->desc;
    const NormW* n0 = q ? &a.qn : &a.kn;
    void* x0 = q ? q : k;
    void* x1 = q ? k : nullptr;
    ProfScope ps(&s->prof, tag, (q && k ? 4.0 : 2.0) * rows * d.model_dim * (double)dtype_size(d.dtype), 0.0, st);
    return launch_row_norm(x0, n0->w ? n0->w : s->ones, n0->b, x1, a.kn.w ? a.kn.w : s->ones, a.kn.b, rows, d.model_dim,
                           kernel_norm_kind(t.norm_kind), d.norm_eps, d.dtype, st);
}

static int ffn_in(lcmhip_session* s, const char* tag, const lcmhip_tower_desc& t, const FfnW& f, const void* a, int rows,
                  hipStream_t st) {
    const int d = s->model->desc.model_dim;
    if (t.ffn_kind == LCMHIP_FFN_SWIGLU)
        return gemm(s, tag, a, d, f.gate, &f.inner, nullptr, s->act_h, t.ffn_dim, rows, LCMHIP_EPI_SWIGLU, st);
    return gemm(s, tag, a, d, f.inner, nullptr, nullptr, s->act_h, t.ffn_dim, rows, LCMHIP_EPI_RELU, st, nullptr, 0, nullptr, 0,
                nullptr, t.ffn_activation, f.act_slope);
}

}  // namespace lcmhip

using namespace lcmhip;

// ================================================================================================ C ABI
extern "C" {

const char* lcmhip_last_error(void) { return g_last_error.c_str(); }
int lcmhip_version(void) { return LCMHIP_ABI_VERSION; }

int lcmhip_model_create(const lcmhip_model_desc* desc, lcmhip_model** out) {
    LCM_REQUIRE(desc && out, "model_create: null argument");
    LCM_REQUIRE(desc->dtype == LCMHIP_F32 || desc->dtype == LCMHIP_BF16, "model_create: dtype must be f32 or bf16");
    LCM_REQUIRE(desc->sonar_dim % 128 == 0 && desc->sonar_dim <= 4096, "model_create: sonar_dim");
    // the frontend and the initial projection stream their K = sonar_dim operands in whole 512-byte chunks
    LCM_REQUIRE(desc->dtype != LCMHIP_BF16 || desc->sonar_dim % 256 == 0, "model_create: sonar_dim must be a multiple of 256 in bf16");
    LCM_REQUIRE(desc->model_dim % 256 == 0 && desc->model_dim <= 8192, "model_create: model_dim");
    LCM_REQUIRE(desc->timestep_freq_dim > 0 && desc->timestep_freq_dim % (desc->dtype == LCMHIP_BF16 ? 256 : 128) == 0,
                "model_create: timestep_freq_dim must be a multiple of 128 (fp32) / 256 (bf16)");
    LCM_TRY(check_tower(desc->context, desc->model_dim, desc->dtype, "context"));
    const bool base = desc->denoiser.num_layers == 0;   // base LCM: no denoiser tower
    if (!base) {
        LCM_TRY(check_tower(desc->denoiser, desc->model_dim, desc->dtype, "denoiser"));
        if (desc->denoiser.pos_kind != LCMHIP_POS_NONE && desc->denoiser.pos_kind != LCMHIP_POS_LEARNED)
            return fail(LCMHIP_ERR_UNSUPPORTED, "denoiser: only learned positions (or none) are implemented for the latents");
        // lcm_denoiser.py:441-443 asserts against post-norm; the normformer extras and a residual scale (which
        // LCMDenoiserTransformerFactory.build_layer, denoisers/factory.py:181-191, never forwards) are not offered
        if (desc->denoiser.norm_order != LCMHIP_ORDER_PRE || desc->denoiser.no_final_norm || desc->denoiser.scale_residual)
            return fail(LCMHIP_ERR_UNSUPPORTED, "denoiser: pre-norm layers with a final norm and no residual scale only");
    }
    LCM_REQUIRE(desc->embed_scale >= 0.f, "model_create: embed_scale");
    LCM_REQUIRE(desc->postnet_activation >= 0 && desc->postnet_activation <= 1 + LCMHIP_ACT_HARDSIGMOID && (base || desc->postnet_activation == 0),
                "model_create: postnet_activation is 0 or 1 + LCMHIP_ACT_* (base LCM only)");
    {
        // the library's set-up call: resolves the static scratch of lcmhip_gemm_static on the current device (a model
        // table can also be built and validated on a host without a GPU: nothing to resolve there)
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0) LCM_TRY(static_scratch_init());
        else (void)hipGetLastError();
    }
    auto* m = new lcmhip_model();
    m->desc = *desc;
    m->base = base;
    if (m->desc.embed_scale == 0.f) m->desc.embed_scale = 1.f;
    *out = m;
    return LCMHIP_OK;
}

int lcmhip_model_bind_weight(lcmhip_model* m, const char* name, const void* ptr, int32_t dtype, int32_t ndim,
                             const int64_t* shape) {
    LCM_REQUIRE(m && name && ptr && shape && ndim >= 1 && ndim <= 4, "bind_weight: bad argument");
    LCM_REQUIRE(dtype == m->desc.dtype, std::string("bind_weight: dtype of ") + name + " differs from the model dtype");
    LCM_REQUIRE(((uintptr_t)ptr % 16) == 0, std::string("bind_weight: ") + name + " is not 16-byte aligned");
    Tensor t;
    t.ptr = ptr;
    t.shape.assign(shape, shape + ndim);
    m->tensors[name] = t;
    m->finalized = false;
    return LCMHIP_OK;
}

int lcmhip_model_finalize(lcmhip_model* m) {
    LCM_REQUIRE(m, "finalize: null model");
    const lcmhip_model_desc& d = m->desc;
    const int dm = d.model_dim, D = d.sonar_dim;
    // tensor names: two-tower = lcm/models/two_tower_diffusion_lcm/builder.py:104-121, base = base_lcm/builder.py:84-88
    const std::string fe = m->base ? "frontend" : "encoder_frontend";
    const std::string dec = m->base ? "lcm" : "context_encoder";
    const std::string nrm = m->base ? "frontend.sonar_normalizer" : "sonar_normalizer";
    m->center = m->scale = nullptr;
    if (!m->base || has_tensor(m, nrm + ".center")) {   // a base LCM may have no normalizer (abstract_lcm/builder.py:92-101)
        LCM_TRY(get_vec(m, nrm + ".center", D, &m->center));
        LCM_TRY(get_vec(m, nrm + ".scale", D, &m->scale));
        m->clip_min = m->clip_max = nullptr;
        if (has_tensor(m, nrm + ".clip_min") || has_tensor(m, nrm + ".clip_max")) {
            LCM_TRY(get_vec(m, nrm + ".clip_min", D, &m->clip_min));
            LCM_TRY(get_vec(m, nrm + ".clip_max", D, &m->clip_max));
        }
    }
    LCM_TRY(get_lin(m, fe + ".pre_linear", dm, D, !m->base || has_tensor(m, fe + ".pre_linear.bias"), &m->pre_linear));
    m->pos_table = nullptr;
    if (d.context.pos_kind == LCMHIP_POS_LEARNED) {
        auto t = m->tensors.find(fe + ".pos_encoder.weight");
        if (t == m->tensors.end()) return fail(LCMHIP_ERR_STATE, "missing weight: " + fe + ".pos_encoder.weight");
        LCM_REQUIRE(t->second.shape.size() == 2 && t->second.shape[0] >= d.max_seq_len && t->second.shape[1] == dm,
                    "bad shape for " + fe + ".pos_encoder.weight");
        m->pos_table = t->second.ptr;
    }
    m->ctx.assign(d.context.num_layers, CtxLayer{});
    for (int i = 0; i < d.context.num_layers; ++i) {
        const std::string p = dec + ".layers." + std::to_string(i);
        CtxLayer& L = m->ctx[i];
        LCM_TRY(get_norm(m, p + ".self_attn_layer_norm", dm, d.context.norm_kind, &L.sa_norm));
        LCM_TRY(get_norm(m, p + ".ffn_layer_norm", dm, d.context.norm_kind, &L.ffn_norm));
        LCM_TRY(get_attn(m, p + ".self_attn", dm, d.context, &L.sa));
        LCM_TRY(get_ffn(m, p + ".ffn", dm, d.context, &L.ffn));
        if (d.context.norm_order == LCMHIP_ORDER_NORMFORMER) {
            LCM_TRY(get_norm(m, p + ".self_attn_norm", dm, d.context.norm_kind, &L.attn_norm));
            // the two-matrix FFN's own LayerNorm is fairseq2's default one (weight + bias) whatever the tower's style:
            // TransformerFactory.build_ffn (factory.py:292-300) passes the order but no layer-norm factory
            if (d.context.ffn_kind == LCMHIP_FFN_RELU)
                LCM_TRY(get_norm(m, p + ".ffn.inner_layer_norm", d.context.ffn_dim, LCMHIP_NORM_LAYERNORM, &L.inner_norm));
        }
        if (d.context.scale_residual) LCM_TRY(get_vec(m, p + ".residual_scale", dm, &L.residual_scale));
    }
    m->ctx_final = NormW{};
    if (!d.context.no_final_norm) LCM_TRY(get_norm(m, dec + ".layer_norm", dm, d.context.norm_kind, &m->ctx_final));
    m->den.clear();
    if (m->base) {
        LCM_TRY(get_lin(m, "postnet.fc", D, dm, has_tensor(m, "postnet.fc.bias"), &m->postnet));
        m->finalized = true;
        return LCMHIP_OK;
    }
    LCM_TRY(get_lin(m, "denoiser.embed_time.fc1", dm, d.timestep_freq_dim, true, &m->t_fc1));
    LCM_TRY(get_lin(m, "denoiser.embed_time.fc2", dm, dm, true, &m->t_fc2));
    LCM_TRY(get_lin(m, "denoiser.initial_proj.fc", dm, D, true, &m->init_proj));
    LCM_TRY(get_lin(m, "denoiser.final_proj.fc", D, dm, true, &m->final_proj));
    m->den_pos_table = nullptr;
    if (d.denoiser.pos_kind == LCMHIP_POS_LEARNED) {   // factory.py:187-209 builds it, lcm_denoiser.py:211-212 adds it
        auto t = m->tensors.find("denoiser.pos_encoder.weight");
        if (t == m->tensors.end()) return fail(LCMHIP_ERR_STATE, "missing weight: denoiser.pos_encoder.weight");
        LCM_REQUIRE(t->second.shape.size() == 2 && t->second.shape[0] >= d.max_seq_len && t->second.shape[1] == dm,
                    "bad shape for denoiser.pos_encoder.weight");
        m->den_pos_table = t->second.ptr;
    }
    m->den.assign(d.denoiser.num_layers, DenLayer{});
    for (int i = 0; i < d.denoiser.num_layers; ++i) {
        const std::string p = "denoiser.layers." + std::to_string(i);
        DenLayer& L = m->den[i];
        LCM_TRY(get_norm(m, p + ".self_attn_layer_norm", dm, d.denoiser.norm_kind, &L.sa_norm));
        LCM_TRY(get_norm(m, p + ".cross_attention_layer_norm", dm, d.denoiser.norm_kind, &L.ca_norm));
        LCM_TRY(get_norm(m, p + ".ffn_layer_norm", dm, d.denoiser.norm_kind, &L.ffn_norm));
        LCM_TRY(get_attn(m, p + ".self_attn", dm, d.denoiser, &L.sa));
        LCM_TRY(get_attn(m, p + ".cross_attention", dm, d.denoiser, &L.ca));
        LCM_TRY(get_ffn(m, p + ".ffn", dm, d.denoiser, &L.ffn));
        LCM_TRY(get_lin(m, p + ".modulator.fc", 9 * dm, dm, true, &L.mod));
        // optional derived weight: self-attention over one position is output_proj(v_proj(h)) = h W_ov^T + b_ov
        L.ov = Lin{};
        if (m->tensors.count(p + ".self_attn.ov_proj.weight")) LCM_TRY(get_lin(m, p + ".self_attn.ov_proj", dm, dm, true, &L.ov));
    }
    LCM_TRY(get_norm(m, "denoiser.layer_norm", dm, d.denoiser.norm_kind, &m->den_final));
    m->finalized = true;
    return LCMHIP_OK;
}

int lcmhip_model_destroy(lcmhip_model* m) {
    delete m;
    return LCMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ session life cycle
static int session_dims(lcmhip_session* s, const lcmhip_model* m, int max_batch, int max_ctx, int max_steps, uint32_t flags) {
    constexpr uint32_t known = LCMHIP_SESSION_KEEP_CONTEXT | LCMHIP_SESSION_SEPARATE_QKV | LCMHIP_SESSION_SEPARATE_ROPE |
                               LCMHIP_SESSION_NO_GRAPH;
    LCM_REQUIRE((flags & ~known) == 0, "session: unknown flag bits");
    LCM_REQUIRE(m && m->finalized, "session: model is not finalized");
    LCM_REQUIRE(max_batch >= 1 && max_batch <= 65536, "session: max_batch");
    LCM_REQUIRE(max_ctx >= 1 && max_ctx <= m->desc.max_seq_len, "session: max_ctx exceeds the model's max_seq_len");
    LCM_REQUIRE(max_ctx <= 3800, "session: the attention kernel supports up to 3800 context positions");
    LCM_REQUIRE(max_steps >= 1 && max_steps <= 4096, "session: max_steps");
    s->model = const_cast<lcmhip_model*>(m);
    s->max_batch = max_batch;
    s->cap = max_ctx;
    s->max_steps = max_steps;
    // rows of one contextualizer pass: large enough for matrix-core-bound GEMMs when a prompt is long
    s->chunk_rows = std::max(2 * max_batch, (int)std::min<int64_t>(4096, std::max<int64_t>(512, (int64_t)max_batch * max_ctx)));
    s->rows_cap = std::max(s->chunk_rows, max_steps);
    s->keep_context = (flags & LCMHIP_SESSION_KEEP_CONTEXT) != 0;
    s->stack_products = (flags & LCMHIP_SESSION_SEPARATE_QKV) == 0;
    s->fuse_rope_store = (flags & LCMHIP_SESSION_SEPARATE_ROPE) == 0;
    s->use_graphs = (flags & LCMHIP_SESSION_NO_GRAPH) == 0;
    return LCMHIP_OK;
}

int lcmhip_session_workspace_bytes(const lcmhip_model* m, int32_t max_batch, int32_t max_ctx, int32_t max_steps,
                                   uint32_t flags, size_t* bytes) {
    LCM_REQUIRE(bytes, "workspace_bytes: null output");
    lcmhip_session tmp;
    LCM_TRY(session_dims(&tmp, m, max_batch, max_ctx, max_steps, flags));
    *bytes = carve(&tmp, nullptr);
    return LCMHIP_OK;
}

// Host -> device copy of set-up tables, complete on return.  It goes through a non-blocking stream of its own and not
// through the legacy stream: a copy on the legacy stream waits for every blocking stream of the process and is refused
// with an error while another thread records a graph on one.
// The stream is created once per device and kept: uploads are rare and serialised.
static int upload_now(void* dst, const void* src, size_t bytes) {
    constexpr int kMaxDevices = 64;
    static std::mutex mu;
    static hipStream_t streams[kMaxDevices] = {};
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    LCM_HIP_CHECK(hipGetDevice(&dev));
    LCM_REQUIRE(dev >= 0 && dev < kMaxDevices, "set-up table upload: device index out of range");
    if (streams[dev] == nullptr) LCM_HIP_CHECK(hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking));
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, streams[dev]);
    if (e == hipSuccess) e = hipStreamSynchronize(streams[dev]);
    if (e != hipSuccess) return fail(LCMHIP_ERR_HIP, std::string("set-up table upload: ") + hipGetErrorString(e));
    return LCMHIP_OK;
}

int lcmhip_session_create(lcmhip_model* m, int32_t max_batch, int32_t max_ctx, int32_t max_steps, uint32_t flags,
                          void* workspace, size_t bytes, lcmhip_session** out) {
    LCM_REQUIRE(out && workspace, "session_create: null argument");
    LCM_REQUIRE(((uintptr_t)workspace % 256) == 0, "session_create: workspace must be 256-byte aligned");
    LCM_TRY(static_scratch_init());     // never inside a capture: see lcmhip_gemm_static
    auto s = std::make_unique<lcmhip_session>();
    LCM_TRY(session_dims(s.get(), m, max_batch, max_ctx, max_steps, flags));
    const size_t need = carve(s.get(), static_cast<char*>(workspace));
    LCM_REQUIRE(bytes >= need, "session_create: workspace too small");
    {
        std::vector<float> one(m->desc.model_dim, 1.0f);
        std::vector<uint16_t> one16(m->desc.model_dim, (uint16_t)0x3f80);   // 1.0 in bf16
        const void* src = m->desc.dtype == LCMHIP_BF16 ? (const void*)one16.data() : (const void*)one.data();
        LCM_TRY(upload_now(s->ones, src, (size_t)m->desc.model_dim * dtype_size(m->desc.dtype)));
    }
    // RoPE angle table, computed the way fairseq2's RotaryEncoder does (fp32 frequencies and angles).
    if (m->desc.context.pos_kind == LCMHIP_POS_ROPE) {
        std::vector<float> c((size_t)s->cap * 64), sn((size_t)s->cap * 64);
        for (int i = 0; i < 64; ++i) {
            const float freq = 1.0f / powf(m->desc.context.rope_theta, (float)(2 * i) / 128.0f);
            for (int p = 0; p < s->cap; ++p) {
                const float ang = (float)p * freq;
                c[(size_t)p * 64 + i] = (float)cos((double)ang);
                sn[(size_t)p * 64 + i] = (float)sin((double)ang);
            }
        }
        LCM_TRY(upload_now(s->rope_cos, c.data(), c.size() * 4));
        LCM_TRY(upload_now(s->rope_sin, sn.data(), sn.size() * 4));
    }
    *out = s.release();
    return LCMHIP_OK;
}

// A recorded loop may still be executing when its session is destroyed, re-scheduled or asked for a seventeenth batch
// size: the executable graphs (which own the kernel arguments of their nodes) are destroyed only after the stream the
// last one was launched on has drained.
static void drop_graphs(lcmhip_session* s) {
    if (s->graph_launched && !s->graphs.empty()) {
        if (hipStreamSynchronize(s->graph_stream) != hipSuccess) (void)hipGetLastError();   // a stream the caller has destroyed since
        s->graph_launched = false;
    }
    for (auto& kv : s->graphs) (void)hipGraphExecDestroy(kv.second);
    s->graphs.clear();
}

int lcmhip_session_destroy(lcmhip_session* s) {
    if (s) drop_graphs(s);
    delete s;
    return LCMHIP_OK;
}

int lcmhip_session_context_len(const lcmhip_session* s) { return s ? s->ctx_len : -1; }
int lcmhip_session_num_active(const lcmhip_session* s) { return s ? s->n_active : -1; }

int lcmhip_session_reset(lcmhip_session* s, int32_t n_docs, void* stream) {
    LCM_REQUIRE(s, "reset: null session");
    LCM_REQUIRE(n_docs >= 1 && n_docs <= s->max_batch, "reset: n_docs exceeds the session's max_batch");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const lcmhip_model* m = s->model;
    const lcmhip_model_desc& d = m->desc;
    s->n_active = n_docs;
    s->ctx_len = 0;
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, s->d_cross_keys, 1);
    s->h_slot.resize(n_docs);
    for (int i = 0; i < n_docs; ++i) s->h_slot[i] = i;
    hipLaunchKernelGGL(k_slot0_rows, dim3((n_docs + 255) / 256), dim3(256), 0, st, s->slot0_rows, s->slot_map, n_docs,
                       s->cap + 1);
    LCM_HIP_CHECK(hipGetLastError());
    // Cross-attention key/value of the prepended zero vector = the k_proj / v_proj biases
    // (lcm_denoiser.py:193-199): position 0 of every document, every layer.
    const size_t es = dtype_size(d.dtype);
    const size_t layer_bytes = (size_t)s->max_batch * (s->cap + 1) * d.model_dim * es;
    for (int l = 0; l < d.denoiser.num_layers; ++l) {
        const void* k0 = m->den[l].ca.k.b;
        if (d.denoiser.qk_layernorm) {   // the zero vector's key is k_layer_norm(k_proj(0)) = k_layer_norm(b_k)
            LCM_HIP_CHECK(hipMemcpyAsync(s->act_k, k0, (size_t)d.model_dim * es, hipMemcpyDeviceToDevice, st));
            LCM_TRY(qk_norm(s, "ctx.qk_norm", d.denoiser, m->den[l].ca, nullptr, s->act_k, 1, st));
            k0 = s->act_k;
        }
        LCM_TRY(launch_fill_rows(s->cross_k + l * layer_bytes, k0, s->slot0_rows, n_docs, d.model_dim,
                                 s->cap + 1, d.dtype, st));
        LCM_TRY(launch_fill_rows(s->cross_v + l * layer_bytes, m->den[l].ca.v.b, s->slot0_rows, n_docs, d.model_dim,
                                 s->cap + 1, d.dtype, st));
    }
    return LCMHIP_OK;
}

int lcmhip_session_reorder(lcmhip_session* s, const int32_t* order, int32_t n, void* stream) {
    LCM_REQUIRE(s && order, "reorder: null argument");
    LCM_REQUIRE(n >= 1 && n <= s->n_active, "reorder: bad row count");
    std::vector<int32_t> ns(n);
    for (int i = 0; i < n; ++i) {
        LCM_REQUIRE(order[i] >= 0 && order[i] < s->n_active, "reorder: index out of range");
        LCM_REQUIRE(i == 0 || order[i] > order[i - 1], "reorder: indices must be strictly increasing");
        ns[i] = s->h_slot[order[i]];
    }
    s->h_slot = ns;
    s->n_active = n;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // pageable source: the runtime stages the copy before returning, so `ns` may go out of scope
    LCM_HIP_CHECK(hipMemcpyAsync(s->slot_map, s->h_slot.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    return LCMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ schedule
int lcmhip_session_set_schedule(lcmhip_session* s, const int32_t* timesteps, int32_t n_steps,
                                const float* alphas_cumprod, int32_t num_train_steps, float final_alpha_cumprod,
                                int32_t step_ratio, const lcmhip_denoise_params* params, void* stream) {
    LCM_REQUIRE(s && timesteps && alphas_cumprod && params, "set_schedule: null argument");
    LCM_REQUIRE(!s->model->base, "set_schedule: a base LCM has no diffusion schedule");
    // ---- validate everything before the session is touched: a rejected call leaves the previous schedule intact
    LCM_REQUIRE(n_steps >= 1 && n_steps <= s->max_steps, "set_schedule: more steps than the session was sized for");
    LCM_REQUIRE(num_train_steps >= 1 && step_ratio >= 0, "set_schedule: bad train-step count or step ratio");
    LCM_REQUIRE(params->ddim_eta >= 0.f, "set_schedule: ddim_eta must not be negative");
    LCM_REQUIRE(params->prediction_type >= 0 && params->prediction_type <= 2, "set_schedule: prediction_type");
    LCM_REQUIRE(!params->thresholding || (params->dynamic_thresholding_ratio >= 0.f && params->dynamic_thresholding_ratio <= 1.f &&
                                          params->sample_max_value >= 1.f),
                "set_schedule: thresholding needs a ratio in [0, 1] and sample_max_value >= 1");
    for (int i = 0; i < n_steps; ++i)
        LCM_REQUIRE(timesteps[i] >= 0 && timesteps[i] < num_train_steps, "set_schedule: timestep out of range");
    // ---- from here on the old schedule is gone; have_schedule is set again only when the new one is complete
    s->have_schedule = false;
    drop_graphs(s);   // recorded loops hold the old schedule's scalars
    hipStream_t st = static_cast<hipStream_t>(stream);
    lcmhip_model* m = s->model;
    const lcmhip_model_desc& d = m->desc;
    const int dm = d.model_dim, FD = d.timestep_freq_dim, L = d.denoiser.num_layers;
    s->params = *params;
    s->do_cfg = params->guidance_scale != 1.0f;
    s->n_steps = n_steps;
    s->timesteps.assign(timesteps, timesteps + n_steps);
    s->scalars.assign(n_steps, DdimScalars{});
    s->step_mode_rows.assign(n_steps, 0);
    for (int i = 0; i < n_steps; ++i) {
        const int t = timesteps[i];
        const int prev = t - step_ratio;                                   // ddim.py:421-423
        const float a_t = alphas_cumprod[t];
        const float a_prev = prev >= 0 ? alphas_cumprod[prev] : final_alpha_cumprod;   // ddim.py:426-431
        const float b_t = 1.0f - a_t;
        DdimScalars& sc = s->scalars[i];
        sc.sqrt_a_t = sqrtf(a_t);
        sc.sqrt_b_t = sqrtf(b_t);
        sc.sqrt_a_prev = sqrtf(a_prev);
        // sigma_t = eta * sqrt(variance_t), variance_t = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)  (ddim.py:209-222, 473-485)
        sc.sigma = 0.f;
        if (params->ddim_eta > 0.f) {
            const float variance = ((1.0f - a_prev) / (1.0f - a_t)) * (1.0f - a_t / a_prev);
            sc.sigma = params->ddim_eta * sqrtf(variance);
        }
        sc.dir_coef = sqrtf(1.0f - a_prev - sc.sigma * sc.sigma);
        sc.guidance_scale = params->guidance_scale;
        sc.guidance_rescale = params->guidance_rescale;
        sc.inv_eps_scaling = params->epsilon_scaling > 0.f ? 1.0f / params->epsilon_scaling : 1.0f;
        sc.clip_noise = params->clip_noise;
        sc.clip_sample_range = params->clip_sample ? params->clip_sample_range : -1.0f;
        sc.thr_ratio = params->thresholding ? params->dynamic_thresholding_ratio : -1.0f;
        sc.thr_max = params->sample_max_value;
        sc.prediction_type = params->prediction_type;
        // builder.py:385-387, 404-449: no guidance on the step whose timestep is T - 1
        const bool is_max = (t == num_train_steps - 1);
        sc.mode = (s->do_cfg && !is_max) ? 1 : 0;
        s->step_mode_rows[i] = sc.mode;
    }
    // ---- timestep embeddings -> adaLN table (timestep_encoder.py:72-122, lcm_denoiser.py:60-64)
    std::vector<float> tf((size_t)n_steps * FD);
    const int half = FD / 2;
    for (int i = 0; i < n_steps; ++i)
        for (int k = 0; k < half; ++k) {
            const float freq = expf(-logf(10000.0f) * (float)k / (float)half);
            const float arg = (float)timesteps[i] * freq;
            tf[(size_t)i * FD + k] = (float)cos((double)arg);
            tf[(size_t)i * FD + half + k] = (float)sin((double)arg);
        }
    LCM_HIP_CHECK(hipMemcpyAsync(s->f32tmp, tf.data(), tf.size() * 4, hipMemcpyHostToDevice, st));
    LCM_TRY(launch_cast(s->f32tmp, s->act_in, (int64_t)n_steps * FD, d.dtype, st));
    LCM_TRY(gemm(s, "setup", s->act_in, FD, m->t_fc1, nullptr, nullptr, s->f32tmp, dm, n_steps, LCMHIP_EPI_STORE_F32, st));
    LCM_TRY(launch_silu_cast(s->f32tmp, s->act_a, (int64_t)n_steps * dm, d.dtype, st));
    LCM_TRY(gemm(s, "setup", s->act_a, dm, m->t_fc2, nullptr, nullptr, s->f32tmp, dm, n_steps, LCMHIP_EPI_STORE_F32, st));
    // the reference keeps the embedding in the model dtype before the modulator's SiLU
    LCM_TRY(launch_cast(s->f32tmp, s->act_q, (int64_t)n_steps * dm, d.dtype, st));
    LCM_TRY(launch_copy_rows_to_f32(s->act_q, s->f32tmp, (int64_t)n_steps * dm, d.dtype, st));
    LCM_TRY(launch_silu_cast(s->f32tmp, s->act_a, (int64_t)n_steps * dm, d.dtype, st));
    for (int l = 0; l < L; ++l)
        LCM_TRY(gemm(s, "setup", s->act_a, dm, m->den[l].mod, nullptr, nullptr, s->mod_table + (size_t)l * 9 * dm,
                     L * 9 * dm, n_steps, LCMHIP_EPI_STORE_F32, st));
    // ---- unconditional rows: cross-attention output = output_proj(v_proj(0)) = W_o b_v + b_o
    for (int l = 0; l < L; ++l)
        LCM_TRY(gemm(s, "setup", m->den[l].ca.v.b, dm, m->den[l].ca.o, nullptr, nullptr, s->uconst + (size_t)l * dm, dm,
                     1, LCMHIP_EPI_STORE_F32, st));
    // ---- learned positions of the latents: the sampling loop's single latent sits at position 0
    if (m->den_pos_table) {
        if (d.dtype == LCMHIP_BF16)
            hipLaunchKernelGGL((k_vec_add<bf16_t>), dim3((dm + 255) / 256), dim3(256), 0, st, (const bf16_t*)m->init_proj.b,
                               (const bf16_t*)m->den_pos_table, (bf16_t*)s->init_bias_pos, dm);
        else
            hipLaunchKernelGGL((k_vec_add<float>), dim3((dm + 255) / 256), dim3(256), 0, st, (const float*)m->init_proj.b,
                               (const float*)m->den_pos_table, (float*)s->init_bias_pos, dm);
        LCM_HIP_CHECK(hipGetLastError());
    }
    LCM_HIP_CHECK(hipStreamSynchronize(st));
    s->have_schedule = true;
    return LCMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ contextualizer
static int append_impl(lcmhip_session* s, const float* x, int32_t n_tokens, int32_t normalized, void* stream,
                       const BaseOut* bo) {
    LCM_REQUIRE(s && x, "context_append: null argument");
    LCM_REQUIRE(s->n_active >= 1, "context_append: call session_reset first");
    LCM_REQUIRE(n_tokens >= 1 && s->ctx_len + n_tokens <= s->cap, "context_append: context exceeds the session's max_ctx");
    hipStream_t st = static_cast<hipStream_t>(stream);
    lcmhip_model* m = s->model;
    const lcmhip_model_desc& d = m->desc;
    const lcmhip_tower_desc& tc = d.context;
    const int dm = d.model_dim, D = d.sonar_dim;
    const size_t es = dtype_size(d.dtype);
    const size_t self_layer = (size_t)s->max_batch * s->cap * dm * es;
    const size_t cross_layer = (size_t)s->max_batch * (s->cap + 1) * dm * es;
    // One pass handles at most chunk_rows rows.  Long appends (prompt prefill) are cut by DOCUMENT first, so that a
    // pass sees long causal query blocks (matrix-core attention, large GEMMs); only a single document longer than
    // a pass is cut by position.  Short appends for many documents (generation steps) go in one pass.
    const int n_total = s->n_active;
    int doc_group = n_total;
    if ((int64_t)n_total * n_tokens > s->chunk_rows && n_tokens >= 16) doc_group = std::max(1, s->chunk_rows / n_tokens);
    const int n_out_pos = (bo && bo->all) ? n_tokens : 1;

    for (int d0 = 0; d0 < n_total; d0 += doc_group) {
        const int n = std::min(doc_group, n_total - d0);          // documents of this pass
        const int per_chunk = std::max(1, s->chunk_rows / n);
        const float* xg = x + (size_t)d0 * n_tokens * D;
        const int32_t* slots = s->slot_map + d0;
        for (int j0 = 0; j0 < n_tokens; j0 += per_chunk) {
            const int sc = std::min(per_chunk, n_tokens - j0);
            const int R = n * sc;
            const int pos0 = s->ctx_len + j0;
            hipLaunchKernelGGL(k_make_rows, dim3((R + 255) / 256), dim3(256), 0, st, slots, s->pos_rows, s->self_rows,
                               s->cross_rows, R, sc, pos0, s->cap);
            LCM_HIP_CHECK(hipGetLastError());
            {
                ProfScope ps(&s->prof, "ctx.normalize", R * (double)D * (4.0 + es), 0.0, st);
                if (d.dtype == LCMHIP_BF16)
                    hipLaunchKernelGGL((k_normalize_gather<bf16_t>), dim3(R), dim3(128), 0, st, xg, (const bf16_t*)m->center,
                                       (const bf16_t*)m->scale, (bf16_t*)s->act_in, D, sc, n_tokens, j0, (normalized || !m->center) ? 0 : 1, d.embed_scale,
                                       (const bf16_t*)m->clip_min, (const bf16_t*)m->clip_max);
                else
                    hipLaunchKernelGGL((k_normalize_gather<float>), dim3(R), dim3(128), 0, st, xg, (const float*)m->center,
                                       (const float*)m->scale, (float*)s->act_in, D, sc, n_tokens, j0, (normalized || !m->center) ? 0 : 1, d.embed_scale,
                                       (const float*)m->clip_min, (const float*)m->clip_max);
                LCM_HIP_CHECK(hipGetLastError());
            }
            LCM_TRY(gemm(s, "ctx.frontend", s->act_in, D, m->pre_linear, nullptr, nullptr, s->xres, dm, R, LCMHIP_EPI_STORE_F32, st));
            if (tc.pos_kind == LCMHIP_POS_LEARNED) LCM_TRY(launch_add_pos(s->xres, m->pos_table, s->pos_rows, R, dm, d.dtype, st));
            if (tc.pos_kind == LCMHIP_POS_SINE) LCM_TRY(launch_add_sine_pos(s->xres, s->pos_rows, R, dm, st));
            // Layer orders (decoder.py:55-88 for the attention block; the FFN block is fairseq2's
            // StandardTransformerDecoderLayer._forward_ffn, which alone applies residual_scale):
            //   pre:        x += attn(norm(x));            x = rs * x + ffn(norm(x))
            //   post:       x  = norm(x + attn(x));        x = norm(rs * x + ffn(x))
            //   normformer: x += attn_norm(attn(norm(x))); x = rs * x + ffn'(norm(x))
            // xres is the fp32 residual stream, act_a the block input in the operand dtype.  After a post-order norm the
            // stream continues from the normalized rows as the operand dtype holds them (what the reference's stream is).
            const bool post = tc.norm_order == LCMHIP_ORDER_POST, normformer = tc.norm_order == LCMHIP_ORDER_NORMFORMER;
            const int64_t n_el = (int64_t)R * dm;
            if (post) LCM_TRY(launch_cast_rows(s->xres, s->act_a, n_el, d.dtype, st));
            for (int l = 0; l < tc.num_layers; ++l) {
                const CtxLayer& L = m->ctx[l];
                if (!post) LCM_TRY(norm(s, "ctx.norm", s->xres, L.sa_norm, nullptr, nullptr, s->act_a, R, R, tc.norm_kind, nullptr, st));
                int rc = LCMHIP_OK;
                if (!tc.qk_layernorm &&
                    qkv_rope_store_fused(s, "ctx.qkv_rope", s->act_a, dm, L.sa, s->act_q, s->self_k + l * self_layer,
                                         s->self_v + l * self_layer, R, s->cap, tc.pos_kind == LCMHIP_POS_ROPE, st, &rc)) {
                    LCM_TRY(rc);
                } else {
                    {
                        const Lin* const w3[3] = {&L.sa.q, &L.sa.k, &L.sa.v};
                        void* const o3[3] = {s->act_q, s->act_k, s->act_v};
                        LCM_TRY(gemm_stack(s, "ctx.qkv", s->act_a, dm, 3, w3, o3, dm, R, st));
                    }
                    if (tc.qk_layernorm) LCM_TRY(qk_norm(s, "ctx.qk_norm", tc, L.sa, s->act_q, s->act_k, R, st));
                    ProfScope ps(&s->prof, "ctx.rope_store", 5.0 * R * dm * es, 0.0, st);
                    LCM_TRY(launch_rope_store(s->act_q, s->act_k, s->act_v, s->self_k + l * self_layer, s->self_v + l * self_layer,
                                              s->self_rows, s->pos_rows, s->rope_cos, s->rope_sin, R, dm, s->cap,
                                              tc.pos_kind == LCMHIP_POS_ROPE, d.dtype, st));
                }
                {
                    ProfScope ps(&s->prof, "ctx.attn", 2.0 * n * (double)(pos0 + sc) * dm * es, 4.0 * R * (double)(pos0 + sc) * dm, st);
                    LCM_TRY(launch_attention(s->act_q, s->self_k + l * self_layer, s->self_v + l * self_layer, slots, s->act_o,
                                             n, sc, tc.num_heads, s->cap, pos0, 1, d.dtype, st));
                }
                if (normformer) {
                    // the block's output passes through its own norm before it joins the stream: project into plain rows
                    // (act_q is free again), normalize them in place, add
                    LCM_TRY(gemm(s, "ctx.attn_out", s->act_o, dm, L.sa.o, nullptr, nullptr, s->act_q, dm, R, LCMHIP_EPI_STORE, st));
                    LCM_TRY(launch_row_norm(s->act_q, L.attn_norm.w ? L.attn_norm.w : s->ones, L.attn_norm.b, nullptr, nullptr, nullptr, R,
                                            dm, kernel_norm_kind(tc.norm_kind), d.norm_eps, d.dtype, st));
                    LCM_TRY(launch_take_rows(s->xres, s->act_q, n_el, d.dtype, 1, st));
                } else {
                    LCM_TRY(gemm(s, "ctx.attn_out", s->act_o, dm, L.sa.o, nullptr, nullptr, s->xres, dm, R, LCMHIP_EPI_RESID, st));
                }
                LCM_TRY(norm(s, "ctx.norm", s->xres, post ? L.sa_norm : L.ffn_norm, nullptr, nullptr, s->act_a, R, R, tc.norm_kind, nullptr, st));
                if (post) LCM_TRY(launch_take_rows(s->xres, s->act_a, n_el, d.dtype, 0, st));
                LCM_TRY(ffn_in(s, "ctx.ffn_in", tc, L.ffn, s->act_a, R, st));
                if (normformer && tc.ffn_kind == LCMHIP_FFN_RELU)
                    LCM_TRY(launch_row_norm(s->act_h, L.inner_norm.w, L.inner_norm.b, nullptr, nullptr, nullptr, R, tc.ffn_dim,
                                            LCMHIP_NORM_LAYERNORM, d.norm_eps, d.dtype, st));
                if (L.residual_scale) LCM_TRY(launch_scale_cols(s->xres, L.residual_scale, R, dm, d.dtype, st));
                LCM_TRY(gemm(s, "ctx.ffn_out", s->act_h, tc.ffn_dim, L.ffn.out, nullptr, nullptr, s->xres, dm, R, LCMHIP_EPI_RESID, st));
                if (post) {
                    LCM_TRY(norm(s, "ctx.norm", s->xres, L.ffn_norm, nullptr, nullptr, s->act_a, R, R, tc.norm_kind, nullptr, st));
                    LCM_TRY(launch_take_rows(s->xres, s->act_a, n_el, d.dtype, 0, st));
                }
            }
            // the stack's own norm (absent when the final order is "post"); a post-order stack without it ends on the
            // rows its last layer normalized
            if (!tc.no_final_norm) LCM_TRY(norm(s, "ctx.norm", s->xres, m->ctx_final, nullptr, nullptr, s->act_a, R, R, tc.norm_kind, nullptr, st));
            else if (!post) LCM_TRY(launch_cast_rows(s->xres, s->act_a, n_el, d.dtype, st));
            // cross-attention keys / values of the new context positions, every denoiser layer
            for (int l = 0; l < d.denoiser.num_layers; ++l) {
                const Lin* const w2[2] = {&m->den[l].ca.k, &m->den[l].ca.v};
                if (d.denoiser.qk_layernorm) {
                    // k passes through its LayerNorm before it is cached: project into plain rows, normalize, scatter
                    void* const o2[2] = {s->act_k, s->act_v};
                    LCM_TRY(gemm_stack(s, "ctx.cross_kv", s->act_a, dm, 2, w2, o2, dm, R, st));
                    LCM_TRY(qk_norm(s, "ctx.qk_norm", d.denoiser, m->den[l].ca, nullptr, s->act_k, R, st));
                    LCM_TRY(launch_rope_store(nullptr, s->act_k, s->act_v, s->cross_k + l * cross_layer, s->cross_v + l * cross_layer,
                                              s->cross_rows, s->pos_rows, s->rope_cos, s->rope_sin, R, dm, s->cap + 1, 0, d.dtype, st));
                    continue;
                }
                void* const o2[2] = {s->cross_k + l * cross_layer, s->cross_v + l * cross_layer};
                LCM_TRY(gemm_stack(s, "ctx.cross_kv", s->act_a, dm, 2, w2, o2, dm, R, st, s->cross_rows, s->cap + 1));
            }
            if (s->keep_context) {
                if (d.dtype == LCMHIP_BF16)
                    hipLaunchKernelGGL((k_keep_rows<bf16_t>), dim3(R), dim3(256), 0, st, (const bf16_t*)s->act_a, s->kept_ctx, s->self_rows, dm);
                else
                    hipLaunchKernelGGL((k_keep_rows<float>), dim3(R), dim3(256), 0, st, (const float*)s->act_a, s->kept_ctx, s->self_rows, dm);
                LCM_HIP_CHECK(hipGetLastError());
            }
            // base LCM: postnet on the decoder output (projection.py:80-86), sampling noise, denormalize
            if (bo && bo->out && (bo->all || j0 + sc == n_tokens)) {
                LCM_TRY(gemm(s, "base.postnet", s->act_a, dm, m->postnet, nullptr, nullptr, s->f32tmp, D, R, LCMHIP_EPI_STORE_F32, st));
                ProfScope ps(&s->prof, "base.out", R * (double)D * 8.0, 0.0, st);
                float* g_out = bo->out + (size_t)d0 * n_out_pos * D;
                const float* g_noise = bo->noise ? bo->noise + (size_t)d0 * n_out_pos * D : nullptr;
                // inputs handed over in normalized space get their predictions back in normalized space: BaseLCModel.forward
                // (builder.py:94-124) has no normalizer in it, predict_next_sentence (:150-165) has it on both sides
                const void* center = normalized ? nullptr : m->center;
                const void* scale = normalized ? nullptr : m->scale;
                const void* cmin = normalized ? nullptr : m->clip_min;
                const void* cmax = normalized ? nullptr : m->clip_max;
                if (d.dtype == LCMHIP_BF16)
                    hipLaunchKernelGGL((k_base_out<bf16_t>), dim3(R), dim3(128), 0, st, s->f32tmp, g_noise, (const bf16_t*)center,
                                       (const bf16_t*)scale, g_out, D, sc, n_tokens, j0, bo->all, bo->temperature,
                                       (const bf16_t*)cmin, (const bf16_t*)cmax, d.postnet_activation);
                else
                    hipLaunchKernelGGL((k_base_out<float>), dim3(R), dim3(128), 0, st, s->f32tmp, g_noise, (const float*)center,
                                       (const float*)scale, g_out, D, sc, n_tokens, j0, bo->all, bo->temperature,
                                       (const float*)cmin, (const float*)cmax, d.postnet_activation);
                LCM_HIP_CHECK(hipGetLastError());
            }
        }
    }
    s->ctx_len += n_tokens;
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), s->d_cross_keys, s->ctx_len + 1);
    LCM_HIP_CHECK(hipGetLastError());
    return LCMHIP_OK;
}

int lcmhip_context_append(lcmhip_session* s, const float* x, int32_t n_tokens, int32_t normalized, void* stream) {
    return append_impl(s, x, n_tokens, normalized, stream, nullptr);
}

int lcmhip_base_forward(lcmhip_session* s, const float* x, int32_t n_tokens, int32_t normalized, const float* noise,
                        float temperature, int32_t all_positions, float* out, void* stream) {
    LCM_REQUIRE(s && s->model->base, "base_forward: the model is not a base LCM (it has a denoiser tower)");
    BaseOut bo;
    bo.out = out;
    bo.noise = (noise != nullptr && temperature > 0.f) ? noise : nullptr;
    bo.temperature = temperature;
    bo.all = all_positions ? 1 : 0;
    return append_impl(s, x, n_tokens, normalized, stream, &bo);
}

int lcmhip_session_read_context(lcmhip_session* s, int32_t row, float* out, void* stream) {
    LCM_REQUIRE(s && out, "read_context: null argument");
    if (!s->keep_context) return fail(LCMHIP_ERR_STATE, "read_context: session was created without LCMHIP_SESSION_KEEP_CONTEXT");
    LCM_REQUIRE(row >= 0 && row < s->n_active, "read_context: row out of range");
    const size_t dm = s->model->desc.model_dim;
    LCM_HIP_CHECK(hipMemcpyAsync(out, s->kept_ctx + (size_t)s->h_slot[row] * s->cap * dm, (size_t)s->ctx_len * dm * 4,
                                 hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return LCMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ denoising loop
// The whole sampling loop of one sentence.  `keys_dev` != nullptr makes the attention launches read the key
// count from device memory (the form that is recorded into a graph and replayed for later sentences).
static int denoise_body(lcmhip_session* s, const float* noise, float* out, float* latents_trace, hipStream_t st,
                        const int32_t* keys_dev, const float* variance_noise = nullptr) {
    lcmhip_model* m = s->model;
    const lcmhip_model_desc& d = m->desc;
    const lcmhip_tower_desc& td = d.denoiser;
    const int dm = d.model_dim, D = d.sonar_dim, B = s->n_active, L = td.num_layers;
    const size_t es = dtype_size(d.dtype);
    const size_t cross_layer = (size_t)s->max_batch * (s->cap + 1) * dm * es;
    const int nkeys = s->ctx_len + 1;

    // the latents also live in the operand dtype (act_lat), written by the kernels that produce them
    LCM_TRY(launch_init_latents(noise, s->lat, (int64_t)B * D, s->params.init_noise_sigma, s->params.clip_noise, st, s->act_lat, d.dtype));
    float* lat = s->lat;
    float* lat_next = s->lat2;
    for (int si = 0; si < s->n_steps; ++si) {
        const DdimScalars& sc = s->scalars[si];
        const int rows = sc.mode == 1 ? 2 * B : B;
        const float* mod_step = s->mod_table + (size_t)si * L * 9 * dm;
        // initial projection: both halves of the doubled batch see the same latents; the product stays
        // in the split-K slab and the first norm kernel writes it (duplicated) into the residual stream
        Pending pend;
        LCM_TRY(gemm_pending(s, "den.initial_proj", s->act_lat, D, m->init_proj, nullptr, B, &pend, st));
        pend.init = 1;
        pend.dup = 1;
        // one latent per call and no state bag: it sits at position 0 of the learned positions (bias + row 0, see set_schedule)
        if (m->den_pos_table) pend.bias = s->init_bias_pos;
        for (int l = 0; l < L; ++l) {
            const DenLayer& Ly = m->den[l];
            const float* mod = mod_step + (size_t)l * 9 * dm;
            // self-attention over a single position: softmax == 1, so the block is output_proj(v_proj(h))
            LCM_TRY(norm(s, "den.norm_mod", s->xres, Ly.sa_norm, mod, mod + dm, s->act_a, rows, rows, td.norm_kind, &pend, st));
            if (Ly.ov.w != nullptr) {
                LCM_TRY(gemm_pending(s, "den.self_ov", s->act_a, dm, Ly.ov, mod + 2 * dm, rows, &pend, st));
            } else {
                LCM_TRY(gemm(s, "den.self_v", s->act_a, dm, Ly.sa.v, nullptr, nullptr, s->act_v, dm, rows, LCMHIP_EPI_STORE, st));
                LCM_TRY(gemm_pending(s, "den.self_o", s->act_v, dm, Ly.sa.o, mod + 2 * dm, rows, &pend, st));
            }
            // cross-attention: conditional rows attend to [zero slot, context]; unconditional rows only
            // see the zero slot, whose value is a per-layer constant
            LCM_TRY(norm(s, "den.norm_mod", s->xres, Ly.ca_norm, mod + 3 * dm, mod + 4 * dm, s->act_a, rows, B, td.norm_kind, &pend, st));
            Pending qp;   // the q projection stays in slab_q; the attention kernel sums it
            if (td.qk_layernorm) {
                // q passes through its LayerNorm first: a finished product, normalized in place
                LCM_TRY(gemm(s, "den.cross_q", s->act_a, dm, Ly.ca.q, nullptr, nullptr, s->act_q, dm, B, LCMHIP_EPI_STORE, st));
                LCM_TRY(qk_norm(s, "den.qk_norm", td, Ly.ca, s->act_q, nullptr, B, st));
            } else {
                LCM_TRY(gemm_pending(s, "den.cross_q", s->act_a, dm, Ly.ca.q, nullptr, B, &qp, st, s->slab_q));
            }
            {
                AttnQ qa{};
                if (td.qk_layernorm) qa.q = s->act_q;
                qa.partial = qp.partial; qa.splits = qp.splits; qa.rows = B; qa.bias = qp.bias;
                ProfScope ps(&s->prof, "den.cross_attn", 2.0 * B * (double)nkeys * dm * es + B * dm * (es + 4.0 * qp.splits),
                             4.0 * B * (double)nkeys * dm, st);
                LCM_TRY(launch_attention_q(qa, s->cross_k + l * cross_layer, s->cross_v + l * cross_layer, s->slot_map,
                                           s->act_o, B, 1, td.num_heads, s->cap + 1, nkeys, 0, d.dtype, st, keys_dev));
            }
            LCM_TRY(gemm_pending(s, "den.cross_o", s->act_o, dm, Ly.ca.o, mod + 5 * dm, B, &pend, st));
            pend.extra_const = s->uconst + (size_t)l * dm;
            LCM_TRY(norm(s, "den.norm_mod", s->xres, Ly.ffn_norm, mod + 6 * dm, mod + 7 * dm, s->act_a, rows, rows, td.norm_kind, &pend, st));
            LCM_TRY(ffn_in(s, "den.ffn_in", td, Ly.ffn, s->act_a, rows, st));
            LCM_TRY(gemm_pending(s, "den.ffn_out", s->act_h, td.ffn_dim, Ly.ffn.out, mod + 8 * dm, rows, &pend, st));
        }
        LCM_TRY(norm(s, "den.norm_mod", s->xres, m->den_final, nullptr, nullptr, s->act_a, rows, rows, td.norm_kind, &pend, st));
        // the final projection stays in its split-K slab: the fused DDIM / guidance step sums it
        Pending fp;
        LCM_TRY(gemm_pending(s, "den.final_proj", s->act_a, dm, m->final_proj, nullptr, rows, &fp, st, s->slab_q));
        {
            DdimIO io;
            io.partial = fp.partial; io.splits = fp.splits; io.rows_total = rows; io.bias = fp.bias;
            io.prev_cast = s->act_lat; io.bf16 = d.dtype == LCMHIP_BF16 ? 1 : 0;
            io.variance_noise = variance_noise ? variance_noise + (size_t)si * B * D : nullptr;
            ProfScope ps(&s->prof, "den.ddim_cfg_step", (double)B * D * 4.0 * (3.0 + fp.splits * (rows == 2 * B ? 2.0 : 1.0)), 0.0, st);
            // the predicted x0 is only read after the last step (denormalize)
            LCM_TRY(launch_ddim_step(lat, nullptr, nullptr, lat_next, si + 1 == s->n_steps ? s->x0 : nullptr, B, D, sc, st, &io));
        }
        std::swap(lat, lat_next);
        if (latents_trace)
            LCM_HIP_CHECK(hipMemcpyAsync(latents_trace + (size_t)si * B * D, lat, (size_t)B * D * 4, hipMemcpyDeviceToDevice, st));
    }
    return launch_denormalize_f32_out(s->x0, m->center, m->scale, out, B, D, d.dtype, st, m->clip_min, m->clip_max);
}

}  // extern "C"

static int denoise_impl(lcmhip_session* s, const float* noise, const float* variance_noise, float* out,
                        float* latents_trace, void* stream) {
    LCM_REQUIRE(s && noise && out, "denoise: null argument");
    LCM_REQUIRE(!s->model->base, "denoise: a base LCM has no denoiser (use lcmhip_base_forward)");
    if (!s->have_schedule) return fail(LCMHIP_ERR_STATE, "denoise: call session_set_schedule first");
    LCM_REQUIRE(s->n_active >= 1 && s->ctx_len >= 1, "denoise: empty context (append at least one position first)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (s->params.ddim_eta > 0.f) {
        // stochastic DDIM: the per-step draws are an input of every call, so the loop is launched, not replayed
        if (variance_noise == nullptr)
            return fail(LCMHIP_ERR_STATE, "denoise: the schedule has ddim_eta > 0; call lcmhip_denoise_stochastic with variance noise");
        return denoise_body(s, noise, out, latents_trace, st, nullptr, variance_noise);
    }
    // Graph replay: the loop is ~6000 launches per sentence whose arguments depend only on the schedule and on
    // the number of active documents, so it is recorded once per session and replayed.  Needs a non-default
    // stream (capture is not defined on the legacy stream); per-launch profiling and latent traces run eagerly.
    if (!s->use_graphs || st == nullptr || s->prof.enabled || latents_trace != nullptr)
        return denoise_body(s, noise, out, latents_trace, st, nullptr);
    const size_t io_bytes = (size_t)s->n_active * s->model->desc.sonar_dim * 4;
    auto it = s->graphs.find(s->n_active);
    if (it == s->graphs.end()) {
        if (s->graphs.size() >= 16) drop_graphs(s);   // documents finishing one by one: keep the cache bounded
        // one recording at a time in the process: sessions on other host threads wait here (a recording is a few
        // thousand captured launches, once per session and batch size), they do not record side by side
        static std::mutex record_mutex;
        std::lock_guard<std::mutex> recording(record_mutex);
        hipGraph_t graph = nullptr;
        LCM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int rc = denoise_body(s, s->noise_buf, s->out_buf, nullptr, st, s->d_cross_keys);
        const hipError_t end = hipStreamEndCapture(st, &graph);
        if (rc != LCMHIP_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        LCM_HIP_CHECK(end);
        hipGraphExec_t exec = nullptr;
        const hipError_t inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        LCM_HIP_CHECK(inst);
        it = s->graphs.emplace(s->n_active, exec).first;
    }
    LCM_HIP_CHECK(hipMemcpyAsync(s->noise_buf, noise, io_bytes, hipMemcpyDeviceToDevice, st));
    LCM_HIP_CHECK(hipGraphLaunch(it->second, st));
    s->graph_stream = st;
    s->graph_launched = true;
    LCM_HIP_CHECK(hipMemcpyAsync(out, s->out_buf, io_bytes, hipMemcpyDeviceToDevice, st));
    return LCMHIP_OK;
}

extern "C" {

int lcmhip_denoise(lcmhip_session* s, const float* noise, float* out, float* latents_trace, void* stream) {
    return denoise_impl(s, noise, nullptr, out, latents_trace, stream);
}

int lcmhip_denoise_stochastic(lcmhip_session* s, const float* noise, const float* variance_noise, float* out,
                              float* latents_trace, void* stream) {
    return denoise_impl(s, noise, variance_noise, out, latents_trace, stream);
}

// ------------------------------------------------------------------------------------------------ all positions at once
// The denoiser over every position of every document in one pass (the training / validation shape of the model):
// position j of document i has its own noisy latent and its own diffusion step, its self-attention sees only
// itself (diagonal mask, so the block is output_proj(v_proj(h)) as in the sampling loop), and its cross-attention
// sees the zero slot plus context positions [0, j + shift_i) (lower-triangular mask moved by the source length),
// cut at the document's context length.  Rows are independent given the cross K/V cache, so the pass is cut into
// document groups (and long documents into position ranges) of at most chunk_rows rows.
int lcmhip_denoise_positions(lcmhip_session* s, const float* noisy, const int32_t* step_index, int32_t n_positions,
                             const int32_t* source_lengths, const int32_t* context_lengths, float* out, void* stream) {
    LCM_REQUIRE(s && noisy && step_index && out, "denoise_positions: null argument");
    LCM_REQUIRE(!s->model->base, "denoise_positions: a base LCM has no denoiser");
    if (!s->have_schedule) return fail(LCMHIP_ERR_STATE, "denoise_positions: call session_set_schedule first (it builds the modulation table)");
    LCM_REQUIRE(s->n_active >= 1, "denoise_positions: call session_reset first");
    LCM_REQUIRE(n_positions >= 1 && n_positions <= 65536, "denoise_positions: bad position count");
    hipStream_t st = static_cast<hipStream_t>(stream);
    lcmhip_model* m = s->model;
    const lcmhip_model_desc& d = m->desc;
    const lcmhip_tower_desc& td = d.denoiser;
    const int dm = d.model_dim, D = d.sonar_dim, L = td.num_layers;
    const size_t es = dtype_size(d.dtype);
    const size_t cross_layer = (size_t)s->max_batch * (s->cap + 1) * dm * es;
    RowSteps rs;
    rs.row_step = s->pos_rows;
    rs.step_stride = (size_t)L * 9 * dm;
    rs.n_steps = s->n_steps;
    KeyWindow win;
    win.shift = nullptr;
    win.limit = nullptr;
    win.max_keys = s->ctx_len + 1;   // zero slot + the positions appended so far

    const int n_total = s->n_active;
    const int doc_group = std::max(1, std::min(n_total, s->chunk_rows / std::min(n_positions, s->chunk_rows)));
    for (int d0 = 0; d0 < n_total; d0 += doc_group) {
        const int n = std::min(doc_group, n_total - d0);
        const int per_chunk = std::max(1, s->chunk_rows / n);
        const float* xg = noisy + (size_t)d0 * n_positions * D;
        const int32_t* slots = s->slot_map + d0;
        win.shift = source_lengths ? source_lengths + d0 : nullptr;
        if (context_lengths) {
            // key limit = context length + 1 (the zero slot); built per pass in cross_rows
            win.limit = s->cross_rows;
        }
        for (int j0 = 0; j0 < n_positions; j0 += per_chunk) {
            const int sc = std::min(per_chunk, n_positions - j0);
            const int R = n * sc;
            hipLaunchKernelGGL(k_gather_rows_int, dim3((R + 255) / 256), dim3(256), 0, st, step_index + (size_t)d0 * n_positions,
                               s->pos_rows, R, sc, n_positions, j0);
            if (context_lengths)
                hipLaunchKernelGGL(k_add_int, dim3((n + 255) / 256), dim3(256), 0, st, context_lengths + d0, s->cross_rows, n, 1);
            LCM_HIP_CHECK(hipGetLastError());
            // latents of this pass, operand dtype (no normalizer: the caller's latents live in normalized space)
            if (d.dtype == LCMHIP_BF16)
                hipLaunchKernelGGL((k_normalize_gather<bf16_t>), dim3(R), dim3(128), 0, st, xg, (const bf16_t*)nullptr,
                                   (const bf16_t*)nullptr, (bf16_t*)s->act_in, D, sc, n_positions, j0, 0, 1.0f,
                                   (const bf16_t*)nullptr, (const bf16_t*)nullptr);
            else
                hipLaunchKernelGGL((k_normalize_gather<float>), dim3(R), dim3(128), 0, st, xg, (const float*)nullptr,
                                   (const float*)nullptr, (float*)s->act_in, D, sc, n_positions, j0, 0, 1.0f,
                                   (const float*)nullptr, (const float*)nullptr);
            LCM_HIP_CHECK(hipGetLastError());
            // The residual stream is updated by the projections' own epilogues (x += gate[row's step] * (h W^T + b)):
            // with many rows no K split is needed, so nothing has to wait in a slab for the next norm kernel.
            LCM_TRY(gemm(s, "pos.initial_proj", s->act_in, D, m->init_proj, nullptr, nullptr, s->xres, dm, R, LCMHIP_EPI_STORE_F32, st));
            if (m->den_pos_table) {   // position j of a document gets row j of the learned positions
                hipLaunchKernelGGL(k_row_positions, dim3((R + 255) / 256), dim3(256), 0, st, s->self_rows, R, sc, j0);
                LCM_HIP_CHECK(hipGetLastError());
                LCM_TRY(launch_add_pos(s->xres, m->den_pos_table, s->self_rows, R, dm, d.dtype, st));
            }
            for (int l = 0; l < L; ++l) {
                const DenLayer& Ly = m->den[l];
                const float* mod = s->mod_table + (size_t)l * 9 * dm;   // step 0; rows add their own step offset
                LCM_TRY(norm(s, "pos.norm_mod", s->xres, Ly.sa_norm, mod, mod + dm, s->act_a, R, R, td.norm_kind, nullptr, st, &rs));
                if (Ly.ov.w != nullptr) {
                    LCM_TRY(gemm(s, "pos.self_ov", s->act_a, dm, Ly.ov, nullptr, mod + 2 * dm, s->xres, dm, R, LCMHIP_EPI_RESID, st,
                                 nullptr, 0, nullptr, 0, &rs));
                } else {
                    LCM_TRY(gemm(s, "pos.self_v", s->act_a, dm, Ly.sa.v, nullptr, nullptr, s->act_v, dm, R, LCMHIP_EPI_STORE, st));
                    LCM_TRY(gemm(s, "pos.self_o", s->act_v, dm, Ly.sa.o, nullptr, mod + 2 * dm, s->xres, dm, R, LCMHIP_EPI_RESID, st,
                                 nullptr, 0, nullptr, 0, &rs));
                }
                LCM_TRY(norm(s, "pos.norm_mod", s->xres, Ly.ca_norm, mod + 3 * dm, mod + 4 * dm, s->act_a, R, R, td.norm_kind, nullptr, st, &rs));
                LCM_TRY(gemm(s, "pos.cross_q", s->act_a, dm, Ly.ca.q, nullptr, nullptr, s->act_q, dm, R, LCMHIP_EPI_STORE, st));
                if (td.qk_layernorm) LCM_TRY(qk_norm(s, "pos.qk_norm", td, Ly.ca, s->act_q, nullptr, R, st));
                {
                    AttnQ qa{};
                    qa.q = s->act_q;
                    ProfScope ps(&s->prof, "pos.cross_attn", 2.0 * n * (double)win.max_keys * dm * es + 2.0 * R * dm * es,
                                 2.0 * R * (double)std::min(win.max_keys, j0 + sc + 1) * dm, st);
                    LCM_TRY(launch_attention_q(qa, s->cross_k + l * cross_layer, s->cross_v + l * cross_layer, slots, s->act_o, n, sc,
                                               td.num_heads, s->cap + 1, j0, 1, d.dtype, st, nullptr, &win));
                }
                LCM_TRY(gemm(s, "pos.cross_o", s->act_o, dm, Ly.ca.o, nullptr, mod + 5 * dm, s->xres, dm, R, LCMHIP_EPI_RESID, st,
                             nullptr, 0, nullptr, 0, &rs));
                LCM_TRY(norm(s, "pos.norm_mod", s->xres, Ly.ffn_norm, mod + 6 * dm, mod + 7 * dm, s->act_a, R, R, td.norm_kind, nullptr, st, &rs));
                LCM_TRY(ffn_in(s, "pos.ffn_in", td, Ly.ffn, s->act_a, R, st));
                LCM_TRY(gemm(s, "pos.ffn_out", s->act_h, td.ffn_dim, Ly.ffn.out, nullptr, mod + 8 * dm, s->xres, dm, R, LCMHIP_EPI_RESID, st,
                             nullptr, 0, nullptr, 0, &rs));
            }
            LCM_TRY(norm(s, "pos.norm_mod", s->xres, m->den_final, nullptr, nullptr, s->act_a, R, R, td.norm_kind, nullptr, st));
            LCM_TRY(gemm(s, "pos.final_proj", s->act_a, dm, m->final_proj, nullptr, nullptr, s->f32tmp, D, R, LCMHIP_EPI_STORE_F32, st));
            float* g_out = out + (size_t)d0 * n_positions * D;
            if (d.dtype == LCMHIP_BF16)
                hipLaunchKernelGGL((k_base_out<bf16_t>), dim3(R), dim3(128), 0, st, s->f32tmp, (const float*)nullptr,
                                   (const bf16_t*)nullptr, (const bf16_t*)nullptr, g_out, D, sc, n_positions, j0, 1, 0.f,
                                   (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0);
            else
                hipLaunchKernelGGL((k_base_out<float>), dim3(R), dim3(128), 0, st, s->f32tmp, (const float*)nullptr,
                                   (const float*)nullptr, (const float*)nullptr, g_out, D, sc, n_positions, j0, 1, 0.f,
                                   (const float*)nullptr, (const float*)nullptr, 0);
            LCM_HIP_CHECK(hipGetLastError());
        }
    }
    return LCMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ profiling
int lcmhip_session_profile_enable(lcmhip_session* s, int32_t on) {
    LCM_REQUIRE(s, "profile_enable: null session");
    s->prof.enabled = on != 0;
    return LCMHIP_OK;
}

int lcmhip_session_profile(lcmhip_session* s, lcmhip_profile_row* rows, int32_t max_rows, int32_t* n_rows, void* stream) {
    LCM_REQUIRE(s && rows && n_rows, "profile: null argument");
    LCM_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    s->prof.collect();
    int n = 0;
    for (const auto& r : s->prof.rows) {
        if (n >= max_rows) break;
        lcmhip_profile_row& o = rows[n++];
        memset(&o, 0, sizeof(o));
        strncpy(o.name, r.name.c_str(), sizeof(o.name) - 1);
        o.launches = r.launches;
        o.total_ms = r.ms;
        o.bytes = r.bytes;
        o.flops = r.flops;
    }
    *n_rows = n;
    s->prof.clear_rows();
    return LCMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ single kernels
int lcmhip_normalize(const void* x, const void* center, const void* scale, const void* clip_min, const void* clip_max,
                     void* y, int64_t rows, int32_t dim, int32_t dtype, void* stream) {
    LCM_REQUIRE(x && center && scale && y, "normalize: null argument");
    return launch_normalize(x, center, scale, clip_min, clip_max, y, rows, dim, dtype, false, static_cast<hipStream_t>(stream));
}

int lcmhip_denormalize(const void* x, const void* center, const void* scale, const void* clip_min, const void* clip_max,
                       void* y, int64_t rows, int32_t dim, int32_t dtype, void* stream) {
    LCM_REQUIRE(x && center && scale && y, "denormalize: null argument");
    return launch_normalize(x, center, scale, clip_min, clip_max, y, rows, dim, dtype, true, static_cast<hipStream_t>(stream));
}

int lcmhip_rfft_pack(const float* x, float* y, int64_t rows, int32_t dim, int32_t inverse, void* stream) {
    LCM_REQUIRE(x && y, "rfft_pack: null argument");
    return launch_rfft_pack(x, y, rows, dim, inverse != 0, static_cast<hipStream_t>(stream));
}

int lcmhip_norm_modulate(const float* x, const void* weight, const void* bias, const float* shift, const float* scale,
                         void* y, int32_t rows, int32_t dim, int32_t norm_kind, float eps, int32_t dtype, void* stream) {
    LCM_REQUIRE(x && y, "norm_modulate: null argument");
    LCM_REQUIRE(norm_kind == LCMHIP_NORM_RMS || norm_kind == LCMHIP_NORM_LAYERNORM, "norm_modulate: unknown norm kind");
    return launch_norm_modulate(const_cast<float*>(x), weight, bias, shift, scale, y, rows, rows, dim, norm_kind, eps, dtype,
                                nullptr, static_cast<hipStream_t>(stream));
}

int lcmhip_gemm(const void* A, int32_t lda, const void* W, const void* W2, const void* bias, const void* bias2,
                const float* gate, void* out, int32_t ldo, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                int32_t dtype, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    LCM_REQUIRE(A && W && out, "gemm: null argument");
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = W; a.W2 = W2; a.bias = bias; a.bias2 = bias2; a.gate = gate;
    a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.epilogue = epilogue & 0xff; a.dtype = dtype;
    a.act = (epilogue >> 8) & 0xff;      // LCMHIP_EPI_RELU | (LCMHIP_ACT_x << 8)
    LCM_REQUIRE((epilogue >> 16) == 0 && a.act <= LCMHIP_ACT_HARDSIGMOID && (a.act == 0 || a.epilogue == LCMHIP_EPI_RELU),
                "gemm: an activation code goes with LCMHIP_EPI_RELU only");
    a.partial = splitk_ws; a.partial_bytes = splitk_ws_bytes; a.max_splits = kMaxSplits;
    return launch_gemm(a, static_cast<hipStream_t>(stream));
}

int lcmhip_gemm_partial(const void* A, int32_t lda, const void* W, float* slab, size_t slab_bytes, int32_t* splits,
                        int32_t M, int32_t N, int32_t K, int32_t dtype, void* stream) {
    LCM_REQUIRE(A && W && slab && splits, "gemm_partial: null argument");
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = W; a.M = M; a.N = N; a.K = K; a.epilogue = LCMHIP_EPI_STORE_F32; a.dtype = dtype;
    a.partial = slab; a.partial_bytes = slab_bytes;
    a.max_splits = (int)std::min<size_t>(kMaxSplits, slab_bytes / ((size_t)M * N * 4));
    LCM_REQUIRE(a.max_splits >= 1, "gemm_partial: slab too small for one split");
    int sp = 0;
    LCM_TRY(launch_gemm_partial(a, &sp, static_cast<hipStream_t>(stream)));
    *splits = sp;
    return LCMHIP_OK;
}

int lcmhip_residual_norm(float* x, const float* slab, int32_t splits, int32_t M, const void* bias, const float* gate,
                         const float* extra_const, int32_t dup, int32_t init, const void* weight, const void* norm_bias,
                         const float* shift, const float* scale, void* y, int32_t rows_total, int32_t rows_norm,
                         int32_t dim, int32_t norm_kind, float eps, int32_t dtype, void* stream) {
    LCM_REQUIRE(x && slab && y, "residual_norm: null argument");
    LCM_REQUIRE(norm_kind == LCMHIP_NORM_RMS || norm_kind == LCMHIP_NORM_LAYERNORM, "residual_norm: unknown norm kind");
    Pending pd;
    pd.partial = slab; pd.splits = splits; pd.M = M; pd.bias = bias; pd.gate = gate; pd.extra_const = extra_const;
    pd.dup = dup; pd.init = init;
    return launch_norm_modulate(x, weight, norm_bias, shift, scale, y, rows_total, rows_norm, dim, norm_kind, eps, dtype, &pd,
                                static_cast<hipStream_t>(stream));
}

int lcmhip_ddim_cfg_step(const float* sample, const float* out_cond, const float* out_uncond, float* prev_sample,
                         float* pred_x0, int32_t rows, int32_t dim, float alpha_prod_t, float alpha_prod_t_prev,
                         const lcmhip_denoise_params* p, const float* variance_noise, void* stream) {
    LCM_REQUIRE(sample && out_cond && prev_sample && p, "ddim_cfg_step: null argument");
    LCM_REQUIRE(p->ddim_eta >= 0.f, "ddim_cfg_step: ddim_eta must not be negative");
    LCM_REQUIRE(p->ddim_eta == 0.f || variance_noise != nullptr, "ddim_cfg_step: ddim_eta > 0 needs variance noise");
    DdimScalars sc{};
    sc.sqrt_a_t = sqrtf(alpha_prod_t);
    sc.sqrt_b_t = sqrtf(1.0f - alpha_prod_t);
    sc.sqrt_a_prev = sqrtf(alpha_prod_t_prev);
    sc.sigma = 0.f;
    if (p->ddim_eta > 0.f)
        sc.sigma = p->ddim_eta * sqrtf(((1.0f - alpha_prod_t_prev) / (1.0f - alpha_prod_t)) * (1.0f - alpha_prod_t / alpha_prod_t_prev));
    sc.dir_coef = sqrtf(1.0f - alpha_prod_t_prev - sc.sigma * sc.sigma);
    sc.guidance_scale = p->guidance_scale;
    sc.guidance_rescale = p->guidance_rescale;
    sc.inv_eps_scaling = p->epsilon_scaling > 0.f ? 1.0f / p->epsilon_scaling : 1.0f;
    sc.clip_noise = p->clip_noise;
    sc.clip_sample_range = p->clip_sample ? p->clip_sample_range : -1.0f;
    sc.thr_ratio = p->thresholding ? p->dynamic_thresholding_ratio : -1.0f;
    sc.thr_max = p->sample_max_value;
    sc.prediction_type = p->prediction_type;
    sc.mode = out_uncond ? 1 : 0;
    DdimIO io;
    io.variance_noise = variance_noise;
    return launch_ddim_step(sample, out_cond, out_uncond, prev_sample, pred_x0, rows, dim, sc, static_cast<hipStream_t>(stream), &io);
}

int lcmhip_stream_copy(const void* src, void* dst, int64_t bytes, int32_t nontemporal, void* stream) {
    LCM_REQUIRE(src && dst, "stream_copy: null argument");
    return launch_stream_copy(src, dst, bytes, nontemporal, static_cast<hipStream_t>(stream));
}

int lcmhip_finalize_step(const float* new_emb, void* seqs, int32_t seqs_dtype, const int32_t* doc_rows,
                         const int32_t* prompt_lens, int32_t n_active, int32_t step, int32_t total_len, int32_t dim,
                         const float* eos_vec, float* cos_out, void* stream) {
    LCM_REQUIRE(new_emb && seqs && cos_out, "finalize_step: null argument");
    return launch_finalize_step(new_emb, seqs, seqs_dtype, doc_rows, prompt_lens, n_active, step, total_len, dim, eos_vec,
                                cos_out, static_cast<hipStream_t>(stream));
}

int lcmhip_attention(const void* q, const void* k_cache, const void* v_cache, const int32_t* slot_map, void* out,
                     int32_t n_docs, int32_t s, int32_t n_heads, int32_t cap, int32_t base_keys, int32_t causal,
                     int32_t dtype, void* stream) {
    LCM_REQUIRE(q && k_cache && v_cache && out, "attention: null argument");
    return launch_attention(q, k_cache, v_cache, slot_map, out, n_docs, s, n_heads, cap, base_keys, causal, dtype,
                            static_cast<hipStream_t>(stream));
}

}  // extern "C"
