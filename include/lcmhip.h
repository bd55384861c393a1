/*
 * lcmhip.h -- C ABI of liblcm_hip.so: the MI355X (gfx950) generation path of the two-tower
 * diffusion Large Concept Model.
 *
 * The reference (viasky657/Chimera-LLM) has no FFI for this path: its boundary is the Python
 * API of lcm/inference/two_tower_diffusion_lcm/generator.py and
 * lcm/models/two_tower_diffusion_lcm/builder.py.  Each entry point below names the reference
 * function whose arithmetic it replaces (file:line relative to the reference root).
 *
 * Conventions
 *  - every function returns 0 on success and a negative LCMHIP_ERR_* code otherwise;
 *    lcmhip_last_error() returns a thread-local message for the last failure on this thread;
 *    nothing throws across the boundary.
 *  - all tensor pointers are BORROWED device pointers (row-major, contiguous unless a stride
 *    is given); weights must outlive the model handle; workspaces are allocated by the caller.
 *  - `stream` is a hipStream_t passed as void*; every kernel is enqueued on it; no call
 *    synchronises the device unless its comment says so.
 *  - a session is not thread-safe: one host thread at a time calls into a given session.  Different sessions of one
 *    model may be driven by different host threads on different streams at the same time (the model is read-only
 *    after lcmhip_model_finalize).  A session records its sampling loop into a HIP graph with a THREAD-LOCAL
 *    stream capture the first time it sees a shape; while that happens no other thread of the process may
 *    synchronise the whole device (hipDeviceSynchronize), which this runtime refuses during any capture and which
 *    invalidates the recording.  Stream and event synchronisation, copies and allocations by other threads are fine.
 */
#ifndef LCMHIP_H
#define LCMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this ABI (struct layouts and signatures).  lcmhip_version() returns the value the library was built
 * with; a caller built against another value must not bind.  400 -> 401: lcmhip_base_forward with normalized != 0
 * returns its predictions in the normalizer's space (it applied the inverse normalizer before).  300 -> 400: lcmhip_tower_desc.norm_order,
 * no_final_norm and scale_residual, LCMHIP_POS_SINE, LCMHIP_ACT_PRELU, lcmhip_model_desc.postnet_activation,
 * lcmhip_rfft_pack; 200 -> 300: lcmhip_session_workspace_bytes and
 * lcmhip_session_create take `flags`, lcmhip_static_init added, lcmhip_tower_desc.ffn_activation; 100 -> 200: lcmhip_tower_desc.qk_layernorm,
 * thresholding fields of lcmhip_denoise_params, variance_noise of lcmhip_ddim_cfg_step. */
#define LCMHIP_ABI_VERSION 401

#define LCMHIP_OK 0
#define LCMHIP_ERR_INVALID (-1)     /* bad argument / shape the kernels do not support */
#define LCMHIP_ERR_HIP (-2)         /* a HIP runtime call failed */
#define LCMHIP_ERR_UNSUPPORTED (-3) /* valid in the reference, not implemented here */
#define LCMHIP_ERR_STATE (-4)       /* call sequence error (e.g. denoise before set_schedule) */

enum { LCMHIP_F32 = 0, LCMHIP_BF16 = 1 };
/* lcm/nn/normalization.py:73-88: "standard" (LayerNorm, weight + bias), "rms" (RMSNorm, weight only), "fp32"
 * (LayerNorm evaluated in fp32, weight only) and "unit" (LayerNorm without learnable parameters).  Every kind keeps
 * its statistics in fp32 here, so "fp32" differs from "standard" only by the missing bias. */
enum { LCMHIP_NORM_LAYERNORM = 0, LCMHIP_NORM_RMS = 1, LCMHIP_NORM_FP32 = 2, LCMHIP_NORM_UNIT = 3 };
enum { LCMHIP_FFN_RELU = 0, LCMHIP_FFN_SWIGLU = 1 };  /* FFN_RELU: the two-matrix FFN with the activation of ffn_activation */
/* Inner activation of the two-matrix FFN (lcm/nn/transformer/factory.py:73, lcm/nn/initialization.py:121-151: the
 * torch.nn modules of these names with their default arguments).  PRELU (torch.nn.PReLU(): one learned slope, tensor
 * <layer>.ffn.inner_activation.weight of shape (1,)) exists as lcmhip_tower_desc.ffn_activation only: the epilogue
 * argument of lcmhip_gemm has no room for the slope. */
enum {
    LCMHIP_ACT_RELU = 0, LCMHIP_ACT_TANH = 1, LCMHIP_ACT_ELU = 2, LCMHIP_ACT_LEAKY_RELU = 3, LCMHIP_ACT_SELU = 4,
    LCMHIP_ACT_GELU = 5, LCMHIP_ACT_SILU = 6, LCMHIP_ACT_SOFTSIGN = 7, LCMHIP_ACT_SIGMOID = 8, LCMHIP_ACT_HARDSIGMOID = 9,
    LCMHIP_ACT_PRELU = 10
};
/* lcm/nn/transformer/factory.py:79-87.  SINE: fairseq2's SinusoidalPositionEncoder, added by the frontend in fp32 --
 * row p = [sin(p f_0) .. sin(p f_{h-1}), cos(p f_0) .. cos(p f_{h-1})], h = model_dim / 2, f_i = exp(-i ln(10000) / (h - 1));
 * the contextualizer / base-LCM stack only. */
enum { LCMHIP_POS_NONE = 0, LCMHIP_POS_LEARNED = 1, LCMHIP_POS_ROPE = 2, LCMHIP_POS_SINE = 3 };
/* lcm/nn/transformer/factory.py:95-98, lcm/nn/transformer/decoder.py:55-88 (the contextualizer / base-LCM stack; the
 * denoiser's layers are pre-norm, lcm/nn/denoisers/lcm_denoiser.py:441-443).
 *  PRE:        x += attn(norm(x));             x += ffn(norm(x))
 *  POST:       x  = norm(x + attn(x));         x  = norm(x + ffn(x))
 *  NORMFORMER: x += attn_norm(attn(norm(x)));  x += ffn'(norm(x)), ffn' with a LayerNorm (weight + bias) on the inner
 *              activations of the two-matrix FFN; tensors <layer>.self_attn_norm.* and <layer>.ffn.inner_layer_norm.* */
enum { LCMHIP_ORDER_PRE = 0, LCMHIP_ORDER_POST = 1, LCMHIP_ORDER_NORMFORMER = 2 };
enum { LCMHIP_PRED_SAMPLE = 0, LCMHIP_PRED_EPSILON = 1, LCMHIP_PRED_V = 2 };

/* GEMM epilogues of lcmhip_gemm (C = A * W^T, W stored (N, K) like torch.nn.Linear.weight) */
enum {
    LCMHIP_EPI_STORE = 0,     /* out[m][n] = acc + bias[n]            (out in the operand dtype) */
    LCMHIP_EPI_STORE_F32 = 1, /* same, fp32 output                                               */
    LCMHIP_EPI_RELU = 2,      /* out = act(acc + bias); act = ReLU, or LCMHIP_ACT_x when the epilogue
                                 argument is LCMHIP_EPI_RELU | (LCMHIP_ACT_x << 8)                */
    LCMHIP_EPI_SWIGLU = 3,    /* out = (acc2 + bias2) * silu(acc + bias); W = gate, W2 = inner   */
    LCMHIP_EPI_RESID = 4      /* resid[m][n] += gate[n] * (acc + bias[n]); fp32, gate optional   */
};

typedef struct lcmhip_model lcmhip_model;
typedef struct lcmhip_session lcmhip_session;

/* One transformer stack (contextualizer or denoiser); lcm/nn/transformer/factory.py:47-123. */
typedef struct {
    int32_t num_layers;
    int32_t num_heads;     /* model_dim / num_heads must be 128 (so the count is even) */
    int32_t ffn_dim;       /* inner width actually allocated (after GLU rounding); a multiple of 128 (fp32) / 256 (bf16) */
    int32_t norm_kind;     /* LCMHIP_NORM_* */
    int32_t ffn_kind;      /* LCMHIP_FFN_*  */
    int32_t pos_kind;      /* LCMHIP_POS_*  */
    int32_t out_proj_bias; /* 0 / 1 */
    float rope_theta;
    int32_t qk_layernorm;  /* 0 / 1: LayerNorm (the tower's norm style, full model_dim) on q and on k after their
                              projections, before the head split and RoPE; lcm/nn/transformer/attention.py:143-199.
                              Tensors <attn>.q_layer_norm.* and <attn>.k_layer_norm.* */
    int32_t ffn_activation; /* LCMHIP_ACT_*: inner activation when ffn_kind == LCMHIP_FFN_RELU (0 = ReLU) */
    int32_t norm_order;     /* LCMHIP_ORDER_* of the layers (context stack; the denoiser takes PRE only) */
    int32_t no_final_norm;  /* 1: the stack does not end with a LayerNorm -- final_norm_order_style (or, when that is
                               None, norm_order_style) is "post"; lcm/models/two_tower_diffusion_lcm/builder.py:583-595 */
    int32_t scale_residual; /* 1: the FFN block's residual is multiplied by the learned vector <layer>.residual_scale
                               (model_dim) before the branch is added (factory.py:120-121, 172-184); context stack */
} lcmhip_tower_desc;

/* lcm/models/two_tower_diffusion_lcm/builder.py:49-81 (TwoTowerDiffusionLCModelConfig). */
typedef struct {
    int32_t sonar_dim;         /* 1024 for SONAR; any multiple of 128 (fp32) / 256 (bf16) up to 4096 */
    int32_t model_dim;         /* a multiple of 256 up to 8192 */
    int32_t max_seq_len;
    int32_t timestep_freq_dim; /* 256, lcm/nn/timestep_encoder.py:28; a multiple of 128 (fp32) / 256 (bf16) */
    int32_t dtype;             /* LCMHIP_F32 or LCMHIP_BF16: dtype of every bound weight and of GEMM operands */
    float norm_eps;
    lcmhip_tower_desc context;   /* two-tower: the contextualizer; base LCM: the causal decoder ("lcm") */
    lcmhip_tower_desc denoiser;  /* num_layers == 0 selects the base LCM (lcm/models/base_lcm/builder.py:64-124):
                                    frontend -> causal decoder -> postnet, tensor names frontend.*, lcm.*, postnet.fc.* */
    float embed_scale;           /* frontend input scale, lcm/models/base_lcm/frontend.py:91, 170; 0 means 1 */
    int32_t postnet_activation;  /* base LCM: activation after the output projection (ProjectionConfig.activation_name,
                                    lcm/nn/projection.py:73-86); 0 = none, otherwise 1 + LCMHIP_ACT_* */
} lcmhip_model_desc;

/* Options of DiffusionLCMGeneratorOptions that reach the arithmetic
 * (lcm/inference/two_tower_diffusion_lcm/generator.py:29-73) plus the scheduler config bits of
 * lcm/nn/schedulers/ddim.py:48-132 that `step` reads. */
typedef struct {
    float guidance_scale;
    float guidance_rescale;
    float ddim_eta;            /* > 0: stochastic DDIM (ddim.py:473-506); lcmhip_denoise then needs variance noise */
    float epsilon_scaling;     /* <= 0: disabled */
    float clip_noise;
    float init_noise_sigma;
    int32_t prediction_type;   /* LCMHIP_PRED_* configured for the model */
    int32_t clip_sample;       /* 0 / 1 */
    float clip_sample_range;
    int32_t thresholding;      /* 0 / 1: dynamic thresholding of the predicted x0 (ddim.py:242-280); wins over clip_sample */
    float dynamic_thresholding_ratio;
    float sample_max_value;
} lcmhip_denoise_params;

/* One row of lcmhip_session_profile(): time spent in one kernel family while profiling was on. */
typedef struct {
    char name[48];
    int64_t launches;
    double total_ms;
    double bytes;  /* algorithmic bytes moved, summed over launches */
    double flops;  /* algorithmic flops, summed over launches */
} lcmhip_profile_row;

const char* lcmhip_last_error(void);
int lcmhip_version(void);

/* ---- model ------------------------------------------------------------------------------ */

/* Replaces TwoTowerDiffusionLCModelBuilder.build_model (builder.py:532-557): creates an empty
 * weight table for `desc`. */
int lcmhip_model_create(const lcmhip_model_desc* desc, lcmhip_model** out);

/* Binds one state-dict tensor by its reference name (e.g.
 * "denoiser.layers.3.cross_attention.q_proj.weight"; names as produced by the modules of
 * builder.py:104-121).  `dtype` must equal desc.dtype.  Replaces load_state_dict of
 * lcm/models/base_lcm/loader.py:24-30 + fairseq2 StandardModelLoader. */
int lcmhip_model_bind_weight(lcmhip_model* m, const char* name, const void* ptr, int32_t dtype,
                             int32_t ndim, const int64_t* shape);

/* Checks that every tensor the configured architecture needs is bound with the right shape. */
int lcmhip_model_finalize(lcmhip_model* m);
int lcmhip_model_destroy(lcmhip_model* m);

/* ---- session: caches + scratch for one batch of documents --------------------------------- */

/* Options of a session (`flags` of the two calls below; both calls of one session must get the same value). */
#define LCMHIP_SESSION_KEEP_CONTEXT 1u  /* keep an fp32 copy of the contextualizer output for lcmhip_session_read_context */
#define LCMHIP_SESSION_SEPARATE_QKV 2u  /* q / k / v products of many rows as three launches, not one stacked launch */
#define LCMHIP_SESSION_SEPARATE_ROPE 4u /* rotation of q / k and the cache scatter as a pass of their own, not in the
                                           stacked product's epilogue (same roundings: same bits) */
#define LCMHIP_SESSION_NO_GRAPH 8u      /* the sampling loop is launched kernel by kernel, never replayed from a HIP graph */

int lcmhip_session_workspace_bytes(const lcmhip_model* m, int32_t max_batch, int32_t max_ctx,
                                   int32_t max_steps, uint32_t flags, size_t* bytes);

/* `workspace` (device, 256-byte aligned, `bytes` long) holds the contextualizer KV cache
 * (lcm/nn/incremental_state.py:14-43), the per-layer cross-attention K/V of the context, the
 * adaLN modulation table and all activations.  Replaces the LCMIncrementalStateBag pair of
 * generator.py:172-181.  The call writes its constant tables into the workspace before it returns, through a
 * stream of its own: no work still pending on any stream may touch the workspace when it is called. */
int lcmhip_session_create(lcmhip_model* m, int32_t max_batch, int32_t max_ctx, int32_t max_steps,
                          uint32_t flags, void* workspace, size_t bytes, lcmhip_session** out);
int lcmhip_session_destroy(lcmhip_session* s);

/* Replaces TwoTowerDiffusionLCModel.prep_for_denoising (builder.py:215-242) together with the
 * timestep-only parts of the denoiser: DiTTimestepEncoder.forward (lcm/nn/timestep_encoder.py:
 * 106-122) and AdaLNModulator.forward (lcm/nn/denoisers/lcm_denoiser.py:60-64) are evaluated
 * once per timestep into a table, and the unconditional rows' cross-attention output
 * (lcm_denoiser.py:505-513 with a zero-length context) into one vector per layer.
 * `timesteps` / `alphas_cumprod` are HOST arrays: the list of DDIMScheduler.set_timesteps
 * (ddim.py:333-347) and the table of ddim.py:190-191; `step_ratio` = T // S (ddim.py:421-423).
 * Synchronises `stream` before returning. */
int lcmhip_session_set_schedule(lcmhip_session* s, const int32_t* timesteps, int32_t n_steps,
                                const float* alphas_cumprod, int32_t num_train_steps,
                                float final_alpha_cumprod, int32_t step_ratio,
                                const lcmhip_denoise_params* params, void* stream);

/* Starts a new batch of `n_docs` documents: empty context, identity row->cache-slot map. */
int lcmhip_session_reset(lcmhip_session* s, int32_t n_docs, void* stream);

/* Replaces SonarNormalizer.normalize + TwoTowerDiffusionLCModel.encode with a state bag
 * (builder.py:161-191, 356-364; frontend.py:127-152; decoder.py:131-176; attention.py:201-298)
 * and the k_proj/v_proj of every denoiser cross-attention layer for the new positions
 * (lcm_denoiser.py:505-513).  `x` is (n_active, s, sonar_dim) fp32 SONAR embeddings for the next
 * `s` positions of every active document: raw when `normalized` == 0 (the normalizer is applied
 * while staging), already normalized otherwise (what TwoTowerDiffusionLCModel.encode receives). */
int lcmhip_context_append(lcmhip_session* s, const float* x, int32_t n_tokens, int32_t normalized,
                          void* stream);

/* Base LCM only.  Replaces BaseLCModel.predict_next_sentence (lcm/models/base_lcm/builder.py:126-167):
 * SonarNormalizer.normalize (skipped when `normalized` != 0 or the model has no normalizer), LCMFrontend.forward
 * (base_lcm/frontend.py:154-183), the KV-cached causal decoder (lcm/nn/transformer/decoder.py:131-176), the
 * postnet Projection (lcm/nn/projection.py:80-86), `+ temperature * noise` when `noise` is not NULL and
 * temperature > 0 (builder.py:158-160, with host-supplied N(0,1) draws instead of randn_like), and denormalize
 * (skipped under the same condition as normalize: `normalized` != 0 is BaseLCModel.forward, builder.py:94-124, which
 * has no normalizer on either side).
 * `x` is (n_active, n_tokens, sonar_dim) fp32: the next n_tokens positions of every active document, appended to
 * the session's KV cache.  `out` receives (n_active, n_tokens, sonar_dim) fp32 when all_positions != 0, else the
 * prediction after the last position only, (n_active, sonar_dim); `noise` has the shape of `out`.
 * out == NULL only fills the cache (the generator's prefill, lcm/inference/lcm/generator.py:226-246). */
int lcmhip_base_forward(lcmhip_session* s, const float* x, int32_t n_tokens, int32_t normalized,
                        const float* noise, float temperature, int32_t all_positions, float* out,
                        void* stream);

/* Replaces the sampling loop of predict_next_sentence[_with_cf_guidance] (builder.py:299-338,
 * 366-461): sample_initial_noise_vectors with host-supplied N(0,1) `noise` (n_active, sonar_dim)
 * fp32, LCMDenoiser.forward per timestep, DDIMScheduler.get_epsilon / step,
 * apply_classifier_free_guidance, latent clipping and SonarNormalizer.denormalize.
 * `out` (n_active, sonar_dim) fp32 receives the next sentence embedding.  If `latents_trace`
 * is not NULL it receives prev_sample after every step, (n_steps, n_active, sonar_dim) fp32. */
int lcmhip_denoise(lcmhip_session* s, const float* noise, float* out, float* latents_trace,
                   void* stream);

/* lcmhip_denoise for a schedule with ddim_eta > 0 (DDIMScheduler.step, ddim.py:473-506: prev_sample +=
 * eta * sqrt(variance_t) * variance_noise).  `variance_noise` (n_steps, n_active, sonar_dim) fp32 holds the N(0, 1)
 * draw of every executed step (the reference draws it inside `step` with randn_tensor); it may be NULL when
 * ddim_eta == 0, where this call equals lcmhip_denoise.  Runs launch by launch (no graph replay). */
int lcmhip_denoise_stochastic(lcmhip_session* s, const float* noise, const float* variance_noise, float* out,
                              float* latents_trace, void* stream);

/* The denoiser over every position of every active document in one pass: the denoising half of
 * TwoTowerDiffusionLCModel.forward (builder.py:128-159, 193-213) = LCMDenoiser.forward with inference=False
 * (lcm_denoiser.py:152-251): diagonal self-attention mask (attention_masks.py:82-113) and the
 * ShiftedCausalAttentionMask over [zero vector, context] (attention_masks.py:158-228, cf_guidance_prob = 0).
 * The context half is lcmhip_context_append on a freshly reset session (the whole clean batch at once).
 * `noisy` (n_active, n_positions, sonar_dim) fp32: noised latents in normalized space;
 * `step_index` (n_active, n_positions) int32 DEVICE array: for every position the index of its diffusion timestep
 * in the timestep list last given to lcmhip_session_set_schedule (whose adaLN table is reused; out-of-range
 * indices are clamped); `source_lengths` (n_active) int32 device array or NULL (= 0): position j sees the zero
 * vector and context positions [0, j + source_length); `context_lengths` (n_active) int32 device array or NULL:
 * valid context positions per document (key padding mask; NULL = all appended positions).
 * `out` (n_active, n_positions, sonar_dim) fp32 receives the model output (normalized space, no scheduler step). */
int lcmhip_denoise_positions(lcmhip_session* s, const float* noisy, const int32_t* step_index,
                             int32_t n_positions, const int32_t* source_lengths,
                             const int32_t* context_lengths, float* out, void* stream);

/* Replaces TwoTowerDiffusionLCMGenerator.reorder_state (generator.py:442-466) for the caches:
 * afterwards active row i is the row that was `order[i]` (HOST array, strictly increasing). No
 * cache bytes move. */
int lcmhip_session_reorder(lcmhip_session* s, const int32_t* order, int32_t n, void* stream);

int lcmhip_session_context_len(const lcmhip_session* s);
int lcmhip_session_num_active(const lcmhip_session* s);

/* Copies the contextualizer output of active row `row`, positions [0, context_len), as fp32
 * (context_len, model_dim) into `out` (device).  Only kept when the session was created with
 * LCMHIP_SESSION_KEEP_CONTEXT; otherwise returns LCMHIP_ERR_STATE. */
int lcmhip_session_read_context(lcmhip_session* s, int32_t row, float* out, void* stream);

/* Per-kernel timing with HIP events on `stream`, for roofline accounting.  While enabled, every
 * launch is bracketed by an event pair (slower end-to-end; never enable inside a timed region
 * whose wall time is reported).  lcmhip_session_profile synchronises the stream, fills at most
 * `max_rows` rows and returns the row count through `n_rows`; it also resets the counters. */
int lcmhip_session_profile_enable(lcmhip_session* s, int32_t on);
int lcmhip_session_profile(lcmhip_session* s, lcmhip_profile_row* rows, int32_t max_rows,
                           int32_t* n_rows, void* stream);

/* ---- single kernels (unit parity tests and roofline microbenchmarks) ----------------------- */

/* SonarNormalizer.normalize / denormalize, lcm/models/sonar_normalizer/builder.py:121-137.
 * x, y: (rows, dim) in `dtype`; center/scale: (dim,) in `dtype`; clip_min/clip_max optional
 * (dim,) or NULL (builder.py:126-127, 131-132). */
int lcmhip_normalize(const void* x, const void* center, const void* scale, const void* clip_min,
                     const void* clip_max, void* y, int64_t rows, int32_t dim, int32_t dtype,
                     void* stream);
int lcmhip_denormalize(const void* x, const void* center, const void* scale, const void* clip_min,
                       const void* clip_max, void* y, int64_t rows, int32_t dim, int32_t dtype,
                       void* stream);

/* FFTInterface.fft_transform / fft_inverse_transform of the "base_fft" / "clipping_fft" normalizers
 * (lcm/models/sonar_normalizer/builder.py:58-84, archs.py:28-40): the transform a with_fft normalizer applies to the raw
 * embeddings before center / scale / clip, and its inverse after them.  x, y: (rows, dim) fp32, dim a power of two in
 * [64, 4096]; y may equal x.  Forward (inverse = 0): X = rfft(x) without scaling, packed as
 * [Re X_0 .. Re X_{dim/2}, Im X_1 .. Im X_{dim/2-1}]; inverse (1): irfft of that packing (Im X_0 = Im X_{dim/2} = 0). */
int lcmhip_rfft_pack(const float* x, float* y, int64_t rows, int32_t dim, int32_t inverse, void* stream);

/* Layer/RMS norm (lcm/nn/normalization.py:73-88) fused with the adaLN modulation
 * `h * (1 + scale) + shift` of lcm_denoiser.py:449-452.  x: (rows, dim) fp32; weight/bias:
 * (dim,) in `dtype` (bias NULL for RMS); shift/scale: (dim,) fp32 or NULL; y: (rows, dim) in
 * `dtype`. */
int lcmhip_norm_modulate(const float* x, const void* weight, const void* bias, const float* shift,
                         const float* scale, void* y, int32_t rows, int32_t dim, int32_t norm_kind,
                         float eps, int32_t dtype, void* stream);

/* C = A * W^T with a fused epilogue (LCMHIP_EPI_*); the Linear layers of fairseq2 used at
 * attention.py:151-185, factory.py:247-254, 277-300, projection.py:80-86, frontend.py:139.
 * A: (M, K) lda; W, W2: (N, K); bias, bias2: (N,) in `dtype` or NULL; gate: (N,) fp32 or NULL;
 * out: (M, N) ldo -- operand dtype for STORE/RELU/SWIGLU, fp32 for STORE_F32/RESID.
 * K must cover whole 512-byte chunks of operands (256 bf16 / 128 f32 elements), N a multiple of
 * 64.  `splitk_ws` (fp32, `splitk_ws_bytes` long) is optional scratch: with it the K range may be
 * split over workgroups (partial tiles summed by a finishing pass; deterministic). */
int lcmhip_gemm(const void* A, int32_t lda, const void* W, const void* W2, const void* bias,
                const void* bias2, const float* gate, void* out, int32_t ldo, int32_t M, int32_t N,
                int32_t K, int32_t epilogue, int32_t dtype, float* splitk_ws, size_t splitk_ws_bytes,
                void* stream);

/* lcmhip_gemm without the workspace argument, for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.
 * The split-K slab lives in a static device array of LCMHIP_STATIC_SCRATCH_BYTES that each GPU the library runs
 * on holds until the process ends; the call allocates and frees nothing.  M * N must not exceed
 * LCMHIP_STATIC_MAX_OUTPUTS (two fp32 partial tiles must fit), otherwise LCMHIP_ERR_INVALID.  All calls on one
 * GPU share that GPU's array: calls that overlap on one GPU (different streams or threads) race; calls on one
 * stream are ordered.  Nothing in the array has to be zero: every element a call reads it has written first.
 * Results equal lcmhip_gemm's with a workspace of LCMHIP_STATIC_SCRATCH_BYTES bit for bit.
 * The array's address on a GPU is resolved once per GPU, which loads its code object and places the array: that
 * first resolution must happen OUTSIDE any stream capture.  lcmhip_static_init() does it for the current device
 * (lcmhip_model_create and lcmhip_session_create call it too); a lcmhip_gemm_static call on a device that has seen
 * none of them resolves it itself and so must not be the first thing inside a capture. */
#define LCMHIP_STATIC_SCRATCH_BYTES (16u << 20)
#define LCMHIP_STATIC_MAX_OUTPUTS (LCMHIP_STATIC_SCRATCH_BYTES / 8u)
int lcmhip_static_init(void);
int lcmhip_gemm_static(const void* A, int32_t lda, const void* W, const void* W2, const void* bias,
                       const void* bias2, const float* gate, void* out, int32_t ldo, int32_t M, int32_t N,
                       int32_t K, int32_t epilogue, int32_t dtype, void* stream);

/* The residual update + pre-norm of one denoiser sub-block, fed from a split-K slab
 * (lcm_denoiser.py:445-473 / 496-521 / 527-546: `seqs = residual + gate * block(...)` followed
 * by the next block's `layer_norm(seqs) * (1 + scale) + shift`).
 * First  x[r] = (init ? 0 : x[r]) + gate * (sum_s slab[s][src(r)] + bias)  for rows r < M
 * (src(r) = r) and, for rows M <= r < rows_total, either the same with src(r) = r - M (dup != 0)
 * or x[r] += gate * extra_const (dup == 0); then rows [0, rows_norm) are normalised like
 * lcmhip_norm_modulate into y.  slab: (splits, M, dim) fp32 -- the output of lcmhip_gemm_partial. */
int lcmhip_residual_norm(float* x, const float* slab, int32_t splits, int32_t M, const void* bias,
                         const float* gate, const float* extra_const, int32_t dup, int32_t init,
                         const void* weight, const void* norm_bias, const float* shift,
                         const float* scale, void* y, int32_t rows_total, int32_t rows_norm,
                         int32_t dim, int32_t norm_kind, float eps, int32_t dtype, void* stream);

/* Split-K partial products only: slab (splits, M, N) fp32 without bias; `*splits` is chosen by
 * the library (<= slab_bytes / (M * N * 4)). */
int lcmhip_gemm_partial(const void* A, int32_t lda, const void* W, float* slab, size_t slab_bytes,
                        int32_t* splits, int32_t M, int32_t N, int32_t K, int32_t dtype, void* stream);

/* One DDIM update for a batch of rows with optional classifier-free guidance:
 * DDIMScheduler.get_epsilon (ddim.py:561-587), apply_classifier_free_guidance
 * (builder.py:463-488), DDIMScheduler.step (ddim.py:358-510, eta = 0) and the latent clip of
 * builder.py:452-454.  sample: (rows, dim) fp32 x_t; out_cond / out_uncond: (rows, dim) fp32
 * model outputs (out_uncond NULL: no guidance, `step` is applied to out_cond with the model's
 * prediction type); prev_sample, pred_x0: (rows, dim) fp32; pred_x0 may be NULL when only prev_sample is wanted.  `variance_noise` (rows, dim) fp32 is required when
 * params->ddim_eta > 0 and ignored otherwise.  With params->thresholding the predicted x0 of every row is divided
 * by s = clamp(quantile(|x0|, ratio), 1, sample_max_value) after clamping to [-s, s] (torch.quantile's linear
 * interpolation). */
int lcmhip_ddim_cfg_step(const float* sample, const float* out_cond, const float* out_uncond,
                         float* prev_sample, float* pred_x0, int32_t rows, int32_t dim,
                         float alpha_prod_t, float alpha_prod_t_prev,
                         const lcmhip_denoise_params* params, const float* variance_noise, void* stream);

/* Roofline calibration, no counterpart in the reference: copies `bytes` (a multiple of 16) from src to dst with 16
 * bytes per lane, four pieces in flight per lane and a grid sized to the compute units; nontemporal != 0 uses
 * non-temporal loads and stores.  The rate it reaches is the yardstick the fused elementwise kernels are held to. */
int lcmhip_stream_copy(const void* src, void* dst, int64_t bytes, int32_t nontemporal, void* stream);

/* The tensor work of finalize_step (lcm/inference/two_tower_diffusion_lcm/generator.py:305-351, 403; the base
 * generator's lcm/inference/lcm/generator.py:288-330 is the same): for active row r of document doc_rows[r] (NULL:
 * doc = r), a document still inside its prompt (prompt_lens[doc] > step; prompt_lens NULL: none) keeps the prompt
 * embedding already stored at seqs[doc][step] (generator.py:322-328), every other row stores new_emb[r] there in the
 * buffer's dtype (:403).  cos_out (n_active, 2) fp32 receives the two cosines of the stopping rules for the value
 * kept: against eos_vec (:331-336; 0 when eos_vec is NULL) and against the previous sentence seqs[doc][step - 1]
 * (:339-351; 0 at step 0).  new_emb (n_active, dim) fp32; seqs (n_docs, total_len, dim) in seqs_dtype;
 * doc_rows, prompt_lens: int32 device arrays; eos_vec (dim) fp32.  The thresholds are applied by the host. */
int lcmhip_finalize_step(const float* new_emb, void* seqs, int32_t seqs_dtype, const int32_t* doc_rows,
                         const int32_t* prompt_lens, int32_t n_active, int32_t step, int32_t total_len, int32_t dim,
                         const float* eos_vec, float* cos_out, void* stream);

/* Single-query-block attention over a per-document key/value cache: fairseq2 TorchSDPA as
 * called at attention.py:272-279 (scale 1/sqrt(128), -inf on masked keys).
 * q: (n_docs * s, H * 128) in `dtype`; k_cache / v_cache: (n_slots, H, cap, 128) -- the keys of one
 * (document, head) are contiguous;
 * slot_map: (n_docs,) int32 or NULL (identity); query j of a document sees keys
 * [0, base_keys + (causal ? j + 1 : 0)).  out: like q. */
int lcmhip_attention(const void* q, const void* k_cache, const void* v_cache, const int32_t* slot_map,
                     void* out, int32_t n_docs, int32_t s, int32_t n_heads, int32_t cap,
                     int32_t base_keys, int32_t causal, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LCMHIP_H */
