/* A caller of the C ABI with nothing but the HIP runtime around it (no Python, no torch): reads a case file written by
 * tests/test_gpu_c_caller.py, builds the model from the tensors in it, runs
 *   reset -> context_append(prompt) -> denoise -> context_append(that sentence) -> denoise
 * (a base LCM, i.e. a description without a denoiser tower: reset -> base_forward(prompt) -> base_forward(that sentence))
 * and writes the two generated sentences to the output file.  The test compares them bit for bit with what the Python
 * host gets from the same library calls.  With a thread count as third argument that many workers run the flow at the
 * same time, three times each, every worker with a session and a stream of its own on the one model; all must produce
 * the same bits.
 *
 * Case file (little endian, no padding):
 *   int32 magic 0x4c434d31
 *   int32 desc_bytes,   lcmhip_model_desc
 *   int32 params_bytes, lcmhip_denoise_params
 *   int32 n_steps, int32 timesteps[n_steps]
 *   int32 num_train_steps, float alphas_cumprod[num_train_steps], float final_alpha_cumprod, int32 step_ratio
 *   int32 n_docs, int32 prompt_len, int32 max_ctx
 *   float prompts[n_docs * prompt_len * sonar_dim], float noise[2 * n_docs * sonar_dim]
 *   int32 n_tensors, then per tensor: int32 name_len, char name[name_len], int32 ndim, int64 shape[ndim],
 *                                      int64 nbytes, bytes
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lcmhip.h"

#define HIP_OK(call)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            exit(3);                                                                            \
        }                                                                                       \
    } while (0)
#define LCM_OK(call)                                                                            \
    do {                                                                                        \
        int r_ = (call);                                                                        \
        if (r_ != LCMHIP_OK) {                                                                  \
            fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, r_, lcmhip_last_error()); \
            exit(4);                                                                            \
        }                                                                                       \
    } while (0)

static void read_exact(FILE* f, void* dst, size_t n) {
    if (n && fread(dst, 1, n, f) != n) {
        fprintf(stderr, "case file is truncated\n");
        exit(2);
    }
}
static int32_t read_i32(FILE* f) {
    int32_t v;
    read_exact(f, &v, sizeof v);
    return v;
}
static void* xmalloc(size_t n) {
    void* p = malloc(n ? n : 1);
    if (!p) {
        fprintf(stderr, "out of host memory\n");
        exit(2);
    }
    return p;
}
static void* to_device(const void* host, size_t n) {
    void* d = NULL;
    HIP_OK(hipMalloc(&d, n ? n : 4));
    if (n) HIP_OK(hipMemcpy(d, host, n, hipMemcpyHostToDevice));
    return d;
}

#define MAX_THREADS 16

/* One worker's share: reset -> prefill -> sampling loop -> append -> sampling loop, `repeats` times on a fresh session
 * each (so sessions and their recorded graphs are created and destroyed while other workers run theirs).  Copies go
 * through the worker's own stream: the legacy stream's copies are refused while another thread records a graph. */
struct flow {
    lcmhip_model* model;
    const lcmhip_denoise_params* params;
    const int32_t* timesteps;
    const float* alphas;
    int32_t n_steps, n_train, step_ratio, n_docs, prompt_len, max_ctx, repeats, base;
    float final_alpha;
    size_t D;
    const float* d_prompts;
    const float* d_noise;
    float* out;         /* host, 2 * n_docs * D */
};

static void* run_flow(void* arg) {
    struct flow* w = (struct flow*)arg;
    const size_t row = (size_t)w->n_docs * w->D;
    HIP_OK(hipSetDevice(0));
    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    float* d_out = NULL;
    HIP_OK(hipMalloc((void**)&d_out, 2 * row * sizeof(float)));
    size_t ws_bytes = 0;
    LCM_OK(lcmhip_session_workspace_bytes(w->model, w->n_docs, w->max_ctx, w->n_steps, 0u, &ws_bytes));
    void* workspace = NULL;
    HIP_OK(hipMalloc(&workspace, ws_bytes));           /* hipMalloc returns memory aligned well beyond 256 bytes */
    for (int r = 0; r < w->repeats; ++r) {
        lcmhip_session* session = NULL;
        LCM_OK(lcmhip_session_create(w->model, w->n_docs, w->max_ctx, w->n_steps, 0u, workspace, ws_bytes, &session));
        LCM_OK(lcmhip_session_reset(session, w->n_docs, stream));
        if (w->base) {
            /* base LCM: the prediction after the prompt, sampled with the case's noise at temperature 0.5, then that
             * sentence as the next position (n_docs, 1, sonar_dim) */
            LCM_OK(lcmhip_base_forward(session, w->d_prompts, w->prompt_len, 0, w->d_noise, 0.5f, 0, d_out, stream));
            LCM_OK(lcmhip_base_forward(session, d_out, 1, 0, w->d_noise + row, 0.5f, 0, d_out + row, stream));
        } else {
            LCM_OK(lcmhip_session_set_schedule(session, w->timesteps, w->n_steps, w->alphas, w->n_train, w->final_alpha,
                                               w->step_ratio, w->params, stream));
            LCM_OK(lcmhip_context_append(session, w->d_prompts, w->prompt_len, 0, stream));
            LCM_OK(lcmhip_denoise(session, w->d_noise, d_out, NULL, stream));
            /* the generated sentence becomes the next context position (n_docs, 1, sonar_dim) */
            LCM_OK(lcmhip_context_append(session, d_out, 1, 0, stream));
            LCM_OK(lcmhip_denoise(session, w->d_noise + row, d_out + row, NULL, stream));
        }
        HIP_OK(hipStreamSynchronize(stream));
        if (lcmhip_session_context_len(session) != w->prompt_len + 1 || lcmhip_session_num_active(session) != w->n_docs) {
            fprintf(stderr, "session reports context %d, %d active\n", lcmhip_session_context_len(session),
                    lcmhip_session_num_active(session));
            exit(5);
        }
        LCM_OK(lcmhip_session_destroy(session));       /* drains the stream its graphs last ran on */
    }
    HIP_OK(hipMemcpyAsync(w->out, d_out, 2 * row * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipFree(workspace));
    HIP_OK(hipFree(d_out));
    HIP_OK(hipStreamDestroy(stream));
    return NULL;
}

int main(int argc, char** argv) {
    if (argc != 3 && argc != 4) {
        fprintf(stderr, "usage: %s <case file> <output file> [threads]\n", argv[0]);
        return 2;
    }
    const int n_threads = argc == 4 ? atoi(argv[3]) : 1;
    if (n_threads < 1 || n_threads > MAX_THREADS) {
        fprintf(stderr, "1 to %d threads\n", MAX_THREADS);
        return 2;
    }
    if (lcmhip_version() != LCMHIP_ABI_VERSION) {
        fprintf(stderr, "library ABI %d, header ABI %d\n", lcmhip_version(), LCMHIP_ABI_VERSION);
        return 2;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 2;
    }
    if (read_i32(f) != 0x4c434d31) {
        fprintf(stderr, "not a case file\n");
        return 2;
    }
    lcmhip_model_desc desc;
    lcmhip_denoise_params params;
    if (read_i32(f) != (int32_t)sizeof desc) {
        fprintf(stderr, "lcmhip_model_desc has another size in the case file\n");
        return 2;
    }
    read_exact(f, &desc, sizeof desc);
    if (read_i32(f) != (int32_t)sizeof params) {
        fprintf(stderr, "lcmhip_denoise_params has another size in the case file\n");
        return 2;
    }
    read_exact(f, &params, sizeof params);
    const int32_t n_steps = read_i32(f);
    int32_t* timesteps = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)n_steps);
    read_exact(f, timesteps, sizeof(int32_t) * (size_t)n_steps);
    const int32_t n_train = read_i32(f);
    float* alphas = (float*)xmalloc(sizeof(float) * (size_t)n_train);
    read_exact(f, alphas, sizeof(float) * (size_t)n_train);
    float final_alpha;
    read_exact(f, &final_alpha, sizeof final_alpha);
    const int32_t step_ratio = read_i32(f);
    const int32_t n_docs = read_i32(f), prompt_len = read_i32(f), max_ctx = read_i32(f);
    const size_t D = (size_t)desc.sonar_dim;
    const size_t prompt_bytes = sizeof(float) * (size_t)n_docs * (size_t)prompt_len * D;
    const size_t row_bytes = sizeof(float) * (size_t)n_docs * D;
    float* prompts = (float*)xmalloc(prompt_bytes);
    read_exact(f, prompts, prompt_bytes);
    float* noise = (float*)xmalloc(2 * row_bytes);
    read_exact(f, noise, 2 * row_bytes);

    HIP_OK(hipSetDevice(0));

    lcmhip_model* model = NULL;
    LCM_OK(lcmhip_model_create(&desc, &model));
    const int32_t n_tensors = read_i32(f);
    void** tensors = (void**)xmalloc(sizeof(void*) * (size_t)n_tensors);
    for (int32_t i = 0; i < n_tensors; ++i) {
        const int32_t name_len = read_i32(f);
        char* name = (char*)xmalloc((size_t)name_len + 1);
        read_exact(f, name, (size_t)name_len);
        name[name_len] = 0;
        const int32_t ndim = read_i32(f);
        int64_t shape[8];
        if (ndim < 0 || ndim > 8) {
            fprintf(stderr, "%s: %d dimensions\n", name, ndim);
            return 2;
        }
        read_exact(f, shape, sizeof(int64_t) * (size_t)ndim);
        int64_t nbytes;
        read_exact(f, &nbytes, sizeof nbytes);
        void* host = xmalloc((size_t)nbytes);
        read_exact(f, host, (size_t)nbytes);
        tensors[i] = to_device(host, (size_t)nbytes);
        free(host);
        LCM_OK(lcmhip_model_bind_weight(model, name, tensors[i], desc.dtype, ndim, shape));
        free(name);
    }
    fclose(f);
    LCM_OK(lcmhip_model_finalize(model));

    float* d_prompts = (float*)to_device(prompts, prompt_bytes);
    float* d_noise = (float*)to_device(noise, 2 * row_bytes);

    /* every worker: a session, a stream and an output of its own on the shared model and inputs */
    struct flow flows[MAX_THREADS];
    pthread_t threads[MAX_THREADS];
    for (int t = 0; t < n_threads; ++t) {
        struct flow* w = &flows[t];
        w->model = model; w->params = &params; w->timesteps = timesteps; w->n_steps = n_steps; w->alphas = alphas;
        w->n_train = n_train; w->final_alpha = final_alpha; w->step_ratio = step_ratio; w->n_docs = n_docs;
        w->prompt_len = prompt_len; w->max_ctx = max_ctx; w->D = D; w->d_prompts = d_prompts; w->d_noise = d_noise;
        w->repeats = n_threads > 1 ? 3 : 1;
        w->base = desc.denoiser.num_layers == 0;
        w->out = (float*)xmalloc(2 * row_bytes);
    }
    if (n_threads == 1) {
        run_flow(&flows[0]);
    } else {
        for (int t = 0; t < n_threads; ++t)
            if (pthread_create(&threads[t], NULL, run_flow, &flows[t]) != 0) {
                fprintf(stderr, "pthread_create failed\n");
                return 2;
            }
        for (int t = 0; t < n_threads; ++t) pthread_join(threads[t], NULL);
    }
    for (int t = 1; t < n_threads; ++t)
        if (memcmp(flows[t].out, flows[0].out, 2 * row_bytes) != 0) {
            fprintf(stderr, "thread %d produced other bits than thread 0\n", t);
            return 6;
        }
    FILE* g = fopen(argv[2], "wb");
    if (!g || fwrite(flows[0].out, 1, 2 * row_bytes, g) != 2 * row_bytes || fclose(g) != 0) {
        perror(argv[2]);
        return 2;
    }

    LCM_OK(lcmhip_model_destroy(model));
    HIP_OK(hipFree(d_prompts));
    HIP_OK(hipFree(d_noise));
    for (int32_t i = 0; i < n_tensors; ++i) HIP_OK(hipFree(tensors[i]));
    for (int t = 0; t < n_threads; ++t) free(flows[t].out);
    free(tensors);
    free(noise);
    free(prompts);
    free(alphas);
    free(timesteps);
    printf("ok %d docs, context %d, %d steps, %d thread%s\n", n_docs, prompt_len + 1, n_steps, n_threads,
           n_threads == 1 ? "" : "s");
    fflush(stdout);
    return 0;
}
