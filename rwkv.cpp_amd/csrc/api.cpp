// api.cpp -- the C ABI of librwkv.so: the reference's rwkv.h entry points (reference rwkv.cpp:71-258, rwkv_eval.inc:38-241)
// re-implemented on the device-resident engine, plus the opt-in rwkv_mi_* extensions (include/rwkv_mi355x.h).
#include "model.h"
#include "rwkv_mi355x.h"

#include <cinttypes>
#include <cstdlib>
#include <cstring>
#include <string>

using namespace rwkvmi;

#define HIP_CTX_OK(CTX, CALL) \
    do { hipError_t e_ = (CALL); RW_CTX_CHECK((CTX), RWKV_ERROR_GRAPH, false, e_ == hipSuccess, "HIP error: %s", hipGetErrorString(e_)); } while (0)

// Sequence calls are cut into pieces of at most this many tokens internally (bounds scratch memory; results do not
// depend on the cut because every kernel is per-token order-preserving).
static const size_t k_max_tokens_per_pass = 1024;

static bool upload_tokens(rwkv_context * ctx, const uint32_t * tokens, size_t n) {
    if ((int64_t) n > ctx->d_tokens_cap) {
        HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_tokens) (void) hipFree(ctx->d_tokens);
        if (ctx->h_tokens) (void) hipHostFree(ctx->h_tokens);
        ctx->d_tokens = nullptr; ctx->h_tokens = nullptr; ctx->d_tokens_cap = 0;
        size_t cap = n < 64 ? 64 : n;
        HIP_CTX_OK(ctx, hipMalloc((void **) &ctx->d_tokens, cap * sizeof(uint32_t)));
        HIP_CTX_OK(ctx, hipHostMalloc((void **) &ctx->h_tokens, cap * sizeof(uint32_t), hipHostMallocDefault));
        ctx->d_tokens_cap = (int64_t) cap;
        // captured graphs hold the old token pointer
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) if (ctx->graph_exec[a][b]) { (void) hipGraphExecDestroy(ctx->graph_exec[a][b]); ctx->graph_exec[a][b] = nullptr; }
    }
    // the previous pass may still be reading h_tokens through an in-flight copy
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(ctx->h_tokens, tokens, n * sizeof(uint32_t));
    HIP_CTX_OK(ctx, hipMemcpyAsync(ctx->d_tokens, ctx->h_tokens, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    return true;
}

namespace rwkvmi { bool upload_tokens_for(rwkv_context * ctx, const uint32_t * tokens, size_t n) { return upload_tokens(ctx, tokens, n); } }

// Runs tokens[0..n) from the device-resident state; logits (of the last token) stay in ctx->d_logits.
static bool run_tokens(rwkv_context * ctx, const uint32_t * tokens, size_t n, bool want_logits) {
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    size_t done = 0;
    while (done < n) {
        const size_t step = (n - done) < k_max_tokens_per_pass ? (n - done) : k_max_tokens_per_pass;
        const bool last = done + step == n;
        if (!upload_tokens(ctx, tokens + done, step)) return false;
        const bool ok = (step == 1) ? forward_decode(ctx, want_logits && last) : forward(ctx, (int64_t) step, want_logits && last);
        if (!ok) return false;
        done += step;
    }
    return true;
}

static const char * k_abort_msg = "persistent decode kernel timed out waiting for a workgroup (is the GPU shared with another process?); "
                                  "the context continues on the per-layer launches";

// Copies the requested outputs and drains the stream. *aborted (when given) reports a poll time-out of the persistent kernel
// since the last check instead of failing: the caller repeats the step on the per-layer path (rwkv_eval). The abort word
// travels with the other copies through a pinned mirror, no extra device round trip.
static bool fetch_outputs(rwkv_context * ctx, float * state_out, float * logits_out, bool * aborted = nullptr) {
    if (aborted) *aborted = false;
    if (state_out && !state_to_host(ctx, state_out)) return false;
    if (logits_out) HIP_CTX_OK(ctx, hipMemcpyAsync(logits_out, ctx->d_logits, (size_t) ctx->model->n_vocab() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    // (a control-word copy that could not even be enqueued leaves a stale "not aborted" mirror: treated as an abort)
    const bool ctl_ok = !ctx->mega || mega_v6_ctl_fetch(ctx->mega, ctx->stream);
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->mega && (!ctl_ok || mega_v6_aborted_cached(ctx->mega))) {
        recover_from_abort(ctx);
        if (aborted) { *aborted = true; return true; }
        RW_CTX_CHECK(ctx, RWKV_ERROR_GRAPH, false, false, "%s", k_abort_msg);
    }
    return true;
}

extern "C" {

RWKV_API struct rwkv_context * rwkv_init_from_file(const char * file_path, const uint32_t n_threads, const uint32_t n_gpu_layers) {
    (void) n_gpu_layers;  // every layer runs on the GPU; the layer pipeline (rwkv_mi_init_stage) supersedes partial offload
    g_last_error = RWKV_ERROR_NONE;
    RW_CHECK(RWKV_ERROR_ARGS, nullptr, file_path != nullptr, "model_file_path is NULL");
    // RWKV_MI_DEVICES=0,1,...: a layer pipeline over the listed devices behind this same ABI (pipeline.cpp); supersedes n_gpu_layers
    if (const char * devs = getenv("RWKV_MI_DEVICES")) { if (devs[0]) return pipeline_create(file_path, n_threads, devs); }
    Model * m = load_model(file_path, 0, UINT32_MAX);
    if (!m) return nullptr;
    rwkv_context * ctx = create_context(m, n_threads);
    return ctx;  // on failure create_context has already released the model
}

RWKV_API struct rwkv_context * rwkv_clone_context(struct rwkv_context * ctx, const uint32_t n_threads) {
    RW_CHECK(RWKV_ERROR_ARGS, nullptr, ctx != nullptr, "ctx is NULL");
    if (!ctx->stages.empty()) return pipeline_clone(ctx, n_threads);
    (void) hipSetDevice(ctx->model->device);
    rwkv_context * clone = create_context(ctx->model, n_threads);
    if (clone) clone->print_errors = ctx->print_errors;
    return clone;
}

RWKV_API bool rwkv_eval(struct rwkv_context * ctx, const uint32_t token, const float * state_in, float * state_out, float * logits_out) {
    ctx->last_error = RWKV_ERROR_NONE;
    const size_t n_vocab = (size_t) ctx->model->n_vocab();
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, token < n_vocab, "Token (%" PRId32 ") is out of range (0 .. %zu)", (int32_t) token, n_vocab - 1);
    if (!ctx->stages.empty()) return pipeline_eval(ctx, &token, 1, 1, state_in, state_out, logits_out);
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    bool aborted = false;
    if ((state_in || state_out) && forward_streamed_eligible(ctx)) {
        // the state slices travel under the layers of the other groups (engine.hip, forward_streamed)
        if (!upload_tokens(ctx, &token, 1)) return false;
        if (!forward_streamed(ctx, logits_out != nullptr, state_in, state_out, logits_out, &aborted)) return false;
        if (!aborted) return true;
        // (poll time-out: the input state is complete on the device in the buffer the step read -- repeat on the per-layer launches)
        ctx->cur ^= 1;
        if (!run_tokens(ctx, &token, 1, logits_out != nullptr)) return false;
        return fetch_outputs(ctx, state_out, logits_out);
    }
    if (!state_from_host(ctx, state_in)) return false;
    if (!run_tokens(ctx, &token, 1, logits_out != nullptr)) return false;
    if (!fetch_outputs(ctx, state_out, logits_out, &aborted)) return false;
    if (aborted) {
        // The persistent kernel gave up (device shared with another process' persistent kernel). It only writes the OTHER state
        // buffer, so the input state is intact: flip back and repeat the token on the per-layer launches.
        ctx->cur ^= 1;
        if (!run_tokens(ctx, &token, 1, logits_out != nullptr)) return false;
        return fetch_outputs(ctx, state_out, logits_out);
    }
    return true;
}

RWKV_API bool rwkv_eval_sequence(struct rwkv_context * ctx, const uint32_t * sequence, const size_t sequence_len,
                                 const float * state_in, float * state_out, float * logits_out) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, sequence_len > 0, "Sequence length is 0");
    if (sequence) {
        const size_t n_vocab = (size_t) ctx->model->n_vocab();
        for (size_t i = 0; i < sequence_len; i++) {
            RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, sequence[i] < n_vocab, "Token at index %zu (%" PRId32 ") is out of range (0 .. %zu)",
                         i, (int32_t) sequence[i], n_vocab - 1);
        }
    } else {
        // "prepare only": the reference builds and caches the graph for this length; nothing to build here.
        return true;
    }
    if (!ctx->stages.empty()) return pipeline_eval(ctx, sequence, sequence_len, k_max_tokens_per_pass, state_in, state_out, logits_out);
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    if (!state_from_host(ctx, state_in)) return false;
    if (!run_tokens(ctx, sequence, sequence_len, logits_out != nullptr)) return false;
    return fetch_outputs(ctx, state_out, logits_out);
}

RWKV_API bool rwkv_eval_sequence_in_chunks(struct rwkv_context * ctx, const uint32_t * tokens, const size_t sequence_len, const size_t chunk_size,
                                           const float * state_in, float * state_out, float * logits_out) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, sequence_len > 0, "Sequence length is 0");
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, chunk_size > 0, "Chunk size is 0");
    if (!tokens) return true;
    const size_t n_vocab = (size_t) ctx->model->n_vocab();
    for (size_t i = 0; i < sequence_len; i++) {
        RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, tokens[i] < n_vocab, "Token at index %zu (%" PRId32 ") is out of range (0 .. %zu)",
                     i, (int32_t) tokens[i], n_vocab - 1);
    }
    if (!ctx->stages.empty()) return pipeline_eval(ctx, tokens, sequence_len, chunk_size, state_in, state_out, logits_out);
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    if (!state_from_host(ctx, state_in)) return false;
    // The state stays in HBM between chunks; only the final chunk produces logits (reference rwkv_eval.inc:183-218).
    size_t done = 0;
    while (done < sequence_len) {
        const size_t step = (sequence_len - done) < chunk_size ? (sequence_len - done) : chunk_size;
        const bool last = done + step == sequence_len;
        if (!run_tokens(ctx, tokens + done, step, last && logits_out != nullptr)) return false;
        done += step;
    }
    return fetch_outputs(ctx, state_out, logits_out);
}

RWKV_API size_t rwkv_get_n_vocab(const struct rwkv_context * ctx) { return (size_t) ctx->model->n_vocab(); }
RWKV_API size_t rwkv_get_n_embed(const struct rwkv_context * ctx) { return (size_t) ctx->model->n_embed(); }
RWKV_API size_t rwkv_get_n_layer(const struct rwkv_context * ctx) { return (size_t) ctx->model->n_layer(); }
RWKV_API size_t rwkv_get_state_len(const struct rwkv_context * ctx) { return (size_t) ctx->model->state_len(); }
RWKV_API size_t rwkv_get_logits_len(const struct rwkv_context * ctx) { return (size_t) ctx->model->n_vocab(); }
RWKV_API uint32_t rwkv_get_state_buffer_element_count(const struct rwkv_context * ctx) { return (uint32_t) rwkv_get_state_len(ctx); }
RWKV_API uint32_t rwkv_get_logits_buffer_element_count(const struct rwkv_context * ctx) { return (uint32_t) rwkv_get_logits_len(ctx); }

RWKV_API void rwkv_init_state(const struct rwkv_context * ctx, float * state) {
    const Model & m = *ctx->model;
    const size_t n = (size_t) m.state_len();
    memset(state, 0, n * sizeof(float));
    if (m.arch_major >= 5) return;
    const size_t D = (size_t) m.n_embed();
    for (size_t l = 0; l < (size_t) m.n_layer(); l++) {
        float * pp = state + l * 5 * D + 4 * D;
        for (size_t i = 0; i < D; i++) pp[i] = -1e30F;
    }
}

RWKV_API void rwkv_free(struct rwkv_context * ctx) {
    if (!ctx) return;
    if (!ctx->stages.empty()) { pipeline_destroy(ctx); return; }
    (void) hipSetDevice(ctx->model->device);
    destroy_context(ctx);
}

RWKV_API void rwkv_set_print_errors(struct rwkv_context * ctx, const bool print_errors) {
    if (ctx) ctx->print_errors = print_errors; else g_print_errors = print_errors;
}

RWKV_API bool rwkv_get_print_errors(const struct rwkv_context * ctx) { return ctx ? ctx->print_errors : g_print_errors; }

RWKV_API enum rwkv_error_flags rwkv_get_last_error(struct rwkv_context * ctx) {
    int * p = ctx ? &ctx->last_error : &g_last_error;
    const int v = *p;
    *p = RWKV_ERROR_NONE;
    return (enum rwkv_error_flags) v;
}

RWKV_API const char * rwkv_get_system_info_string(void) {
    static std::string s;
    if (s.empty()) {
        // Same key order as the reference (rwkv.cpp:239-258); the host SIMD flags are informational only here.
#define RW_HAS(F) std::to_string((int) (__builtin_cpu_supports(F) != 0))
        s += "AVX=" + RW_HAS("avx") + " ";
        s += "AVX2=" + RW_HAS("avx2") + " ";
        s += "AVX512=" + RW_HAS("avx512f") + " ";
        s += "FMA=" + RW_HAS("fma") + " ";
        s += "NEON=0 ARM_FMA=0 ";
        s += "F16C=" + RW_HAS("f16c") + " ";
        s += "FP16_VA=0 WASM_SIMD=0 ";
        s += "SSE3=" + RW_HAS("sse3") + " ";
        s += "VSX=0";
#undef RW_HAS
        int n = 0;
        if (hipGetDeviceCount(&n) == hipSuccess && n > 0) {
            hipDeviceProp_t p;
            int dev = 0;
            (void) hipGetDevice(&dev);
            if (hipGetDeviceProperties(&p, dev) == hipSuccess) {
                s += " | HIP=1 DEVICES=" + std::to_string(n) + " ARCH=" + std::string(p.gcnArchName) + " CU=" + std::to_string(p.multiProcessorCount);
                // the one-launch-per-token kernels need every CU of an unpartitioned part (rwkv_mi_persist_info says what a context got and why)
                s += p.multiProcessorCount == 256 ? " PERSISTENT_DECODE=available" : " PERSISTENT_DECODE=unavailable(needs_256_CUs)";
            }
        } else {
            s += " | HIP=0";
        }
    }
    return s.c_str();
}

// ---------------------------------------------------------------------------------------------------------------
// rwkv_mi_* extensions (include/rwkv_mi355x.h)
// ---------------------------------------------------------------------------------------------------------------

#define RW_NO_PIPELINE(CTX, RET) RW_CTX_CHECK((CTX), RWKV_ERROR_ARGS | RWKV_ERROR_UNSUPPORTED, RET, (CTX)->stages.empty(), \
    "this rwkv_mi_* extension works on a single-device context; with RWKV_MI_DEVICES use the rwkv.h entry points")

RWKV_API bool rwkv_mi_state_load(struct rwkv_context * ctx, const float * state_in) {
    ctx->last_error = RWKV_ERROR_NONE;
    if (!ctx->stages.empty()) return pipeline_state_load(ctx, state_in);
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    if (!state_from_host(ctx, state_in)) return false;
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    return true;
}

RWKV_API bool rwkv_mi_state_store(struct rwkv_context * ctx, float * state_out) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, state_out != nullptr, "state_out is NULL");
    if (!ctx->stages.empty()) return pipeline_state_store(ctx, state_out);
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    return fetch_outputs(ctx, state_out, nullptr);
}

RWKV_API bool rwkv_mi_eval_resident(struct rwkv_context * ctx, const uint32_t * tokens, size_t n_tokens, float * logits_out) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_NO_PIPELINE(ctx, false);
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, tokens != nullptr && n_tokens > 0, "tokens is NULL or empty");
    const size_t n_vocab = (size_t) ctx->model->n_vocab();
    for (size_t i = 0; i < n_tokens; i++) RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, tokens[i] < n_vocab, "Token at index %zu is out of range", i);
    if (!run_tokens(ctx, tokens, n_tokens, logits_out != nullptr)) return false;
    return fetch_outputs(ctx, nullptr, logits_out);
}

RWKV_API bool rwkv_mi_decode_greedy(struct rwkv_context * ctx, uint32_t first_token, size_t n_tokens, uint32_t * tokens_out, float * elapsed_ms) {
    ctx->last_error = RWKV_ERROR_NONE;
    const size_t n_vocab = (size_t) ctx->model->n_vocab();
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, first_token < n_vocab, "Token is out of range");
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, n_tokens > 0, "n_tokens is 0");
    if (!ctx->stages.empty()) { struct rwkv_context * one[1] = {ctx}; return pipeline_decode_greedy(one, 1, &first_token, n_tokens, tokens_out, elapsed_ms); }
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    if (!upload_tokens(ctx, &first_token, 1)) return false;
    struct DevBuf { uint32_t * p = nullptr; ~DevBuf() { if (p) (void) hipFree(p); } } hist;   // freed on every exit
    HIP_CTX_OK(ctx, hipMalloc((void **) &hist.p, n_tokens * sizeof(uint32_t)));
    uint32_t * d_hist = hist.p;
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    // persist_v47.hip: the launch itself picks the token, leaves it where its own embedding lookup reads it and appends it to the history
    const bool in_launch = folded_argmax_target(ctx) == ctx->d_tokens && mega_v6_set_history(ctx->mega, d_hist, n_tokens, ctx->stream);
    // (every exit from here on takes the history pointer out of the kernel's control words again: d_hist is freed when this function returns)
    struct HistGuard { rwkv_context * c; bool on; ~HistGuard() { if (on && c->mega) (void) mega_v6_set_history(c->mega, nullptr, 0, c->stream); } } hist_guard{ctx, in_launch};
    HIP_CTX_OK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    bool ok = true;
    for (size_t i = 0; i < n_tokens && ok; i++) {
        ok = forward_decode(ctx, true);
        if (!ok) break;
        if (in_launch) continue;
        // next token = argmax(logits), written where the embedding kernel reads it; no host round trip
        if (folded_argmax_target(ctx) != ctx->d_tokens) launch_argmax(ctx->d_logits, (int64_t) n_vocab, ctx->d_tokens, ctx->stream);
        if (hipMemcpyAsync(d_hist + i, ctx->d_tokens, sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) ok = false;
    }
    if (in_launch && ctx->mega) {
        const bool drained = hipEventRecord(ctx->ev1, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
        ok = mega_v6_set_history(ctx->mega, nullptr, 0, ctx->stream) && drained && ok;
        hist_guard.on = false;
        if (ok && elapsed_ms) ok = hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1) == hipSuccess;
        if (ok && tokens_out) ok = hipMemcpyAsync(tokens_out, d_hist, n_tokens * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    } else if (ok) {
        ok = hipEventRecord(ctx->ev1, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (ok && elapsed_ms) ok = hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1) == hipSuccess;
        if (ok && tokens_out) ok = hipMemcpyAsync(tokens_out, d_hist, n_tokens * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    }
    RW_CTX_CHECK(ctx, RWKV_ERROR_GRAPH, false, ok, "greedy decode failed: %s", hipGetErrorString(hipGetLastError()));
    return fetch_outputs(ctx, nullptr, nullptr);   // drains the stream; a poll time-out invalidates the whole run (state included)
}

static bool ensure_sampler(rwkv_context * ctx) {
    if (!ctx->d_probs) HIP_CTX_OK(ctx, hipMalloc((void **) &ctx->d_probs, (size_t) ctx->model->n_vocab() * sizeof(float)));
    if (!ctx->d_rng_counter) {
        HIP_CTX_OK(ctx, hipMalloc((void **) &ctx->d_rng_counter, 64));
        HIP_CTX_OK(ctx, hipMemsetAsync(ctx->d_rng_counter, 0, 64, ctx->stream));
    }
    return true;
}

// Samples one token from the logits of the last evaluation on the device (reference: python/sampling.py sample_logits, run there on
// the host after downloading the logits). u in [0, 1): the caller's uniform random number; u < 0: the context's generator (seed).
RWKV_API bool rwkv_mi_sample(struct rwkv_context * ctx, float temperature, float top_p, float u, uint64_t seed, uint32_t * token_out) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_NO_PIPELINE(ctx, false);
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, temperature >= 0.0f && top_p >= 0.0f && top_p <= 1.0f && u < 1.0f && token_out, "bad sampling arguments");
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, ctx->model->has_head, "this stage has no head");
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    if (!ensure_sampler(ctx)) return false;
    launch_sample(ctx->d_logits, (int) ctx->model->n_vocab(), temperature, top_p, u, seed, ctx->d_rng_counter, ctx->d_probs, ctx->d_next_token, nullptr, 0, ctx->stream);
    HIP_CTX_OK(ctx, hipMemcpyAsync(token_out, ctx->d_next_token, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    return true;
}

// Sampling decode loop entirely on the device: feeds first_token, then n_tokens - 1 times a token sampled (temperature, top_p, generator
// seeded with `seed`) from the previous logits. tokens_out[i] = token sampled after step i.
RWKV_API bool rwkv_mi_decode_sample(struct rwkv_context * ctx, uint32_t first_token, size_t n_tokens, float temperature, float top_p, uint64_t seed,
                                    uint32_t * tokens_out, float * elapsed_ms) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_NO_PIPELINE(ctx, false);
    const size_t n_vocab = (size_t) ctx->model->n_vocab();
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, first_token < n_vocab && n_tokens > 0, "bad arguments");
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, temperature >= 0.0f && top_p >= 0.0f && top_p <= 1.0f, "bad sampling arguments");
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    if (!upload_tokens(ctx, &first_token, 1) || !ensure_sampler(ctx)) return false;
    HIP_CTX_OK(ctx, hipMemsetAsync(ctx->d_rng_counter, 0, 8, ctx->stream));
    struct DevBuf { uint32_t * p = nullptr; ~DevBuf() { if (p) (void) hipFree(p); } } hist;
    HIP_CTX_OK(ctx, hipMalloc((void **) &hist.p, n_tokens * sizeof(uint32_t)));
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    HIP_CTX_OK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (size_t i = 0; i < n_tokens; i++) {
        if (!forward_decode(ctx, true)) return false;
        // the sampled token is written where the embedding kernel of the next step reads it
        launch_sample(ctx->d_logits, (int) n_vocab, temperature, top_p, -1.0f, seed, ctx->d_rng_counter, ctx->d_probs, ctx->d_tokens, hist.p, (int) i, ctx->stream);
    }
    HIP_CTX_OK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    if (tokens_out) HIP_CTX_OK(ctx, hipMemcpyAsync(tokens_out, hist.p, n_tokens * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (!fetch_outputs(ctx, nullptr, nullptr)) return false;
    if (elapsed_ms) HIP_CTX_OK(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return true;
}

// Eager (graph-free) greedy decode with a HIP-event pair around every launch of the dominant kernel.
// out[0] = summed kernel time (ms), out[1] = launches, out[2] = summed algorithmic bytes, out[3] = wall ms of the loop.
RWKV_API bool rwkv_mi_profile_decode(struct rwkv_context * ctx, uint32_t first_token, size_t n_tokens, double * out) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_NO_PIPELINE(ctx, false);
    const size_t n_vocab = (size_t) ctx->model->n_vocab();
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, first_token < n_vocab && n_tokens > 0 && out, "bad arguments");
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    if (!upload_tokens(ctx, &first_token, 1)) return false;
    auto & pf = ctx->prof;
    pf.total_ms = 0.0; pf.launches = 0; pf.total_bytes = 0;
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    HIP_CTX_OK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (size_t i = 0; i < n_tokens; i++) {
        pf.on = true; pf.used = 0;
        const bool ok = forward(ctx, 1, true);
        pf.on = false;
        if (!ok) return false;
        if (folded_argmax_target(ctx) != ctx->d_tokens) launch_argmax(ctx->d_logits, (int64_t) n_vocab, ctx->d_tokens, ctx->stream);
        HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t k = 0; k < pf.used; k++) {
            float ms = 0.0f;
            HIP_CTX_OK(ctx, hipEventElapsedTime(&ms, pf.events[2 * k], pf.events[2 * k + 1]));
            pf.total_ms += ms; pf.launches++; pf.total_bytes += pf.bytes[k];
        }
    }
    HIP_CTX_OK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    float wall = 0.0f;
    HIP_CTX_OK(ctx, hipEventElapsedTime(&wall, ctx->ev0, ctx->ev1));
    out[0] = pf.total_ms; out[1] = (double) pf.launches; out[2] = (double) pf.total_bytes; out[3] = wall;
    return true;
}

// One sequence pass over `tokens` from the resident state with a HIP-event pair (on the context's stream) around every launch of the
// sequence-mode GEMM (k_mmq_mfma). out[0] = summed kernel ms, out[1] = launches, out[2] = summed integer operations (2 T N K), out[3] = wall ms.
RWKV_API bool rwkv_mi_profile_prefill(struct rwkv_context * ctx, const uint32_t * tokens, size_t n_tokens, double * out) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_NO_PIPELINE(ctx, false);
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, tokens && n_tokens > 0 && n_tokens <= k_max_tokens_per_pass && out, "bad arguments");
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    if (!upload_tokens(ctx, tokens, n_tokens)) return false;
    auto & pf = ctx->prof;
    pf.total_ms = 0.0; pf.launches = 0; pf.total_bytes = 0;
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    HIP_CTX_OK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    pf.on = true; pf.used = 0;
    const bool ok = forward(ctx, (int64_t) n_tokens, true);
    pf.on = false;
    if (!ok) return false;
    HIP_CTX_OK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t k = 0; k < pf.used; k++) {
        float ms = 0.0f;
        HIP_CTX_OK(ctx, hipEventElapsedTime(&ms, pf.events[2 * k], pf.events[2 * k + 1]));
        pf.total_ms += ms; pf.launches++; pf.total_bytes += pf.bytes[k];
    }
    float wall = 0.0f;
    HIP_CTX_OK(ctx, hipEventElapsedTime(&wall, ctx->ev0, ctx->ev1));
    out[0] = pf.total_ms; out[1] = (double) pf.launches; out[2] = (double) pf.total_bytes; out[3] = wall;
    return true;
}

// (a RWKV_MI_DEVICES front context: the sum over its stages)
RWKV_API uint64_t rwkv_mi_bytes_per_token(const struct rwkv_context * ctx) {
    if (ctx->stages.empty()) return ctx->model->bytes_per_token;
    uint64_t t = 0; for (const rwkv_context * s : ctx->stages) t += s->model->bytes_per_token; return t;
}
RWKV_API uint64_t rwkv_mi_weight_bytes(const struct rwkv_context * ctx) {
    if (ctx->stages.empty()) return ctx->model->weight_bytes;
    uint64_t t = 0; for (const rwkv_context * s : ctx->stages) t += s->model->weight_bytes; return t;
}

// Arithmetic of one sequence pass over T tokens: 2 * T * (elements of every 2-D layer matrix) + 2 * V * D
// (the head runs on the last token only). Embedding, vectors and the elementwise v7 r_k table are not matrices of the pass.
RWKV_API uint64_t rwkv_mi_prefill_flops(const struct rwkv_context * ctx, size_t n_tokens) {
    const Model & m = *ctx->model;
    uint64_t w = 0;
    for (const auto & t : m.tensors) {
        if (t->ndim != 2 || t.get() == m.emb || t.get() == m.head) continue;
        if (t->name.find("att.r_k") != std::string::npos) continue;
        w += (uint64_t) t->ne[0] * (uint64_t) t->ne[1];
    }
    uint64_t f = 2ull * (uint64_t) n_tokens * w;
    if (m.has_head && m.head) f += 2ull * (uint64_t) m.head->ne[0] * (uint64_t) m.head->ne[1];
    return f;
}

RWKV_API void rwkv_mi_get_arch(const struct rwkv_context * ctx, uint32_t * major, uint32_t * minor, uint32_t * head_count, uint32_t * head_size) {
    if (major) *major = (uint32_t) ctx->model->arch_major;
    if (minor) *minor = (uint32_t) ctx->model->arch_minor;
    if (head_count) *head_count = (uint32_t) ctx->model->head_count;
    if (head_size) *head_size = (uint32_t) ctx->model->head_size;
}

RWKV_API void rwkv_mi_set_graph_enabled(struct rwkv_context * ctx, bool enabled) { ctx->use_graph = enabled; }

RWKV_API bool rwkv_mi_decode_healthy(struct rwkv_context * ctx) {
    if (hipSetDevice(ctx->model->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
    return !(ctx->mega && mega_v6_aborted(ctx->mega, ctx->stream));
}

RWKV_API int rwkv_mi_decode_path(const struct rwkv_context * ctx) { return ctx->mega ? 2 : ((ctx->fused_v6 || ctx->fused_v7 || ctx->fused_v4) ? 1 : 0); }
RWKV_API int rwkv_mi_persist_kind(const struct rwkv_context * ctx) { return mega_v6_kind((ctx->stages.empty() ? ctx : ctx->stages.front())->mega); }   // (a chain: its first stage's)
// "persist: ring | regs | k47 | none; <why>": which persistent kernel serves this context's single-token steps and what decided it
// (geometry / device / environment at creation, the calibration's figures, a fall-back after a poll time-out). Valid until the next call on ctx.
RWKV_API const char * rwkv_mi_persist_info(struct rwkv_context * ctx) {
    static thread_local std::string out;
    rwkv_context * c = ctx->stages.empty() ? ctx : ctx->stages.front();
    const int k = mega_v6_kind(c->mega);
    out = std::string("persist: ") + (k == 2 ? "ring" : (k == 1 ? "regs" : (k == 3 ? "k47" : "none")));
    if (!c->persist_note.empty()) out += "; " + c->persist_note;
    return out.c_str();
}

// seconds the payload of the model file took to reach HBM (reads + host-to-device copies + re-pack kernels), and its bytes
RWKV_API void rwkv_mi_load_stats(const struct rwkv_context * ctx, double * seconds, uint64_t * bytes) {
    double s = 0.0; uint64_t b = 0;
    if (ctx->stages.empty()) { s = ctx->model->load_seconds; b = ctx->model->weight_bytes; }
    else for (const rwkv_context * st : ctx->stages) { s += st->model->load_seconds; b += st->model->weight_bytes; }
    if (seconds) *seconds = s;
    if (bytes) *bytes = b;
}

// ---------------------------------------------------------------------------------------------------------------
// Layer pipeline (one process per GPU; the hand-off itself is done by the caller with RCCL send/recv)
// ---------------------------------------------------------------------------------------------------------------

RWKV_API struct rwkv_context * rwkv_mi_init_stage(const char * file_path, uint32_t n_threads, uint32_t layer_begin, uint32_t layer_end) {
    g_last_error = RWKV_ERROR_NONE;
    RW_CHECK(RWKV_ERROR_ARGS, nullptr, file_path != nullptr, "model_file_path is NULL");
    Model * m = load_model(file_path, layer_begin, layer_end);
    if (!m) return nullptr;
    return create_context(m, n_threads);
}

RWKV_API bool rwkv_mi_set_stream(struct rwkv_context * ctx, void * hip_stream) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_NO_PIPELINE(ctx, false);
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream));
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) if (ctx->graph_exec[a][b]) { (void) hipGraphExecDestroy(ctx->graph_exec[a][b]); ctx->graph_exec[a][b] = nullptr; }
    if (ctx->owns_stream && ctx->stream) (void) hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t) hip_stream;
    ctx->owns_stream = false;
    return true;
}

RWKV_API size_t rwkv_mi_handoff_len(const struct rwkv_context * ctx) { return (size_t) handoff_len(*ctx->model); }

RWKV_API void rwkv_mi_stage_range(const struct rwkv_context * ctx, uint32_t * layer_begin, uint32_t * layer_end) {
    if (layer_begin) *layer_begin = ctx->model->layer_begin;
    if (layer_end) *layer_end = ctx->model->layer_end;
}

// One single-token step of this stage, everything on the context's stream, nothing synchronised:
//   first stage : reads the token id from device memory (d_token), runs embedding + its layers
//   other stages: start from x_in (device, rwkv_mi_handoff_len floats)
//   not last    : writes the outgoing residual stream to x_out (device)
//   last stage  : ln_out + head into the context's logits, argmax into d_next_token (device, may be NULL)
RWKV_API bool rwkv_mi_stage_step(struct rwkv_context * ctx, const uint32_t * d_token, const float * x_in, float * x_out, uint32_t * d_next_token) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_NO_PIPELINE(ctx, false);
    Model & m = *ctx->model;
    HIP_CTX_OK(ctx, hipSetDevice(m.device));
    const size_t D = (size_t) m.n_embed();
    if (!ctx->d_tokens) {  // first use: allocate the token slot
        const uint32_t zero = 0;
        if (!upload_tokens(ctx, &zero, 1)) return false;
    }
    if (!ensure_scratch(ctx, 1)) return false;
    if (m.has_embed) {
        RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, d_token != nullptr, "first stage needs a token");
        HIP_CTX_OK(ctx, hipMemcpyAsync(ctx->d_tokens, d_token, sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, x_in != nullptr, "stage needs x_in");
        HIP_CTX_OK(ctx, hipMemcpyAsync(ctx->b.x, x_in, D * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        if (m.arch_major == 7) HIP_CTX_OK(ctx, hipMemcpyAsync(ctx->b.v_first, x_in + D, D * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (!forward_decode(ctx, m.has_head)) return false;
    if (!m.has_head) {
        RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, x_out != nullptr, "stage needs x_out");
        HIP_CTX_OK(ctx, hipMemcpyAsync(x_out, ctx->b.x, D * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        if (m.arch_major == 7) HIP_CTX_OK(ctx, hipMemcpyAsync(x_out + D, ctx->b.v_first, D * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    } else if (d_next_token) {
        launch_argmax(ctx->d_logits, m.n_vocab(), d_next_token, ctx->stream);
    }
    return true;
}

// copies the context's logits (of the last step that produced any) to host memory, synchronising the context's stream
RWKV_API bool rwkv_mi_logits_store(struct rwkv_context * ctx, float * logits_out) {
    ctx->last_error = RWKV_ERROR_NONE;
    RW_NO_PIPELINE(ctx, false);
    RW_CTX_CHECK(ctx, RWKV_ERROR_ARGS, false, logits_out != nullptr, "logits_out is NULL");
    HIP_CTX_OK(ctx, hipSetDevice(ctx->model->device));
    return fetch_outputs(ctx, nullptr, logits_out);
}

// device pointer of the context's logits buffer (valid after a last-stage step / any eval that produced logits)
RWKV_API const float * rwkv_mi_logits_device_ptr(const struct rwkv_context * ctx) { return ctx->stages.empty() ? ctx->d_logits : ctx->stages.back()->d_logits; }

RWKV_API bool rwkv_mi_trace_phases(struct rwkv_context * ctx, uint32_t token, int layer, int n, long long * out) {
    if (!ctx->mega) return false;
    const bool g = ctx->use_graph; ctx->use_graph = false;
    bool ok = mega_v6_trace(ctx->mega, layer, out, false);
    for (int i = 0; i < n && ok; i++) ok = run_tokens(ctx, &token, 1, true);
    (void) hipStreamSynchronize(ctx->stream);
    ctx->use_graph = g;
    return ok && mega_v6_trace(ctx->mega, layer, out, true);
}

// The persistent decode kernel's hand-over generation (it advances by 8 per layer and launch; the kernel compares its low 16 bits).
// Diagnostic: bench.py positions its parity run across the 16-bit wrap with it. 0 when decode path 2 is off.
RWKV_API uint32_t rwkv_mi_decode_generation(struct rwkv_context * ctx) {
    rwkv_context * c = ctx->stages.empty() ? ctx : ctx->stages.front();
    if (!c->mega || hipSetDevice(c->model->device) != hipSuccess) return 0;
    return mega_v6_generation(c->mega, c->stream);
}

}  // extern "C"
