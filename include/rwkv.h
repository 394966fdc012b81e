/*
 * rwkv.h -- C ABI of librwkv.so for AMD Instinct MI355X (gfx950).
 *
 * Drop-in boundary: the declarations below are binary-compatible with the public header of RWKV/rwkv.cpp
 * (reference rwkv.h @ 2025-02-19; each entry cites the reference declaration it replaces), so existing callers --
 * the ctypes binding python/rwkv_cpp/rwkv_cpp_shared_library.py:49-107, the C tests under tests/, extras/quantize.c,
 * the Go / Node bindings -- link or dlopen this library unchanged.
 *
 * What differs behind the boundary: there is no ggml graph.  Weights are uploaded once to HBM, the recurrent state
 * lives on the device, and every call runs hand-written HIP kernels.  `n_threads` is accepted and
 * ignored; `n_gpu_layers` is accepted and ignored (all layers always run on the GPU -- there is no CPU path in this
 * library, and rwkv_init_from_file fails with RWKV_ERROR_CTX | RWKV_ERROR_UNSUPPORTED when no gfx950 device is visible).
 */
#ifndef RWKV_H
#define RWKV_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#if defined(RWKV_SHARED)
#    define RWKV_API __attribute__((visibility("default")))
#else
#    define RWKV_API
#endif

/* File magic 'ggmf' and supported container versions (reference rwkv.h:23-30). */
#define RWKV_FILE_MAGIC 0x67676d66
#define RWKV_FILE_VERSION_0 100
#define RWKV_FILE_VERSION_1 101
#define RWKV_FILE_VERSION_MIN RWKV_FILE_VERSION_0
#define RWKV_FILE_VERSION_MAX RWKV_FILE_VERSION_1
#define RWKV_FILE_VERSION RWKV_FILE_VERSION_MAX

#if defined(__cplusplus)
extern "C" {
#endif

/* Error flags: (category << 8) | code, OR-accumulated (reference rwkv.h:38-62). */
enum rwkv_error_flags {
    RWKV_ERROR_NONE = 0,

    RWKV_ERROR_ARGS = 1 << 8,
    RWKV_ERROR_FILE = 2 << 8,
    RWKV_ERROR_MODEL = 3 << 8,
    RWKV_ERROR_MODEL_PARAMS = 4 << 8,
    RWKV_ERROR_GRAPH = 5 << 8,
    RWKV_ERROR_CTX = 6 << 8,

    RWKV_ERROR_ALLOC = 1,
    RWKV_ERROR_FILE_OPEN = 2,
    RWKV_ERROR_FILE_STAT = 3,
    RWKV_ERROR_FILE_READ = 4,
    RWKV_ERROR_FILE_WRITE = 5,
    RWKV_ERROR_FILE_MAGIC = 6,
    RWKV_ERROR_FILE_VERSION = 7,
    RWKV_ERROR_DATA_TYPE = 8,
    RWKV_ERROR_UNSUPPORTED = 9,
    RWKV_ERROR_SHAPE = 10,
    RWKV_ERROR_DIMENSION = 11,
    RWKV_ERROR_KEY = 12,
    RWKV_ERROR_DATA = 13,
    RWKV_ERROR_PARAM_MISSING = 14
};

/* Opaque inference context (reference rwkv.h:64-68). One eval at a time per context; contexts may move between
 * threads; parallel inference = one rwkv_clone_context per thread (clones share the weights in HBM). */
struct rwkv_context;

/* Error printing switch; ctx == NULL addresses the thread-local global used by load / quantise (rwkv.h:70-80). */
RWKV_API void rwkv_set_print_errors(struct rwkv_context * ctx, const bool print_errors);
RWKV_API bool rwkv_get_print_errors(const struct rwkv_context * ctx);

/* Returns AND clears the error flags of ctx, or of the calling thread when ctx == NULL (rwkv.h:82-84). */
RWKV_API enum rwkv_error_flags rwkv_get_last_error(struct rwkv_context * ctx);

/* Loads an rwkv.cpp-format model file (docs/FILE_FORMAT.md) and uploads it to the MI355X. NULL on error (rwkv.h:86-91). */
RWKV_API struct rwkv_context * rwkv_init_from_file(const char * model_file_path, const uint32_t n_threads, const uint32_t n_gpu_layers);

/* New context on the same weights (refcounted); own state, scratch and launch graph (rwkv.h:93-99). */
RWKV_API struct rwkv_context * rwkv_clone_context(struct rwkv_context * ctx, const uint32_t n_threads);

/* One token (rwkv.h:101-115). state_in: FP32[rwkv_get_state_len] or NULL for a fresh state; state_out / logits_out are
 * written when non-NULL (state_in may alias state_out). logits_out == NULL skips ln_out + head. false on error. */
RWKV_API bool rwkv_eval(
    struct rwkv_context * ctx,
    const uint32_t token,
    const float * state_in,
    float * state_out,
    float * logits_out
);

/* A sequence of tokens in one pass (rwkv.h:117-143). Logits are those of the LAST token. tokens == NULL only prepares
 * (returns true, writes nothing). sequence_len == 0 is RWKV_ERROR_ARGS. State/logits bit-identical to calling rwkv_eval
 * token by token. There is no ggml node limit here: any sequence_len is accepted in one call. */
RWKV_API bool rwkv_eval_sequence(
    struct rwkv_context * ctx,
    const uint32_t * tokens,
    const size_t sequence_len,
    const float * state_in,
    float * state_out,
    float * logits_out
);

/* Same, split into chunks of chunk_size tokens; outputs come from the final chunk only (rwkv.h:145-168). */
RWKV_API bool rwkv_eval_sequence_in_chunks(
    struct rwkv_context * ctx,
    const uint32_t * tokens,
    const size_t sequence_len,
    const size_t chunk_size,
    const float * state_in,
    float * state_out,
    float * logits_out
);

/* Model geometry (rwkv.h:170-192). state_len = n_layer * n_embed * 5 (v4) or n_layer * n_embed * (2 + head_size) (v5+). */
RWKV_API size_t rwkv_get_n_vocab(const struct rwkv_context * ctx);
RWKV_API size_t rwkv_get_n_embed(const struct rwkv_context * ctx);
RWKV_API size_t rwkv_get_n_layer(const struct rwkv_context * ctx);
RWKV_API size_t rwkv_get_state_len(const struct rwkv_context * ctx);
RWKV_API size_t rwkv_get_logits_len(const struct rwkv_context * ctx);

/* Fills `state` so that passing it equals passing NULL: zeros, v4 pp slots = -1e30 (rwkv.h:194-198). */
RWKV_API void rwkv_init_state(const struct rwkv_context * ctx, float * state);

/* Releases the context; the weights go with the last context referencing them (rwkv.h:200-202). NULL is a no-op. */
RWKV_API void rwkv_free(struct rwkv_context * ctx);

/* FP32/FP16 model file -> Q4_0 | Q4_1 | Q5_0 | Q5_1 | Q8_0 file, byte-compatible with the reference's quantiser
 * (rwkv.h:204-216). */
RWKV_API bool rwkv_quantize_model_file(const char * model_file_path_in, const char * model_file_path_out, const char * format_name);

/* "AVX=0 AVX2=0 ... VSX=0"-style capability string of the reference plus the HIP device line (rwkv.h:218-219). */
RWKV_API const char * rwkv_get_system_info_string(void);

/* Legacy getters still bound by the reference's Python wrapper (reference rwkv.cpp:145-153). */
RWKV_API uint32_t rwkv_get_state_buffer_element_count(const struct rwkv_context * ctx);
RWKV_API uint32_t rwkv_get_logits_buffer_element_count(const struct rwkv_context * ctx);

#if defined(__cplusplus)
}
#endif

#endif
