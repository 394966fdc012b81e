/*
 * rwkv_mi355x.h -- opt-in extensions of librwkv.so for MI355X. None of these change the behaviour of the rwkv.h entry
 * points; they expose what the device-resident engine can do beyond the reference ABI:
 *
 *  - the recurrent state can stay in HBM between calls (the reference ABI hands the whole state in and out through host
 *    memory on every call, rwkv_eval.inc:2-22 -- 34.6 MB each way for RWKV-6 7B, i.e. as long as the token itself);
 *  - a greedy decode loop that never leaves the device (argmax on the GPU feeds the next embedding lookup);
 *  - the numbers the benchmark needs (algorithmic bytes per token).
 */
#ifndef RWKV_MI355X_H
#define RWKV_MI355X_H

#include "rwkv.h"

#if defined(__cplusplus)
extern "C" {
#endif

/* Copies a host state (or a fresh state when NULL) into the context's device-resident state (on a RWKV_MI_DEVICES chain: every stage
 * takes the slice of its layers; rwkv_mi_state_store and rwkv_mi_decode_greedy work on chains as well). */
RWKV_API bool rwkv_mi_state_load(struct rwkv_context * ctx, const float * state_in);
/* Copies the device-resident state to host memory (FP32[rwkv_get_state_len]). */
RWKV_API bool rwkv_mi_state_store(struct rwkv_context * ctx, float * state_out);

/* Like rwkv_eval_sequence, but continues from and updates the device-resident state; no state traffic over PCIe.
 * logits_out (of the last token) may be NULL. */
RWKV_API bool rwkv_mi_eval_resident(struct rwkv_context * ctx, const uint32_t * tokens, size_t n_tokens, float * logits_out);

/* Greedy single-stream decode entirely on the device: feeds first_token, then n_tokens - 1 times the argmax of the
 * previous logits. tokens_out[i] = argmax after step i (may be NULL). elapsed_ms (may be NULL) receives the HIP-event
 * time of the whole loop measured on the context's stream. */
RWKV_API bool rwkv_mi_decode_greedy(struct rwkv_context * ctx, uint32_t first_token, size_t n_tokens, uint32_t * tokens_out, float * elapsed_ms);

/* Temperature / top-p sampling ON THE DEVICE from the logits of the last evaluation (the reference samples on the host:
 * python/sampling.py:10-52 -- same statements: softmax, top-p cut-off, p^(1/temperature), renormalise, draw). temperature == 0: argmax;
 * top_p == 0 means 1. u in [0, 1): the caller's uniform random number; u < 0: the context's counter-based generator with `seed`. */
RWKV_API bool rwkv_mi_sample(struct rwkv_context * ctx, float temperature, float top_p, float u, uint64_t seed, uint32_t * token_out);
/* Sampling decode loop entirely on the device (like rwkv_mi_decode_greedy, the sampled token feeds the next embedding lookup). */
RWKV_API bool rwkv_mi_decode_sample(struct rwkv_context * ctx, uint32_t first_token, size_t n_tokens, float temperature, float top_p, uint64_t seed,
                                    uint32_t * tokens_out, float * elapsed_ms);

/* Measurement aid: eager greedy decode with a HIP-event pair (on the context's stream) around every launch of the dominant
 * kernel -- the single-token projection of the model's quantised format. out[0] = summed kernel ms, out[1] = launches,
 * out[2] = summed algorithmic bytes (weight rows + quantised activation + outputs), out[3] = wall ms of the loop. */
RWKV_API bool rwkv_mi_profile_decode(struct rwkv_context * ctx, uint32_t first_token, size_t n_tokens, double * out);

/* The same for sequence mode: one pass over `tokens` (at most 1024) from the resident state, HIP events around every launch of the int8
 * MFMA GEMM. out[0] = summed kernel ms, out[1] = launches, out[2] = summed integer operations (2 T N K per launch), out[3] = wall ms of the pass. */
RWKV_API bool rwkv_mi_profile_prefill(struct rwkv_context * ctx, const uint32_t * tokens, size_t n_tokens, double * out);

/* Algorithmic HBM bytes one decoded token must move on this context's layers: every parameter once (file dtype), one
 * embedding row, state read + write, logits write. */
RWKV_API uint64_t rwkv_mi_bytes_per_token(const struct rwkv_context * ctx);
RWKV_API uint64_t rwkv_mi_weight_bytes(const struct rwkv_context * ctx);
/* Arithmetic of one rwkv_eval_sequence pass over n_tokens: 2 * n_tokens * (elements of every 2-D layer matrix) + 2 * n_vocab * n_embed. */
RWKV_API uint64_t rwkv_mi_prefill_flops(const struct rwkv_context * ctx, size_t n_tokens);

/* Detected architecture (4, 5.1, 5.2, 6, 7) and head geometry. Any pointer may be NULL. */
RWKV_API void rwkv_mi_get_arch(const struct rwkv_context * ctx, uint32_t * major, uint32_t * minor, uint32_t * head_count, uint32_t * head_size);

/* Single-token steps are replayed from a captured hipGraph by default; disable for debugging / profiling per kernel. */
RWKV_API void rwkv_mi_set_graph_enabled(struct rwkv_context * ctx, bool enabled);

/* Which single-token path this context runs: 0 = one kernel per graph op, 1 = fused RWKV-6 layer (seven launches per layer),
 * 2 = persistent whole-stage kernel (one launch per token; needs the device to itself). The choice is made
 * at context creation from the model geometry, the weight format and the environment (RWKV_MI_NO_FUSED / RWKV_MI_NO_MEGA). */
RWKV_API int rwkv_mi_decode_path(const struct rwkv_context * ctx);

/* Which persistent kernel serves decode path 2: 2 = weights streamed through an LDS ring by a loader wave (LDS-DMA; the default where the
 * model qualifies), 1 = weights prefetched into the registers of the waves that use them (RWKV_MI_PERSIST=regs), 0 = path 2 not active. */
RWKV_API int rwkv_mi_persist_kind(const struct rwkv_context * ctx);
/* "persist: ring|regs|k47|none; <why>" -- the kernel above by name and what decided it: the device (the persistent kernels need all 256 CUs of an
 * unpartitioned MI355X; rwkv_get_system_info_string() carries PERSISTENT_DECODE=available|unavailable for the current device), the model's
 * format / geometry, the environment, the calibration's two figures, or a fall-back after a poll time-out at run time (a second process on the
 * GPU). The string is valid until the next call on this thread. */
RWKV_API const char * rwkv_mi_persist_info(struct rwkv_context * ctx);

/* How long the payload of the model file took to reach HBM at rwkv_init_from_file (parallel reads into pinned staging buffers,
 * asynchronous copies, re-pack kernels; the reference reads one tensor at a time, rwkv_file_format.inc:302-313), and its bytes. */
RWKV_API void rwkv_mi_load_stats(const struct rwkv_context * ctx, double * seconds, uint64_t * bytes);

/* Waits for the context's stream and reports whether every persistent-kernel step so far completed (false: a poll timed out
 * because not all workgroups could be resident -- the device is shared, or other kernels held CUs for seconds; results since
 * then are invalid and the context should be re-created with RWKV_MI_NO_MEGA=1). Always true on paths 0 and 1. */
RWKV_API bool rwkv_mi_decode_healthy(struct rwkv_context * ctx);

/* Diagnostic for decode path 2: runs `n` eager single-token steps of `token` and returns the shader-clock stamps the
 * persistent kernel took in layer `layer`: out[(workgroup * 8 + wave) * 32 + k], 256 workgroups, k < 30 (k < 17 shader-clock stamps, 17..29 stamps of the 100 MHz real-time counter; wave 0: the
 * polling wave's phases, waves 1..7: the row workers' phases; tools/trace.py prints them). false if path 2 is not active. */
RWKV_API bool rwkv_mi_trace_phases(struct rwkv_context * ctx, uint32_t token, int layer, int n, long long * out);

/* ---- layer pipeline: one process per GPU, each owning layers [layer_begin, layer_end) and their slice of the state ----
 * (supersedes the reference's n_gpu_layers CPU/GPU split, rwkv_model_loading.inc:129-142). The hand-off of the residual
 * stream between stages is the caller's job (RCCL send/recv over xGMI, see rwkv.cpp_amd/pipeline.py). */

/* Loads only the tensors of layers [layer_begin, layer_end) (plus emb + ln0 on the first stage, ln_out + head on the last). */
RWKV_API struct rwkv_context * rwkv_mi_init_stage(const char * model_file_path, uint32_t n_threads, uint32_t layer_begin, uint32_t layer_end);
/* Runs every later call of this context on the given hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
RWKV_API bool rwkv_mi_set_stream(struct rwkv_context * ctx, void * hip_stream);
/* Floats in one hand-off message: n_embed, or 2 * n_embed for RWKV-7 (x and v_first). */
RWKV_API size_t rwkv_mi_handoff_len(const struct rwkv_context * ctx);
RWKV_API void rwkv_mi_stage_range(const struct rwkv_context * ctx, uint32_t * layer_begin, uint32_t * layer_end);
/* One single-token step of the stage on its stream, not synchronised. First stage: token id read from device memory
 * (d_token). Other stages: residual stream from x_in (device). Not last: writes x_out (device). Last: ln_out + head,
 * argmax into d_next_token (device, may be NULL). State stays resident (use rwkv_mi_state_load(ctx, NULL) to reset). */
RWKV_API bool rwkv_mi_stage_step(struct rwkv_context * ctx, const uint32_t * d_token, const float * x_in, float * x_out, uint32_t * d_next_token);
/* The greedy decode loop of a whole pipeline, enqueued from C++ (runner.cpp) -- no host language between tokens.
 * rwkv_mi_decode_greedy_streams: n_streams contexts of ONE process (a RWKV_MI_DEVICES chain and its clones, or a one-device context
 * and its clones), interleaved stream by stream so that every stage of the chain has work; per-stage launches replay per-device
 * hipGraphs, the residual stream travels device to device. tokens_out: [n_streams][n_tokens] (may be NULL); elapsed_ms: host wall time
 * of the loop including the final drain (may be NULL). State: resident (rwkv_mi_state_load works on chains too). */
RWKV_API bool rwkv_mi_decode_greedy_streams(struct rwkv_context * const * ctxs, size_t n_streams, const uint32_t * first_tokens, size_t n_tokens,
                                            uint32_t * tokens_out, float * elapsed_ms);
/* One process per GPU: this rank's stage (rwkv_mi_init_stage + clones, one per decode stream, all bound to one stream) runs its share
 * of the same loop with ncclSend / ncclRecv of librccl.so on the stage's stream. librccl.so is dlopen'ed by the first rwkv_mi_comm_*
 * call (librwkv.so itself does not depend on it). comm_fwd: residual stream rank -> rank + 1; comm_fb: the chosen token from the last
 * rank to rank 0 -- two communicators of the same ranks (a single one would dead-lock, see runner.cpp). The 128-byte id of
 * rwkv_mi_comm_unique_id (rank 0) is distributed by the caller (e.g. torch.distributed.broadcast). tokens_out is filled on the last rank. */
RWKV_API bool rwkv_mi_comm_available(void);
RWKV_API bool rwkv_mi_comm_unique_id(void * id_out, size_t capacity);
RWKV_API void * rwkv_mi_comm_init(const void * id128, int rank, int world);
RWKV_API void rwkv_mi_comm_free(void * comm);
/* The same kind of handle WITHOUT RCCL, for ranks that share one GPU (RCCL refuses that; tests of the multi-process loop): mailboxes in
 * device memory shared through HIP IPC, hand-shakes through a POSIX shared-memory segment `name` ("/...", the same on every rank, unique
 * per communicator and run). A hop synchronises the stream on both sides: correct and slow, not a production transport. */
RWKV_API void * rwkv_mi_comm_init_ipc(const char * name, int rank, int world);
RWKV_API bool rwkv_mi_stage_run(struct rwkv_context * const * handles, size_t n_streams, const uint32_t * first_tokens, size_t n_tokens,
                                int rank, int world, void * comm_fwd, void * comm_fb, uint32_t * tokens_out, float * elapsed_ms);
/* Copies the context's logits (n_vocab floats, from the last step that produced any) to host memory. */
RWKV_API bool rwkv_mi_logits_store(struct rwkv_context * ctx, float * logits_out);
/* Device pointer of the context's logits (n_vocab floats), valid after a step that produced logits. */
RWKV_API const float * rwkv_mi_logits_device_ptr(const struct rwkv_context * ctx);

/* The hand-over generation of decode path 2 (the persistent kernel compares its low 16 bits; it advances by 8 per layer and launch).
 * Diagnostic, read-only: bench.py places its parity run across the 16-bit wrap with it. 0 if path 2 is off. */
RWKV_API uint32_t rwkv_mi_decode_generation(struct rwkv_context * ctx);

#if defined(__cplusplus)
}
#endif

#endif
