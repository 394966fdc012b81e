/*
 * rwkv_oracle.h -- CPU ORACLE for the rwkv.cpp hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm (RWKV/rwkv.cpp @ 2025-02-19) for
 *   rwkv_init_from_file -> rwkv_eval / rwkv_eval_sequence, RWKV v4 / v5.1 / v5.2 / v6 / v7,
 *   FP32 / FP16 / Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q8_0 weights,
 * used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.
 * The product (librwkv.so, HIP) never links, loads or calls anything in oracle/.
 *
 * Parity pin: checked against the reference's own golden vectors (tests/golden/expected-logits-*.bin,
 * the recorded logit-difference sums of tests/test_tiny_rwkv.c:38-134 and
 * tests/test_quantization_format_compatibility.c:22-35, and byte-level against the shipped
 * *-Q5_0.bin / *-Q5_1.bin fixtures) by tests/test_oracle_*.py.
 * The reference itself cannot be built here: its arithmetic lives in the un-vendored ggml submodule
 * (github.com/ggerganov/ggml, branch master, pinned commit unknown, absent from /root/reference).
 */
#ifndef RWKV_ORACLE_H
#define RWKV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rwkv.cpp on-disk type ids (rwkv_file_format.inc:5-24) */
enum orc_type { ORC_F32 = 0, ORC_F16 = 1, ORC_Q4_0 = 2, ORC_Q4_1 = 3, ORC_Q5_0 = 7, ORC_Q5_1 = 8, ORC_Q8_0 = 9 };

typedef struct orc_model orc_model;

/* info[]: 0 arch_major, 1 arch_minor, 2 n_vocab, 3 n_embed, 4 n_layer, 5 head_count, 6 head_size,
 *         7 header data_type, 8 file version, 9 ffn size */
#define ORC_INFO_LEN 10

orc_model * orc_load(const char * path);
void        orc_free(orc_model * m);
void        orc_info(const orc_model * m, int64_t * info);
size_t      orc_state_len(const orc_model * m);
/* bytes of every tensor one decoded token touches: all tensors once, minus emb, plus one emb row,
 * plus state read+write, plus logits write. */
uint64_t    orc_bytes_per_token(const orc_model * m);
void        orc_init_state(const orc_model * m, float * state);
void        orc_set_threads(int n);
/* 1: the projections use the SIMD row kernels of rwkv_oracle_fast.c (AVX2, AVX-512-VNNI when the CPU has it): bit-identical results */
void        orc_set_fast(int on);
int         orc_fast_uses_vnni(void);

/* state_in may be NULL (fresh state); state_out / logits_out may be NULL. in/out may alias. Returns 0 on success. */
int orc_eval(orc_model * m, uint32_t token, const float * state_in, float * state_out, float * logits_out);
int orc_eval_sequence(orc_model * m, const uint32_t * tokens, size_t n, const float * state_in, float * state_out, float * logits_out);

/* One pipeline stage (layers [layer_begin, layer_end)) of a single-token step; `state` (full layout) is updated in place.
 * layer_begin == 0 starts from `token`, otherwise from xio (D floats, + D floats of v_first for v7); unless the stage is the
 * last one the outgoing residual stream is written back to xio; the last stage writes logits_out (may be NULL). */
int orc_eval_stage(orc_model * m, uint32_t layer_begin, uint32_t layer_end, uint32_t token, float * xio, float * state, float * logits_out);

/* primitives (row-wise, n multiple of 32 for quantised types) */
size_t orc_type_size(int type);   /* bytes per block */
int    orc_block_size(int type);  /* elements per block */
void   orc_quantize_row(int type, const float * x, void * y, int64_t n);
void   orc_dequantize_row(int type, const void * x, float * y, int64_t n);
/* activation quantisation used inside mul_mat: q8_0 -> int8 q[n], f32 d[n/32] (already fp16-rounded);
 * q8_1 additionally s[n/32] = fp16(d * sum q). */
void   orc_quantize_act(const float * x, int64_t n, int8_t * q, float * d, float * s);
/* y[t*N + n] = sum_k W[n][k] * x[t*K + k] with ggml CPU mul_mat semantics */
void   orc_mul_mat(int wtype, const void * W, int64_t K, int64_t N, const float * x, int64_t T, float * y);

/* file -> file quantiser (rwkv_quantize.inc:16-171). Returns 0 on success. */
int orc_quantize_file(const char * in_path, const char * out_path, const char * format_name);

/* deterministic scalar functions: 0 exp, 1 tanh, 2 sigmoid, 3 silu, 4 exp(-exp), 5 v7 decay, 6 1/sqrt(x+1e-5) */
void orc_unary(int op, const float * x, float * y, int64_t n);

uint16_t orc_f32_to_f16(float f);
float    orc_f16_to_f32(uint16_t h);

#ifdef __cplusplus
}
#endif
#endif
