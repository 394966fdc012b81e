/* rwkv_testhooks.h -- entry points of lib/librwkv_testhooks.so, for tests/ ONLY (kernel-level parity checks against the CPU oracle).
 * librwkv.so does not export them: that library carries the reference's rwkv.h symbols (rwkv.h:38-221) + the rwkv_mi_* extensions of
 * rwkv_mi355x.h and nothing else. librwkv_testhooks.so is the same objects + csrc/testhooks.cpp. */
#ifndef RWKV_TESTHOOKS_H
#define RWKV_TESTHOOKS_H

#include "rwkv.h"

#if defined(__cplusplus)
extern "C" {
#endif

/* Test hook (used by tests/ only): presets the rolling hand-over tag of decode path 2 (the kernel
 * compares its low 16 bits; it advances by 8 per layer), so that a short run crosses the 16-bit wrap. false if path 2 is off. */
RWKV_API bool rwkv_mi_test_set_tag(struct rwkv_context * ctx, uint32_t base);

/* Test hook (used by tests/ only): the next n state initialisations inside rwkv_eval / rwkv_init fail (error-path tests). */
RWKV_API void rwkv_mi_test_fail_state_init(int n);

/* Sets the persistent kernel's abort word as a timed-out poll would: the next single-token step drains at once and the context falls back. */
RWKV_API bool rwkv_mi_test_force_abort(struct rwkv_context * ctx);

/* Test hook (used by tests/ only): how often the F16 matrix-core sequence kernel has been launched by this process. */
RWKV_API uint64_t rwkv_mi_test_mmf16_launches(void);
/* Test hook (used by tests/ only): launches of the EXACT F16 / F32 sequence kernel on the matrix cores (k_mmfx_seq, kernels.hip) so far. */
RWKV_API uint64_t rwkv_mi_test_mmfx_launches(void);

/* Test hook (used by tests/ only): how often the plain-order quantised sequence GEMM has been launched by this process. */
RWKV_API uint64_t rwkv_mi_test_mmq_fast_launches(void);

/* Test hook (used by tests/ only): the activation quantiser; n multiple of 32; d, s, isum have n/32 entries. */
RWKV_API bool rwkv_mi_test_quantize_act(const float * x, int64_t n, int8_t * q, float * d, float * s, int32_t * isum);

/* Test hook (used by tests/ only): y[i] = f(x[i]) with the kernels' deterministic scalar routines.
 * op: 0 exp, 1 tanh, 2 sigmoid, 3 silu, 4 exp(-exp(x)), 5 exp(-0.606531*sigmoid(x)), 6 1/sqrt(x + 1e-5). */
RWKV_API bool rwkv_mi_test_unary(int op, const float * x, float * y, int64_t n);

/* Test hook (used by tests/ only): y[T][N] = W[N][K] . x[T][K] through the production projection kernels. `w` holds N rows
 * in the FILE layout of `type` (rwkv.cpp type id: 0 FP32, 1 FP16, 2 Q4_0, 3 Q4_1, 7 Q5_0, 8 Q5_1, 9 Q8_0). */
RWKV_API bool rwkv_mi_test_mul_mat(int type, const void * w, int64_t K, int64_t N, const float * x, int64_t T, float * y);

#if defined(__cplusplus)
}
#endif

#endif
