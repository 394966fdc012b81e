// testhooks.cpp -- entry points for tests/ ONLY. NOT part of librwkv.so: `make` links them into a second library,
// lib/librwkv_testhooks.so (every product object + this file), which the kernel-level parity tests load (tests/gpu_lib.py). The product
// library exports the rwkv.h / rwkv_mi355x.h symbols and nothing else (csrc/rwkv.map).
#include "model.h"
#include "kdev.h"
#include "prefill_mm.h"
#include "rwkv_mi355x.h"
#include "rwkv_testhooks.h"

#include <cstdlib>

using namespace rwkvmi;

namespace rwkvmi {
// Test hook: the deterministic scalar functions applied elementwise (compared against the oracle's on the CPU).
__global__ __launch_bounds__(256) void k_test_unary(int op, const float * __restrict__ x, float * __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) {
        const float v = x[i];
        float r;
        switch (op) {
            case 0: r = det_expf(v); break;
            case 1: r = det_tanhf(v); break;
            case 2: r = sigmoid_f(v); break;
            case 3: r = v / (1.0f + det_expf(-v)); break;
            case 4: r = det_expf(-det_expf(v)); break;
            case 5: r = det_expf(sigmoid_f(v) * -0.606531f); break;
            case 6: r = 1.0f / sqrtf(v + 1e-5f); break;
            case 7: { const float a = wave_sum_f(v), b = wave_sum_f_ref(v); r = (__float_as_uint(a) == __float_as_uint(b)) ? 1.0f : 0.0f; break; }
            case 8: { const double xd = (double) v * (1.0 + 1e-9 * (double) (threadIdx.x & 63));
                      const double a = wave_sum_d(xd), b = wave_sum_d_ref(xd); r = (a == b) ? 1.0f : 0.0f; break; }
            case 9: { float m = fabsf(v); for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
                      int q = (int) v, sref = q; for (int o = 16; o > 0; o >>= 1) sref += __shfl_xor(sref, o, WAVE);
                      r = (half_max_f(fabsf(v)) == m && half_sum_i(q) == sref) ? 1.0f : 0.0f; break; }
            default: r = v; break;
        }
        y[i] = r;
    }
}
static void launch_test_unary(int op, const float * x, float * y, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_test_unary, dim3(1024), dim3(256), 0, st, op, x, y, n);
}

}  // namespace rwkvmi

extern "C" {

// Test hook: presets the persistent kernel's rolling 16-bit hand-over tag (e.g. a few tokens below its wrap).
RWKV_API bool rwkv_mi_test_set_tag(struct rwkv_context * ctx, uint32_t base) {
    if (!ctx->mega) return false;
    if (hipSetDevice(ctx->model->device) != hipSuccess) return false;
    return mega_v6_set_tag(ctx->mega, base, ctx->stream);
}

// Test hook: the persistent kernel's abort word set from the host (what a poll that timed out leaves behind).
RWKV_API bool rwkv_mi_test_force_abort(struct rwkv_context * ctx) {
    if (!ctx->mega || hipSetDevice(ctx->model->device) != hipSuccess) return false;
    return mega_v6_force_abort(ctx->mega, ctx->stream);
}

// Test hook: the next n state initialisations (state_from_host) of this process fail -- the error paths of rwkv_eval.
RWKV_API void rwkv_mi_test_fail_state_init(int n) { g_test_fail_state_init.store(n); }

// Test hook: launches of the F16 matrix-core sequence kernel (k_mmf16_seq) by this process so far.
RWKV_API uint64_t rwkv_mi_test_mmf16_launches(void) { return (uint64_t) g_mmf16_launches.load(); }
RWKV_API uint64_t rwkv_mi_test_mmfx_launches(void) { return (uint64_t) g_mmfx_launches.load(); }

// Test hook: launches of the plain-order quantised GEMM (k_mmq_fast) by this process so far.
RWKV_API uint64_t rwkv_mi_test_mmq_fast_launches(void) { return (uint64_t) g_mmq_fast_launches.load(); }

// Test hook: the activation quantiser (f32 -> Q8_0/Q8_1 blocks) on standalone buffers.
RWKV_API bool rwkv_mi_test_quantize_act(const float * x, int64_t n, int8_t * q, float * d, float * s, int32_t * isum) {
    g_last_error = RWKV_ERROR_NONE;
    RW_CHECK(RWKV_ERROR_ARGS, false, x && q && d && s && isum && n > 0 && n % 32 == 0, "bad arguments");
    const size_t nb = (size_t) n / 32;
    float * dx = nullptr; uint8_t * dq = nullptr;
    bool ok = hipMalloc((void **) &dx, (size_t) n * 4) == hipSuccess && hipMalloc((void **) &dq, (size_t) n + 3 * nb * 4 + 1024) == hipSuccess &&
              hipMemcpy(dx, x, (size_t) n * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        QAct qa;
        qa.q = (int8_t *) dq;
        qa.d = (float *) (dq + ((size_t) n + 255) / 256 * 256);
        qa.s = qa.d + nb;
        qa.isum = (int *) (qa.s + nb);
        launch_quantize_act(dx, 1, n, qa, nullptr);
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(q, qa.q, (size_t) n, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(d, qa.d, nb * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(s, qa.s, nb * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(isum, qa.isum, nb * 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (dx) (void) hipFree(dx);
    if (dq) (void) hipFree(dq);
    RW_CHECK(RWKV_ERROR_GRAPH, false, ok, "HIP error: %s", hipGetErrorString(hipGetLastError()));
    return true;
}

// Test hook: elementwise deterministic scalar functions on the device.
RWKV_API bool rwkv_mi_test_unary(int op, const float * x, float * y, int64_t n) {
    g_last_error = RWKV_ERROR_NONE;
    RW_CHECK(RWKV_ERROR_ARGS, false, x && y && n > 0, "bad arguments");
    float *dx = nullptr, *dy = nullptr;
    bool ok = hipMalloc((void **) &dx, (size_t) n * 4) == hipSuccess && hipMalloc((void **) &dy, (size_t) n * 4) == hipSuccess &&
              hipMemcpy(dx, x, (size_t) n * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        launch_test_unary(op, dx, dy, n, nullptr);
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(y, dy, (size_t) n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (dx) (void) hipFree(dx);
    if (dy) (void) hipFree(dy);
    RW_CHECK(RWKV_ERROR_GRAPH, false, ok, "HIP error: %s", hipGetErrorString(hipGetLastError()));
    return true;
}

// Test hook: y[T][N] = W[N][K] . x[T][K] through the production projection kernels (load-time re-pack, activation
// quantiser, single-token or token-tiled kernel) on standalone buffers. W is in the FILE layout of `type`.
RWKV_API bool rwkv_mi_test_mul_mat(int type, const void * w, int64_t K, int64_t N, const float * x, int64_t T, float * y) {
    g_last_error = RWKV_ERROR_NONE;
    RW_CHECK(RWKV_ERROR_ARGS, false, dtype_supported(type) && w && x && y && K > 0 && N > 0 && T > 0, "bad arguments");
    RW_CHECK(RWKV_ERROR_ARGS, false, K % 32 == 0, "K must be a multiple of 32");
    const uint64_t wbytes = tensor_nbytes(type, K, N, 1);
    const int64_t nblk = K * N / 32;
    void *d_raw = nullptr, *d_x = nullptr, *d_y = nullptr, *d_q = nullptr, *d_planes = nullptr;
    bool ok = true;
    auto chk = [&](hipError_t e) { if (e != hipSuccess) { global_fail(RWKV_ERROR_GRAPH, __FILE__, __LINE__, "hip call", "HIP error: %s", hipGetErrorString(e)); ok = false; } return ok; };
    DevTensor W;
    W.type = type; W.ndim = 2; W.ne[0] = K; W.ne[1] = N; W.nbytes = wbytes;
    QAct qa;
    const size_t nbk = (size_t) T * (size_t)(K / 32);
    if (chk(hipMalloc(&d_raw, wbytes)) && chk(hipMalloc(&d_x, (size_t) T * K * 4)) && chk(hipMalloc(&d_y, (size_t) T * N * 4)) &&
        chk(hipMalloc(&d_q, (size_t) T * K + 3 * nbk * 4 + 1024)) && chk(hipMalloc(&d_planes, (size_t) nblk * 40 + 1024)) &&
        chk(hipMemcpy(d_raw, w, wbytes, hipMemcpyHostToDevice)) && chk(hipMemcpy(d_x, x, (size_t) T * K * 4, hipMemcpyHostToDevice))) {
        hipStream_t st = nullptr;
        if (dtype_quantized(type)) {
            W.qs = (uint8_t *) d_planes;
            W.qh = (uint32_t *) ((uint8_t *) d_planes + (size_t) nblk * 32);
            W.sc = (uint8_t *) d_planes + (size_t) nblk * 36;
            launch_repack(type, (const uint8_t *) d_raw, nblk, W.qs, W.qh, W.sc, st);
            qa.q = (int8_t *) d_q;
            qa.d = (float *) ((uint8_t *) d_q + (((size_t) T * K + 255) / 256) * 256);
            qa.s = qa.d + nbk;
            qa.isum = (int *) (qa.s + nbk);
            if (T >= k_mfma_min_tokens) {
                // sequence mode: tile-major quantiser + int8 GEMM on the matrix cores (what the engine does for T >= 32)
                void * d_tile = nullptr;
                if (chk(hipMalloc(&d_tile, tile_act_bytes(T, K)))) {
                    const TileAct ta = tile_act_at(d_tile, T, K);
                    launch_quantize_act_tiles((const float *) d_x, T, K, type, ta, st);
                    // (with the workspace of the split walk, as the engine runs it: few-tile shapes take that path)
                    MmqWs ws;
                    void * d_ws = nullptr;
                    const size_t ws_part = (size_t) 16 << 20;
                    if (chk(hipMalloc(&d_ws, ws_part + 1024 * sizeof(int))) && chk(hipMemsetAsync((uint8_t *) d_ws + ws_part, 0, 1024 * sizeof(int), st))) {
                        ws.part = (float *) d_ws; ws.part_bytes = ws_part; ws.counters = (int *) ((uint8_t *) d_ws + ws_part); ws.n_counters = 1024;
                    }
                    if (!launch_mmq_mfma(W, ta, T, (float *) d_y, N, Epi(), st, &ws)) ok = false;
                    chk(hipDeviceSynchronize());
                    if (const char * rep = getenv("RWKV_MI_TIME_MM")) {   // kernel timing aid (tools/gemm_bench.py): average of n back-to-back launches
                        const int n = atoi(rep);
                        hipEvent_t e0, e1;
                        if (n > 0 && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                            (void) hipEventRecord(e0, st);
                            for (int i = 0; i < n; i++) (void) launch_mmq_mfma(W, ta, T, (float *) d_y, N, Epi(), st, &ws);
                            (void) hipEventRecord(e1, st);
                            (void) hipEventSynchronize(e1);
                            float ms = 0.0f;
                            (void) hipEventElapsedTime(&ms, e0, e1);
                            fprintf(stderr, "[time_mm] type %d K %lld N %lld T %lld: %.2f us per launch, %.1f TOP/s\n", type, (long long) K, (long long) N, (long long) T,
                                    ms * 1e3 / n, 2.0 * K * N * T / (ms * 1e-3 / n) / 1e12);
                            (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
                        }
                    }
                    (void) hipFree(d_tile);
                    if (d_ws) (void) hipFree(d_ws);
                    free_pf(W);
                }
            } else {
                launch_quantize_act((const float *) d_x, T, K, qa, st);
                launch_matvec_q(W, qa, T, (float *) d_y, N, Epi(), st);
            }
        } else {
            W.data = d_raw;
            launch_matvec_f(W, (const float *) d_x, K, T, (float *) d_y, N, Epi(), st);
        }
        chk(hipDeviceSynchronize());
        chk(hipGetLastError());
        if (ok) chk(hipMemcpy(y, d_y, (size_t) T * N * 4, hipMemcpyDeviceToHost));
    }
    for (void * p : {d_raw, d_x, d_y, d_q, d_planes}) if (p) (void) hipFree(p);
    return ok;
}

}  // extern "C"
