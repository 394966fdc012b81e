// kernels.h -- host-callable launchers of the gfx950 kernels (kernels.hip). All launches are asynchronous on `st`.
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>
#include "common.h"

namespace rwkvmi {

// Activations quantised to ggml's Q8_0/Q8_1 form for the integer dot:
//   q [T][K] int8, d [T][K/32] = f32(fp16(amax/127)), s [T][K/32] = f32(fp16(d * sum q)), isum [T][K/32] = sum q.
struct QAct {
    int8_t * q = nullptr;
    float *  d = nullptr;
    float *  s = nullptr;
    int *    isum = nullptr;
};

// Fused output transforms of the projection kernels (what follows each ggml_mul_mat in rwkv_graph.inc).
enum EpiOp : int {
    EPI_NONE = 0,
    EPI_SIGMOID,          // sigmoid(acc)
    EPI_RELU_SQ,          // relu(acc)^2
    EPI_SILU,             // acc * sigmoid(acc)
    EPI_TANH,             // tanh(acc)
    EPI_ADD_RES,          // res[t,n] + acc
    EPI_SIGMUL_ADD_RES,   // res[t,n] + sigmoid(aux[t,n]) * acc            (v4-v6 channel mixing output)
    EPI_BIAS_SIGMOID,     // sigmoid(acc + bias[n])                         (v7 a, v-gate)
    EPI_V6_DECAY,         // exp(-exp(acc + bias[n]))                       (rwkv_graph.inc:365-367)
    EPI_V7_DECAY,         // exp(-0.606531 * sigmoid(acc + bias[n]))        (rwkv_graph.inc:425-430)
};

struct Epi {
    int op = EPI_NONE;
    const float * bias = nullptr;  // [N]
    const float * res = nullptr;   // [T][ldy]
    const float * aux = nullptr;   // [T][ldy]
};

// y[t*ldy + n] = epi( W[n,:] . x[t,:] ),  W quantised (planes), x pre-quantised.
void launch_matvec_q(const DevTensor & W, const QAct & x, int64_t T, float * y, int64_t ldy, const Epi & epi, hipStream_t st);
// Same for F32 / F16 weights with f32 activations x[t*ldx + k] (F16 weights: activations rounded to fp16 on load).
void launch_matvec_f(const DevTensor & W, const float * x, int64_t ldx, int64_t T, float * y, int64_t ldy, const Epi & epi, hipStream_t st);

extern std::atomic<unsigned long long> g_mmfx_launches;    // launches of the exact F16 / F32 matrix-core sequence kernel (k_mmfx_seq) so far (test hook reads it)
extern std::atomic<unsigned long long> g_mmf16_launches;   // launches of the F16 matrix-core sequence kernel so far (test hook reads it)
void matvec_f_release_stream(hipStream_t st);   // frees the split-K workspace kept for this stream (a context's own stream, at its destruction)

// x[T][K] f32 -> QAct
void launch_quantize_act(const float * x, int64_t T, int64_t K, const QAct & out, hipStream_t st);

// Embedding gather (+ convert) + LayerNorm(ln0): x[t,:] = LN(emb[tokens[t], :]) * w + b     (rwkv_graph.inc:655-658)
void launch_embed_ln0(const DevTensor & emb, const uint32_t * tokens, int64_t T, int64_t D, const float * w, const float * b, float * x, hipStream_t st);
// y[t,:] = norm(x[t,:], 1e-5) * w + b                                                     (rwkv_operators.inc:93-97)
void launch_layernorm(const float * x, int64_t T, int64_t D, const float * w, const float * b, float * y, hipStream_t st);

// Token-shift mixes (rwkv_carry_x + the lerps, rwkv_graph.inc:56-82,93-109,214-241,402-413,488-501,516-521,536-538).
// x_prev[t] = t ? xn[t-1] : carry_in.   mode 0: out_f = xn*c_f + (x_prev - x_prev*c_f)   (v4, v5)
//                                       mode 1: out_f = (x_prev - xn)*c_f + xn           (v6 xxx / ffn, v7)
// Also writes carry_out = xn[T-1] and (optionally) sx = x_prev - xn.
struct MixArgs {
    const float * xn = nullptr;
    const float * carry_in = nullptr;
    float * carry_out = nullptr;
    int n_out = 0;
    const float * coef[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float * out[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float * sx = nullptr;
    int mode = 0;
};
void launch_mix(const MixArgs & a, int64_t T, int64_t D, hipStream_t st);

// v6 data-dependent mix, second stage (rwkv_graph.inc:323-346): for f in (w,k,v,r,g):
//   out_f[t,d] = (sum_m W2[f][d][m] * tl[t][f*R+m] + maa_f[d]) * sx[t,d] + xn[t,d]
struct V6Mix2Args {
    const float * w2 = nullptr;   // [5][R][D] f32 (transposed at load)
    const float * tl = nullptr;   // [T][5R] (already tanh'ed)
    const float * maa[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const float * sx = nullptr;
    const float * xn = nullptr;
    float * out[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};
void launch_v6_mix2(const V6Mix2Args & a, int64_t T, int64_t D, int64_t R, hipStream_t st);

// RWKV-4 WKV recurrence fused with r * wkv (rwkv_graph.inc:119-161,178-195). r is already sigmoid'ed.
void launch_wkv4(const float * k, const float * v, const float * r, const float * time_first, const float * time_decay,
                 const float * aa_in, const float * bb_in, const float * pp_in, float * aa_out, float * bb_out, float * pp_out,
                 float * out, int64_t T, int64_t D, hipStream_t st);

// RWKV-5/6 WKV recurrence (ggml_rwkv_wkv6; rwkv_graph.inc:275,370). state[h][i=key][j=value].
// u: [H] (u_per_chan = 0) or [H*S]; w: [H] (w_mode 0), [H*S] (w_mode 1) or per token [T][H*S] (w_mode 2).
void launch_wkv6(const float * r, const float * k, const float * v, const float * u, int u_per_chan, const float * w, int w_mode,
                 const float * state_in, float * state_out, float * out, int64_t T, int64_t H, int64_t S, hipStream_t st);

// RWKV-7 WKV recurrence (rwkv_operators_wkv_v7.inc:37-107). state[h][i=value][j=key]; a = -kk, b = kk*a_gate.
void launch_wkv7(const float * r, const float * w, const float * k, const float * v, const float * a, const float * b,
                 const float * state_in, float * state_out, float * out, int64_t T, int64_t H, int64_t S, hipStream_t st);
// sequence form, head size 64 (prefill.hip): four rows per wave, the row's two ordered sums as DPP chains
bool launch_wkv7_seq(const float * r, const float * w, const float * k, const float * v, const float * a, const float * b,
                     const float * state_in, float * state_out, float * out, int64_t T, int64_t H, hipStream_t st);

// Per-head group norm * ln_x.w + ln_x.b, optional v7 bonus (+ v * sum_head(k*r*r_k)), optional gate multiply
// (rwkv_graph.inc:280-289, 375-382, 465-479). In place on x[T][H*S].
void launch_groupnorm(float * x, const float * lw, const float * lb, float eps, const float * gate,
                      const float * v7_k, const float * v7_r, const float * v7_v, const float * v7_rk,
                      int64_t T, int64_t H, int64_t S, hipStream_t st);

// v7 key path (rwkv_graph.inc:432-437,460): kk = l2norm_head(k*k_k); k' = k + (a*k*k_a - k*k_a); na = -kk; nb = kk*a
void launch_v7_kprep(const float * k, const float * a, const float * k_k, const float * k_a,
                     float * k_out, float * neg_kk, float * kk_a, int64_t T, int64_t H, int64_t S, hipStream_t st);

// v7 value residual (rwkv_graph.inc:439-453): v = v + (v_first - v) * gate   (gate already sigmoid'ed)
void launch_v7_vmix(float * v, const float * v_first, const float * gate, int64_t n, hipStream_t st);

// Plain elementwise helpers
void launch_mul(float * y, const float * a, const float * b, int64_t n, hipStream_t st);   // y = a * b
void launch_copy_f32(float * dst, const float * src, int64_t n, hipStream_t st);
void launch_fill_state_v4(float * state, int64_t n_layer, int64_t D, hipStream_t st);      // zeros + pp = -1e30

// argmax over logits[n] -> *out (first index of the maximum), used by the on-device greedy decode loop
void launch_argmax(const float * logits, int64_t n, uint32_t * out, hipStream_t st);

// load-time transpose of att.time_maa_w2: [5][D][R] -> [5][R][D]
void launch_transpose_w2(const float * src, float * dst, int64_t D, int64_t R, hipStream_t st);

// load-time re-pack of quantised blocks (file layout) into planes; see DevTensor
void launch_repack(int type, const uint8_t * raw, int64_t n_blocks, uint8_t * qs, uint32_t * qh, void * sc, hipStream_t st);

}  // namespace rwkvmi
