// fused_v7.hip -- the RWKV-7 single-token (decode) layer as FIVE launches instead of ~30 graph-op kernels
// (rwkv_att_v7, rwkv_graph.inc:387-482; rwkv_wkv_v7_impl, rwkv_operators_wkv_v7.inc:37-107; rwkv_ffn_v7, rwkv_graph.inc:533-543):
//
//   A  k7_att_in    LN1 + token shift + the ONE static mix this workgroup's matrix consumes (x_rwkvag) -> quantised image or f16-rounded
//                   vector in LDS -> rows of R / K / V (quantised) or of the first low-rank stages W1 (tanh), A1, G1 (sigmoid), V1
//   B  k7_head      per head: second low-rank stages W2 / A2 / G2 / V2 for the head's 64 channels (decay, a, g, value gate), key
//                   path (l2-norm, k += (a - 1) k k_a), value residual, WKV7 recurrence, GroupNorm * ln_x + bonus, gate, quantise
//   C  k6_proj_res  output projection + residual add                                   (shared with the RWKV-6 path, fused_v6.hip)
//   D  k6_ffn_kr    LN2 + token shift + mix + quantise -> key rows (relu^2, quantised per 32 rows); no receptance in RWKV-7
//   E  k6_proj_res  value projection + residual add
//
// Every launch boundary is an all-to-all dependency (a full-vector LayerNorm, or a projection consuming a whole vector). Arithmetic
// and reduction orders are those of the per-op kernels (kernels.hip) and of the CPU oracle: bit-identical.
// The low-rank matrices stay in their file dtype (F16 in a quantised checkpoint, F32 when quantised from an FP32 file): their rows
// run in ggml's 32-partial AVX2 dot order like k_mvf (4 lanes per row), activations rounded to fp16 first for F16 weights.
#include "fused_blocks.h"

#include <hip/hip_ext.h>

namespace rwkvmi {

// 16 rows per wave (lane = 4 * row + q; lane q of a row keeps partials 8q .. 8q+7 of ggml's 32): the k_mvf inner loop on an
// activation vector already staged (and, for F16 weights, fp16-rounded) in LDS. Returns the row sum in every lane of the row's quad.
template <bool F16> struct LrBatch { static constexpr int UB = F16 ? 16 : 8; int4 ra[UB], rb[F16 ? 1 : UB]; };

template <bool F16>
__device__ __forceinline__ void lr_issue(LrBatch<F16> & bt, const void * __restrict__ W, int64_t row, int K, int s0, int q) {
    const int nsteps = K / 32;
#pragma unroll
    for (int u = 0; u < LrBatch<F16>::UB; u++) {
        const int sidx = s0 + u < nsteps ? s0 + u : nsteps - 1;
        const int64_t e0 = row * K + 32 * sidx + 8 * q;
        if constexpr (F16) bt.ra[u] = ldw16(reinterpret_cast<const uint16_t *>(W) + e0);
        else { bt.ra[u] = ldw16(reinterpret_cast<const float *>(W) + e0); bt.rb[u] = ldw16(reinterpret_cast<const float *>(W) + e0 + 4); }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <bool F16>
__device__ __forceinline__ void lr_consume(const LrBatch<F16> & bt, int K, int s0, int q, const float * l_x, float (&acc)[8]) {
    const int nsteps = K / 32;
#pragma unroll
    for (int u = 0; u < LrBatch<F16>::UB; u++) {
        if (s0 + u < nsteps) {
            const int sidx = s0 + u;
            float w[8];
            if constexpr (F16) {
                const unsigned uu[4] = {(unsigned) bt.ra[u].x, (unsigned) bt.ra[u].y, (unsigned) bt.ra[u].z, (unsigned) bt.ra[u].w};
#pragma unroll
                for (int i = 0; i < 4; i++) { w[2 * i] = h2f_bits((uint16_t) (uu[i] & 0xFFFFu)); w[2 * i + 1] = h2f_bits((uint16_t) (uu[i] >> 16)); }
            } else {
                w[0] = __int_as_float(bt.ra[u].x); w[1] = __int_as_float(bt.ra[u].y); w[2] = __int_as_float(bt.ra[u].z); w[3] = __int_as_float(bt.ra[u].w);
                w[4] = __int_as_float(bt.rb[u].x); w[5] = __int_as_float(bt.rb[u].y); w[6] = __int_as_float(bt.rb[u].z); w[7] = __int_as_float(bt.rb[u].w);
            }
            const float4 xa = *reinterpret_cast<const float4 *>(l_x + 32 * sidx + 8 * q);
            const float4 xb = *reinterpret_cast<const float4 *>(l_x + 32 * sidx + 8 * q + 4);
            acc[0] = fmaf(w[0], xa.x, acc[0]); acc[1] = fmaf(w[1], xa.y, acc[1]); acc[2] = fmaf(w[2], xa.z, acc[2]); acc[3] = fmaf(w[3], xa.w, acc[3]);
            acc[4] = fmaf(w[4], xb.x, acc[4]); acc[5] = fmaf(w[5], xb.y, acc[5]); acc[6] = fmaf(w[6], xb.z, acc[6]); acc[7] = fmaf(w[7], xb.w, acc[7]);
        }
    }
}

// `first` holds the batch of steps [0, UB), already in flight (issued before the caller's prologue); two batches stay in flight
// (DB = false: short rows -- one batch in flight, the buffer is reused; saves the second buffer's registers)
template <bool F16, bool DB = true>
__device__ __forceinline__ float lr_row16(LrBatch<F16> & first, const void * __restrict__ W, int64_t row, int K, const float * l_x, int lane) {
    constexpr int UB = LrBatch<F16>::UB;
    const int q = lane & 3;
    const int nsteps = K / 32;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.0f;
    if constexpr (DB) {
        LrBatch<F16> second;
        for (int s0 = 0;; s0 += 2 * UB) {
            const bool more1 = s0 + UB < nsteps;
            if (more1) lr_issue<F16>(second, W, row, K, s0 + UB, q);
            lr_consume<F16>(first, K, s0, q, l_x, acc);
            if (!more1) break;
            const bool more2 = s0 + 2 * UB < nsteps;
            if (more2) lr_issue<F16>(first, W, row, K, s0 + 2 * UB, q);
            lr_consume<F16>(second, K, s0 + UB, q, l_x, acc);
            if (!more2) break;
        }
    } else {
        for (int s0 = 0;; s0 += UB) {
            lr_consume<F16>(first, K, s0, q, l_x, acc);
            if (s0 + UB >= nsteps) break;
            lr_issue<F16>(first, W, row, K, s0 + UB, q);
        }
    }
    float ps[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        float v = acc[e];
        v += __shfl_xor(v, 2, WAVE);  // ps[i] += ps[i + 16]
        v += __shfl_xor(v, 1, WAVE);  // ps[i] += ps[i + 8]
        ps[e] = v;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) ps[e] += ps[e + 4];
    return (ps[0] + ps[1]) + (ps[2] + ps[3]);
}

// The same dot with EIGHT lanes per row (lane = 8 * row + q; lane q keeps partials 4q .. 4q+3 of ggml's 32) for the long rows of the
// first low-rank stages (K = n_embed): a partial is a chain of K / 32 single FMAs in increasing k, so a row cannot be split along k,
// but its 32 chains can be spread over more lanes. At 8 bytes per lane and step (F16) two batches of 24 steps (48 of the 80 steps of a 2560-long row)
// are in flight before the prologue within the registers the quantised row groups of the same launch use anyway, where four lanes per
// row (16 bytes per step, 2 x 16 steps) paid three more memory round trips
// behind it; and a wave carries 8 rows instead of 16, so the rows spread over twice the waves.
template <bool F16> struct Lr8Batch { static constexpr int UB = F16 ? 24 : 12; typename std::conditional<F16, int2, int4>::type r[UB]; };

template <bool F16>
__device__ __forceinline__ void lr8_issue(Lr8Batch<F16> & bt, const void * __restrict__ W, int64_t row, int K, int s0, int q) {
    const int nsteps = K / 32;
#pragma unroll
    for (int u = 0; u < Lr8Batch<F16>::UB; u++) {
        const int sidx = s0 + u < nsteps ? s0 + u : nsteps - 1;
        const int64_t e0 = row * K + 32 * sidx + 4 * q;
        if constexpr (F16) bt.r[u] = *reinterpret_cast<const int2 *>(reinterpret_cast<const uint16_t *>(W) + e0);
        else bt.r[u] = ldw16(reinterpret_cast<const float *>(W) + e0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <bool F16>
__device__ __forceinline__ void lr8_consume(const Lr8Batch<F16> & bt, int K, int s0, int q, const float * l_x, float (&acc)[4]) {
    const int nsteps = K / 32;
#pragma unroll
    for (int u = 0; u < Lr8Batch<F16>::UB; u++) {
        if (s0 + u < nsteps) {
            const int sidx = s0 + u;
            float w[4];
            if constexpr (F16) {
                const unsigned u0 = (unsigned) bt.r[u].x, u1 = (unsigned) bt.r[u].y;
                w[0] = h2f_bits((uint16_t) (u0 & 0xFFFFu)); w[1] = h2f_bits((uint16_t) (u0 >> 16));
                w[2] = h2f_bits((uint16_t) (u1 & 0xFFFFu)); w[3] = h2f_bits((uint16_t) (u1 >> 16));
            } else {
                w[0] = __int_as_float(bt.r[u].x); w[1] = __int_as_float(bt.r[u].y); w[2] = __int_as_float(bt.r[u].z); w[3] = __int_as_float(bt.r[u].w);
            }
            const float4 xa = *reinterpret_cast<const float4 *>(l_x + 32 * sidx + 4 * q);
            acc[0] = fmaf(w[0], xa.x, acc[0]); acc[1] = fmaf(w[1], xa.y, acc[1]); acc[2] = fmaf(w[2], xa.z, acc[2]); acc[3] = fmaf(w[3], xa.w, acc[3]);
        }
    }
}

// `first` and `second` hold steps [0, UB) and [UB, 2 UB), already in flight (issued before the caller's prologue)
template <bool F16>
__device__ __forceinline__ float lr8_row(Lr8Batch<F16> & first, Lr8Batch<F16> & second, const void * __restrict__ W, int64_t row, int K, const float * l_x, int lane) {
    constexpr int UB = Lr8Batch<F16>::UB;
    const int q = lane & 7;
    const int nsteps = K / 32;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s0 = 0;; s0 += 2 * UB) {
        lr8_consume<F16>(first, K, s0, q, l_x, acc);
        if (s0 + UB >= nsteps) break;
        if (s0 + 2 * UB < nsteps) lr8_issue<F16>(first, W, row, K, s0 + 2 * UB, q);
        lr8_consume<F16>(second, K, s0 + UB, q, l_x, acc);
        if (s0 + 2 * UB >= nsteps) break;
        if (s0 + 3 * UB < nsteps) lr8_issue<F16>(second, W, row, K, s0 + 3 * UB, q);
    }
    // ggml's fold of the 32 partials: ps[i] += ps[i + 16], ps[i] += ps[i + 8], ps[i] += ps[i + 4], (ps0 + ps1) + (ps2 + ps3)
    float ps[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float v = acc[e];
        v += __shfl_xor(v, 4, WAVE);
        v += __shfl_xor(v, 2, WAVE);
        v += __shfl_xor(v, 1, WAVE);
        ps[e] = v;
    }
    return (ps[0] + ps[1]) + (ps[2] + ps[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// A: LN1 + shift + one mix -> R / K / V rows or first low-rank stage rows
// ---------------------------------------------------------------------------------------------------------------

struct P7A {
    const float * x; const float * ln_w; const float * ln_b; const float * att_xx_in;
    const float * coef_q[3];      // token-shift mix coefficients of the quantised matrices' inputs (v7: x_rwkvag rows r, k, v; v4: time_mix_r / _k / _v)
    const float * coef_lr[4];     // ... of the low-rank first stages' inputs (v7: x_rwkvag rows w, a, g, v)
    int mix_mode;                 // 1: (x_prev - xn) * c + xn (v6, v7);  0: xn * c + (x_prev - x_prev * c) (v4, v5)
    int epi_q[3];                 // epilogue of the quantised rows: 0 none, 1 sigmoid (v4 receptance)
    float * att_xx_out;
    WPl wq[3];                    // receptance, key, value (quantised)
    const void * lr[4];           // w1, a1, g1, v1 (F16 / F32, [rank][D]); v1 may be null (layer 0)
    int rank[4];
    float * out_q[3];             // r, k, v   [D]
    float * out_lr[4];            // tanh(W1 xw), A1 xa, sigmoid(G1 xg), V1 xv
    int64_t D;
    int lr_groups;                // low-rank workgroups come FIRST in the grid: their rows are the longest dependent chain of the launch
};

template <int FMT, bool LRF16>
__global__ __launch_bounds__(256) void k7_att_in(P7A p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t D = p.D;
    const int nb = (int) (D / 32);
    float * l_row = reinterpret_cast<float *>(smem);                       // D floats: x, then x - mean, then the mix (low-rank groups)
    unsigned char * l_qv = smem + D * 4;                                    // lohi image (quantised groups)
    double * red = reinterpret_cast<double *>(l_qv + ((qvec_bytes(D) + 15) / 16) * 16);
    const QVec lq = qvec_at(l_qv, D);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // job of this workgroup: quantised groups of 32 rows (r, k, v), then low-rank groups of 32 rows (8 per wave)
    const int64_t G = D / 32;
    int mat = -1, lrm = -1;
    int64_t grp = blockIdx.x;
    if (grp >= p.lr_groups) { grp -= p.lr_groups; mat = (int) (grp / G); grp -= (int64_t) mat * G; if (mat > 2) return; }
    else {
        for (int m = 0; m < 4; m++) {
            const int64_t g = p.lr[m] ? (p.rank[m] + 31) / 32 : 0;
            if (grp < g) { lrm = m; break; }
            grp -= g;
        }
        if (lrm < 0) return;
    }
    const float * coef = mat >= 0 ? p.coef_q[mat] : p.coef_lr[lrm];

    // weights of the quantised job go in flight before the prologue
    const int64_t row0 = grp * 32 + wave * 8;
    Batch<FMT, 8, 2> bt;
    Lr8Batch<LRF16> lb0, lb1;
    const int64_t lr_row = grp * 32 + wave * 8 + (lane >> 3);
    const int64_t lr_rowc = lrm >= 0 ? (lr_row < p.rank[lrm] ? lr_row : p.rank[lrm] - 1) : 0;
    if (mat >= 0) batch_issue<FMT, 8, 2>(bt, p.wq[mat].qs, p.wq[mat].qh, p.wq[mat].sc, row0, D, nb, 0, lane);
    else {
        lr8_issue<LRF16>(lb0, p.lr[lrm], lr_rowc, (int) D, 0, lane & 7);
        lr8_issue<LRF16>(lb1, p.lr[lrm], lr_rowc, (int) D, Lr8Batch<LRF16>::UB, lane & 7);
    }

    fill_row(l_row, p.x, D);
    __syncthreads();
    const double sacc = ln_partial_sum(l_row, D);
    const float mean = (float) (block_sum_d_1b(sacc, red) / (double) D);
    const double s2 = ln_partial_var(l_row, D, mean);
    const float var = (float) (block_sum_d_1b(s2, red + 256) / (double) D);
    const float scale = 1.0f / sqrtf(var + 1e-5f);
    // D % 256 == 0: whole waves, half-wave = one 32-block. Parameter loads of several elements go out together.
    auto fin = [&](int64_t i, float lw, float lb, float pv, float cf) -> float {
        const float y = l_row[i] * scale;
        const float yw = y * lw;
        const float xn = yw + lb;
        if (blockIdx.x == 0) p.att_xx_out[i] = xn;
        if (p.mix_mode == 1) {
            const float sx = pv - xn;
            const float sm = sx * cf;
            return sm + xn;
        }
        const float xc = xn * cf, pc = pv * cf;
        return xc + (pv - pc);
    };
    auto put = [&](int64_t i, float mx) {
        if (mat >= 0) {
            int qi, isum; float d16, s16;
            quant_block32(mx, qi, d16, s16, isum);
            qvec_store(lq, nb, (int) (i >> 5), (int) (i & 31), qi, d16, s16, isum);
        } else {
            l_row[i] = LRF16 ? round_f16(mx) : mx;       // own element: no hazard
        }
    };
    int64_t i0 = threadIdx.x;
    for (; i0 + 4 * 256 < D; i0 += 5 * 256) {
        float lw[5], lb[5], pv[5], cf[5], mx[5];
#pragma unroll
        for (int u = 0; u < 5; u++) { const int64_t i = i0 + u * 256; lw[u] = p.ln_w[i]; lb[u] = p.ln_b[i]; pv[u] = p.att_xx_in[i]; cf[u] = coef[i]; }
#pragma unroll
        for (int u = 0; u < 5; u++) mx[u] = fin(i0 + u * 256, lw[u], lb[u], pv[u], cf[u]);
#pragma unroll
        for (int u = 0; u < 5; u++) put(i0 + u * 256, mx[u]);
    }
    for (; i0 < D; i0 += 256) put(i0, fin(i0, p.ln_w[i0], p.ln_b[i0], p.att_xx_in[i0], coef[i0]));
    __syncthreads();
    if (mat >= 0) {
        if (row0 >= D) return;
        float res[8];
        rows_finish<FMT, 8, 2>(bt, p.wq[mat].qs, p.wq[mat].qh, p.wq[mat].sc, row0, D, nb, lq, lane, res);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < 8; r++) if (row0 + r < D) p.out_q[mat][row0 + r] = p.epi_q[mat] == 1 ? sigmoid_f(res[r]) : res[r];
        }
    } else {
        float v = lr8_row<LRF16>(lb0, lb1, p.lr[lrm], lr_rowc, (int) D, l_row, lane);
        if (lrm == 0) v = det_tanhf(v);
        else if (lrm == 2) v = sigmoid_f(v);
        if ((lane & 7) == 0 && lr_row < p.rank[lrm]) p.out_lr[lrm][lr_row] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// B: one workgroup (4 waves) per head
// ---------------------------------------------------------------------------------------------------------------

struct P7B {
    const float * lr_in[4];       // tanh(W1 xw), A1 xa, sigmoid(G1 xg), V1 xv
    const void * lr2[4];          // w2, a2, g2, v2 ([D][rank]); v2 null on layer 0
    int rank[4];
    const float * w0; const float * a0; const float * v0;
    const float * r; const float * k; const float * v;
    const float * k_k; const float * k_a; const float * r_k;
    const float * lnx_w; const float * lnx_b;
    float * v_first; int layer0;
    const float * state_in; float * state_out;
    void * y_out;                 // lohi image of D elements for the output projection
    int64_t D;
};

template <bool LRF16>
__global__ __launch_bounds__(512) void k7_head(P7B p) {
    constexpr int S = 64;
    __shared__ __attribute__((aligned(16))) float l_lr[4][512];     // the four low-rank vectors (fp16-rounded for F16 weights)
    __shared__ float l_ch[4][S];                                     // second-stage results of the head's channels: w, a, g, value gate
    __shared__ float l_r[S], l_w[S], l_k[S], l_a[S], l_b[S];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t h = blockIdx.x, c = h * S + lane;
    const int64_t D = p.D;
    const int mtx = wave & 3, half = wave >> 2;
    LrBatch<LRF16> lbA, lbB;
    if (p.lr2[mtx]) {
        lr_issue<LRF16>(lbA, p.lr2[mtx], h * S + (2 * half) * 16 + (lane >> 2), p.rank[mtx], 0, lane & 3);
        lr_issue<LRF16>(lbB, p.lr2[mtx], h * S + (2 * half + 1) * 16 + (lane >> 2), p.rank[mtx], 0, lane & 3);
    }
    // per-channel operands of the second stages' epilogues and of wave 0's recurrence: in flight now, not behind the phase that uses them
    // (each of those loads was a memory round trip of its own on the head's critical path)
    float e0[2];
#pragma unroll
    for (int ps = 0; ps < 2; ps++) {
        const int64_t ci = h * S + (2 * half + ps) * 16 + (lane >> 2);
        e0[ps] = mtx == 0 ? p.w0[ci] : (mtx == 1 ? p.a0[ci] : ((mtx == 3 && p.lr2[3]) ? p.v0[ci] : 0.0f));
    }
    float rv = 0.0f, kv0 = 0.0f, vv = 0.0f, c_kk = 0.0f, c_ka = 0.0f, c_rk = 0.0f, c_lw = 0.0f, c_lb = 0.0f, vf = 0.0f;
    if (wave == 0) {
        rv = p.r[c]; kv0 = p.k[c]; vv = p.v[c]; c_kk = p.k_k[c]; c_ka = p.k_a[c]; c_rk = p.r_k[c]; c_lw = p.lnx_w[c]; c_lb = p.lnx_b[c];
        if (!p.layer0) vf = p.v_first[c];
    }
    // the state row of this lane's value index goes in flight early (wave 0 only)
    float s[S];
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < S; j += 4) {
            const float4 q4 = *reinterpret_cast<const float4 *>(p.state_in + h * S * S + (int64_t) lane * S + j);
            s[j] = q4.x; s[j + 1] = q4.y; s[j + 2] = q4.z; s[j + 3] = q4.w;
        }
    }
    for (int m = 0; m < 4; m++) {
        if (!p.lr2[m]) continue;
        for (int i = threadIdx.x; i < p.rank[m]; i += 512) l_lr[m][i] = LRF16 ? round_f16(p.lr_in[m][i]) : p.lr_in[m][i];
    }
    __syncthreads();
    // second low-rank stages: 8 waves = 4 matrices (w2, a2, g2, v2) x 2 halves of the head's 64 rows, 16 rows per pass; the weights
    // of both passes were put in flight at the top of the kernel
    {
        if (p.lr2[mtx]) {
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                const int i = (2 * half + ps) * 16 + (lane >> 2);
                float v = lr_row16<LRF16, false>(ps == 0 ? lbA : lbB, p.lr2[mtx], h * S + i, p.rank[mtx], l_lr[mtx], lane);
                if (mtx == 0) v = det_expf(sigmoid_f(v + e0[ps]) * -0.606531f);
                else if (mtx == 1) v = sigmoid_f(v + e0[ps]);
                else if (mtx == 3) v = sigmoid_f(v + e0[ps]);
                if ((lane & 3) == 0) l_ch[mtx][i] = v;
            }
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // ---- key path, value residual (lane = channel) ----
    const float av = l_ch[1][lane], wv = l_ch[0][lane], gv = l_ch[2][lane];
    const float kkr = kv0 * c_kk;
    const float ssum = wave_sum_f(kkr * kkr);
    const float kscale = 1.0f / fmaxf(sqrtf(ssum), 1e-12f);
    const float kk = kkr * kscale;
    const float ka = kv0 * c_ka;
    const float aka = av * ka;
    const float kn = kv0 + (aka - ka);
    if (p.layer0) p.v_first[c] = vv;
    else { const float dv = (vf - vv) * l_ch[3][lane]; vv = vv + dv; }
    l_r[lane] = rv; l_w[lane] = wv; l_k[lane] = kn; l_a[lane] = -kk; l_b[lane] = kk * av;
    __builtin_amdgcn_wave_barrier();
    // ---- WKV7 (rwkv_operators_wkv_v7.inc:37-107): lane i = value row i of state[h][i][:] ----
    float sa = 0.0f;
#pragma unroll
    for (int j = 0; j < S; j++) sa += l_a[j] * s[j];
    float res = 0.0f;
#pragma unroll
    for (int j = 0; j < S; j++) {
        const float kvj = vv * l_k[j];
        const float ns = (s[j] * l_w[j] + kvj) + sa * l_b[j];
        s[j] = ns;
        res += ns * l_r[j];
    }
#pragma unroll
    for (int j = 0; j < S; j += 4)
        *reinterpret_cast<float4 *>(p.state_out + h * S * S + (int64_t) lane * S + j) = make_float4(s[j], s[j + 1], s[j + 2], s[j + 3]);
    // ---- GroupNorm over the head * ln_x, + v * sum_head(k r r_k), gate (rwkv_graph.inc:465-479) ----
    const float mean = (float) (wave_sum_d((double) res) / (double) S);
    const float dv2 = res - mean;
    const float var = (float) (wave_sum_d((double) (dv2 * dv2)) / (double) S);
    const float scale = 1.0f / sqrtf(var + 64e-5f);
    const float bonus = wave_sum_f((kn * rv) * c_rk);
    float y = dv2 * scale;
    y = y * c_lw;
    y = y + c_lb;
    y += vv * bonus;
    y *= gv;
    int qi, isum; float d16, s16;
    quant_block32(y, qi, d16, s16, isum);
    qvec_store(qvec_at(p.y_out, D), (int) (D / 32), (int) (2 * h + (lane >> 5)), lane & 31, qi, d16, s16, isum);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

bool fused_v7_supported(const Model & m) {
    if (m.arch_major != 7 || m.head_size != 64) return false;
    const int64_t D = m.n_embed();
    if (D % 256 != 0) return false;
    const int fmt = (int) m.header.data_type;
    if (!dtype_quantized(fmt)) return false;
    int lrt = -1;
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        const DevTensor * mats[] = {L.att_receptance, L.att_key, L.att_value, L.att_output, L.ffn_key, L.ffn_value};
        for (const DevTensor * t : mats) if (!t || t->type != fmt) return false;
        const DevTensor * lrs[] = {L.att_w1, L.att_w2, L.att_a1, L.att_a2, L.att_g1, L.att_g2, L.att_v1, L.att_v2};
        for (int j = 0; j < 8; j++) {
            const DevTensor * t = lrs[j];
            if (!t) { if (j >= 6 && i == 0) continue; return false; }   // v1 / v2 do not exist on layer 0
            if (t->type != T_F16 && t->type != T_F32) return false;
            if (lrt < 0) lrt = t->type;
            if (t->type != lrt) return false;
            const int64_t rank = (j & 1) ? t->ne[0] : t->ne[1];
            if (rank % 32 != 0 || rank > 512) return false;
        }
        if (L.ffn_key->ne[1] % 32 != 0) return false;
    }
    return lrt >= 0;
}

size_t fused_v7_scratch_bytes(const Model & m) {
    const size_t D = (size_t) m.n_embed(), F = (size_t) m.ffn_size;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    return 4 * up(D * 4) + 4 * up(512 * 4) + up(qvec_bytes(D)) + up(qvec_bytes(F)) + 4096;
}

template <typename Kern, typename Param>
static void launch7(rwkv_context::Prof * pf, uint64_t bytes, Kern kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const Param & prm) {
    if (pf && pf->on && bytes) {
        if (pf->used * 2 + 2 > pf->events.size()) {
            hipEvent_t a = nullptr, c = nullptr;
            (void) hipEventCreate(&a); (void) hipEventCreate(&c);
            pf->events.push_back(a); pf->events.push_back(c); pf->bytes.push_back(0);
        }
        pf->bytes[pf->used] = bytes;
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t) shmem, st, pf->events[pf->used * 2], pf->events[pf->used * 2 + 1], 0, prm);
        pf->used++;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, shmem, st, prm);
    }
}

// the shared projection / channel-mixing kernels live in fused_v6.hip
void fused_proj_res(int fmt, const DevTensor * W, const void * act, float * x, const float * rgate, int64_t N, int64_t K, bool long_rows, hipStream_t st, rwkv_context::Prof * pf);
void fused_ffn_kr(int fmt, const float * x, const float * ln_w, const float * ln_b, const float * xx_in, float * xx_out, const float * maa_k, const float * maa_r, int mix_mode,
                  const DevTensor * wk, const DevTensor * wr, void * k_out, float * r_out, int64_t D, int64_t F, hipStream_t st, rwkv_context::Prof * pf);

template <int FMT, bool LRF16>
static void fused_v7_layer_t(const Model & m, const LayerW & L, int layer, float * x, float * v_first, const float * sin, float * sout, void * scratch,
                             hipStream_t st, rwkv_context::Prof * pf) {
    const int64_t D = m.n_embed(), F = L.ffn_key->ne[1], H = m.head_count;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    unsigned char * p = (unsigned char *) scratch;
    auto takef = [&](size_t n) { float * r = (float *) p; p += up(n * 4); return r; };
    float * r = takef(D), * k = takef(D), * v = takef(D); takef(D);
    float * lr[4] = {takef(512), takef(512), takef(512), takef(512)};
    void * yq = p; p += up(qvec_bytes(D));
    void * kq = p; p += up(qvec_bytes(F));
    auto f = [](const DevTensor * t) { return t ? (const float *) t->data : nullptr; };
    const size_t qbD = ((qvec_bytes(D) + 15) / 16) * 16;

    P7A a{};
    a.x = x; a.ln_w = f(L.ln1_w); a.ln_b = f(L.ln1_b); a.att_xx_in = sin + D; a.att_xx_out = sout + D;
    {   // x_rwkvag rows: r, w, k, v, a, g
        const float * xm = f(L.att_x_rwkvag);
        a.coef_q[0] = xm; a.coef_q[1] = xm + 2 * D; a.coef_q[2] = xm + 3 * D;
        a.coef_lr[0] = xm + D; a.coef_lr[1] = xm + 4 * D; a.coef_lr[2] = xm + 5 * D; a.coef_lr[3] = xm + 3 * D;
        a.mix_mode = 1; a.epi_q[0] = a.epi_q[1] = a.epi_q[2] = 0;
    }
    a.wq[0] = planes(L.att_receptance); a.wq[1] = planes(L.att_key); a.wq[2] = planes(L.att_value);
    const DevTensor * l1[4] = {L.att_w1, L.att_a1, L.att_g1, L.att_v1}, * l2[4] = {L.att_w2, L.att_a2, L.att_g2, L.att_v2};
    int64_t lr_groups = 0;
    uint64_t lr_bytes = 0;
    for (int i = 0; i < 4; i++) {
        a.lr[i] = l1[i] ? l1[i]->data : nullptr; a.rank[i] = l1[i] ? (int) l1[i]->ne[1] : 0; a.out_lr[i] = lr[i];
        lr_groups += l1[i] ? (a.rank[i] + 31) / 32 : 0;
        lr_bytes += l1[i] ? l1[i]->nbytes : 0;
    }
    a.out_q[0] = r; a.out_q[1] = k; a.out_q[2] = v; a.D = D;
    a.lr_groups = (int) lr_groups;
    launch7(pf, L.att_receptance->nbytes + L.att_key->nbytes + L.att_value->nbytes + lr_bytes + 7 * D * 4, k7_att_in<FMT, LRF16>,
            dim3((unsigned) (3 * (D / 32) + lr_groups)), dim3(256), (size_t) D * 4 + qbD + 512 * 8, st, a);

    P7B b{};
    for (int i = 0; i < 4; i++) { b.lr_in[i] = lr[i]; b.lr2[i] = l2[i] ? l2[i]->data : nullptr; b.rank[i] = l2[i] ? (int) l2[i]->ne[0] : 0; }
    b.w0 = f(L.att_w0); b.a0 = f(L.att_a0); b.v0 = f(L.att_v0);
    b.r = r; b.k = k; b.v = v; b.k_k = f(L.att_k_k); b.k_a = f(L.att_k_a); b.r_k = f(L.att_r_k);
    b.lnx_w = f(L.att_ln_x_w); b.lnx_b = f(L.att_ln_x_b);
    b.v_first = v_first; b.layer0 = layer == 0 ? 1 : 0;
    b.state_in = sin + 2 * D; b.state_out = sout + 2 * D; b.y_out = yq; b.D = D;
    launch7(pf, 0, k7_head<LRF16>, dim3((unsigned) H), dim3(512), 0, st, b);

    fused_proj_res(FMT, L.att_output, yq, x, nullptr, D, D, false, st, pf);
    fused_ffn_kr(FMT, x, f(L.ln2_w), f(L.ln2_b), sin, sout, f(L.ffn_x_k), f(L.ffn_x_k), 1, L.ffn_key, nullptr, kq, nullptr, D, F, st, pf);
    fused_proj_res(FMT, L.ffn_value, kq, x, nullptr, D, F, true, st, pf);
}

void fused_v7_layer(const Model & m, const LayerW & L, int layer, float * x, float * v_first, const float * sin, float * sout, void * scratch,
                    hipStream_t st, rwkv_context::Prof * pf) {
    const bool f16 = L.att_w1->type == T_F16;
#define V7_CASE(FMT) if (f16) fused_v7_layer_t<FMT, true>(m, L, layer, x, v_first, sin, sout, scratch, st, pf); \
                     else fused_v7_layer_t<FMT, false>(m, L, layer, x, v_first, sin, sout, scratch, st, pf); break
    switch ((int) m.header.data_type) {
        case T_Q4_0: V7_CASE(T_Q4_0);
        case T_Q4_1: V7_CASE(T_Q4_1);
        case T_Q5_0: V7_CASE(T_Q5_0);
        case T_Q5_1: V7_CASE(T_Q5_1);
        case T_Q8_0: V7_CASE(T_Q8_0);
        default: break;
    }
#undef V7_CASE
}


// ---------------------------------------------------------------------------------------------------------------
// RWKV-4 (rwkv_att_v4, rwkv_graph.inc:84-197; rwkv_ffn_v4_v5, :484-511) in FOUR launches per layer:
//   k7_att_in (mode 0 lerps, no low-rank groups: K, V, sigmoid(R) rows) -> k4_wkv_out (the log-space WKV of every channel, r * wkv,
//   quantise -> output rows + residual) -> k6_ffn_kr (mode 0 lerps, key + receptance rows) -> k6_proj_res (value rows, sigmoid gate).
// ---------------------------------------------------------------------------------------------------------------

struct P4C {
    const float * k; const float * v; const float * r; const float * tf; const float * td;
    const float * aa_in; const float * bb_in; const float * pp_in; float * aa_out; float * bb_out; float * pp_out;
    WPl w; float * x; int64_t D;
};

template <int FMT>
__global__ __launch_bounds__(256) void k4_wkv_out(P4C p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t D = p.D;
    const int nb = (int) (D / 32);
    const QVec lq = qvec_at(smem, D);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t) blockIdx.x * 4 + wave) * 4;
    Batch<FMT, 4, 2> bt;
    batch_issue<FMT, 4, 2>(bt, p.w.qs, p.w.qh, p.w.sc, row0 < D ? row0 : D - 1, D, nb, 0, lane);
    // every workgroup runs the (cheap, elementwise) recurrence of all D channels: it needs the whole r * wkv vector (k_wkv4's statements)
    for (int64_t i = threadIdx.x; i < D; i += 256) {   // D % 256 == 0
        const float aa = p.aa_in[i], bb = p.bb_in[i], pp = p.pp_in[i], u = p.tf[i], w = p.td[i];
        const float kk = p.k[i], vv = p.v[i];
        float ww = u + kk;
        float qq = fmaxf(pp, ww);
        float e1 = det_expf(pp - qq), e2 = det_expf(ww - qq);
        const float a = e1 * aa + e2 * vv;
        const float b = e1 * bb + e2;
        ww = pp + w;
        qq = fmaxf(ww, kk);
        e1 = det_expf(ww - qq); e2 = det_expf(kk - qq);
        if (blockIdx.x == 0) { p.aa_out[i] = e1 * aa + e2 * vv; p.bb_out[i] = e1 * bb + e2; p.pp_out[i] = qq; }
        const float y = p.r[i] * (a / b);
        int qi, isum; float d16, s16;
        quant_block32(y, qi, d16, s16, isum);
        qvec_store(lq, nb, (int) (i >> 5), (int) (i & 31), qi, d16, s16, isum);
    }
    __syncthreads();
    if (row0 >= D) return;
    float res[4];
    rows_finish<FMT, 4, 2>(bt, p.w.qs, p.w.qh, p.w.sc, row0, D, nb, lq, lane, res);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) if (row0 + r < D) p.x[row0 + r] = p.x[row0 + r] + res[r];
    }
}

bool fused_v4_supported(const Model & m) {
    if (m.arch_major != 4) return false;
    const int64_t D = m.n_embed();
    if (D % 256 != 0) return false;
    const int fmt = (int) m.header.data_type;
    if (!dtype_quantized(fmt)) return false;
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        const DevTensor * mats[] = {L.att_receptance, L.att_key, L.att_value, L.att_output, L.ffn_key, L.ffn_value, L.ffn_receptance};
        for (const DevTensor * t : mats) if (!t || t->type != fmt) return false;
        if (L.ffn_key->ne[1] % 32 != 0) return false;
    }
    return true;
}

size_t fused_v4_scratch_bytes(const Model & m) {
    const size_t D = (size_t) m.n_embed(), F = (size_t) m.ffn_size;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    return 4 * up(D * 4) + up(qvec_bytes(F)) + 4096;
}

template <int FMT>
static void fused_v4_layer_t(const Model & m, const LayerW & L, float * x, const float * sin, float * sout, void * scratch, hipStream_t st, rwkv_context::Prof * pf) {
    const int64_t D = m.n_embed(), F = L.ffn_key->ne[1];
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    unsigned char * p = (unsigned char *) scratch;
    auto takef = [&](size_t n) { float * r = (float *) p; p += up(n * 4); return r; };
    float * r = takef(D), * k = takef(D), * v = takef(D), * rr = takef(D);
    void * kq = p; p += up(qvec_bytes(F));
    auto f = [](const DevTensor * t) { return t ? (const float *) t->data : nullptr; };
    const size_t qbD = ((qvec_bytes(D) + 15) / 16) * 16;

    P7A a{};
    a.x = x; a.ln_w = f(L.ln1_w); a.ln_b = f(L.ln1_b); a.att_xx_in = sin + D; a.att_xx_out = sout + D;
    a.wq[0] = planes(L.att_receptance); a.wq[1] = planes(L.att_key); a.wq[2] = planes(L.att_value);
    a.coef_q[0] = f(L.att_time_mix_r); a.coef_q[1] = f(L.att_time_mix_k); a.coef_q[2] = f(L.att_time_mix_v);
    a.mix_mode = 0; a.epi_q[0] = 1; a.epi_q[1] = a.epi_q[2] = 0;
    a.out_q[0] = r; a.out_q[1] = k; a.out_q[2] = v; a.D = D; a.lr_groups = 0;
    launch7(pf, L.att_receptance->nbytes + L.att_key->nbytes + L.att_value->nbytes + 7 * D * 4, k7_att_in<FMT, true>,
            dim3((unsigned) (3 * (D / 32))), dim3(256), (size_t) D * 4 + qbD + 512 * 8, st, a);

    P4C c{k, v, r, f(L.att_time_first), f(L.att_time_decay), sin + 2 * D, sin + 3 * D, sin + 4 * D, sout + 2 * D, sout + 3 * D, sout + 4 * D,
          planes(L.att_output), x, D};
    launch7(pf, L.att_output->nbytes + 12 * D * 4, k4_wkv_out<FMT>, dim3((unsigned) ((D + 15) / 16)), dim3(256), qbD, st, c);

    fused_ffn_kr(FMT, x, f(L.ln2_w), f(L.ln2_b), sin, sout, f(L.ffn_time_mix_k), f(L.ffn_time_mix_r), 0, L.ffn_key, L.ffn_receptance, kq, rr, D, F, st, pf);
    fused_proj_res(FMT, L.ffn_value, kq, x, rr, D, F, true, st, pf);
}

void fused_v4_layer(const Model & m, const LayerW & L, float * x, const float * sin, float * sout, void * scratch, hipStream_t st, rwkv_context::Prof * pf) {
    switch ((int) m.header.data_type) {
        case T_Q4_0: fused_v4_layer_t<T_Q4_0>(m, L, x, sin, sout, scratch, st, pf); break;
        case T_Q4_1: fused_v4_layer_t<T_Q4_1>(m, L, x, sin, sout, scratch, st, pf); break;
        case T_Q5_0: fused_v4_layer_t<T_Q5_0>(m, L, x, sin, sout, scratch, st, pf); break;
        case T_Q5_1: fused_v4_layer_t<T_Q5_1>(m, L, x, sin, sout, scratch, st, pf); break;
        case T_Q8_0: fused_v4_layer_t<T_Q8_0>(m, L, x, sin, sout, scratch, st, pf); break;
        default: break;
    }
}

}  // namespace rwkvmi
