// kdev.h -- device-side building blocks shared by the kernel files: wave/block reductions in the specified order, the
// deterministic scalar routines, quantised weight-block decode and the per-block fma. See kernels.hip for the contract.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "kernels.h"

namespace rwkvmi {

#define WAVE 64

// ---------------------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------------------------
// Cross-lane primitives. __shfl_xor lowers to ds_bpermute (LDS crossbar, ~100+ cycles per dependent step); the
// sequences below use DPP and the gfx950 permlane swaps instead (a few cycles per step) and implement EXACTLY the
// xor-butterfly pairing (lane l with lane l ^ o, o = 32, 16, 8, 4, 2, 1), i.e. the halving tree of the numerics spec.
// ---------------------------------------------------------------------------------------------------------------

template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// value of lane (l ^ 4): row_shl:4 feeds banks 0 and 2 (lanes reading l + 4), row_shr:4 banks 1 and 3 (lanes reading l - 4)
__device__ __forceinline__ int lane_xor4_i(int v) {
    int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);
}
__device__ __forceinline__ int lane_xor1_i(int v) { return dpp_i<0xB1>(v); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ int lane_xor2_i(int v) { return dpp_i<0x4E>(v); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ int lane_xor8_i(int v) { return dpp_i<0x128>(v); }  // row_ror:8

__device__ __forceinline__ float wave_sum_f(float v) {
    { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    v = v + __int_as_float(lane_xor8_i(__float_as_int(v)));
    v = v + __int_as_float(lane_xor4_i(__float_as_int(v)));
    v = v + __int_as_float(lane_xor2_i(__float_as_int(v)));
    v = v + __int_as_float(lane_xor1_i(__float_as_int(v)));
    return v;
}

__device__ __forceinline__ double mk_double(unsigned lo, unsigned hi) { return __hiloint2double((int) hi, (int) lo); }
__device__ __forceinline__ double wave_sum_d(double v) {
    {
        const unsigned lo = (unsigned) __double2loint(v), hi = (unsigned) __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = mk_double(a[0], b[0]) + mk_double(a[1], b[1]);
    }
    {
        const unsigned lo = (unsigned) __double2loint(v), hi = (unsigned) __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = mk_double(a[0], b[0]) + mk_double(a[1], b[1]);
    }
    v = v + mk_double((unsigned) lane_xor8_i(__double2loint(v)), (unsigned) lane_xor8_i(__double2hiint(v)));
    v = v + mk_double((unsigned) lane_xor4_i(__double2loint(v)), (unsigned) lane_xor4_i(__double2hiint(v)));
    v = v + mk_double((unsigned) lane_xor2_i(__double2loint(v)), (unsigned) lane_xor2_i(__double2hiint(v)));
    v = v + mk_double((unsigned) lane_xor1_i(__double2loint(v)), (unsigned) lane_xor1_i(__double2hiint(v)));
    return v;
}

// reference forms (ds_bpermute), kept for the self-test of the fast sequences
__device__ __forceinline__ float wave_sum_f_ref(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ double wave_sum_d_ref(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// order-free reductions over the 32 lanes of a half-wave (max and integer sum are exact in any order)
__device__ __forceinline__ float half_max_f(float v) {
    v = fmaxf(v, __int_as_float(lane_xor1_i(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(lane_xor2_i(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(lane_xor4_i(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(lane_xor8_i(__float_as_int(v))));
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ int half_sum_i(int v) {
    v += lane_xor1_i(v);
    v += lane_xor2_i(v);
    v += lane_xor4_i(v);
    v += lane_xor8_i(v);
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned) v, (unsigned) v, false, false);
    return (int) r[0] + (int) r[1];
}

// exp in double (ln2 hi/lo reduction, degree-13 Taylor, fma Horner), rounded once to float; same routine as the oracle's.
// A double constant held in a scalar register pair, re-materialised (two s_mov) where it is used. Left to itself the
// compiler hoists the sixteen constants of det_exp_d out of a kernel's main loop into vector registers and, under pressure,
// spills them: a scratch reload in the middle of a dependent exp chain is a memory round trip.
__device__ __forceinline__ double kconst(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int) (b & 0xFFFFFFFFll), hi = (int) (b >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double det_exp_d(double x) {
    const double n = rint(x * kconst(1.4426950408889634074));
    double r = fma(n, kconst(-6.93147180369123816490e-01), x);
    r = fma(n, kconst(-1.90821492927058770002e-10), r);
    double p = kconst(1.6059043836821613e-10);
    p = fma(p, r, kconst(2.08767569878681e-09));
    p = fma(p, r, kconst(2.505210838544172e-08));
    p = fma(p, r, kconst(2.755731922398589e-07));
    p = fma(p, r, kconst(2.7557319223985893e-06));
    p = fma(p, r, kconst(2.48015873015873e-05));
    p = fma(p, r, kconst(1.984126984126984e-04));
    p = fma(p, r, kconst(1.388888888888889e-03));
    p = fma(p, r, kconst(8.333333333333333e-03));
    p = fma(p, r, kconst(4.1666666666666664e-02));
    p = fma(p, r, kconst(1.6666666666666666e-01));
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int) n);
}
__device__ __forceinline__ float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -103.97208404541016f) return 0.0f;
    return (float) det_exp_d((double) x);
}
__device__ __forceinline__ float det_tanhf(float x) {
    if (x != x) return x;
    const double xd = (double) x;
    const double ax = fabs(xd);
    if (ax < 1e-4) return (float) (xd * fma(xd * xd, -1.0 / 3.0, 1.0));
    if (ax > 20.0) return x > 0.0f ? 1.0f : -1.0f;
    const double sv = det_exp_d(ax + ax);
    const double t = 1.0 - 2.0 / (sv + 1.0);
    return (float) (x > 0.0f ? t : -t);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + det_expf(-x)); }
__device__ __forceinline__ float h2f_bits(uint16_t h) { return __half2float(__ushort_as_half(h)); }
// f32 -> f16 -> f32 with the f32 value materialised first: without the barrier LLVM folds a preceding f32 multiply into
// v_fma_mixlo_f16 (ONE rounding to f16), whereas the reference rounds the f32 product and then converts (two roundings).
__device__ __forceinline__ float round_f16(float x) {
    asm volatile("" : "+v"(x));
    return __half2float(__float2half_rn(x));
}

// LayerNorm partials of thread t < 256: elements t, t + 256, ... of a row in LDS, accumulated in that order (DESIGN.md
// section 4). The LDS reads go out eight at a time: as a plain loop every iteration waits for its own read.
__device__ __forceinline__ double ln_partial_sum(const float * l_row, int64_t D) {
    double s = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 7 * 256 < D; i += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = l_row[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; u++) s += (double) v[u];
    }
    for (; i < D; i += 256) s += (double) l_row[i];
    return s;
}
// second pass: replaces x by x - mean and returns the partial of (x - mean)^2
__device__ __forceinline__ double ln_partial_var(float * l_row, int64_t D, float mean) {
    double s2 = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 7 * 256 < D; i += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = l_row[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; u++) { const float d = v[u] - mean; l_row[i + u * 256] = d; s2 += (double) (d * d); }
    }
    for (; i < D; i += 256) { const float d = l_row[i] - mean; l_row[i] = d; s2 += (double) (d * d); }
    return s2;
}

// Sum of one double per thread over a 256-thread workgroup, as a halving tree over the 256 partials
// (p[i] += p[i+128]; p[i] += p[i+64]; then the 64-entry butterfly): the order the oracle's fold_d(.., 256) uses.
__device__ __forceinline__ double block_sum_d(double v, double * red /* [257] */) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        double t = (red[i] + red[i + 128]) + (red[i + 64] + red[i + 192]);
        t = wave_sum_d(t);
        if (i == 0) red[256] = t;
    }
    __syncthreads();
    return red[256];
}

// Same tree for a 512-thread workgroup in which only threads 0..255 carry a partial (the others pass anything and just
// keep the barriers): identical result to block_sum_d.
__device__ __forceinline__ double block_sum_d_8w(double v, double * red /* [257] */) {
    __syncthreads();
    if (threadIdx.x < 256) red[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        double t = (red[i] + red[i + 128]) + (red[i + 64] + red[i + 192]);
        t = wave_sum_d(t);
        if (i == 0) red[256] = t;
    }
    __syncthreads();
    return red[256];
}

// The same tree with ONE workgroup barrier: the caller passes a scratch array of 256 doubles that no wave can still be
// reading (a different array per reduction of a kernel), and every wave folds the 256 partials itself -- identical additions
// in identical order in every wave -- instead of waiting for wave 0 to publish the total. Any workgroup size >= 256.
__device__ __forceinline__ double block_sum_d_1b(double v, double * red /* [256], fresh */) {
    if (threadIdx.x < 256) red[threadIdx.x] = v;
    __syncthreads();
    const int i = threadIdx.x & 63;
    const double t = (red[i] + red[i + 128]) + (red[i + 64] + red[i + 192]);
    return wave_sum_d(t);
}

__device__ __forceinline__ float apply_epi(const Epi & e, float acc, int64_t t, int64_t n, int64_t ldy) {
    switch (e.op) {
        case EPI_NONE: return acc;
        case EPI_SIGMOID: return sigmoid_f(acc);
        case EPI_RELU_SQ: { const float r = acc > 0.0f ? acc : 0.0f; return r * r; }
        case EPI_SILU: return acc / (1.0f + det_expf(-acc));
        case EPI_TANH: return det_tanhf(acc);
        case EPI_ADD_RES: return e.res[t * ldy + n] + acc;
        case EPI_SIGMUL_ADD_RES: { const float g = sigmoid_f(e.aux[t * ldy + n]) * acc; return e.res[t * ldy + n] + g; }
        case EPI_BIAS_SIGMOID: return sigmoid_f(acc + e.bias[n]);
        case EPI_V6_DECAY: return det_expf(-det_expf(acc + e.bias[n]));
        case EPI_V7_DECAY: return det_expf(sigmoid_f(acc + e.bias[n]) * -0.606531f);
        default: return acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Quantised weight blocks (planes, see DevTensor). One block = 32 weights.
// ---------------------------------------------------------------------------------------------------------------

template <int FMT> struct QF;
template <> struct QF<T_Q4_0> { static constexpr int QS = 16; static constexpr bool QH = false, HM = false; static constexpr int OFF = 8; };
template <> struct QF<T_Q4_1> { static constexpr int QS = 16; static constexpr bool QH = false, HM = true;  static constexpr int OFF = 0; };
template <> struct QF<T_Q5_0> { static constexpr int QS = 16; static constexpr bool QH = true,  HM = false; static constexpr int OFF = 16; };
template <> struct QF<T_Q5_1> { static constexpr int QS = 16; static constexpr bool QH = true,  HM = true;  static constexpr int OFF = 0; };
template <> struct QF<T_Q8_0> { static constexpr int QS = 32; static constexpr bool QH = false, HM = false; static constexpr int OFF = 0; };

// Raw bytes of one block as they sit in the planes. Loading is split from decoding so that a kernel can issue the
// loads of ALL the blocks of a step (codes, fifth bits and scales) before the first use -- otherwise the compiler
// sinks the scale loads behind the dot products and every step pays two dependent memory round trips.
template <int FMT>
struct RawBlk {
    int4 q[QF<FMT>::QS / 16];   // 16 B of codes (32 B for Q8_0)
    unsigned qh;                // Q5 only (never touched otherwise)
    unsigned sc;                // fp16 d in the low half (+ fp16 m in the high half for Q4_1 / Q5_1)
};

// Weight-stream loads. A decode step reads every weight byte exactly once, from one CU: with the non-temporal policy (`nt`) the
// lines are not kept in L2 / MALL for a re-use that never comes (MI355X_MICROARCH.md, row nt-weights: issued -> landed -18 %,
// a decode layer -5...10 %). RWKV_NT_WEIGHTS=0 builds the default-policy variant for A/B runs.
#ifndef RWKV_NT_WEIGHTS
#define RWKV_NT_WEIGHTS 1
#endif
typedef int wv4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 ldw16(const void * p) {
#if RWKV_NT_WEIGHTS
    const wv4i v = __builtin_nontemporal_load(reinterpret_cast<const wv4i *>(p));
    return make_int4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const int4 *>(p);
#endif
}
__device__ __forceinline__ uint32_t ldw4(const uint32_t * p) {
#if RWKV_NT_WEIGHTS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ uint32_t ldw2(const uint16_t * p) {
#if RWKV_NT_WEIGHTS
    return (uint32_t) __builtin_nontemporal_load(p);
#else
    return (uint32_t) *p;
#endif
}

template <int FMT, bool ONCE = true>   // ONCE = false: weights that many workgroups re-read (token-tiled sequence kernels) keep the default policy
__device__ __forceinline__ void load_raw(RawBlk<FMT> & r, const uint8_t * __restrict__ qs, const uint32_t * __restrict__ qh,
                                         const void * __restrict__ sc, int64_t blk) {
    if constexpr (!ONCE) {
        if constexpr (QF<FMT>::HM) r.sc = reinterpret_cast<const uint32_t *>(sc)[blk];
        else r.sc = reinterpret_cast<const uint16_t *>(sc)[blk];
        if constexpr (QF<FMT>::QH) r.qh = qh[blk];
        r.q[0] = *reinterpret_cast<const int4 *>(qs + blk * QF<FMT>::QS);
        if constexpr (QF<FMT>::QS == 32) r.q[1] = *reinterpret_cast<const int4 *>(qs + blk * 32 + 16);
        return;
    }
    if constexpr (QF<FMT>::HM) r.sc = ldw4(reinterpret_cast<const uint32_t *>(sc) + blk);
    else r.sc = ldw2(reinterpret_cast<const uint16_t *>(sc) + blk);
    if constexpr (QF<FMT>::QH) r.qh = ldw4(qh + blk);
    if constexpr (QF<FMT>::QS == 32) {
        r.q[0] = ldw16(qs + blk * 32);
        r.q[1] = ldw16(qs + blk * 32 + 16);
    } else {
        r.q[0] = ldw16(qs + blk * 16);
    }
}

// Codes of one block as 8 dwords of 4 x int8: c[0..3] = elements 0..15, c[4..7] = elements 16..31.
// 4/5-bit codes are left unsigned (0..15 / 0..31); the -8 / -16 offset is applied through the activation sum.
template <int FMT>
struct WBlk {
    int c[8];
    float d, m;
};

template <int FMT>
__device__ __forceinline__ void unpack_raw(WBlk<FMT> & w, const RawBlk<FMT> & r) {
    if constexpr (QF<FMT>::QS == 32) {
        w.c[0] = r.q[0].x; w.c[1] = r.q[0].y; w.c[2] = r.q[0].z; w.c[3] = r.q[0].w;
        w.c[4] = r.q[1].x; w.c[5] = r.q[1].y; w.c[6] = r.q[1].z; w.c[7] = r.q[1].w;
    } else {
        const int raw[4] = {r.q[0].x, r.q[0].y, r.q[0].z, r.q[0].w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int lo = raw[i] & 0x0F0F0F0F;
            int hi = (raw[i] >> 4) & 0x0F0F0F0F;
            if constexpr (QF<FMT>::QH) {
                // bit j of qh -> bit 4 of byte j (elements 0..15), bit 16+j -> elements 16..31.
                const unsigned nl = (r.qh >> (4 * i)) & 0xFu, nh = (r.qh >> (16 + 4 * i)) & 0xFu;
                lo |= (int)(((nl * 0x00204081u) & 0x01010101u) << 4);
                hi |= (int)(((nh * 0x00204081u) & 0x01010101u) << 4);
            }
            w.c[i] = lo;
            w.c[4 + i] = hi;
        }
    }
    w.d = h2f_bits((uint16_t)(r.sc & 0xFFFFu));
    if constexpr (QF<FMT>::HM) w.m = h2f_bits((uint16_t)(r.sc >> 16)); else w.m = 0.0f;
}

template <int FMT, bool ONCE = true>
__device__ __forceinline__ void load_wblk(WBlk<FMT> & w, const uint8_t * __restrict__ qs, const uint32_t * __restrict__ qh,
                                          const void * __restrict__ sc, int64_t blk) {
    RawBlk<FMT> r;
    load_raw<FMT, ONCE>(r, qs, qh, sc, blk);
    unpack_raw<FMT>(w, r);
}

// acc <- acc + contribution of one weight block against one activation block. `valid` = false turns the step into
// acc + 0 (used by lanes whose block index is past the row end: loads are clamped, nothing is branched around).
template <int FMT>
__device__ __forceinline__ float blk_fma(const WBlk<FMT> & w, const int4 alo, const int4 ahi, float dx, float sx, int asum, float acc, bool valid = true) {
    int s = 0;
    s = __builtin_amdgcn_sdot4(w.c[0], alo.x, s, false);
    s = __builtin_amdgcn_sdot4(w.c[1], alo.y, s, false);
    s = __builtin_amdgcn_sdot4(w.c[2], alo.z, s, false);
    s = __builtin_amdgcn_sdot4(w.c[3], alo.w, s, false);
    s = __builtin_amdgcn_sdot4(w.c[4], ahi.x, s, false);
    s = __builtin_amdgcn_sdot4(w.c[5], ahi.y, s, false);
    s = __builtin_amdgcn_sdot4(w.c[6], ahi.z, s, false);
    s = __builtin_amdgcn_sdot4(w.c[7], ahi.w, s, false);
    if constexpr (QF<FMT>::OFF != 0) s -= QF<FMT>::OFF * asum;
    const float dd = w.d * dx;
    acc = fmaf(dd, valid ? (float) s : 0.0f, acc);
    if constexpr (QF<FMT>::HM) acc = fmaf(w.m, valid ? sx : 0.0f, acc);
    return acc;
}

}  // namespace rwkvmi
