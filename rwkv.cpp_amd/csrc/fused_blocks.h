// fused_blocks.h -- device building blocks shared by the fused single-token paths (fused_v6.hip, mega_v6.hip):
// the "lohi" image of a quantised activation vector, the activation quantiser over half-waves, batched weight-row loads
// (issue / consume split) and the row finish. Arithmetic and orders follow DESIGN.md section 4.
#pragma once

#include "kdev.h"
#include "model.h"

namespace rwkvmi {

// ---------------------------------------------------------------------------------------------------------------
// building blocks
// ---------------------------------------------------------------------------------------------------------------

struct QVec {      // one quantised activation vector in the lohi image (global or LDS)
    int8_t * q;    // 32 * nb bytes
    float * d;
    float * s;
    int * isum;
};

__host__ __device__ inline size_t qvec_bytes(int64_t K) { return (size_t) K + (size_t)(K / 32) * 12; }
__host__ __device__ inline QVec qvec_at(void * base, int64_t K) {
    QVec v;
    v.q = (int8_t *) base;
    v.d = (float *) ((uint8_t *) base + K);
    v.s = v.d + K / 32;
    v.isum = (int *) (v.s + K / 32);
    return v;
}

// Quantise the 32-element block whose elements sit in the 32 lanes of a half-wave (ggml quantize_row_q8_0 / q8_1).
__device__ __forceinline__ void quant_block32(float v, int & qi, float & d16, float & s16, int & isum) {
    const float amax = half_max_f(fabsf(v));
    const float dd = amax / 127.0f;
    const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
    qi = (int) roundf(v * id);
    const int sum = half_sum_i(qi);
    d16 = round_f16(dd);
    s16 = round_f16((float) sum * dd);
    isum = sum;
}

// NQ independent blocks at once (one per array slot, each across the half-wave): same arithmetic per block, written stage by
// stage so that the NQ dependent chains (DPP reductions, two f32 divisions, roundf) interleave instead of running back to back.
template <int NQ>
__device__ __forceinline__ void quant_blocks(const float (&v)[NQ], int (&qi)[NQ], float (&d16)[NQ], float (&s16)[NQ], int (&isum)[NQ]) {
    float am[NQ], dd[NQ], id[NQ];
#pragma unroll
    for (int u = 0; u < NQ; u++) am[u] = fabsf(v[u]);
#pragma unroll
    for (int u = 0; u < NQ; u++) am[u] = fmaxf(am[u], __int_as_float(lane_xor1_i(__float_as_int(am[u]))));
#pragma unroll
    for (int u = 0; u < NQ; u++) am[u] = fmaxf(am[u], __int_as_float(lane_xor2_i(__float_as_int(am[u]))));
#pragma unroll
    for (int u = 0; u < NQ; u++) am[u] = fmaxf(am[u], __int_as_float(lane_xor4_i(__float_as_int(am[u]))));
#pragma unroll
    for (int u = 0; u < NQ; u++) am[u] = fmaxf(am[u], __int_as_float(lane_xor8_i(__float_as_int(am[u]))));
#pragma unroll
    for (int u = 0; u < NQ; u++) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(am[u]), __float_as_uint(am[u]), false, false);
        am[u] = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
#pragma unroll
    for (int u = 0; u < NQ; u++) dd[u] = am[u] / 127.0f;
#pragma unroll
    for (int u = 0; u < NQ; u++) id[u] = dd[u] != 0.0f ? 1.0f / dd[u] : 0.0f;
#pragma unroll
    for (int u = 0; u < NQ; u++) qi[u] = (int) roundf(v[u] * id[u]);
#pragma unroll
    for (int u = 0; u < NQ; u++) isum[u] = qi[u] + lane_xor1_i(qi[u]);
#pragma unroll
    for (int u = 0; u < NQ; u++) isum[u] += lane_xor2_i(isum[u]);
#pragma unroll
    for (int u = 0; u < NQ; u++) isum[u] += lane_xor4_i(isum[u]);
#pragma unroll
    for (int u = 0; u < NQ; u++) isum[u] += lane_xor8_i(isum[u]);
#pragma unroll
    for (int u = 0; u < NQ; u++) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned) isum[u], (unsigned) isum[u], false, false);
        isum[u] = (int) r[0] + (int) r[1];
    }
#pragma unroll
    for (int u = 0; u < NQ; u++) { d16[u] = round_f16(dd[u]); s16[u] = round_f16((float) isum[u] * dd[u]); }
}

// store one quantised element (block blk, element e) + the block scalars into a lohi image
__device__ __forceinline__ void qvec_store(const QVec & v, int nb, int blk, int e, int qi, float d16, float s16, int isum) {
    v.q[(e < 16 ? 0 : nb * 16) + blk * 16 + (e & 15)] = (int8_t) qi;
    if (e == 0) { v.d[blk] = d16; v.s[blk] = s16; v.isum[blk] = isum; }
}

// copy a lohi image global -> LDS with 16-byte accesses (K multiple of 32; the scalar tail is 12 * nb bytes)
__device__ __forceinline__ void qvec_stage(const void * __restrict__ g, void * l, int64_t K) {
    const int n16 = (int) (qvec_bytes(K) / 16);  // K + 12*K/32 = K*11/8 bytes; a sub-16-byte tail remains when K % 128 != 0
    const int4 * src = reinterpret_cast<const int4 *>(g);
    int4 * dst = reinterpret_cast<int4 *>(l);
    for (int i0 = threadIdx.x; i0 < n16; i0 += 4 * (int) blockDim.x) {
        int4 t[4];
        int idx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + u * (int) blockDim.x; idx[u] = i < n16 ? i : n16 - 1; t[u] = src[idx[u]]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) dst[idx[u]] = t[u];  // clamped duplicates rewrite the last chunk with the same bytes
    }
    const int tail0 = n16 * 4, words = (int) (qvec_bytes(K) / 4);
    for (int i = tail0 + threadIdx.x; i < words; i += blockDim.x) reinterpret_cast<int *>(l)[i] = reinterpret_cast<const int *>(g)[i];
}

// R consecutive rows of a quantised matrix against an activation image in LDS, U block-steps (64 blocks each) per batch.
// batch_issue puts every load of the batch in flight (codes, fifth bits, scales of R x U blocks per lane);
// batch_consume decodes and accumulates them in increasing block order (the specified order). Kernels issue the first
// batch BEFORE their prologue (LayerNorm / activation staging), so the weight stream overlaps it.
template <int FMT, int R, int U>
struct Batch { RawBlk<FMT> raw[U][R]; };

template <int FMT, int R, int U>
__device__ __forceinline__ void batch_issue(Batch<FMT, R, U> & bt, const uint8_t * __restrict__ qs, const uint32_t * __restrict__ qh,
                                            const void * __restrict__ sc, int64_t row0, int64_t N, int nb, int bbase, int lane) {
    // Row bases are formed first (wave-uniform when row0 is: scalar base + 32-bit lane offset addressing, no 64-bit
    // address pair per load), then every load of the batch is issued.
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int bb = bbase + u * WAVE + lane;
        unsigned b = (unsigned) (bb < nb ? bb : nb - 1);
        // (opaque copy: instruction selection works per basic block and only forms "scalar base + 32-bit lane offset"
        //  addresses when the zero-extension of the offset sits in the block of the load)
        asm volatile("" : "+v"(b));
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int64_t row = (row0 + r < N) ? row0 + r : N - 1;
            const uint8_t * rq = qs + row * nb * QF<FMT>::QS;
            RawBlk<FMT> & o = bt.raw[u][r];
            if constexpr (QF<FMT>::HM) o.sc = ldw4(reinterpret_cast<const uint32_t *>(sc) + row * nb + b);
            else o.sc = ldw2(reinterpret_cast<const uint16_t *>(sc) + row * nb + b);
            if constexpr (QF<FMT>::QH) o.qh = ldw4(qh + row * nb + b);
            o.q[0] = ldw16(rq + b * QF<FMT>::QS);
            if constexpr (QF<FMT>::QS == 32) o.q[1] = ldw16(rq + b * QF<FMT>::QS + 16);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int FMT, int R, int U>
__device__ __forceinline__ void batch_consume(const Batch<FMT, R, U> & bt, int nb, int bbase, int lane, const QVec & a, float (&acc)[R]) {
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int bb = bbase + u * WAVE + lane;
        const bool valid = bb < nb;
        const int b = valid ? bb : nb - 1;
        const int4 alo = *reinterpret_cast<const int4 *>(a.q + b * 16);
        const int4 ahi = *reinterpret_cast<const int4 *>(a.q + nb * 16 + b * 16);
        const float dx = a.d[b], sx = a.s[b];
        const int asum = a.isum[b];
#pragma unroll
        for (int r = 0; r < R; r++) {
            WBlk<FMT> w;
            unpack_raw<FMT>(w, bt.raw[u][r]);
            acc[r] = blk_fma<FMT>(w, alo, ahi, dx, sx, asum, acc[r], valid);
        }
    }
}

// remaining batches after the first one, then the butterfly; every lane returns the R row sums.
// DB = true keeps TWO batches in flight (the next batch is issued before the current one is consumed): for long rows
// (channel-mixing value projection, K = 14336) at the price of twice the staging registers.
template <int FMT, int R, int U, bool DB = false>
__device__ __forceinline__ void rows_finish(Batch<FMT, R, U> & bt, const uint8_t * __restrict__ qs, const uint32_t * __restrict__ qh,
                                            const void * __restrict__ sc, int64_t row0, int64_t N, int nb, const QVec & a, int lane, float (&res)[R]) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0f;
    if constexpr (!DB) {
        batch_consume<FMT, R, U>(bt, nb, 0, lane, a, acc);
        for (int bbase = U * WAVE; bbase < nb; bbase += U * WAVE) {  // further batches live in their own registers
            Batch<FMT, R, U> nx;
            batch_issue<FMT, R, U>(nx, qs, qh, sc, row0, N, nb, bbase, lane);
            batch_consume<FMT, R, U>(nx, nb, bbase, lane, a, acc);
        }
    } else {
        Batch<FMT, R, U> b2;
        for (int bbase = 0;; bbase += 2 * U * WAVE) {
            const bool more1 = bbase + U * WAVE < nb;
            if (more1) batch_issue<FMT, R, U>(b2, qs, qh, sc, row0, N, nb, bbase + U * WAVE, lane);
            batch_consume<FMT, R, U>(bt, nb, bbase, lane, a, acc);
            if (!more1) break;
            const bool more2 = bbase + 2 * U * WAVE < nb;
            if (more2) batch_issue<FMT, R, U>(bt, qs, qh, sc, row0, N, nb, bbase + 2 * U * WAVE, lane);
            batch_consume<FMT, R, U>(b2, nb, bbase + U * WAVE, lane, a, acc);
            if (!more2) break;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) res[r] = wave_sum_f(acc[r]);
}

// copy a row of D floats (D % 4 == 0) global -> LDS with batched 16-byte loads
__device__ __forceinline__ void fill_row(float * l_row, const float * __restrict__ x, int64_t D) {
    const int n4 = (int) (D / 4);
    for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * 256) {
        float4 t[4];
        int idx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + u * 256; idx[u] = i < n4 ? i : n4 - 1; t[u] = reinterpret_cast<const float4 *>(x)[idx[u]]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) reinterpret_cast<float4 *>(l_row)[idx[u]] = t[u];
    }
}

// LayerNorm of one row by a 256-thread workgroup: the row is in l_row (D floats, overwritten by x - mean);
// returns scale; afterwards xn_i = ((l_row[i] * scale) * w[i]) + b[i]   (same steps as block_layernorm in kernels.hip)
__device__ __forceinline__ float block_ln_stats(float * l_row, int64_t D, double * red) {
    double s = 0.0;
    s = ln_partial_sum(l_row, D);
    const float mean = (float)(block_sum_d(s, red) / (double) D);
    double s2 = 0.0;
    s2 = ln_partial_var(l_row, D, mean);
    const float var = (float)(block_sum_d(s2, red) / (double) D);
    return 1.0f / sqrtf(var + 1e-5f);
}

struct WPl {  // planes of one quantised matrix
    const uint8_t * qs;
    const uint32_t * qh;
    const void * sc;
};
static inline WPl planes(const DevTensor * t) { return WPl{t->qs, t->qh, t->sc}; }

}  // namespace rwkvmi
