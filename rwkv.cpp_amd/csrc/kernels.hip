// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the RWKV hot path.
//
// Numerics contract (DESIGN.md "Numerics"):
//  * Projections with quantised weights follow ggml's CPU mul_mat: activations are quantised per
//    32-element block to int8 (d = amax/127 stored fp16-rounded), the dot is an exact integer dot (v_dot4_i32_i8),
//    and blocks are accumulated in f32:  acc = fma(d_w*d_x, isum, acc) [+ fma(m_w, s_x, acc)].
//  * Lane l of the wave that owns an output row accumulates blocks l, l+64, l+128, ... in that order; the 64 partials
//    are folded by an xor-butterfly (32,16,8,4,2,1). The single-token kernel and the token-tiled (sequence) kernel use
//    the SAME order, so rwkv_eval_sequence is bit-identical to repeated rwkv_eval (reference test
//    tests/test_eval_sequence_in_chunks.c:54 checks this with memcmp).
//  * F16 weights: activations are rounded to fp16 first (what ggml does), products/accumulation in f32.
//  * Norm statistics are accumulated in double (ggml_norm), with a fixed reduction tree.
//  * Everything outside the dot products is ONE ROUNDING PER ggml GRAPH OP (the reference evaluates mul, add, sub ...
//    as separate graph nodes), so this file is compiled with -ffp-contract=off and fused multiply-adds appear only as
//    explicit fmaf() inside dot products. exp / tanh are the deterministic double-precision routines below. Together
//    with the fixed reduction orders this makes the GPU results bit-identical to the CPU oracle (oracle/rwkv_oracle.c),
//    which the tests assert with array_equal.
#include "kdev.h"

#include <atomic>

#include <mutex>
#include <vector>

namespace rwkvmi {

// ---------------------------------------------------------------------------------------------------------------
// Projection, quantised weights, single token (decode): the HBM-roofline kernel.
// One wave owns R consecutive rows; lane l streams blocks l, l+64, ... of each row straight into VGPRs (16 B/lane,
// 1 KiB per wave-instruction, no LDS round trip -- every weight byte is used exactly once).
// ---------------------------------------------------------------------------------------------------------------

template <int FMT, int R>
__global__ __launch_bounds__(256) void k_mvq_t1(const uint8_t * __restrict__ qs, const uint32_t * __restrict__ qh, const void * __restrict__ sc,
                                                int64_t N, int nb, const int8_t * __restrict__ xq, const float * __restrict__ xd,
                                                const float * __restrict__ xs, const int * __restrict__ xi, float * __restrict__ y, Epi epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t) blockIdx.x * 4 + wave) * R;
    if (row0 >= N) return;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0f;
    for (int b0 = lane; b0 < nb; b0 += 2 * WAVE) {
        RawBlk<FMT> raw[2][R];
        int4 alo[2], ahi[2];
        float dx[2], sx[2];
        int asum[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int b = b0 + u * WAVE < nb ? b0 + u * WAVE : nb - 1;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int64_t row = (row0 + r < N) ? row0 + r : N - 1;
                load_raw<FMT>(raw[u][r], qs, qh, sc, row * nb + b);
            }
            alo[u] = reinterpret_cast<const int4 *>(xq)[2 * b];
            ahi[u] = reinterpret_cast<const int4 *>(xq)[2 * b + 1];
            dx[u] = xd[b]; sx[u] = xs[b]; asum[u] = xi[b];
        }
        __builtin_amdgcn_sched_barrier(0);  // every load of the step is in flight before the first use
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const bool valid = b0 + u * WAVE < nb;
#pragma unroll
            for (int r = 0; r < R; r++) {
                WBlk<FMT> w;
                unpack_raw<FMT>(w, raw[u][r]);
                acc[r] = blk_fma<FMT>(w, alo[u], ahi[u], dx[u], sx[u], asum[u], acc[r], valid);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const float v = wave_sum_f(acc[r]);
        if (lane == 0 && row0 + r < N) y[row0 + r] = apply_epi(epi, v, 0, row0 + r, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Projection, quantised weights, token tile (sequence mode). Each weight block is unpacked once and used for TT tokens;
// the activation tile of the current K-step (64 blocks x TT tokens) is staged in LDS in a lane-linear, conflict-free
// image [tt][half][lane][16 B]. Per (row, token) the accumulation order equals k_mvq_t1's.
// ---------------------------------------------------------------------------------------------------------------

template <int FMT, int R, int TT>
__global__ __launch_bounds__(256) void k_mvq_tn(const uint8_t * __restrict__ qs, const uint32_t * __restrict__ qh, const void * __restrict__ sc,
                                                int64_t N, int nb, const int8_t * __restrict__ xq, const float * __restrict__ xd,
                                                const float * __restrict__ xs, const int * __restrict__ xi, int64_t T,
                                                float * __restrict__ y, int64_t ldy, Epi epi) {
    __shared__ __attribute__((aligned(16))) int8_t l_q[TT * 2048];
    __shared__ float l_d[TT * 64];
    __shared__ float l_s[TT * 64];
    __shared__ int l_i[TT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t) blockIdx.x * 4 + wave) * R;
    const int64_t t0 = (int64_t) blockIdx.y * TT;
    const int64_t K = (int64_t) nb * 32;
    float acc[R][TT];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int tt = 0; tt < TT; tt++) acc[r][tt] = 0.0f;

    for (int b0 = 0; b0 < nb; b0 += 64) {
        __syncthreads();
        // stage q: TT * 128 chunks of 16 B
        for (int c = threadIdx.x; c < TT * 128; c += 256) {
            const int tt = c >> 7, cc = c & 127, blk = cc >> 1, half = cc & 1;
            int4 v = make_int4(0, 0, 0, 0);
            if (t0 + tt < T && b0 + blk < nb) v = *reinterpret_cast<const int4 *>(xq + (t0 + tt) * K + (int64_t)(b0 + blk) * 32 + half * 16);
            *reinterpret_cast<int4 *>(l_q + tt * 2048 + half * 1024 + blk * 16) = v;
        }
        for (int c = threadIdx.x; c < TT * 64; c += 256) {
            const int tt = c >> 6, blk = c & 63;
            const bool ok = (t0 + tt < T) && (b0 + blk < nb);
            const int64_t idx = (t0 + tt) * nb + b0 + blk;
            l_d[c] = ok ? xd[idx] : 0.0f;
            l_s[c] = ok ? xs[idx] : 0.0f;
            l_i[c] = ok ? xi[idx] : 0;
        }
        __syncthreads();
        const int b = b0 + lane;
        if (b < nb && row0 < N) {
            WBlk<FMT> w[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int64_t row = (row0 + r < N) ? row0 + r : N - 1;
                load_wblk<FMT, false>(w[r], qs, qh, sc, row * nb + b);
            }
#pragma unroll
            for (int tt = 0; tt < TT; tt++) {
                const int4 alo = *reinterpret_cast<const int4 *>(l_q + tt * 2048 + lane * 16);
                const int4 ahi = *reinterpret_cast<const int4 *>(l_q + tt * 2048 + 1024 + lane * 16);
                const float dx = l_d[tt * 64 + lane], sx = l_s[tt * 64 + lane];
                const int asum = l_i[tt * 64 + lane];
#pragma unroll
                for (int r = 0; r < R; r++) acc[r][tt] = blk_fma<FMT>(w[r], alo, ahi, dx, sx, asum, acc[r][tt]);
            }
        }
    }
    if (row0 >= N) return;
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
        for (int tt = 0; tt < TT; tt++) {
            const float v = wave_sum_f(acc[r][tt]);
            if (lane == 0 && row0 + r < N && t0 + tt < T) y[(t0 + tt) * ldy + row0 + r] = apply_epi(epi, v, t0 + tt, row0 + r, ldy);
        }
    }
}

template <int FMT>
static void launch_mvq_fmt(const DevTensor & W, const QAct & x, int64_t T, float * y, int64_t ldy, const Epi & epi, hipStream_t st) {
    const int64_t N = W.rows();
    const int nb = (int)(W.cols() / 32);
    if (T == 1) {
        constexpr int R = 4;
        const dim3 grid((unsigned)((N + 4 * R - 1) / (4 * R)));
        hipLaunchKernelGGL((k_mvq_t1<FMT, R>), grid, dim3(256), 0, st, W.qs, W.qh, W.sc, N, nb, x.q, x.d, x.s, x.isum, y, epi);
    } else {
        constexpr int R = 4, TT = 8;
        const dim3 grid((unsigned)((N + 4 * R - 1) / (4 * R)), (unsigned)((T + TT - 1) / TT));
        hipLaunchKernelGGL((k_mvq_tn<FMT, R, TT>), grid, dim3(256), 0, st, W.qs, W.qh, W.sc, N, nb, x.q, x.d, x.s, x.isum, T, y, ldy, epi);
    }
}

void launch_matvec_q(const DevTensor & W, const QAct & x, int64_t T, float * y, int64_t ldy, const Epi & epi, hipStream_t st) {
    switch (W.type) {
        case T_Q4_0: launch_mvq_fmt<T_Q4_0>(W, x, T, y, ldy, epi, st); break;
        case T_Q4_1: launch_mvq_fmt<T_Q4_1>(W, x, T, y, ldy, epi, st); break;
        case T_Q5_0: launch_mvq_fmt<T_Q5_0>(W, x, T, y, ldy, epi, st); break;
        case T_Q5_1: launch_mvq_fmt<T_Q5_1>(W, x, T, y, ldy, epi, st); break;
        case T_Q8_0: launch_mvq_fmt<T_Q8_0>(W, x, T, y, ldy, epi, st); break;
        default: break;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Projection, F32 / F16 weights.
// The f32 accumulation order is the one of ggml's AVX2 ggml_vec_dot_f32 / ggml_vec_dot_f16 (and of the CPU oracle):
// 32 partial sums, partial p accumulating k = p, p+32, p+64, ... with fma; then ps[i] += ps[i+16], ps[i] += ps[i+8],
// ps[i] += ps[i+4], result (ps0+ps1)+(ps2+ps3). With fp16-rounded activations (what ggml feeds an F16 matrix) the
// logits are sensitive to this order at the level the reference's FP16 thresholds test, so it is pinned, not chosen
// for speed: 4 lanes own one row (lane q keeps partials 8q..8q+7 = one 16/32-byte chunk per 32-element step), 16 rows
// per wave. Activations for the current K tile are staged (and fp16-rounded) once per workgroup in LDS.
// The same kernel serves T = 1 and token tiles, so sequence mode is bit-identical to serial mode.
// ---------------------------------------------------------------------------------------------------------------

template <bool F16, int TT>
__global__ __launch_bounds__(256) void k_mvf(const void * __restrict__ W, int64_t N, int64_t K, const float * __restrict__ x, int64_t ldx,
                                             int64_t T, float * __restrict__ y, int64_t ldy, Epi epi) {
    constexpr int KT = 2048;
    __shared__ __attribute__((aligned(16))) float l_x[TT * KT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & 3, rloc = lane >> 2;
    const int64_t row = ((int64_t) blockIdx.x * 4 + wave) * 16 + rloc;
    const int64_t rowc = row < N ? row : N - 1;
    const int64_t t0 = (int64_t) blockIdx.y * TT;
    float acc[TT][8];
#pragma unroll
    for (int tt = 0; tt < TT; tt++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[tt][e] = 0.0f;

    for (int64_t k0 = 0; k0 < K; k0 += KT) {
        const int kt = (int) ((K - k0) < KT ? (K - k0) : KT);
        __syncthreads();
        for (int i = threadIdx.x; i < TT * (kt / 4); i += 256) {
            const int tt = i / (kt / 4), c = i % (kt / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 + tt < T) v = *reinterpret_cast<const float4 *>(x + (t0 + tt) * ldx + k0 + 4 * c);
            if constexpr (F16) { v.x = round_f16(v.x); v.y = round_f16(v.y); v.z = round_f16(v.z); v.w = round_f16(v.w); }
            *reinterpret_cast<float4 *>(l_x + tt * KT + 4 * c) = v;
        }
        __syncthreads();
        const int steps = kt / 32;
#pragma unroll 4
        for (int s = 0; s < steps; s++) {
            float w[8];
            const int64_t e0 = rowc * K + k0 + 32 * s + 8 * q;
            if constexpr (F16) {
                const int4 raw = TT == 1 ? ldw16(reinterpret_cast<const uint16_t *>(W) + e0) : *reinterpret_cast<const int4 *>(reinterpret_cast<const uint16_t *>(W) + e0);
                const unsigned u[4] = {(unsigned) raw.x, (unsigned) raw.y, (unsigned) raw.z, (unsigned) raw.w};
#pragma unroll
                for (int i = 0; i < 4; i++) { w[2 * i] = h2f_bits((uint16_t)(u[i] & 0xFFFFu)); w[2 * i + 1] = h2f_bits((uint16_t)(u[i] >> 16)); }
            } else {
                const int4 a = TT == 1 ? ldw16(reinterpret_cast<const float *>(W) + e0) : *reinterpret_cast<const int4 *>(reinterpret_cast<const float *>(W) + e0);
                const int4 b = TT == 1 ? ldw16(reinterpret_cast<const float *>(W) + e0 + 4) : *reinterpret_cast<const int4 *>(reinterpret_cast<const float *>(W) + e0 + 4);
                w[0] = __int_as_float(a.x); w[1] = __int_as_float(a.y); w[2] = __int_as_float(a.z); w[3] = __int_as_float(a.w);
                w[4] = __int_as_float(b.x); w[5] = __int_as_float(b.y); w[6] = __int_as_float(b.z); w[7] = __int_as_float(b.w);
            }
#pragma unroll
            for (int tt = 0; tt < TT; tt++) {
                const float4 xa = *reinterpret_cast<const float4 *>(l_x + tt * KT + 32 * s + 8 * q);
                const float4 xb = *reinterpret_cast<const float4 *>(l_x + tt * KT + 32 * s + 8 * q + 4);
                acc[tt][0] = fmaf(w[0], xa.x, acc[tt][0]); acc[tt][1] = fmaf(w[1], xa.y, acc[tt][1]);
                acc[tt][2] = fmaf(w[2], xa.z, acc[tt][2]); acc[tt][3] = fmaf(w[3], xa.w, acc[tt][3]);
                acc[tt][4] = fmaf(w[4], xb.x, acc[tt][4]); acc[tt][5] = fmaf(w[5], xb.y, acc[tt][5]);
                acc[tt][6] = fmaf(w[6], xb.z, acc[tt][6]); acc[tt][7] = fmaf(w[7], xb.w, acc[tt][7]);
            }
        }
    }
#pragma unroll
    for (int tt = 0; tt < TT; tt++) {
        float ps[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float v = acc[tt][e];
            v += __shfl_xor(v, 2, WAVE);  // ps[i] += ps[i + 16]
            v += __shfl_xor(v, 1, WAVE);  // ps[i] += ps[i + 8]
            ps[e] = v;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) ps[e] += ps[e + 4];
        const float r = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        if (q == 0 && row < N && t0 + tt < T) y[(t0 + tt) * ldy + row] = apply_epi(epi, r, t0 + tt, row, ldy);
    }
}

template <bool F16>
static void launch_mvf_t(const DevTensor & W, const float * x, int64_t ldx, int64_t T, float * y, int64_t ldy, const Epi & epi, hipStream_t st) {
    const int64_t N = W.rows(), K = W.cols();
    const unsigned gx = (unsigned)((N + 63) / 64);
    if (T == 1) {
        hipLaunchKernelGGL((k_mvf<F16, 1>), dim3(gx), dim3(256), 0, st, W.data, N, K, x, ldx, T, y, ldy, epi);
    } else {
        constexpr int TT = 4;
        hipLaunchKernelGGL((k_mvf<F16, TT>), dim3(gx, (unsigned)((T + TT - 1) / TT)), dim3(256), 0, st, W.data, N, K, x, ldx, T, y, ldy, epi);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Sequence mode, F16 weights: y[T][N] = epi(W[N][K] . fp16(x[T][K])) on the matrix cores (v_mfma_f32_32x32x16_f16).
//
// What it keeps of ggml's F16 product (rwkv_graph.inc:416-447 low-rank stages, :744-866 the sequential graph on an FP16 file): the
// activations are rounded to fp16 first, every product is exact in f32, the sum is accumulated in f32. What it does NOT keep is the
// ORDER of the additions (ggml: 32 partial sums k mod 32, each a chain of single FMAs; the matrix core adds eight products at a time
// into one accumulator) -- so this path is not bit-identical to the single-token kernel / the CPU oracle, it agrees with them to
// rounding: measured <= 3e-6 relative on the logits. The contract allows exactly that: the reference promises serial == sequence
// bit for bit only for FP32 files (tests/test_eval_sequence_in_chunks.c:54 runs on an FP32 model) and validates FP16 against recorded
// thresholds (tests/test_tiny_rwkv.c:38-54). FP32 matrices therefore stay on k_mvf (memcmp equality), F16 matrices take this kernel
// from k_mfma_min_tokens tokens per pass on WHEN RWKV_MI_SEQ_F16=mfma asks for it (round 6: the default keeps them on k_mvf, bit-identical to the serial path).
//
// These products are short (K = 64 .. 320 for the second low-rank stages, N = 64 .. 320 for the first) and were 22 % of an RWKV-7 2.9B
// pass on the VALU kernel (62 of 281 ms, round-3 review).
// ---------------------------------------------------------------------------------------------------------------
typedef _Float16 mf_h8 __attribute__((ext_vector_type(8)));
typedef float mf_f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ mf_h8 mf_cvt8(const float4 a, const float4 b) {
    mf_h8 h;
    h[0] = (_Float16) a.x; h[1] = (_Float16) a.y; h[2] = (_Float16) a.z; h[3] = (_Float16) a.w;
    h[4] = (_Float16) b.x; h[5] = (_Float16) b.y; h[6] = (_Float16) b.z; h[7] = (_Float16) b.w;
    return h;
}

// LDS-tiled: a workgroup (4 waves) owns 64 tokens x 64 weight rows, K walks in chunks of 64 through two LDS buffers. Global loads
// are row-contiguous (activations: 256 bytes of f32 per token and chunk, rounded to fp16 on the way into LDS; weights: 128 bytes per
// row); the MFMA operands come out of LDS as 16-byte reads on a 144-byte row pitch (conflict-free for the 32 rows of a lane group).
// (The first version fed the operands straight from global memory: 64 different cache lines per activation load instruction -- 20.0 k
// tokens/s on the RWKV-7 2.9B pass against 18.4 k on the VALU kernel.) Few-tile shapes (the first low-rank stages: N = 64 .. 320)
// additionally split K over blockIdx.z; k_mmf16_combine adds the parts in part order (deterministic) and applies the epilogue.
constexpr int MF_KC = 64, MF_PITCH = 72;      // chunk of K, LDS row pitch in halfs

struct MfArgs {
    const uint16_t * W; const float * x; float * y; float * part;
    int64_t N, K, ldx, T, ldy; int ksplit;
    Epi epi;
};

__global__ __launch_bounds__(256) void k_mmf16_seq(MfArgs p) {
    __shared__ __attribute__((aligned(16))) _Float16 l_x[2][64 * MF_PITCH];
    __shared__ __attribute__((aligned(16))) _Float16 l_w[2][64 * MF_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const int64_t n0 = (int64_t) blockIdx.x * 64, t0 = (int64_t) blockIdx.y * 64;
    const int tw = wave & 1, nw = wave >> 1;                      // this wave's 32-token x 32-row quadrant
    // K range of this part
    const int64_t chunks = (p.K + MF_KC - 1) / MF_KC;
    const int64_t c_lo = chunks * blockIdx.z / p.ksplit, c_hi = chunks * (blockIdx.z + 1) / p.ksplit;
    // staging roles: activations 4 x float4 per thread (row tid / 16 + 16 j, columns 4 (tid % 16) ..), weights 2 x 16 bytes (row tid / 8 + 32 j)
    const int xr = tid >> 4, xc = (tid & 15) * 4, wr = tid >> 3, wc = (tid & 7) * 8;
    float4 xv[4]; int4 wv[2];
    auto fetch = [&](int64_t c) {
        const int64_t k0 = c * MF_KC;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t t = t0 + xr + 16 * j;
            const bool ok = t < p.T && k0 + xc < p.K;
            xv[j] = ok ? *reinterpret_cast<const float4 *>(p.x + t * p.ldx + k0 + xc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int64_t n = n0 + wr + 32 * j;
            const bool ok = n < p.N && k0 + wc < p.K;
            wv[j] = ok ? *reinterpret_cast<const int4 *>(p.W + n * p.K + k0 + wc) : make_int4(0, 0, 0, 0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 h; h[0] = (_Float16) xv[j].x; h[1] = (_Float16) xv[j].y; h[2] = (_Float16) xv[j].z; h[3] = (_Float16) xv[j].w;   // round to nearest even, like ggml's fp32 -> fp16
            *reinterpret_cast<h4 *>(&l_x[buf][(xr + 16 * j) * MF_PITCH + xc]) = h;
        }
#pragma unroll
        for (int j = 0; j < 2; j++) *reinterpret_cast<int4 *>(&l_w[buf][(wr + 32 * j) * MF_PITCH + wc]) = wv[j];
    };
    mf_f16v acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    if (c_lo < c_hi) { fetch(c_lo); stash(0); }
    __syncthreads();
    for (int64_t c = c_lo; c < c_hi; c++) {
        const int buf = (int) ((c - c_lo) & 1);
        if (c + 1 < c_hi) fetch(c + 1);
        const _Float16 * ax = &l_x[buf][(32 * tw + i) * MF_PITCH + 8 * kh];
        const _Float16 * bw = &l_w[buf][(32 * nw + i) * MF_PITCH + 8 * kh];
#pragma unroll
        for (int kk = 0; kk < MF_KC / 16; kk++) {
            const mf_h8 a = *reinterpret_cast<const mf_h8 *>(ax + 16 * kk);
            const mf_h8 b = *reinterpret_cast<const mf_h8 *>(bw + 16 * kk);
            // D[token][row] += A[token][k] * B[k][row]: A = the activations, B = the weights (k of lane l: 8 (l >> 5) .. + 7 of the 16-wide step)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        if (c + 1 < c_hi) stash(buf ^ 1);
        __syncthreads();
    }
    // C / D layout: column (= weight row) lane & 31, row (= token) (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int64_t n = n0 + 32 * nw + i;
    if (p.ksplit == 1) {
        if (n < p.N) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t t = t0 + 32 * tw + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (t < p.T) p.y[t * p.ldy + n] = apply_epi(p.epi, acc[r], t, n, p.ldy);
            }
        }
        return;
    }
    // split K: every part stores its tile ([tile][part][reg][thread]); k_mmf16_combine adds the parts in order (a second launch: partial
    // tiles written on one XCD are only guaranteed visible to another at a kernel boundary)
    const int64_t tile = (int64_t) blockIdx.y * gridDim.x + blockIdx.x;
    float * mine = p.part + ((tile * p.ksplit + blockIdx.z) * 16) * 256;
#pragma unroll
    for (int r = 0; r < 16; r++) mine[r * 256 + tid] = acc[r];
}

// (blockIdx.z = register r of the thread's 16 outputs: sixteen times the workgroups, each thread one output -- its parts are loaded
//  together, not in `ksplit` dependent rounds of sixteen, and added in part order as before)
__global__ __launch_bounds__(256) void k_mmf16_combine(MfArgs p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5, tw = wave & 1, nw = wave >> 1;
    const int64_t n0 = (int64_t) blockIdx.x * 64, t0 = (int64_t) blockIdx.y * 64;
    const int64_t tile = (int64_t) blockIdx.y * gridDim.x + blockIdx.x;
    const int r = blockIdx.z;
    const float * base = p.part + (tile * p.ksplit * 16 + r) * 256 + tid;
    float part[8];                                               // (MF_MAX_SPLIT)
#pragma unroll
    for (int z = 0; z < 8; z++) part[z] = z < p.ksplit ? base[(int64_t) z * 16 * 256] : 0.0f;
    float sum = part[0];
#pragma unroll
    for (int z = 1; z < 8; z++) if (z < p.ksplit) sum += part[z];
    const int64_t n = n0 + 32 * nw + i;
    const int64_t t = t0 + 32 * tw + (r & 3) + 8 * (r >> 2) + 4 * kh;
    if (n < p.N && t < p.T) p.y[t * p.ldy + n] = apply_epi(p.epi, sum, t, n, p.ldy);
}

// ---------------------------------------------------------------------------------------------------------------
// Sequence mode, F16 / F32 weights, EXACT: y[T][N] = epi(W[N][K] . x[T][K]) on the matrix cores in ggml's own addition order (round 6).
//
// ggml's vec_dot (and k_mvf above, and the oracle) keeps 32 partial sums, partial p = the chain fma(w[k], x[k], .) over k = p, p + 32,
// p + 64, ... and folds them at the end. v_mfma_f32_16x16x4_f32 adds its four k slices onto the accumulator as a chain of single fused
// multiply-adds in k order (tools/mfma_f32_chain.hip: bit for bit against fmaf on 1 M random outputs) -- so ONE instruction with the k
// slices p, p + 32, p + 64, p + 96 of a 128-element step is four links of partial p's chain for a 16-token x 16-row tile, and 32 such
// instructions (32 accumulators of 4 registers) are the whole step. Lane l = (k slice l / 16, token or row l % 16) reads the 32 consecutive
// elements of its slice of its token (128 bytes, rounded to fp16 for F16 weights as ggml does) and of its weight row (64 / 128 bytes):
// no LDS, no barrier; the two token groups and two row groups of a workgroup share their lines in the CU's L1. The fold of the 32
// partials (16, 8, 4, then (p0 + p1) + (p2 + p3)) is k_mvf's, inside the lane. Bit-identical to k_mvf, hence to the serial path and the
// oracle (tests/test_gpu_seq_f16.py); RWKV_MI_SEQ_F=valu keeps sequence passes on k_mvf's token tiles (A/B, tests).
// What it is for: FP16 / FP32 files (every matrix) and the low-rank stages of RWKV-7 in a quantised file -- until now 48.8 us per launch on
// the VALU token tiles, 12 ms of a 52.7 ms pass of the 2.9B.
// ---------------------------------------------------------------------------------------------------------------
typedef float mf_f4 __attribute__((ext_vector_type(4)));
std::atomic<unsigned long long> g_mmfx_launches{0};

// Workgroup = 4 waves = 2 token groups x 2 row groups; a wave owns 16 tokens x 16 RW rows (RW = 1, 2 MFMA tiles sharing the token operand),
// the workgroup 32 tokens x 32 RW rows. K walks in steps of 128 through two LDS buffers: global loads are row-contiguous (512 bytes of f32
// per token and step, rounded to fp16 VALUES on the way in for F16 weights; 256 / 512 bytes per weight row), the operands come out of LDS as
// 16-byte reads on a pitch of 132 floats / 136 halfs (the 16 rows of a lane group start 4 banks apart: conflict-free).
// (The first version of this kernel read its operands straight from global memory, 128 bytes per lane: 64 different cache lines per load
//  instruction -- 98.6 ms per pass of an FP16 1.6B file against 67.5 on k_mvf's token tiles.)
constexpr int FX_KC = 128, FX_PX = 132, FX_PW16 = 136, FX_PW32 = 132;
// PH = 2: the 32 partials are independent chains until the fold, so a SECOND set of four waves owns partials 16 .. 31 of the same tiles (half the
// accumulators per wave: 64 RW registers instead of 128 RW, twice the waves per tile -- short products no longer sit on one wave per SIMD);
// the fold's first level, ps[i] += ps[i + 16], joins the two sets through LDS.
template <bool F16, int RW, int PH> struct FxLds {
    static constexpr int X_FLOATS = 32 * FX_PX;
    static constexpr int W_BYTES = 32 * RW * FX_PW32 * 4;               // weights sit in LDS as f32 for both types (F16: converted once, on the way in)
    static constexpr int BUF = X_FLOATS * 4 + W_BYTES;
    static constexpr int XCH = PH == 2 ? 4 * RW * 16 * 64 * 16 : 0;        // [wave][row tile][partial][lane] float4
    static constexpr int BYTES = 2 * BUF > XCH ? 2 * BUF : XCH;
};

template <bool F16, int RW, int PH>
__global__ __launch_bounds__(256 * PH) void k_mmfx_seq(const void * __restrict__ W, int64_t N, int64_t K, const float * __restrict__ x, int64_t ldx,
                                                       int64_t T, float * __restrict__ y, int64_t ldy, Epi epi) {
    typedef FxLds<F16, RW, PH> L;
    constexpr int NT = 256 * PH, NP = 32 / PH;
    extern __shared__ __attribute__((aligned(16))) unsigned char fx_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w4 = wave & 3, ph = wave >> 2;
    const int j = lane & 15, kq = lane >> 4, tw = w4 & 1, nw = w4 >> 1;
    const int64_t n0 = (int64_t) blockIdx.x * 32 * RW, t0 = (int64_t) blockIdx.y * 32;
    auto lx = [&](int buf) { return reinterpret_cast<float *>(fx_lds + buf * L::BUF); };
    auto lw = [&](int buf) { return fx_lds + buf * L::BUF + L::X_FLOATS * 4; };
    // staging roles (idx = tid + NT i). Activations: 1024 float4 per step (token idx / 32, floats 4 (idx % 32) ..). Weights, F16: 512 RW int4 (row
    // idx / 16, halfs 8 (idx % 16) ..); F32: 1024 RW float4 (row idx / 32, floats 4 (idx % 32) ..)
    constexpr int NX = 1024 / NT, NW16 = 512 * RW / NT, NW32 = 1024 * RW / NT;
    float4 xv[NX]; int4 wv[F16 ? NW16 : NW32];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < NX; i++) {
            const int idx = tid + NT * i;
            const int64_t t = t0 + (idx >> 5), k = k0 + 4 * (idx & 31);
            xv[i] = (t < T && k < K) ? *reinterpret_cast<const float4 *>(x + t * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);   // (K % 32 == 0: absent slices are zeros, fma(0, 0, a) = a)
        }
        if constexpr (F16) {
#pragma unroll
            for (int i = 0; i < NW16; i++) {
                const int idx = tid + NT * i;
                const int64_t n = n0 + (idx >> 4), k = k0 + 8 * (idx & 15);
                wv[i] = (n < N && k < K) ? *reinterpret_cast<const int4 *>(reinterpret_cast<const uint16_t *>(W) + n * K + k) : make_int4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NW32; i++) {
                const int idx = tid + NT * i;
                const int64_t n = n0 + (idx >> 5), k = k0 + 4 * (idx & 31);
                wv[i] = (n < N && k < K) ? *reinterpret_cast<const int4 *>(reinterpret_cast<const float *>(W) + n * K + k) : make_int4(0, 0, 0, 0);
            }
        }
    };
    auto stash = [&](int buf) {
        float * const bx = lx(buf);
#pragma unroll
        for (int i = 0; i < NX; i++) {
            const int idx = tid + NT * i;
            float4 v = xv[i];
            if constexpr (F16) { v.x = round_f16(v.x); v.y = round_f16(v.y); v.z = round_f16(v.z); v.w = round_f16(v.w); }     // what ggml feeds an F16 matrix
            *reinterpret_cast<float4 *>(bx + (idx >> 5) * FX_PX + 4 * (idx & 31)) = v;
        }
        if constexpr (F16) {
            // converted here, once per element (the first version kept halfs in LDS and converted in every wave that read them -- two waves per
            // element, unpacking shifts included: 6.6 VALU instructions per MFMA and 182 us per launch against 4.1 and 128 for F32 weights)
            float * const bw = reinterpret_cast<float *>(lw(buf));
#pragma unroll
            for (int i = 0; i < NW16; i++) {
                const int idx = tid + NT * i;
                const unsigned u[4] = {(unsigned) wv[i].x, (unsigned) wv[i].y, (unsigned) wv[i].z, (unsigned) wv[i].w};
                float f[8];
#pragma unroll
                for (int e = 0; e < 4; e++) { f[2 * e] = h2f_bits((uint16_t) (u[e] & 0xFFFFu)); f[2 * e + 1] = h2f_bits((uint16_t) (u[e] >> 16)); }
                float * dst = bw + (idx >> 4) * FX_PW32 + 8 * (idx & 15);
                *reinterpret_cast<float4 *>(dst) = make_float4(f[0], f[1], f[2], f[3]);
                *reinterpret_cast<float4 *>(dst + 4) = make_float4(f[4], f[5], f[6], f[7]);
            }
        } else {
            float * const bw = reinterpret_cast<float *>(lw(buf));
#pragma unroll
            for (int i = 0; i < NW32; i++) { const int idx = tid + NT * i; *reinterpret_cast<int4 *>(bw + (idx >> 5) * FX_PW32 + 4 * (idx & 31)) = wv[i]; }
        }
    };
    mf_f4 acc[RW][NP];
#pragma unroll
    for (int r = 0; r < RW; r++)
#pragma unroll
        for (int p = 0; p < NP; p++) acc[r][p] = (mf_f4){0.0f, 0.0f, 0.0f, 0.0f};
    fetch(0); stash(0);
    __syncthreads();
    int buf = 0;
    for (int64_t k0 = 0; k0 < K; k0 += FX_KC, buf ^= 1) {
        const bool more = k0 + FX_KC < K;
        if (more) fetch(k0 + FX_KC);                                  // the next step's lines are in flight under this step's instructions
        float xa[NP];
        {
            const float4 * src = reinterpret_cast<const float4 *>(lx(buf) + (16 * tw + j) * FX_PX + 32 * kq + NP * ph);
#pragma unroll
            for (int i = 0; i < NP / 4; i++) { const float4 v = src[i]; xa[4 * i] = v.x; xa[4 * i + 1] = v.y; xa[4 * i + 2] = v.z; xa[4 * i + 3] = v.w; }
        }
#pragma unroll
        for (int r = 0; r < RW; r++) {
            float wb[NP];
            const int row = 16 * (RW * nw + r) + j;
            {
                const float4 * src = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(lw(buf)) + row * FX_PW32 + 32 * kq + NP * ph);
#pragma unroll
                for (int i = 0; i < NP / 4; i++) { const float4 v = src[i]; wb[4 * i] = v.x; wb[4 * i + 1] = v.y; wb[4 * i + 2] = v.z; wb[4 * i + 3] = v.w; }
            }
            // D[token][row] += A[token][k] B[k][row]: A = the activations (lane: token l % 16, slice l / 16), B = the weights (row l % 16, slice l / 16);
            // the instruction of partial q = NP ph + p adds the links k = k0 + q, + 32, + 64, + 96 to q's chain
#pragma unroll
            for (int p = 0; p < NP; p++) acc[r][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[p], wb[p], acc[r][p], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
    }
    // The fold of k_mvf: ps[i] += ps[i + 16], += ps[i + 8], += ps[i + 4], (ps0 + ps1) + (ps2 + ps3). PH = 2: the upper partials come from the
    // second wave set through LDS (the staging buffers are free: every wave is behind the loop's last barrier).
    if constexpr (PH == 2) {
        mf_f4 * const xch = reinterpret_cast<mf_f4 *>(fx_lds) + (size_t) w4 * RW * 16 * 64;
        if (ph == 1) {
#pragma unroll
            for (int r = 0; r < RW; r++)
#pragma unroll
                for (int p = 0; p < 16; p++) xch[(r * 16 + p) * 64 + lane] = acc[r][p];
        }
        __syncthreads();
        if (ph == 1) return;
#pragma unroll
        for (int r = 0; r < RW; r++)
#pragma unroll
            for (int p = 0; p < 16; p++) acc[r][p] = acc[r][p] + xch[(r * 16 + p) * 64 + lane];
    } else {
#pragma unroll
        for (int r = 0; r < RW; r++)
#pragma unroll
            for (int p = 0; p < 16; p++) acc[r][p] = acc[r][p] + acc[r][p + 16 < NP ? p + 16 : p];
    }
    // C / D layout: column (= weight row) lane % 16, row (= token) 4 (lane / 16) + register
#pragma unroll
    for (int r = 0; r < RW; r++) {
#pragma unroll
        for (int p = 0; p < 8; p++) acc[r][p] = acc[r][p] + acc[r][p + 8];
#pragma unroll
        for (int p = 0; p < 4; p++) acc[r][p] = acc[r][p] + acc[r][p + 4];
        const mf_f4 res = (acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]);
        const int64_t n = n0 + 16 * (RW * nw + r) + j;
        if (n < N) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int64_t t = t0 + 16 * tw + 4 * kq + e;
                if (t < T) y[t * ldy + n] = apply_epi(epi, res[e], t, n, ldy);
            }
        }
    }
}

template <bool F16, int RW, int PH>
static void launch_mmfx_t(const DevTensor & W, const float * x, int64_t ldx, int64_t T, float * y, int64_t ldy, const Epi & epi, hipStream_t st) {
    typedef FxLds<F16, RW, PH> L;
    static std::atomic<unsigned long long> prepared{0};              // per device: the dynamic-LDS limit of this instantiation
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !(prepared.load() & (1ull << dev))) {
        (void) hipFuncSetAttribute((const void *) k_mmfx_seq<F16, RW, PH>, hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
        (void) hipGetLastError();
        prepared.fetch_or(1ull << dev);
    }
    const int64_t N = W.rows(), K = W.cols();
    const dim3 grid((unsigned) ((N + 32 * RW - 1) / (32 * RW)), (unsigned) ((T + 31) / 32));
    hipLaunchKernelGGL((k_mmfx_seq<F16, RW, PH>), grid, dim3(256 * PH), (size_t) L::BYTES, st, W.data, N, K, x, ldx, T, y, ldy, epi);
}

// RWKV_MI_SEQ_F = valu: sequence passes of F16 / F32 matrices stay on k_mvf's token tiles (round 5's exact arm; A/B, tests). Read per call.
static bool seq_f_on_valu() {
    const char * e = getenv("RWKV_MI_SEQ_F");
    return e && e[0] == 'v';
}

std::atomic<unsigned long long> g_mmf16_launches{0};   // launches of k_mmf16_seq by this process (tests assert that the arm they mean to test ran)

// RWKV_MI_SEQ_F16 = valu (default since round 6: ggml's addition order, sequence == serial bit for bit) | mfma (opt-in: k_mmf16_seq on the matrix cores)
static bool seq_f16_on_mfma() {   // (read per call: the test suite runs both arms in one process)
    const char * e = getenv("RWKV_MI_SEQ_F16");
    return e && e[0] == 'm';
}

// workspace of the split-K form (partial tiles), one per device and STREAM SLOT: launches of one stream are ordered, so a stream can
// reuse its workspace launch after launch; contexts (= streams) of one device get different slots
static std::mutex g_mf_mu;
constexpr int64_t MF_MAX_TILES = 128, MF_MAX_SPLIT = 8;     // (k_mmf16_combine holds MF_MAX_SPLIT parts in registers)
struct MfWs { int dev; hipStream_t st; float * part; };
static std::vector<MfWs> g_mf_ws;
static float * mf_workspace(int dev, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mf_mu);
    for (const MfWs & w : g_mf_ws) if (w.dev == dev && w.st == st) return w.part;
    float * part = nullptr;
    if (hipMalloc((void **) &part, (size_t) MF_MAX_TILES * MF_MAX_SPLIT * 16 * 256 * sizeof(float)) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    g_mf_ws.push_back({dev, st, part});
    return part;
}

// a context that owns its stream drops the stream's workspace when it goes (destroy_context): processes that create and destroy many
// contexts do not accumulate 16 MB per stream handle
void matvec_f_release_stream(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mf_mu);
    for (size_t i = 0; i < g_mf_ws.size();) {
        if (g_mf_ws[i].st == st) { (void) hipFree(g_mf_ws[i].part); g_mf_ws.erase(g_mf_ws.begin() + (long) i); }
        else i++;
    }
}

void launch_matvec_f(const DevTensor & W, const float * x, int64_t ldx, int64_t T, float * y, int64_t ldy, const Epi & epi, hipStream_t st) {
    if (W.type == T_F16 && T >= 32 && W.cols() % 8 == 0 && ldx % 4 == 0 && seq_f16_on_mfma()) {
        const int64_t N = W.rows(), K = W.cols();
        MfArgs a{};
        a.W = (const uint16_t *) W.data; a.x = x; a.y = y; a.N = N; a.K = K; a.ldx = ldx; a.T = T; a.ldy = ldy; a.epi = epi; a.ksplit = 1;
        g_mmf16_launches.fetch_add(1, std::memory_order_relaxed);
        const int64_t tiles = ((N + 63) / 64) * ((T + 63) / 64), chunks = (K + MF_KC - 1) / MF_KC;
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (tiles < 128 && chunks >= 8 && tiles <= MF_MAX_TILES) {
            int ks = (int) (256 / tiles);
            ks = ks > (int) MF_MAX_SPLIT ? (int) MF_MAX_SPLIT : ks;
            ks = ks > (int) (chunks / 2) ? (int) (chunks / 2) : ks;
            float * part = ks >= 2 ? mf_workspace(dev, st) : nullptr;
            if (part) { a.ksplit = ks; a.part = part; }
        }
        const dim3 grid((unsigned) ((N + 63) / 64), (unsigned) ((T + 63) / 64), (unsigned) a.ksplit);
        hipLaunchKernelGGL(k_mmf16_seq, grid, dim3(256), 0, st, a);
        if (a.ksplit > 1) hipLaunchKernelGGL(k_mmf16_combine, dim3(grid.x, grid.y, 16), dim3(256), 0, st, a);
        return;
    }
    if ((W.type == T_F16 || W.type == T_F32) && T >= 32 && W.cols() % 32 == 0 && ldx % 4 == 0 && !seq_f_on_valu()) {
        // the exact arm on the matrix cores (k_mmfx_seq): the same bits as k_mvf
        g_mmfx_launches.fetch_add(1, std::memory_order_relaxed);
        // One row tile per wave, the partials split over two wave sets (121 registers: two workgroups of eight waves per CU). Measured against
        // two row tiles per wave on four waves (424 - 512 registers, one wave per SIMD) and against both at once, 1024-token passes, same box
        // (profiles/r06_seq_f_exact.txt): FP16 1.6B file 39.9 / 56.8 / 43.4 ms, FP32 36.9 / 47.6 / 41.2, RWKV-7 2.9B Q5_1 47.7 / 52.9 / 50.7.
        if (W.type == T_F16) launch_mmfx_t<true, 1, 2>(W, x, ldx, T, y, ldy, epi, st);
        else launch_mmfx_t<false, 1, 2>(W, x, ldx, T, y, ldy, epi, st);
        return;
    }
    if (W.type == T_F16) launch_mvf_t<true>(W, x, ldx, T, y, ldy, epi, st);
    else launch_mvf_t<false>(W, x, ldx, T, y, ldy, epi, st);
}

// ---------------------------------------------------------------------------------------------------------------
// Activation quantiser f32 -> Q8_0/Q8_1 blocks (ggml quantize_row_q8_0 / q8_1 reference semantics)
// one half-wave (32 lanes) per block
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_quant_act(const float * __restrict__ x, int64_t n_blocks, int8_t * __restrict__ q,
                                                   float * __restrict__ d, float * __restrict__ s, int * __restrict__ isum) {
    const int64_t blk = ((int64_t) blockIdx.x * 256 + threadIdx.x) >> 5;
    const int e = threadIdx.x & 31;
    const bool live = blk < n_blocks;  // keep whole waves alive: the reductions below use DPP / permlane across the wave
    const float v = live ? x[blk * 32 + e] : 0.0f;
    const float amax = half_max_f(fabsf(v));
    const float dd = amax / 127.0f;
    const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
    const int qi = (int) roundf(v * id);
    const int sum = half_sum_i(qi);
    if (!live) return;
    q[blk * 32 + e] = (int8_t) qi;
    if (e == 0) {
        d[blk] = round_f16(dd);
        s[blk] = round_f16((float) sum * dd);
        isum[blk] = sum;
    }
}

void launch_quantize_act(const float * x, int64_t T, int64_t K, const QAct & out, hipStream_t st) {
    const int64_t n_blocks = T * K / 32;
    const unsigned grid = (unsigned)((n_blocks * 32 + 255) / 256);
    hipLaunchKernelGGL(k_quant_act, dim3(grid), dim3(256), 0, st, x, n_blocks, out.q, out.d, out.s, out.isum);
}

// ---------------------------------------------------------------------------------------------------------------
// Embedding gather + LayerNorm, LayerNorm. One 256-thread block per token; the row lives in LDS.
// ---------------------------------------------------------------------------------------------------------------

struct EmbView {
    int type;
    const void * data;
    const uint8_t * qs;
    const uint32_t * qh;
    const void * sc;
    int64_t rows;      // n_vocab: a token id that comes from device memory (on-device argmax / sampling, a pipeline's feedback) is clamped to it
};

__device__ __forceinline__ float emb_elem(const EmbView & e, int64_t row, int64_t D, int64_t k) {
    switch (e.type) {
        case T_F32: return reinterpret_cast<const float *>(e.data)[row * D + k];
        case T_F16: return h2f_bits(reinterpret_cast<const uint16_t *>(e.data)[row * D + k]);
        default: break;
    }
    // quantised embedding (never produced by the reference quantiser, supported for completeness)
    const int64_t blk = row * (D / 32) + k / 32;
    const int j = (int)(k & 31);
    float d, m = 0.0f;
    int code;
    if (e.type == T_Q8_0) {
        code = reinterpret_cast<const int8_t *>(e.qs)[blk * 32 + j];
        d = h2f_bits(reinterpret_cast<const uint16_t *>(e.sc)[blk]);
        return code * d;
    }
    const uint8_t byte = e.qs[blk * 16 + (j & 15)];
    code = (j < 16) ? (byte & 0x0F) : (byte >> 4);
    if (e.type == T_Q5_0 || e.type == T_Q5_1) code |= (int)((e.qh[blk] >> j) & 1u) << 4;
    if (e.type == T_Q4_1 || e.type == T_Q5_1) {
        const uint32_t dm = reinterpret_cast<const uint32_t *>(e.sc)[blk];
        d = h2f_bits((uint16_t)(dm & 0xFFFF)); m = h2f_bits((uint16_t)(dm >> 16));
        return code * d + m;
    }
    d = h2f_bits(reinterpret_cast<const uint16_t *>(e.sc)[blk]);
    return (code - (e.type == T_Q4_0 ? 8 : 16)) * d;
}

// normalises the row held in l_row (D floats): ((x - mean) * scale) * w + b, one rounding per step, written to out
__device__ __forceinline__ void block_layernorm(float * l_row, int64_t D, const float * __restrict__ w, const float * __restrict__ b,
                                                float eps, float * __restrict__ out, double * red) {
    double s = 0.0;
    s = ln_partial_sum(l_row, D);
    const float mean = (float)(block_sum_d(s, red) / (double) D);
    double s2 = 0.0;
    s2 = ln_partial_var(l_row, D, mean);
    const float var = (float)(block_sum_d(s2, red) / (double) D);
    const float scale = 1.0f / sqrtf(var + eps);
    int64_t i = threadIdx.x;
    for (; i + 3 * 256 < D; i += 4 * 256) {
        float wv[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { wv[u] = w[i + u * 256]; bv[u] = b[i + u * 256]; }
#pragma unroll
        for (int u = 0; u < 4; u++) { const float y = l_row[i + u * 256] * scale; const float yw = y * wv[u]; out[i + u * 256] = yw + bv[u]; }
    }
    for (; i < D; i += 256) { const float y = l_row[i] * scale; const float yw = y * w[i]; out[i] = yw + b[i]; }
}

__global__ __launch_bounds__(256) void k_embed_ln0(EmbView emb, const uint32_t * __restrict__ tokens, int64_t D,
                                                   const float * __restrict__ w, const float * __restrict__ b, float * __restrict__ x) {
    extern __shared__ __attribute__((aligned(16))) float l_row[];
    __shared__ double red[257];
    const int64_t t = blockIdx.x;
    const int64_t tok = tokens[t];
    const int64_t row = tok < emb.rows ? tok : 0;     // (host-side token ids are range-checked by the API; this guards device-side ones)
    int64_t i = threadIdx.x;
    if (emb.type == T_F16 || emb.type == T_F32) {
        for (; i + 3 * 256 < D; i += 4 * 256) {   // 4 loads in flight per trip
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = emb_elem(emb, row, D, i + u * 256);
#pragma unroll
            for (int u = 0; u < 4; u++) l_row[i + u * 256] = v[u];
        }
    }
    for (; i < D; i += 256) l_row[i] = emb_elem(emb, row, D, i);
    __syncthreads();
    block_layernorm(l_row, D, w, b, 1e-5f, x + t * D, red);
}

__global__ __launch_bounds__(256) void k_layernorm(const float * __restrict__ x, int64_t D, const float * __restrict__ w,
                                                   const float * __restrict__ b, float * __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float l_row[];
    __shared__ double red[257];
    const int64_t t = blockIdx.x;
    int64_t i = threadIdx.x;
    for (; i + 3 * 256 < D; i += 4 * 256) {   // 4 loads in flight per trip
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = x[t * D + i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; u++) l_row[i + u * 256] = v[u];
    }
    for (; i < D; i += 256) l_row[i] = x[t * D + i];
    __syncthreads();
    block_layernorm(l_row, D, w, b, 1e-5f, y + t * D, red);
}

void launch_embed_ln0(const DevTensor & emb, const uint32_t * tokens, int64_t T, int64_t D, const float * w, const float * b, float * x, hipStream_t st) {
    EmbView e{emb.type, emb.data, emb.qs, emb.qh, emb.sc, emb.rows()};
    hipLaunchKernelGGL(k_embed_ln0, dim3((unsigned) T), dim3(256), (size_t) D * sizeof(float), st, e, tokens, D, w, b, x);
}

void launch_layernorm(const float * x, int64_t T, int64_t D, const float * w, const float * b, float * y, hipStream_t st) {
    hipLaunchKernelGGL(k_layernorm, dim3((unsigned) T), dim3(256), (size_t) D * sizeof(float), st, x, D, w, b, y);
}

// ---------------------------------------------------------------------------------------------------------------
// Token-shift mixes
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_mix(MixArgs a, int64_t T, int64_t D) {
    const int64_t n = T * D;
    for (int64_t idx = (int64_t) blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t) gridDim.x * 256) {
        const int64_t t = idx / D, d = idx - t * D;
        const float x = a.xn[idx];
        const float xp = (t == 0) ? a.carry_in[d] : a.xn[idx - D];
        if (a.mode == 0) {
            for (int f = 0; f < a.n_out; f++) { const float c = a.coef[f][d]; const float xc = x * c, pc = xp * c; a.out[f][idx] = xc + (xp - pc); }
        } else {
            const float sx = xp - x;
            if (a.sx) a.sx[idx] = sx;
            for (int f = 0; f < a.n_out; f++) { const float sc = sx * a.coef[f][d]; a.out[f][idx] = sc + x; }
        }
        if (t == T - 1) a.carry_out[d] = x;
    }
}

void launch_mix(const MixArgs & a, int64_t T, int64_t D, hipStream_t st) {
    const int64_t n = T * D;
    const unsigned grid = (unsigned) ((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, st, a, T, D);
}

__global__ __launch_bounds__(256) void k_v6_mix2(V6Mix2Args a, int64_t T, int64_t D, int64_t R) {
    const int64_t n = T * 5 * D;
    for (int64_t idx = (int64_t) blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t) gridDim.x * 256) {
        const int64_t d = idx % D;
        const int64_t f = (idx / D) % 5;
        const int64_t t = idx / (5 * D);
        const float * col = a.w2 + f * R * D + d;  // W2 is stored transposed at load: [5][R][D]
        const float * tl = a.tl + t * 5 * R + f * R;
        float acc = 0.0f;
        for (int64_t m = 0; m < R; m++) acc += col[m * D] * tl[m];
        const int64_t o = t * D + d;
        const float mm = (acc + a.maa[f][d]) * a.sx[o];
        a.out[f][o] = mm + a.xn[o];
    }
}

void launch_v6_mix2(const V6Mix2Args & a, int64_t T, int64_t D, int64_t R, hipStream_t st) {
    const int64_t n = T * 5 * D;
    const unsigned grid = (unsigned) ((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_v6_mix2, dim3(grid), dim3(256), 0, st, a, T, D, R);
}

// ---------------------------------------------------------------------------------------------------------------
// WKV recurrences. Sequential in t inside the kernel; the state stays in registers across the tokens of a call.
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_wkv4(const float * __restrict__ k, const float * __restrict__ v, const float * __restrict__ r,
                                              const float * __restrict__ tf, const float * __restrict__ td,
                                              const float * __restrict__ aa_in, const float * __restrict__ bb_in, const float * __restrict__ pp_in,
                                              float * __restrict__ aa_out, float * __restrict__ bb_out, float * __restrict__ pp_out,
                                              float * __restrict__ out, int64_t T, int64_t D) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    float aa = aa_in[i], bb = bb_in[i], pp = pp_in[i];
    const float u = tf[i], w = td[i];
    for (int64_t t = 0; t < T; t++) {
        const float kk = k[t * D + i], vv = v[t * D + i];
        float ww = u + kk;
        float qq = fmaxf(pp, ww);
        float e1 = det_expf(pp - qq), e2 = det_expf(ww - qq);
        const float a = e1 * aa + e2 * vv;
        const float b = e1 * bb + e2;
        ww = pp + w;
        qq = fmaxf(ww, kk);
        e1 = det_expf(ww - qq); e2 = det_expf(kk - qq);
        aa = e1 * aa + e2 * vv;
        bb = e1 * bb + e2;
        pp = qq;
        out[t * D + i] = r[t * D + i] * (a / b);
    }
    aa_out[i] = aa; bb_out[i] = bb; pp_out[i] = pp;
}

void launch_wkv4(const float * k, const float * v, const float * r, const float * time_first, const float * time_decay,
                 const float * aa_in, const float * bb_in, const float * pp_in, float * aa_out, float * bb_out, float * pp_out,
                 float * out, int64_t T, int64_t D, hipStream_t st) {
    hipLaunchKernelGGL(k_wkv4, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, st, k, v, r, time_first, time_decay,
                       aa_in, bb_in, pp_in, aa_out, bb_out, pp_out, out, T, D);
}

// wkv6: one wave per head, lane j owns value column j of state[h][:, j] (SREG = S when S is a supported compile-time
// size, else the state column is kept in global memory -- generic path for unusual head sizes).
template <int S>
__global__ __launch_bounds__(64) void k_wkv6(const float * __restrict__ r, const float * __restrict__ k, const float * __restrict__ v,
                                             const float * __restrict__ u, int u_per_chan, const float * __restrict__ w, int w_mode,
                                             const float * __restrict__ state_in, float * __restrict__ state_out, float * __restrict__ out,
                                             int64_t T, int64_t H) {
    __shared__ float l_r[S], l_k[S], l_u[S], l_w[S];
    const int64_t h = blockIdx.x;
    const int j = threadIdx.x;
    const int64_t D = H * S;
    float s[S];
    if (j < S) {
#pragma unroll
        for (int i = 0; i < S; i++) s[i] = state_in[h * S * S + i * S + j];
    }
    if (j < S) {
        l_u[j] = u_per_chan ? u[h * S + j] : u[h];
        if (w_mode < 2) l_w[j] = (w_mode == 1) ? w[h * S + j] : w[h];
    }
    for (int64_t t = 0; t < T; t++) {
        __syncthreads();
        if (j < S) {
            l_r[j] = r[t * D + h * S + j];
            l_k[j] = k[t * D + h * S + j];
            if (w_mode == 2) l_w[j] = w[t * D + h * S + j];
        }
        __syncthreads();
        if (j < S) {
            const float vj = v[t * D + h * S + j];
            float o = 0.0f;
#pragma unroll
            for (int i = 0; i < S; i++) {
                const float kv = vj * l_k[i];
                const float prev = s[i];
                const float temp = kv * l_u[i] + prev;
                o += temp * l_r[i];
                s[i] = prev * l_w[i] + kv;
            }
            out[t * D + h * S + j] = o;
        }
    }
    if (j < S) {
#pragma unroll
        for (int i = 0; i < S; i++) state_out[h * S * S + i * S + j] = s[i];
    }
}

__global__ __launch_bounds__(256) void k_wkv6_generic(const float * __restrict__ r, const float * __restrict__ k, const float * __restrict__ v,
                                                      const float * __restrict__ u, int u_per_chan, const float * __restrict__ w, int w_mode,
                                                      const float * __restrict__ state_in, float * __restrict__ state_out, float * __restrict__ out,
                                                      int64_t T, int64_t H, int64_t S) {
    const int64_t h = blockIdx.x;
    const int64_t D = H * S;
    for (int64_t j = threadIdx.x; j < S; j += blockDim.x) {
        for (int64_t t = 0; t < T; t++) {
            const float * sin = (t == 0) ? state_in : state_out;
            const float vj = v[t * D + h * S + j];
            float o = 0.0f;
            for (int64_t i = 0; i < S; i++) {
                const float ki = k[t * D + h * S + i], ri = r[t * D + h * S + i];
                const float ui = u_per_chan ? u[h * S + i] : u[h];
                const float wi = (w_mode == 2) ? w[t * D + h * S + i] : (w_mode == 1 ? w[h * S + i] : w[h]);
                const float kv = vj * ki;
                const float prev = sin[h * S * S + i * S + j];
                const float temp = kv * ui + prev;
                o += temp * ri;
                state_out[h * S * S + i * S + j] = prev * wi + kv;
            }
            out[t * D + h * S + j] = o;
        }
    }
}

void launch_wkv6(const float * r, const float * k, const float * v, const float * u, int u_per_chan, const float * w, int w_mode,
                 const float * state_in, float * state_out, float * out, int64_t T, int64_t H, int64_t S, hipStream_t st) {
    switch (S) {
        case 64: hipLaunchKernelGGL((k_wkv6<64>), dim3((unsigned) H), dim3(64), 0, st, r, k, v, u, u_per_chan, w, w_mode, state_in, state_out, out, T, H); break;
        case 32: hipLaunchKernelGGL((k_wkv6<32>), dim3((unsigned) H), dim3(64), 0, st, r, k, v, u, u_per_chan, w, w_mode, state_in, state_out, out, T, H); break;
        case 16: hipLaunchKernelGGL((k_wkv6<16>), dim3((unsigned) H), dim3(64), 0, st, r, k, v, u, u_per_chan, w, w_mode, state_in, state_out, out, T, H); break;
        case 8:  hipLaunchKernelGGL((k_wkv6<8>),  dim3((unsigned) H), dim3(64), 0, st, r, k, v, u, u_per_chan, w, w_mode, state_in, state_out, out, T, H); break;
        default: hipLaunchKernelGGL(k_wkv6_generic, dim3((unsigned) H), dim3(256), 0, st, r, k, v, u, u_per_chan, w, w_mode, state_in, state_out, out, T, H, S); break;
    }
}

// wkv7: one wave per head, lane i owns value row i of state[h][i, :].
template <int S>
__global__ __launch_bounds__(64) void k_wkv7(const float * __restrict__ r, const float * __restrict__ w, const float * __restrict__ k,
                                             const float * __restrict__ v, const float * __restrict__ a, const float * __restrict__ b,
                                             const float * __restrict__ state_in, float * __restrict__ state_out, float * __restrict__ out,
                                             int64_t T, int64_t H) {
    __shared__ float l_r[S], l_w[S], l_k[S], l_a[S], l_b[S];
    const int64_t h = blockIdx.x;
    const int i = threadIdx.x;
    const int64_t D = H * S;
    float s[S];
    if (i < S) {
#pragma unroll
        for (int j = 0; j < S; j += 4) {
            const float4 q = *reinterpret_cast<const float4 *>(state_in + h * S * S + (int64_t) i * S + j);
            s[j] = q.x; s[j + 1] = q.y; s[j + 2] = q.z; s[j + 3] = q.w;
        }
    }
    for (int64_t t = 0; t < T; t++) {
        __syncthreads();
        if (i < S) {
            const int64_t o = t * D + h * S + i;
            l_r[i] = r[o]; l_w[i] = w[o]; l_k[i] = k[o]; l_a[i] = a[o]; l_b[i] = b[o];
        }
        __syncthreads();
        if (i < S) {
            const float vi = v[t * D + h * S + i];
            float sa = 0.0f;
#pragma unroll
            for (int j = 0; j < S; j++) sa += l_a[j] * s[j];
            float res = 0.0f;
#pragma unroll
            for (int j = 0; j < S; j++) {
                const float kv = vi * l_k[j];
                const float ns = (s[j] * l_w[j] + kv) + sa * l_b[j];
                s[j] = ns;
                res += ns * l_r[j];
            }
            out[t * D + h * S + i] = res;
        }
    }
    if (i < S) {
#pragma unroll
        for (int j = 0; j < S; j += 4)
            *reinterpret_cast<float4 *>(state_out + h * S * S + (int64_t) i * S + j) = make_float4(s[j], s[j + 1], s[j + 2], s[j + 3]);
    }
}

__global__ __launch_bounds__(256) void k_wkv7_generic(const float * __restrict__ r, const float * __restrict__ w, const float * __restrict__ k,
                                                      const float * __restrict__ v, const float * __restrict__ a, const float * __restrict__ b,
                                                      const float * __restrict__ state_in, float * __restrict__ state_out, float * __restrict__ out,
                                                      int64_t T, int64_t H, int64_t S) {
    const int64_t h = blockIdx.x;
    const int64_t D = H * S;
    for (int64_t i = threadIdx.x; i < S; i += blockDim.x) {
        for (int64_t t = 0; t < T; t++) {
            const float * sin = (t == 0) ? state_in : state_out;
            const int64_t base = t * D + h * S;
            const float vi = v[base + i];
            float sa = 0.0f;
            for (int64_t j = 0; j < S; j++) sa += a[base + j] * sin[h * S * S + i * S + j];
            float res = 0.0f;
            for (int64_t j = 0; j < S; j++) {
                const float kv = vi * k[base + j];
                const float ns = (sin[h * S * S + i * S + j] * w[base + j] + kv) + sa * b[base + j];
                state_out[h * S * S + i * S + j] = ns;
                res += ns * r[base + j];
            }
            out[base + i] = res;
        }
    }
}

void launch_wkv7(const float * r, const float * w, const float * k, const float * v, const float * a, const float * b,
                 const float * state_in, float * state_out, float * out, int64_t T, int64_t H, int64_t S, hipStream_t st) {
    switch (S) {
        case 64: hipLaunchKernelGGL((k_wkv7<64>), dim3((unsigned) H), dim3(64), 0, st, r, w, k, v, a, b, state_in, state_out, out, T, H); break;
        case 32: hipLaunchKernelGGL((k_wkv7<32>), dim3((unsigned) H), dim3(64), 0, st, r, w, k, v, a, b, state_in, state_out, out, T, H); break;
        default: hipLaunchKernelGGL(k_wkv7_generic, dim3((unsigned) H), dim3(256), 0, st, r, w, k, v, a, b, state_in, state_out, out, T, H, S); break;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Group norm (+ v7 bonus, + gate), v7 key prep, small elementwise kernels. One wave per (token, head).
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_groupnorm(float * __restrict__ x, const float * __restrict__ lw, const float * __restrict__ lb, float eps,
                                                   const float * __restrict__ gate, const float * __restrict__ v7_k, const float * __restrict__ v7_r,
                                                   const float * __restrict__ v7_v, const float * __restrict__ v7_rk, int64_t TH, int64_t H, int64_t S) {
    const int lane = threadIdx.x & 63;
    const int64_t th = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (th >= TH) return;
    const int64_t h = th % H;
    const int64_t base = th * S;  // == t*D + h*S
    double s = 0.0;
    for (int64_t j = lane; j < S; j += WAVE) s += (double) x[base + j];
    const float mean = (float)(wave_sum_d(s) / (double) S);
    double s2 = 0.0;
    for (int64_t j = lane; j < S; j += WAVE) { const float v = x[base + j] - mean; s2 += (double)(v * v); }
    const float var = (float)(wave_sum_d(s2) / (double) S);
    const float scale = 1.0f / sqrtf(var + eps);
    float bonus = 0.0f;
    if (v7_k) {
        float p = 0.0f;
        for (int64_t j = lane; j < S; j += WAVE) p += (v7_k[base + j] * v7_r[base + j]) * v7_rk[h * S + j];
        bonus = wave_sum_f(p);
    }
    for (int64_t j = lane; j < S; j += WAVE) {
        float y = (x[base + j] - mean) * scale;
        y = y * lw[h * S + j];
        y = y + lb[h * S + j];
        if (v7_k) y += v7_v[base + j] * bonus;
        if (gate) y *= gate[base + j];
        x[base + j] = y;
    }
}

void launch_groupnorm(float * x, const float * lw, const float * lb, float eps, const float * gate,
                      const float * v7_k, const float * v7_r, const float * v7_v, const float * v7_rk,
                      int64_t T, int64_t H, int64_t S, hipStream_t st) {
    const int64_t TH = T * H;
    hipLaunchKernelGGL(k_groupnorm, dim3((unsigned)((TH + 3) / 4)), dim3(256), 0, st, x, lw, lb, eps, gate, v7_k, v7_r, v7_v, v7_rk, TH, H, S);
}

__global__ __launch_bounds__(256) void k_v7_kprep(const float * __restrict__ k, const float * __restrict__ a, const float * __restrict__ k_k,
                                                  const float * __restrict__ k_a, float * __restrict__ k_out, float * __restrict__ neg_kk,
                                                  float * __restrict__ kk_a, int64_t TH, int64_t H, int64_t S) {
    const int lane = threadIdx.x & 63;
    const int64_t th = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (th >= TH) return;
    const int64_t h = th % H;
    const int64_t base = th * S;
    float p = 0.0f;
    for (int64_t j = lane; j < S; j += WAVE) { const float t = k[base + j] * k_k[h * S + j]; p += t * t; }
    const float sum = wave_sum_f(p);
    const float scale = 1.0f / fmaxf(sqrtf(sum), 1e-12f);
    for (int64_t j = lane; j < S; j += WAVE) {
        const float kv = k[base + j], av = a[base + j];
        const float kk = (kv * k_k[h * S + j]) * scale;
        const float ka = kv * k_a[h * S + j];
        const float aka = av * ka;
        k_out[base + j] = kv + (aka - ka);
        neg_kk[base + j] = -kk;
        kk_a[base + j] = kk * av;
    }
}

void launch_v7_kprep(const float * k, const float * a, const float * k_k, const float * k_a, float * k_out, float * neg_kk, float * kk_a,
                     int64_t T, int64_t H, int64_t S, hipStream_t st) {
    const int64_t TH = T * H;
    hipLaunchKernelGGL(k_v7_kprep, dim3((unsigned)((TH + 3) / 4)), dim3(256), 0, st, k, a, k_k, k_a, k_out, neg_kk, kk_a, TH, H, S);
}

__global__ __launch_bounds__(256) void k_v7_vmix(float * __restrict__ v, const float * __restrict__ v_first, const float * __restrict__ gate, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) { const float dv = (v_first[i] - v[i]) * gate[i]; v[i] = v[i] + dv; }
}
void launch_v7_vmix(float * v, const float * v_first, const float * gate, int64_t n, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_v7_vmix, dim3(grid), dim3(256), 0, st, v, v_first, gate, n);
}

__global__ __launch_bounds__(256) void k_mul(float * __restrict__ y, const float * __restrict__ a, const float * __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) y[i] = a[i] * b[i];
}
void launch_mul(float * y, const float * a, const float * b, int64_t n, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_mul, dim3(grid), dim3(256), 0, st, y, a, b, n);
}

__global__ __launch_bounds__(256) void k_copy(float * __restrict__ dst, const float * __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) dst[i] = src[i];
}
void launch_copy_f32(float * dst, const float * src, int64_t n, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st, dst, src, n);
}

__global__ __launch_bounds__(256) void k_fill_state_v4(float * __restrict__ state, int64_t n_layer, int64_t D) {
    const int64_t n = n_layer * 5 * D;
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) state[i] = ((i / D) % 5 == 4) ? -1e30f : 0.0f;
}
void launch_fill_state_v4(float * state, int64_t n_layer, int64_t D, hipStream_t st) {
    const int64_t n = n_layer * 5 * D;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_fill_state_v4, dim3(grid), dim3(256), 0, st, state, n_layer, D);
}

__global__ __launch_bounds__(1024) void k_argmax(const float * __restrict__ logits, int64_t n, uint32_t * __restrict__ out) {
    __shared__ float l_v[16];
    __shared__ int l_i[16];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    int64_t i = threadIdx.x;
    for (; i + 7 * (int64_t) blockDim.x < n; i += 8 * (int64_t) blockDim.x) {   // 8 loads in flight per trip
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = logits[i + u * (int64_t) blockDim.x];
#pragma unroll
        for (int u = 0; u < 8; u++) if (v[u] > best) { best = v[u]; bi = (int) (i + u * (int64_t) blockDim.x); }
    }
    for (; i < n; i += blockDim.x) {
        const float v = logits[i];
        if (v > best) { best = v; bi = (int) i; }  // strided scan keeps the smallest index per thread for ties
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, WAVE);
        const int oi = __shfl_xor(bi, o, WAVE);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { l_v[wave] = best; l_i[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++)
            if (l_v[w] > best || (l_v[w] == best && l_i[w] < bi)) { best = l_v[w]; bi = l_i[w]; }
        // (no element compared greater than -inf: every logit is NaN or -inf, e.g. after a poll time-out of the persistent kernel on a
        //  shared GPU. The token feeds the next embedding lookup on the device: it must stay a row of the table.)
        *out = bi == 0x7fffffff ? 0u : (uint32_t) bi;
    }
}
void launch_argmax(const float * logits, int64_t n, uint32_t * out, hipStream_t st) {
    hipLaunchKernelGGL(k_argmax, dim3(1), dim3(1024), 0, st, logits, n, out);
}

// load-time transpose of the v6 mix matrix: [5][D][R] (file) -> [5][R][D], so that lanes read consecutive d
__global__ __launch_bounds__(256) void k_transpose_w2(const float * __restrict__ src, float * __restrict__ dst, int64_t D, int64_t R) {
    const int64_t n = 5 * D * R;
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) {
        const int64_t d = i % D, m = (i / D) % R, f = i / (D * R);
        dst[i] = src[(f * D + d) * R + m];
    }
}
void launch_transpose_w2(const float * src, float * dst, int64_t D, int64_t R, hipStream_t st) {
    hipLaunchKernelGGL(k_transpose_w2, dim3(1024), dim3(256), 0, st, src, dst, D, R);
}

// ---------------------------------------------------------------------------------------------------------------
// Load-time re-pack: file blocks (d [m] [qh] qs) -> planes. One thread per block.
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_repack(int type, const uint8_t * __restrict__ raw, int64_t n_blocks, uint8_t * __restrict__ qs,
                                                uint32_t * __restrict__ qh, uint8_t * __restrict__ sc) {
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    int bsz, qoff, qbytes, scb, hoff = -1;
    switch (type) {
        case T_Q4_0: bsz = 18; scb = 2; qoff = 2; qbytes = 16; break;
        case T_Q4_1: bsz = 20; scb = 4; qoff = 4; qbytes = 16; break;
        case T_Q5_0: bsz = 22; scb = 2; hoff = 2; qoff = 6; qbytes = 16; break;
        case T_Q5_1: bsz = 24; scb = 4; hoff = 4; qoff = 8; qbytes = 16; break;
        default:     bsz = 34; scb = 2; qoff = 2; qbytes = 32; break;  // Q8_0
    }
    const uint8_t * src = raw + b * bsz;
    for (int i = 0; i < scb; i++) sc[b * scb + i] = src[i];
    if (hoff >= 0) qh[b] = (uint32_t) src[hoff] | ((uint32_t) src[hoff + 1] << 8) | ((uint32_t) src[hoff + 2] << 16) | ((uint32_t) src[hoff + 3] << 24);
    for (int i = 0; i < qbytes; i++) qs[b * qbytes + i] = src[qoff + i];
}

void launch_repack(int type, const uint8_t * raw, int64_t n_blocks, uint8_t * qs, uint32_t * qh, void * sc, hipStream_t st) {
    hipLaunchKernelGGL(k_repack, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, st, type, raw, n_blocks, qs, qh, (uint8_t *) sc);
}

}  // namespace rwkvmi
