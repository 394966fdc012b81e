/*
 * rwkv_oracle_fast.c -- SIMD row kernels of the CPU oracle (TEST INFRASTRUCTURE, see rwkv_oracle.h), used by bench.py's
 * cpu_baseline leg so that the CPU number quoted beside the GPU's is a credible one (ggml's own CPU path is AVX2 / AVX-512-VNNI
 * code, reference README.md:21-31). Same arithmetic as the scalar loops of rwkv_oracle.c, vectorised where that cannot change a
 * bit: the per-block integer dot products are exact in any order (vpmaddubsw / vpdpbusd), the f32 accumulation keeps the scalar
 * code's order (block b -> partial b mod 64, one fmaf per block, halving tree); F16 / F32 rows keep ggml's 32 partial sums as four
 * 8-lane FMA accumulators -- the partial index IS the lane. tests/test_oracle_golden.py asserts bit-equality with the scalar
 * oracle on the fixtures and on random rows.
 *
 * Compiled with -mavx2 -mfma -mf16c (and the VNNI kernels with target attributes, chosen at run time by cpuid).
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "rwkv_oracle.h"

static inline uint16_t rd16(const uint8_t * p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t rd32(const uint8_t * p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline float h2f(uint16_t h) { return _cvtsh_ss(h); }

static inline int hsum_i32(__m256i v) {
    __m128i s = _mm_add_epi32(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0x4E));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0xB1));
    return _mm_cvtsi128_si32(s);
}

/* sum over 32 of u8 * i8 (u <= 31: pair sums stay far below the i16 saturation of vpmaddubsw) */
static inline int dot_u8_i8(__m256i u, __m256i x) {
    const __m256i p16 = _mm256_maddubs_epi16(u, x);
    return hsum_i32(_mm256_madd_epi16(p16, _mm256_set1_epi16(1)));
}
__attribute__((target("avx512vnni,avx512vl")))
static inline int dot_u8_i8_vnni(__m256i u, __m256i x) { return hsum_i32(_mm256_dpbusd_epi32(_mm256_setzero_si256(), u, x)); }

/* 32 4-bit codes of a block -> 32 bytes (elements 0..15 low nibbles, 16..31 high nibbles) */
static inline __m256i nibbles(const uint8_t * qs) {
    const __m128i raw = _mm_loadu_si128((const __m128i *) qs);
    const __m128i lo = _mm_and_si128(raw, _mm_set1_epi8(0x0F));
    const __m128i hi = _mm_and_si128(_mm_srli_epi16(raw, 4), _mm_set1_epi8(0x0F));
    return _mm256_set_m128i(hi, lo);
}
/* bit j of qh -> 0x10 in byte j */
static inline __m256i fifth_bits(uint32_t qh) {
    const __m256i sh = _mm256_setr_epi64x(0x0000000000000000LL, 0x0101010101010101LL, 0x0202020202020202LL, 0x0303030303030303LL);
    __m256i v = _mm256_shuffle_epi8(_mm256_set1_epi32((int) qh), sh);
    const __m256i bit = _mm256_set1_epi64x((long long) 0x8040201008040201ULL);
    v = _mm256_cmpeq_epi8(_mm256_and_si256(v, bit), bit);
    return _mm256_and_si256(v, _mm256_set1_epi8(0x10));
}

static int g_vnni = -1;
static int have_vnni(void) {
    if (g_vnni < 0) g_vnni = (__builtin_cpu_supports("avx512vnni") && __builtin_cpu_supports("avx512vl")) ? 1 : 0;
    return g_vnni;
}
int orc_fast_uses_vnni(void) { return have_vnni(); }

/* unsigned (or |w|) code bytes of block b of a row + its scales; for Q8_0 the activation vector gets the weights' signs */
static inline __m256i block_u(int wtype, const uint8_t * row, int64_t b, __m256i * x, uint16_t * dh, uint16_t * mh) {
    switch (wtype) {
    case ORC_Q4_0: { const uint8_t * blk = row + b * 18; *dh = rd16(blk); return nibbles(blk + 2); }
    case ORC_Q4_1: { const uint8_t * blk = row + b * 20; *dh = rd16(blk); *mh = rd16(blk + 2); return nibbles(blk + 4); }
    case ORC_Q5_0: { const uint8_t * blk = row + b * 22; *dh = rd16(blk); return _mm256_or_si256(nibbles(blk + 6), fifth_bits(rd32(blk + 2))); }
    case ORC_Q5_1: { const uint8_t * blk = row + b * 24; *dh = rd16(blk); *mh = rd16(blk + 2); return _mm256_or_si256(nibbles(blk + 8), fifth_bits(rd32(blk + 4))); }
    default: {
        const uint8_t * blk = row + b * 34; *dh = rd16(blk);
        const __m256i w = _mm256_loadu_si256((const __m256i *) (blk + 2));
        *x = _mm256_sign_epi8(*x, w);
        return _mm256_sign_epi8(w, w); }
    }
}

#define ROW_Q_BODY(DOT8)                                                                                                              \
    float lanes[64];                                                                                                                  \
    for (int i = 0; i < 64; i++) lanes[i] = 0.0f;                                                                                     \
    const int has_m = (wtype == ORC_Q4_1 || wtype == ORC_Q5_1);                                                                       \
    const int off = wtype == ORC_Q4_0 ? 8 : (wtype == ORC_Q5_0 ? 16 : 0);                                                             \
    int64_t b = 0;                                                                                                                    \
    for (; b + 8 <= nb; b += 8) {   /* eight blocks: eight exact integer dots, then ONE vector fma on partials b mod 64 .. +7 */       \
        __m256i p[8];                                                                                                                 \
        uint16_t dh[8], mh[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                                                             \
        for (int j = 0; j < 8; j++) {                                                                                                 \
            __m256i x = _mm256_loadu_si256((const __m256i *) (q + (b + j) * 32));                                                     \
            const __m256i u = block_u(wtype, row, b + j, &x, &dh[j], &mh[j]);                                                         \
            p[j] = DOT8(u, x);                                                                                                        \
        }                                                                                                                             \
        const __m256i t0 = _mm256_hadd_epi32(p[0], p[1]), t1 = _mm256_hadd_epi32(p[2], p[3]);                                         \
        const __m256i t2 = _mm256_hadd_epi32(p[4], p[5]), t3 = _mm256_hadd_epi32(p[6], p[7]);                                         \
        const __m256i u0 = _mm256_hadd_epi32(t0, t1), u1 = _mm256_hadd_epi32(t2, t3);                                                 \
        __m256i isum = _mm256_add_epi32(_mm256_permute2x128_si256(u0, u1, 0x20), _mm256_permute2x128_si256(u0, u1, 0x31));            \
        if (off) isum = _mm256_sub_epi32(isum, _mm256_mullo_epi32(_mm256_set1_epi32(off), _mm256_loadu_si256((const __m256i *) (xsum + b)))); \
        const __m256 dw = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) dh));                                                     \
        const __m256 dd = _mm256_mul_ps(dw, _mm256_loadu_ps(dq + b));                                                                 \
        __m256 a = _mm256_loadu_ps(lanes + (b & 63));                                                                                 \
        a = _mm256_fmadd_ps(dd, _mm256_cvtepi32_ps(isum), a);                                                                         \
        if (has_m) a = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) mh)), _mm256_loadu_ps(sq + b), a);            \
        _mm256_storeu_ps(lanes + (b & 63), a);                                                                                        \
    }                                                                                                                                 \
    for (; b < nb; b++) {                                                                                                             \
        __m256i x = _mm256_loadu_si256((const __m256i *) (q + b * 32));                                                               \
        uint16_t dh = 0, mh = 0;                                                                                                      \
        const __m256i u = block_u(wtype, row, b, &x, &dh, &mh);                                                                       \
        const int32_t isum = hsum_i32(DOT8(u, x)) - off * xsum[b];                                                                    \
        float a = lanes[b & 63];                                                                                                      \
        a = fmaf(h2f(dh) * dq[b], (float) isum, a);                                                                                   \
        if (has_m) a = fmaf(h2f(mh), sq[b], a);                                                                                       \
        lanes[b & 63] = a;                                                                                                            \
    }                                                                                                                                 \
    for (int o = 32; o > 0; o >>= 1) for (int i = 0; i < o; i++) lanes[i] += lanes[i + o];                                            \
    return lanes[0];

static inline __m256i dot8_avx2(__m256i u, __m256i x) { return _mm256_madd_epi16(_mm256_maddubs_epi16(u, x), _mm256_set1_epi16(1)); }
__attribute__((target("avx512vnni,avx512vl")))
static inline __m256i dot8_vnni(__m256i u, __m256i x) { return _mm256_dpbusd_epi32(_mm256_setzero_si256(), u, x); }

static float row_q_avx2(int wtype, const uint8_t * row, const int8_t * q, const float * dq, const float * sq, const int32_t * xsum, int64_t nb) {
    ROW_Q_BODY(dot8_avx2)
}
__attribute__((target("avx512vnni,avx512vl")))
static float row_q_vnni(int wtype, const uint8_t * row, const int8_t * q, const float * dq, const float * sq, const int32_t * xsum, int64_t nb) {
    ROW_Q_BODY(dot8_vnni)
}

/* One row of a quantised matrix against one quantised activation vector: returns the row sum exactly as orc_mul_mat does.
 * q: int8 codes [K], dq / sq: per-block d and s (f32, fp16-rounded), xsum: per-block integer code sums. */
float orc_fast_row_q(int wtype, const uint8_t * row, const int8_t * q, const float * dq, const float * sq, const int32_t * xsum, int64_t nb) {
    return have_vnni() ? row_q_vnni(wtype, row, q, dq, sq, xsum, nb) : row_q_avx2(wtype, row, q, dq, sq, xsum, nb);
}

/* One F16 / F32 row: ggml's 32 partial sums (k mod 32) as four 8-lane accumulators, then the 16 / 8 / 4 / (0+1)+(2+3) fold. */
float orc_fast_row_f(int wtype, const uint8_t * row, const float * x, int64_t K) {
    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    if (wtype == ORC_F16) {
        for (int64_t k = 0; k < K; k += 32) {
            a0 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (row + 2 * k))), _mm256_loadu_ps(x + k), a0);
            a1 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (row + 2 * k + 16))), _mm256_loadu_ps(x + k + 8), a1);
            a2 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (row + 2 * k + 32))), _mm256_loadu_ps(x + k + 16), a2);
            a3 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (row + 2 * k + 48))), _mm256_loadu_ps(x + k + 24), a3);
        }
    } else {
        const float * w = (const float *) row;
        for (int64_t k = 0; k < K; k += 32) {
            a0 = _mm256_fmadd_ps(_mm256_loadu_ps(w + k), _mm256_loadu_ps(x + k), a0);
            a1 = _mm256_fmadd_ps(_mm256_loadu_ps(w + k + 8), _mm256_loadu_ps(x + k + 8), a1);
            a2 = _mm256_fmadd_ps(_mm256_loadu_ps(w + k + 16), _mm256_loadu_ps(x + k + 16), a2);
            a3 = _mm256_fmadd_ps(_mm256_loadu_ps(w + k + 24), _mm256_loadu_ps(x + k + 24), a3);
        }
    }
    /* ps[i] += ps[i+16]: (a0,a1) += (a2,a3); ps[i] += ps[i+8]: a0 += a1; ps[i] += ps[i+4]; (p0+p1)+(p2+p3) */
    a0 = _mm256_add_ps(a0, a2); a1 = _mm256_add_ps(a1, a3);
    a0 = _mm256_add_ps(a0, a1);
    const __m128 s4 = _mm_add_ps(_mm256_castps256_ps128(a0), _mm256_extractf128_ps(a0, 1));
    float p[4];
    _mm_storeu_ps(p, s4);
    return (p[0] + p[1]) + (p[2] + p[3]);
}
