/*
 * rwkv_oracle.c -- CPU ORACLE (test infrastructure, see rwkv_oracle.h). Plain C11 + optional OpenMP.
 *
 * Every function cites the reference file:line it restates (paths relative to RWKV/rwkv.cpp @ 2025-02-19).
 * ggml itself is absent from the reference mount; its op semantics are restated from SURVEY.md Appendix A
 * (published ggml algorithms: block formats, reference quantisers, CPU mul_mat with Q8_0/Q8_1 activations,
 * ggml_norm, ggml_rwkv_wkv6) and pinned by the reference's golden vectors (tests/test_oracle_*.py).
 */
#define _FILE_OFFSET_BITS 64
#include "rwkv_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <unistd.h>

#ifdef _OPENMP
#include <omp.h>
#endif
#ifdef __F16C__
#include <immintrin.h>
#endif

/* ------------------------------------------------------------------------------------------------ */
/* fp16 <-> fp32 (IEEE binary16, round-to-nearest-even; ggml's GGML_FP32_TO_FP16 / F16C semantics)    */
/* ------------------------------------------------------------------------------------------------ */

uint16_t orc_f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u | ((absx >> 13) & 0x3ffu) : 0u));
    }
    if (absx >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (absx < 0x33000001u) { /* < 2^-25 (or == 2^-25 tie to even 0) -> 0 */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(absx >> 23) - 127;
    uint32_t m = (absx & 0x7fffffu) | 0x800000u; /* 24-bit significand */
    uint32_t shift;
    uint32_t hexp;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t hm = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1u))) hm++;
    /* hm may carry into the exponent: normal case hm has implicit bit at 0x400 */
    uint32_t h;
    if (hexp == 0) {
        h = hm; /* carry into 0x400 makes it the smallest normal, which is correct */
    } else {
        h = ((hexp - 1) << 10) + hm; /* hm includes the implicit 1 at bit 10 */
    }
    return (uint16_t)(sign | h);
}

static inline float f16_to_f32_sw(uint16_t h);
float orc_f16_to_f32(uint16_t h) { return f16_to_f32_sw(h); }

/* hot-loop conversion: hardware F16C when available (bit-identical to the software routine) */
static inline float h2f(uint16_t h) {
#ifdef __F16C__
    return _cvtsh_ss(h);
#else
    return f16_to_f32_sw(h);
#endif
}

static inline float f16_to_f32_sw(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else { /* subnormal */
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; sh++; }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 127 - 15) << 23) | (m << 13);
    }
    float f; memcpy(&f, &x, 4);
    return f;
}

/* ------------------------------------------------------------------------------------------------ */
/* Deterministic arithmetic (DESIGN.md "Numerics"). The reference calls the platform's expf / tanhf and leaves */
/* fma contraction and reduction order to the compiler, so its float results are only reproducible on one      */
/* platform; several of its recorded thresholds sit on single rounding ties (a 1-ulp change upstream flips one  */
/* int8 activation code and moves the 5v1 Q8_0 sum from +0.585 to +1.351). Oracle and GPU kernels therefore     */
/* implement ONE spelled-out arithmetic: every fused multiply-add is an explicit fmaf (both sides are compiled  */
/* with -ffp-contract=off), reductions use fixed trees, and exp/tanh are the double-precision routines below          */
/* (correctly rounded to float but for ~1e-8 of arguments). GPU == oracle bit for bit follows.                  */
/* ------------------------------------------------------------------------------------------------ */

/* exp in double (argument reduction by ln2 hi/lo, degree-13 Taylor polynomial, fma Horner), rounded once to float: the
 * float result is the correctly rounded expf(x) except for a ~1e-8 fraction of arguments, i.e. it agrees with a good
 * libm while being reproducible bit for bit on CPU and GPU (only IEEE double fma / mul / add / rint / ldexp). */
static inline double det_exp_d(double x) {
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;          /* 1/13! */
    p = fma(p, r, 2.08767569878681e-09);        /* 1/12! */
    p = fma(p, r, 2.505210838544172e-08);       /* 1/11! */
    p = fma(p, r, 2.755731922398589e-07);       /* 1/10! */
    p = fma(p, r, 2.7557319223985893e-06);      /* 1/9!  */
    p = fma(p, r, 2.48015873015873e-05);        /* 1/8!  */
    p = fma(p, r, 1.984126984126984e-04);       /* 1/7!  */
    p = fma(p, r, 1.388888888888889e-03);       /* 1/6!  */
    p = fma(p, r, 8.333333333333333e-03);       /* 1/5!  */
    p = fma(p, r, 4.1666666666666664e-02);      /* 1/4!  */
    p = fma(p, r, 1.6666666666666666e-01);      /* 1/3!  */
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int) n);
}

static inline float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -103.97208404541016f) return 0.0f;
    return (float) det_exp_d((double) x);
}

static inline float det_tanhf(float x) {
    if (x != x) return x;
    const double xd = (double) x;
    const double ax = fabs(xd);
    if (ax < 1e-4) return (float) (xd * fma(xd * xd, -1.0 / 3.0, 1.0));
    if (ax > 20.0) return x > 0.0f ? 1.0f : -1.0f;
    const double s = det_exp_d(ax + ax);
    const double t = 1.0 - 2.0 / (s + 1.0);
    return (float) (x > 0.0f ? t : -t);
}

/* the same scalar functions, exported so the tests can compare the GPU's against them */
void orc_unary(int op, const float * x, float * y, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        const float v = x[i];
        float r;
        switch (op) {
            case 0: r = det_expf(v); break;
            case 1: r = det_tanhf(v); break;
            case 2: r = 1.0f / (1.0f + det_expf(-v)); break;
            case 3: r = v / (1.0f + det_expf(-v)); break;
            case 4: r = det_expf(-det_expf(v)); break;
            case 5: r = det_expf((1.0f / (1.0f + det_expf(-v))) * -0.606531f); break;
            case 6: r = 1.0f / sqrtf(v + 1e-5f); break;
            default: r = v; break;
        }
        y[i] = r;
    }
}

/* halving-tree folds: for o = n/2 .. 1: p[i] += p[i + o] */
static inline double fold_d(double * p, int n) { for (int o = n / 2; o > 0; o >>= 1) for (int i = 0; i < o; i++) p[i] += p[i + o]; return p[0]; }
static inline float  fold_f(float * p, int n)  { for (int o = n / 2; o > 0; o >>= 1) for (int i = 0; i < o; i++) p[i] += p[i + o]; return p[0]; }

/* ------------------------------------------------------------------------------------------------ */
/* Block formats (ggml block_q4_0 / q4_1 / q5_0 / q5_1 / q8_0), 32 elements per block  */
/* ------------------------------------------------------------------------------------------------ */

#define QK 32

size_t orc_type_size(int type) {
    switch (type) {
        case ORC_F32: return 4; case ORC_F16: return 2;
        case ORC_Q4_0: return 18; case ORC_Q4_1: return 20;
        case ORC_Q5_0: return 22; case ORC_Q5_1: return 24; case ORC_Q8_0: return 34;
        default: return 0;
    }
}

int orc_block_size(int type) {
    switch (type) {
        case ORC_F32: case ORC_F16: return 1;
        case ORC_Q4_0: case ORC_Q4_1: case ORC_Q5_0: case ORC_Q5_1: case ORC_Q8_0: return QK;
        default: return 0;
    }
}

static inline uint16_t rd16(const uint8_t * p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t rd32(const uint8_t * p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline void wr16(uint8_t * p, uint16_t v) { memcpy(p, &v, 2); }
static inline void wr32(uint8_t * p, uint32_t v) { memcpy(p, &v, 4); }

#define MIN_(a, b) ((a) < (b) ? (a) : (b))

/* Reference quantisers: ggml quantize_row_qX_Y_ref, reached from rwkv_quantize.inc:149 (ggml_quantize_chunk). */
void orc_quantize_row(int type, const float * x, void * yv, int64_t n) {
    uint8_t * y = (uint8_t *) yv;
    const int64_t nb = n / QK;
    switch (type) {
    case ORC_F32: memcpy(y, x, (size_t) n * 4); return;
    case ORC_F16: for (int64_t i = 0; i < n; i++) wr16(y + 2 * i, orc_f32_to_f16(x[i])); return;
    case ORC_Q4_0:
        for (int64_t b = 0; b < nb; b++, x += QK, y += 18) {
            float amax = 0.0f, max = 0.0f;
            for (int j = 0; j < QK; j++) { const float v = x[j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
            const float d = max / -8;
            const float id = d ? 1.0f / d : 0.0f;
            wr16(y, orc_f32_to_f16(d));
            for (int j = 0; j < QK / 2; j++) {
                const float x0 = x[j] * id, x1 = x[QK / 2 + j] * id;
                const uint8_t xi0 = MIN_(15, (int8_t)(x0 + 8.5f));
                const uint8_t xi1 = MIN_(15, (int8_t)(x1 + 8.5f));
                y[2 + j] = (uint8_t)(xi0 | (xi1 << 4));
            }
        }
        return;
    case ORC_Q4_1:
        for (int64_t b = 0; b < nb; b++, x += QK, y += 20) {
            float min = INFINITY, max = -INFINITY;
            for (int j = 0; j < QK; j++) { const float v = x[j]; if (v < min) min = v; if (v > max) max = v; }
            const float d = (max - min) / 15;
            const float id = d ? 1.0f / d : 0.0f;
            wr16(y, orc_f32_to_f16(d)); wr16(y + 2, orc_f32_to_f16(min));
            for (int j = 0; j < QK / 2; j++) {
                const float x0 = (x[j] - min) * id, x1 = (x[QK / 2 + j] - min) * id;
                const uint8_t xi0 = MIN_(15, (int8_t)(x0 + 0.5f));
                const uint8_t xi1 = MIN_(15, (int8_t)(x1 + 0.5f));
                y[4 + j] = (uint8_t)(xi0 | (xi1 << 4));
            }
        }
        return;
    case ORC_Q5_0:
        for (int64_t b = 0; b < nb; b++, x += QK, y += 22) {
            float amax = 0.0f, max = 0.0f;
            for (int j = 0; j < QK; j++) { const float v = x[j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
            const float d = max / -16;
            const float id = d ? 1.0f / d : 0.0f;
            wr16(y, orc_f32_to_f16(d));
            uint32_t qh = 0;
            for (int j = 0; j < QK / 2; j++) {
                const float x0 = x[j] * id, x1 = x[QK / 2 + j] * id;
                const uint8_t xi0 = MIN_(31, (int8_t)(x0 + 16.5f));
                const uint8_t xi1 = MIN_(31, (int8_t)(x1 + 16.5f));
                y[6 + j] = (uint8_t)((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
                qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
                qh |= ((xi1 & 0x10u) >> 4) << (j + QK / 2);
            }
            wr32(y + 2, qh);
        }
        return;
    case ORC_Q5_1:
        for (int64_t b = 0; b < nb; b++, x += QK, y += 24) {
            float min = INFINITY, max = -INFINITY;
            for (int j = 0; j < QK; j++) { const float v = x[j]; if (v < min) min = v; if (v > max) max = v; }
            const float d = (max - min) / 31;
            const float id = d ? 1.0f / d : 0.0f;
            wr16(y, orc_f32_to_f16(d)); wr16(y + 2, orc_f32_to_f16(min));
            uint32_t qh = 0;
            for (int j = 0; j < QK / 2; j++) {
                const float x0 = (x[j] - min) * id, x1 = (x[QK / 2 + j] - min) * id;
                const uint8_t xi0 = (uint8_t)(x0 + 0.5f);
                const uint8_t xi1 = (uint8_t)(x1 + 0.5f);
                y[8 + j] = (uint8_t)((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
                qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
                qh |= ((xi1 & 0x10u) >> 4) << (j + QK / 2);
            }
            wr32(y + 4, qh);
        }
        return;
    case ORC_Q8_0:
        for (int64_t b = 0; b < nb; b++, x += QK, y += 34) {
            float amax = 0.0f;
            for (int j = 0; j < QK; j++) { const float v = fabsf(x[j]); if (v > amax) amax = v; }
            const float d = amax / 127;
            const float id = d ? 1.0f / d : 0.0f;
            wr16(y, orc_f32_to_f16(d));
            for (int j = 0; j < QK; j++) ((int8_t *) y)[2 + j] = (int8_t) roundf(x[j] * id);
        }
        return;
    default: return;
    }
}

/* Decode one block into signed integer codes q[32] plus (d, m): value = q*d + m. */
static inline void block_codes(int type, const uint8_t * blk, int8_t * q, float * d, float * m) {
    switch (type) {
    case ORC_Q4_0:
        *d = orc_f16_to_f32(rd16(blk)); *m = 0.0f;
        for (int j = 0; j < 16; j++) { q[j] = (int8_t)((blk[2 + j] & 0x0F) - 8); q[16 + j] = (int8_t)((blk[2 + j] >> 4) - 8); }
        break;
    case ORC_Q4_1:
        *d = orc_f16_to_f32(rd16(blk)); *m = orc_f16_to_f32(rd16(blk + 2));
        for (int j = 0; j < 16; j++) { q[j] = (int8_t)(blk[4 + j] & 0x0F); q[16 + j] = (int8_t)(blk[4 + j] >> 4); }
        break;
    case ORC_Q5_0: {
        *d = orc_f16_to_f32(rd16(blk)); *m = 0.0f;
        const uint32_t qh = rd32(blk + 2);
        for (int j = 0; j < 16; j++) {
            const int h0 = (int)((qh >> j) & 1u) << 4, h1 = (int)((qh >> (j + 16)) & 1u) << 4;
            q[j] = (int8_t)(((blk[6 + j] & 0x0F) | h0) - 16); q[16 + j] = (int8_t)(((blk[6 + j] >> 4) | h1) - 16);
        }
        break; }
    case ORC_Q5_1: {
        *d = orc_f16_to_f32(rd16(blk)); *m = orc_f16_to_f32(rd16(blk + 2));
        const uint32_t qh = rd32(blk + 4);
        for (int j = 0; j < 16; j++) {
            const int h0 = (int)((qh >> j) & 1u) << 4, h1 = (int)((qh >> (j + 16)) & 1u) << 4;
            q[j] = (int8_t)((blk[8 + j] & 0x0F) | h0); q[16 + j] = (int8_t)((blk[8 + j] >> 4) | h1);
        }
        break; }
    case ORC_Q8_0:
        *d = orc_f16_to_f32(rd16(blk)); *m = 0.0f;
        memcpy(q, blk + 2, 32);
        break;
    default: *d = 0; *m = 0; memset(q, 0, 32); break;
    }
}

void orc_dequantize_row(int type, const void * xv, float * y, int64_t n) {
    const uint8_t * x = (const uint8_t *) xv;
    if (type == ORC_F32) { memcpy(y, x, (size_t) n * 4); return; }
    if (type == ORC_F16) { for (int64_t i = 0; i < n; i++) y[i] = orc_f16_to_f32(rd16(x + 2 * i)); return; }
    const size_t ts = orc_type_size(type);
    for (int64_t b = 0; b < n / QK; b++) {
        int8_t q[32]; float d, m;
        block_codes(type, x + b * ts, q, &d, &m);
        /* ggml dequantize_row: Q4_0/Q5_0/Q8_0: q*d ; Q4_1/Q5_1: q*d + m */
        for (int j = 0; j < 32; j++) y[b * QK + j] = (type == ORC_Q4_1 || type == ORC_Q5_1) ? q[j] * d + m : q[j] * d;
    }
}

/* Activation quantisation inside ggml's CPU mul_mat: quantize_row_q8_0 / q8_1. */
void orc_quantize_act(const float * x, int64_t n, int8_t * q, float * dq, float * sq) {
    for (int64_t b = 0; b < n / QK; b++) {
        float amax = 0.0f;
        for (int j = 0; j < QK; j++) { const float v = fabsf(x[b * QK + j]); if (v > amax) amax = v; }
        const float d = amax / 127;
        const float id = d ? 1.0f / d : 0.0f;
        int sum = 0;
        for (int j = 0; j < QK; j++) { const int8_t v = (int8_t) roundf(x[b * QK + j] * id); q[b * QK + j] = v; sum += v; }
        dq[b] = orc_f16_to_f32(orc_f32_to_f16(d));
        if (sq) sq[b] = orc_f16_to_f32(orc_f32_to_f16((float) sum * d));
    }
}

/* Knob for experiments only: 0 = keep activations f32 for F16 weights; 1 (default) = ggml's behaviour. */
static int g_f16_round_act = 1;
void orc_set_f16_act_rounding(int on) { g_f16_round_act = on; }
/* SIMD row kernels (rwkv_oracle_fast.c): bit-identical, used by bench.py's cpu_baseline leg; off by default */
static int g_fast = 0;
void orc_set_fast(int on) { g_fast = on; }
float orc_fast_row_q(int wtype, const uint8_t * row, const int8_t * q, const float * dq, const float * sq, const int32_t * xsum, int64_t nb);
float orc_fast_row_f(int wtype, const uint8_t * row, const float * x, int64_t K);
static int g_threads = 0;
void orc_set_threads(int n) {
    g_threads = n;
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}

/* f32 dot in the order of ggml's AVX2 ggml_vec_dot_f32 / _f16: 32 partial sums (k mod 32) accumulated with fma,
 * then folded 16, 8, 4 and (0+1)+(2+3). (The order is not observable in the reference's tests; it is fixed here
 * so the oracle is deterministic and documented.) */
static inline float dot32_finish(float * ps) {
    for (int i = 0; i < 16; i++) ps[i] += ps[i + 16];
    for (int i = 0; i < 8; i++) ps[i] += ps[i + 8];
    for (int i = 0; i < 4; i++) ps[i] += ps[i + 4];
    return (ps[0] + ps[1]) + (ps[2] + ps[3]);
}

/* ggml CPU mul_mat (call sites: every ggml_mul_mat in rwkv_graph.inc).
 *   F32 W: f32 dot.
 *   F16 W: activations are rounded to fp16 first (ggml converts src1 to the weight's vec_dot_type), products and
 *          accumulation in f32.  This is what reproduces the reference's recorded 7v0 FP16->Qx sums to 6 digits
 *          (tests/test_tiny_rwkv.c:128-133); see DESIGN.md "F16 weights".
 *   Q4_0/Q5_0/Q8_0: x -> Q8_0; y = sum_b (d_w*d_x) * isum.   Q4_1/Q5_1: x -> Q8_1; + m_w * s_x.
 *   f32 accumulation over blocks: 64 interleaved partial sums + halving tree (DESIGN.md "Numerics"). */
void orc_mul_mat(int wtype, const void * Wv, int64_t K, int64_t N, const float * x, int64_t T, float * y) {
    const uint8_t * W = (const uint8_t *) Wv;
    if (wtype == ORC_F32 || wtype == ORC_F16) {
        float * xr = NULL;
        if (wtype == ORC_F16 && g_f16_round_act) xr = (float *) malloc((size_t) K * sizeof(float));
        for (int64_t t = 0; t < T; t++) {
            const float * xt = x + t * K;
            if (xr) { for (int64_t k = 0; k < K; k++) xr[k] = orc_f16_to_f32(orc_f32_to_f16(xt[k])); xt = xr; }
            #pragma omp parallel for schedule(static) if (N * K > 65536)
            for (int64_t n = 0; n < N; n++) {
                if (g_fast && K % 32 == 0) { y[t * N + n] = orc_fast_row_f(wtype, W + (size_t) n * K * (wtype == ORC_F32 ? 4 : 2), xt, K); continue; }
                float ps[32];
                for (int i = 0; i < 32; i++) ps[i] = 0.0f;
                if (wtype == ORC_F32) {
                    const float * w = (const float *)(W + (size_t) n * K * 4);
                    for (int64_t k = 0; k < K; k++) ps[k & 31] = fmaf(w[k], xt[k], ps[k & 31]);
                } else {
                    const uint8_t * w = W + (size_t) n * K * 2;
                    for (int64_t k = 0; k < K; k++) ps[k & 31] = fmaf(h2f(rd16(w + 2 * k)), xt[k], ps[k & 31]);
                }
                y[t * N + n] = dot32_finish(ps);
            }
        }
        free(xr);
        return;
    }
    const int64_t nb = K / QK;
    const size_t ts = orc_type_size(wtype);
    const int has_m = (wtype == ORC_Q4_1 || wtype == ORC_Q5_1);
    int8_t * q = (int8_t *) malloc((size_t) K);
    float * dq = (float *) malloc((size_t) nb * sizeof(float) * 2);
    float * sq = dq + nb;
    int32_t * xsum = g_fast ? (int32_t *) malloc((size_t) nb * sizeof(int32_t)) : NULL;
    for (int64_t t = 0; t < T; t++) {
        orc_quantize_act(x + t * K, K, q, dq, sq);
        if (xsum) for (int64_t b = 0; b < nb; b++) { int32_t sm = 0; for (int j = 0; j < QK; j++) sm += q[b * QK + j]; xsum[b] = sm; }
        #pragma omp parallel for schedule(static) if (N * K > 65536)
        for (int64_t n = 0; n < N; n++) {
            const uint8_t * row = W + (size_t) n * nb * ts;
            if (xsum) { y[t * N + n] = orc_fast_row_q(wtype, row, q, dq, sq, xsum, nb); continue; }
            float lanes[64];
            for (int i = 0; i < 64; i++) lanes[i] = 0.0f;
            for (int64_t b = 0; b < nb; b++) {
                const uint8_t * blk = row + b * ts;
                const int8_t * qx = q + b * QK;
                int32_t isum = 0;
                float d, m = 0.0f;
                switch (wtype) {
                case ORC_Q4_0: {
                    d = h2f(rd16(blk));
                    int32_t s0 = 0, s1 = 0;
                    for (int j = 0; j < 16; j++) { s0 += ((blk[2 + j] & 0x0F) - 8) * qx[j]; s1 += ((blk[2 + j] >> 4) - 8) * qx[16 + j]; }
                    isum = s0 + s1; break; }
                case ORC_Q8_0: {
                    d = h2f(rd16(blk));
                    const int8_t * qw = (const int8_t *)(blk + 2);
                    for (int j = 0; j < 32; j++) isum += qw[j] * qx[j];
                    break; }
                default: {
                    int8_t qw[32];
                    block_codes(wtype, blk, qw, &d, &m);
                    for (int j = 0; j < 32; j++) isum += qw[j] * qx[j];
                    break; }
                }
                /* 64 partial sums (block b -> partial b mod 64, increasing b), fma per block, halving-tree fold */
                float a = lanes[b & 63];
                a = fmaf(d * dq[b], (float) isum, a);
                if (has_m) a = fmaf(m, sq[b], a);
                lanes[b & 63] = a;
            }
            y[t * N + n] = fold_f(lanes, 64);
        }
    }
    free(q); free(dq); free(xsum);
}

/* ------------------------------------------------------------------------------------------------ */
/* Model file (rwkv_file_format.inc:102-197, docs/FILE_FORMAT.md) and parameter table                  */
/* (rwkv_model_loading.inc:127-285)                                                                    */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
    char name[96];
    int type;
    int ndim;
    int64_t ne[3];
    const uint8_t * data;
    size_t nbytes;
} orc_tensor;

typedef struct {
    const orc_tensor *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    /* v4/v5 */
    const orc_tensor *att_mix_k, *att_mix_v, *att_mix_r, *att_mix_g, *att_time_first, *att_time_decay, *att_time_faaaa;
    const orc_tensor *att_key, *att_value, *att_receptance, *att_output, *att_gate, *att_ln_x_w, *att_ln_x_b;
    /* v6 */
    const orc_tensor *maa_x, *maa_w, *maa_k, *maa_v, *maa_r, *maa_g, *maa_w1, *maa_w2, *decay_w1, *decay_w2;
    /* v7 */
    const orc_tensor *x_rwkvag, *w0, *w1, *w2, *a0, *a1, *a2, *g1, *g2, *v0, *v1, *v2, *r_k, *k_k, *k_a;
    /* ffn */
    const orc_tensor *ffn_mix_k, *ffn_mix_r, *ffn_maa_k, *ffn_maa_r, *ffn_x_k, *ffn_key, *ffn_value, *ffn_receptance;
} orc_layer;

struct orc_model {
    uint8_t * blob; size_t blob_size;
    uint32_t version, n_vocab, n_embed, n_layer, data_type;
    int arch_major, arch_minor;
    int64_t head_count, head_size, ffn_size;
    orc_tensor * tensors; int n_tensors;
    const orc_tensor *emb, *ln0_w, *ln0_b, *ln_out_w, *ln_out_b, *head;
    orc_layer * layers;
    float * scratch; /* work vectors */
};

static const orc_tensor * find_tensor(const orc_model * m, const char * name) {
    for (int i = 0; i < m->n_tensors; i++) if (strcmp(m->tensors[i].name, name) == 0) return &m->tensors[i];
    return NULL;
}

static const orc_tensor * need(const orc_model * m, int layer, const char * suffix, int * ok) {
    char buf[128];
    if (layer >= 0) snprintf(buf, sizeof buf, "blocks.%d.%s", layer, suffix); else snprintf(buf, sizeof buf, "%s", suffix);
    const orc_tensor * t = find_tensor(m, buf);
    if (!t) { fprintf(stderr, "oracle: parameter %s not found\n", buf); *ok = 0; }
    return t;
}

orc_model * orc_load(const char * path) {
    FILE * f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "oracle: cannot open %s\n", path); return NULL; }
    struct stat st;
    if (fstat(fileno(f), &st) != 0) { fclose(f); return NULL; }
    orc_model * m = (orc_model *) calloc(1, sizeof *m);
    m->blob_size = (size_t) st.st_size;
    /* map the file: multi-GB benchmark models are read straight from the page cache */
    void * map = mmap(NULL, m->blob_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
    fclose(f);
    if (map == MAP_FAILED) { free(m); return NULL; }
    m->blob = (uint8_t *) map;
    const uint8_t * p = m->blob, * end = m->blob + m->blob_size;
    if (m->blob_size < 24 || rd32(p) != 0x67676d66u) { fprintf(stderr, "oracle: bad magic\n"); orc_free(m); return NULL; }
    m->version = rd32(p + 4); m->n_vocab = rd32(p + 8); m->n_embed = rd32(p + 12); m->n_layer = rd32(p + 16); m->data_type = rd32(p + 20);
    if (m->version < 100 || m->version > 101) { orc_free(m); return NULL; }
    p += 24;
    int cap = 64; m->tensors = (orc_tensor *) calloc((size_t) cap, sizeof(orc_tensor));
    while (p < end) {
        if (p + 12 > end) { orc_free(m); return NULL; }
        const uint32_t dc = rd32(p), kl = rd32(p + 4), ty = rd32(p + 8); p += 12;
        if (dc < 1 || dc > 3 || orc_type_size((int) ty) == 0 || kl >= 96) { fprintf(stderr, "oracle: bad tensor header\n"); orc_free(m); return NULL; }
        if (m->n_tensors == cap) { cap *= 2; m->tensors = (orc_tensor *) realloc(m->tensors, (size_t) cap * sizeof(orc_tensor)); }
        orc_tensor * t = &m->tensors[m->n_tensors++];
        memset(t, 0, sizeof *t);
        t->type = (int) ty; t->ndim = (int) dc; t->ne[0] = t->ne[1] = t->ne[2] = 1;
        for (uint32_t i = 0; i < dc; i++) { t->ne[i] = rd32(p); p += 4; }
        memcpy(t->name, p, kl); t->name[kl] = 0; p += kl;
        /* rwkv_utilities.inc:1-3: nbytes = type_size * n / blck */
        t->nbytes = orc_type_size(t->type) * (size_t)(t->ne[0] * t->ne[1] * t->ne[2]) / (size_t) orc_block_size(t->type);
        t->data = p; p += t->nbytes;
        if (p > end) { fprintf(stderr, "oracle: truncated tensor %s\n", t->name); orc_free(m); return NULL; }
    }
    /* architecture detection: rwkv_model_loading.inc:319-340 */
    m->arch_major = 4; m->arch_minor = 0;
    if (find_tensor(m, "blocks.0.att.ln_x.weight")) { m->arch_major = 5; m->arch_minor = find_tensor(m, "blocks.0.att.gate.weight") ? 2 : 1; }
    if (find_tensor(m, "blocks.0.att.time_maa_x")) { m->arch_major = 6; m->arch_minor = 0; }
    if (find_tensor(m, "blocks.0.att.r_k")) { m->arch_major = 7; m->arch_minor = 0; }

    int ok = 1;
    m->emb = need(m, -1, "emb.weight", &ok);
    m->ln0_w = need(m, 0, "ln0.weight", &ok); m->ln0_b = need(m, 0, "ln0.bias", &ok);
    m->ln_out_w = need(m, -1, "ln_out.weight", &ok); m->ln_out_b = need(m, -1, "ln_out.bias", &ok);
    m->head = need(m, -1, "head.weight", &ok);
    m->layers = (orc_layer *) calloc(m->n_layer, sizeof(orc_layer));
    for (int i = 0; i < (int) m->n_layer; i++) {
        orc_layer * L = &m->layers[i];
        L->ln1_w = need(m, i, "ln1.weight", &ok); L->ln1_b = need(m, i, "ln1.bias", &ok);
        L->ln2_w = need(m, i, "ln2.weight", &ok); L->ln2_b = need(m, i, "ln2.bias", &ok);
        L->att_key = need(m, i, "att.key.weight", &ok); L->att_value = need(m, i, "att.value.weight", &ok);
        L->att_receptance = need(m, i, "att.receptance.weight", &ok); L->att_output = need(m, i, "att.output.weight", &ok);
        L->ffn_key = need(m, i, "ffn.key.weight", &ok); L->ffn_value = need(m, i, "ffn.value.weight", &ok);
        if (m->arch_major != 7) L->ffn_receptance = need(m, i, "ffn.receptance.weight", &ok);
        switch (m->arch_major) {
        case 4:
            L->att_mix_k = need(m, i, "att.time_mix_k", &ok); L->att_mix_v = need(m, i, "att.time_mix_v", &ok); L->att_mix_r = need(m, i, "att.time_mix_r", &ok);
            L->att_time_first = need(m, i, "att.time_first", &ok); L->att_time_decay = need(m, i, "att.time_decay", &ok);
            L->ffn_mix_k = need(m, i, "ffn.time_mix_k", &ok); L->ffn_mix_r = need(m, i, "ffn.time_mix_r", &ok);
            break;
        case 5:
            L->att_mix_k = need(m, i, "att.time_mix_k", &ok); L->att_mix_v = need(m, i, "att.time_mix_v", &ok); L->att_mix_r = need(m, i, "att.time_mix_r", &ok);
            if (m->arch_minor >= 2) {
                L->att_time_faaaa = need(m, i, "att.time_faaaa", &ok); L->att_mix_g = need(m, i, "att.time_mix_g", &ok); L->att_gate = need(m, i, "att.gate.weight", &ok);
            } else {
                L->att_time_first = need(m, i, "att.time_first", &ok);
            }
            L->att_time_decay = need(m, i, "att.time_decay", &ok);
            L->att_ln_x_w = need(m, i, "att.ln_x.weight", &ok); L->att_ln_x_b = need(m, i, "att.ln_x.bias", &ok);
            L->ffn_mix_k = need(m, i, "ffn.time_mix_k", &ok); L->ffn_mix_r = need(m, i, "ffn.time_mix_r", &ok);
            break;
        case 6:
            L->maa_x = need(m, i, "att.time_maa_x", &ok); L->maa_w = need(m, i, "att.time_maa_w", &ok); L->maa_k = need(m, i, "att.time_maa_k", &ok);
            L->maa_v = need(m, i, "att.time_maa_v", &ok); L->maa_r = need(m, i, "att.time_maa_r", &ok); L->maa_g = need(m, i, "att.time_maa_g", &ok);
            L->maa_w1 = need(m, i, "att.time_maa_w1", &ok); L->maa_w2 = need(m, i, "att.time_maa_w2", &ok);
            L->att_time_faaaa = need(m, i, "att.time_faaaa", &ok); L->att_time_decay = need(m, i, "att.time_decay", &ok);
            L->decay_w1 = need(m, i, "att.time_decay_w1", &ok); L->decay_w2 = need(m, i, "att.time_decay_w2", &ok);
            L->att_gate = need(m, i, "att.gate.weight", &ok);
            L->att_ln_x_w = need(m, i, "att.ln_x.weight", &ok); L->att_ln_x_b = need(m, i, "att.ln_x.bias", &ok);
            L->ffn_maa_k = need(m, i, "ffn.time_maa_k", &ok); L->ffn_maa_r = need(m, i, "ffn.time_maa_r", &ok);
            break;
        case 7:
            L->x_rwkvag = need(m, i, "att.x_rwkvag", &ok);
            L->w0 = need(m, i, "att.w0", &ok); L->w1 = need(m, i, "att.w1", &ok); L->w2 = need(m, i, "att.w2", &ok);
            L->a0 = need(m, i, "att.a0", &ok); L->a1 = need(m, i, "att.a1", &ok); L->a2 = need(m, i, "att.a2", &ok);
            L->g1 = need(m, i, "att.g1", &ok); L->g2 = need(m, i, "att.g2", &ok);
            if (i != 0) { L->v0 = need(m, i, "att.v0", &ok); L->v1 = need(m, i, "att.v1", &ok); L->v2 = need(m, i, "att.v2", &ok); }
            L->r_k = need(m, i, "att.r_k", &ok); L->k_k = need(m, i, "att.k_k", &ok); L->k_a = need(m, i, "att.k_a", &ok);
            L->att_ln_x_w = need(m, i, "att.ln_x.weight", &ok); L->att_ln_x_b = need(m, i, "att.ln_x.bias", &ok);
            L->ffn_x_k = need(m, i, "ffn.x_k", &ok);
            break;
        }
        if (!ok) break;
    }
    if (!ok) { orc_free(m); return NULL; }
    /* head_count / head_size: rwkv_model_loading.inc:403-409 */
    if (m->arch_major == 7) { m->head_count = m->layers[0].r_k->ne[1]; m->head_size = m->n_embed / m->head_count; }
    else if (m->arch_major >= 5) { m->head_count = m->layers[0].att_time_decay->ne[2]; m->head_size = m->n_embed / m->head_count; }
    m->ffn_size = m->layers[0].ffn_key->ne[1];
    /* emb shape check: rwkv_model_loading.inc:411-416 */
    if (m->emb->ndim != 2 || m->emb->ne[0] != m->n_embed || m->emb->ne[1] != m->n_vocab) { fprintf(stderr, "oracle: bad emb shape\n"); orc_free(m); return NULL; }
    return m;
}

void orc_free(orc_model * m) {
    if (!m) return;
    if (m->blob) munmap(m->blob, m->blob_size);
    free(m->tensors); free(m->layers); free(m->scratch); free(m);
}

void orc_info(const orc_model * m, int64_t * info) {
    info[0] = m->arch_major; info[1] = m->arch_minor; info[2] = m->n_vocab; info[3] = m->n_embed; info[4] = m->n_layer;
    info[5] = m->head_count; info[6] = m->head_size; info[7] = m->data_type; info[8] = m->version; info[9] = m->ffn_size;
}

/* rwkv.cpp:171-179 */
size_t orc_state_len(const orc_model * m) {
    if (m->arch_major >= 5) return (size_t) m->n_embed * (2 + (size_t) m->head_size) * m->n_layer;
    return (size_t) m->n_embed * 5 * m->n_layer;
}

uint64_t orc_bytes_per_token(const orc_model * m) {
    uint64_t b = 0;
    for (int i = 0; i < m->n_tensors; i++) b += m->tensors[i].nbytes;
    b -= m->emb->nbytes;
    b += (uint64_t) m->n_embed * orc_type_size(m->emb->type);
    b += 2ull * orc_state_len(m) * 4ull + (uint64_t) m->n_vocab * 4ull;
    return b;
}

/* rwkv_eval.inc:224-241 */
void orc_init_state(const orc_model * m, float * state) {
    const size_t n = orc_state_len(m);
    memset(state, 0, n * sizeof(float));
    if (m->arch_major >= 5) return;
    const size_t D = m->n_embed;
    for (size_t l = 0; l < m->n_layer; l++) for (size_t i = 0; i < D; i++) state[l * 5 * D + 4 * D + i] = -1e30f;
}

/* ------------------------------------------------------------------------------------------------ */
/* Elementwise pieces                                                                  */
/* ------------------------------------------------------------------------------------------------ */

static const float * f32data(const orc_tensor * t) { return (const float *) t->data; }

/* ggml_norm: mean and sum of squared deviations accumulated in double, scale = 1/sqrtf(var + eps).
 * Reduction order: NP partial sums (element i goes to partial i mod NP), folded by a halving tree. NP = 256 for a
 * LayerNorm row (one partial per thread of the GPU workgroup), 64 for a per-head GroupNorm row (one per lane). */
static void norm_row(const float * x, float * y, int64_t n, int np, float eps) {
    double ps[256];
    for (int i = 0; i < np; i++) ps[i] = 0.0;
    for (int64_t i = 0; i < n; i++) ps[i % np] += (double) x[i];
    const float mean = (float)(fold_d(ps, np) / (double) n);
    for (int i = 0; i < np; i++) ps[i] = 0.0;
    for (int64_t i = 0; i < n; i++) { const float v = x[i] - mean; y[i] = v; ps[i % np] += (double)(v * v); }
    const float variance = (float)(fold_d(ps, np) / (double) n);
    const float scale = 1.0f / sqrtf(variance + eps);
    for (int64_t i = 0; i < n; i++) y[i] *= scale;
}

/* rwkv_operators.inc:93-97: norm(1e-5) * w + b -- three graph ops, three roundings (norm, mul, add) */
static void layer_norm(const float * x, const orc_tensor * w, const orc_tensor * b, float * y, int64_t n) {
    norm_row(x, y, n, 256, 1e-5f);
    const float * wd = f32data(w), * bd = f32data(b);
    for (int64_t i = 0; i < n; i++) y[i] = y[i] * wd[i] + bd[i];
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + det_expf(-x)); }
static inline float siluf_(float x) { return x / (1.0f + det_expf(-x)); }

static void mm(const orc_tensor * W, const float * x, float * y) {
    orc_mul_mat(W->type, W->data, W->ne[0], W->ne[1], x, 1, y);
}

/* v4/v5 lerp: x*mix + (x_prev - x_prev*mix)   (rwkv_graph.inc:94-97, 214-231, 490-501) */
static void lerp_v4(const float * x, const float * xp, const orc_tensor * mix, float * y, int64_t n) {
    const float * mk = f32data(mix);
    for (int64_t i = 0; i < n; i++) y[i] = x[i] * mk[i] + (xp[i] - xp[i] * mk[i]);
}

/* ggml_rwkv_wkv6 (call sites rwkv_graph.inc:275, 370), one token.
 * state[h][i][j], i = key index, j = value index. u indexed [h*S+i] (u_stride 1) or [h] (u_stride 0). */
static void wkv6_token(float * state, const float * r, const float * k, const float * v, const float * u, int u_per_chan,
                       const float * w, int w_per_chan, float * out, int64_t H, int64_t S) {
    for (int64_t h = 0; h < H; h++) {
        float * s = state + h * S * S;
        float * o = out + h * S;
        for (int64_t j = 0; j < S; j++) o[j] = 0.0f;
        for (int64_t i = 0; i < S; i++) {
            const float k_val = k[h * S + i], r_val = r[h * S + i];
            const float u_val = u_per_chan ? u[h * S + i] : u[h];
            const float w_val = w_per_chan ? w[h * S + i] : w[h];
            for (int64_t j = 0; j < S; j++) {
                const float kv = v[h * S + j] * k_val;
                const float prev = s[i * S + j];
                const float temp = kv * u_val + prev;
                o[j] += temp * r_val;
                s[i * S + j] = prev * w_val + kv;
            }
        }
    }
}

/* group norm over each head (ggml_norm on [S,H]) then *ln_x.w + ln_x.b (rwkv_graph.inc:280-285, 375-380, 465-470) */
static void group_norm(float * x, const orc_tensor * w, const orc_tensor * b, int64_t H, int64_t S, float eps) {
    const float * wd = f32data(w), * bd = f32data(b);
    for (int64_t h = 0; h < H; h++) {
        norm_row(x + h * S, x + h * S, S, 64, eps);
        for (int64_t j = 0; j < S; j++) x[h * S + j] = x[h * S + j] * wd[h * S + j] + bd[h * S + j];
    }
}

/* work buffers */
typedef struct {
    float *x, *xn, *xk, *xv, *xr, *xg, *xw, *xa, *r, *k, *v, *g, *w, *a, *kk, *tmp, *tmp2, *out, *ffk, *v_first, *sx;
} orc_work;

static float * carve(float ** p, size_t n) { float * r = *p; *p += n; return r; }

static void get_work(orc_model * m, orc_work * wk) {
    const size_t D = m->n_embed, F = (size_t) m->ffn_size;
    size_t lr = 0; /* widest low-rank intermediate */
    for (int i = 0; i < m->n_tensors; i++) if (m->tensors[i].ndim >= 2 && (size_t) m->tensors[i].ne[1] > lr && m->tensors[i].ne[1] < (int64_t) m->n_vocab) lr = (size_t) m->tensors[i].ne[1];
    if (lr < D) lr = D;
    if (lr < F) lr = F;
    const size_t total = 20 * D + 2 * lr + F + 5 * D + 64;
    if (!m->scratch) m->scratch = (float *) malloc(total * sizeof(float));
    float * p = m->scratch;
    wk->x = carve(&p, D); wk->xn = carve(&p, D); wk->xk = carve(&p, D); wk->xv = carve(&p, D); wk->xr = carve(&p, D);
    wk->xg = carve(&p, D); wk->xw = carve(&p, D); wk->xa = carve(&p, D); wk->r = carve(&p, D); wk->k = carve(&p, D);
    wk->v = carve(&p, D); wk->g = carve(&p, D); wk->w = carve(&p, D); wk->a = carve(&p, D); wk->kk = carve(&p, D);
    wk->out = carve(&p, D); wk->v_first = carve(&p, D); wk->sx = carve(&p, D); wk->tmp = carve(&p, lr); wk->tmp2 = carve(&p, lr);
    wk->ffk = carve(&p, F);
}

/* ------------------------------------------------------------------------------------------------ */
/* Per-architecture time mixing (rwkv_graph.inc:84-482) and channel mixing (:484-543), one token       */
/* ------------------------------------------------------------------------------------------------ */

/* rwkv_att_v4 (:163-197) with rwkv_att_rkv_v4 (:84-117) and rwkv_att_wkv_v4 (:119-161). st = layer state base. */
static void att_v4(orc_model * m, const orc_layer * L, orc_work * wk, float * st) {
    const int64_t D = m->n_embed;
    float * att_xx = st + D, * aa = st + 2 * D, * bb = st + 3 * D, * pp = st + 4 * D;
    layer_norm(wk->x, L->ln1_w, L->ln1_b, wk->xn, D);
    lerp_v4(wk->xn, att_xx, L->att_mix_k, wk->xk, D);
    lerp_v4(wk->xn, att_xx, L->att_mix_v, wk->xv, D);
    lerp_v4(wk->xn, att_xx, L->att_mix_r, wk->xr, D);
    memcpy(att_xx, wk->xn, (size_t) D * 4);
    mm(L->att_receptance, wk->xr, wk->r);
    mm(L->att_key, wk->xk, wk->k);
    mm(L->att_value, wk->xv, wk->v);
    const float * tf = f32data(L->att_time_first), * td = f32data(L->att_time_decay);
    for (int64_t i = 0; i < D; i++) {
        const float r = sigmoidf_(wk->r[i]);
        const float k = wk->k[i], v = wk->v[i];
        float ww = tf[i] + k;
        float qq = fmaxf(pp[i], ww);
        float e1 = det_expf(pp[i] - qq), e2 = det_expf(ww - qq);
        const float a = e1 * aa[i] + e2 * v;
        const float b = e1 * bb[i] + e2;
        ww = pp[i] + td[i];
        qq = fmaxf(ww, k);
        e1 = det_expf(ww - qq); e2 = det_expf(k - qq);
        aa[i] = e1 * aa[i] + e2 * v;
        bb[i] = e1 * bb[i] + e2;
        pp[i] = qq;
        wk->tmp[i] = r * (a / b);
    }
    mm(L->att_output, wk->tmp, wk->out);
    for (int64_t i = 0; i < D; i++) wk->x[i] += wk->out[i];
}

/* rwkv_att_v5 (:199-292) */
static void att_v5(orc_model * m, const orc_layer * L, orc_work * wk, float * st) {
    const int64_t D = m->n_embed, H = m->head_count, S = m->head_size;
    float * att_xx = st + D, * heads = st + 2 * D;
    layer_norm(wk->x, L->ln1_w, L->ln1_b, wk->xn, D);
    lerp_v4(wk->xn, att_xx, L->att_mix_k, wk->xk, D);
    lerp_v4(wk->xn, att_xx, L->att_mix_v, wk->xv, D);
    lerp_v4(wk->xn, att_xx, L->att_mix_r, wk->xr, D);
    if (m->arch_minor >= 2) lerp_v4(wk->xn, att_xx, L->att_mix_g, wk->xg, D);
    memcpy(att_xx, wk->xn, (size_t) D * 4);
    mm(L->att_receptance, wk->xr, wk->r);
    mm(L->att_key, wk->xk, wk->k);
    mm(L->att_value, wk->xv, wk->v);
    if (m->arch_minor >= 2) { mm(L->att_gate, wk->xg, wk->g); for (int64_t i = 0; i < D; i++) wk->g[i] = siluf_(wk->g[i]); }
    const int per_chan = m->arch_minor >= 2;
    const float * u = per_chan ? f32data(L->att_time_faaaa) : f32data(L->att_time_first);
    wkv6_token(heads, wk->r, wk->k, wk->v, u, per_chan, f32data(L->att_time_decay), per_chan, wk->tmp, H, S);
    group_norm(wk->tmp, L->att_ln_x_w, L->att_ln_x_b, H, S, 1e-5f);
    if (m->arch_minor >= 2) for (int64_t i = 0; i < D; i++) wk->tmp[i] *= wk->g[i];
    mm(L->att_output, wk->tmp, wk->out);
    for (int64_t i = 0; i < D; i++) wk->x[i] += wk->out[i];
}

/* rwkv_att_v6 (:294-385) */
static void att_v6(orc_model * m, const orc_layer * L, orc_work * wk, float * st) {
    const int64_t D = m->n_embed, H = m->head_count, S = m->head_size;
    float * att_xx = st + D, * heads = st + 2 * D;
    layer_norm(wk->x, L->ln1_w, L->ln1_b, wk->xn, D);
    const float * mx = f32data(L->maa_x);
    for (int64_t i = 0; i < D; i++) { wk->sx[i] = att_xx[i] - wk->xn[i]; wk->xa[i] = wk->sx[i] * mx[i] + wk->xn[i]; } /* xxx */
    memcpy(att_xx, wk->xn, (size_t) D * 4);
    const int64_t R5 = L->maa_w1->ne[1], R = R5 / 5;
    mm(L->maa_w1, wk->xa, wk->tmp);
    for (int64_t i = 0; i < R5; i++) wk->tmp[i] = det_tanhf(wk->tmp[i]);
    /* batched W2: time_maa_w2 ne = (R, D, 5), F32 x F32 (:326-334); slice order w,k,v,r,g (:336-340) */
    const float * w2 = f32data(L->maa_w2);
    const float * maa[5] = { f32data(L->maa_w), f32data(L->maa_k), f32data(L->maa_v), f32data(L->maa_r), f32data(L->maa_g) };
    float * dst[5] = { wk->xw, wk->xk, wk->xv, wk->xr, wk->xg };
    for (int f = 0; f < 5; f++) {
        for (int64_t d = 0; d < D; d++) {
            const float * row = w2 + ((size_t) f * D + d) * R;
            float acc = 0.0f;
            for (int64_t q = 0; q < R; q++) acc += row[q] * wk->tmp[f * R + q];
            dst[f][d] = (acc + maa[f][d]) * wk->sx[d] + wk->xn[d];
        }
    }
    mm(L->att_receptance, wk->xr, wk->r);
    mm(L->att_key, wk->xk, wk->k);
    mm(L->att_value, wk->xv, wk->v);
    mm(L->att_gate, wk->xg, wk->g);
    for (int64_t i = 0; i < D; i++) wk->g[i] = siluf_(wk->g[i]);
    const int64_t DR = L->decay_w1->ne[1];
    mm(L->decay_w1, wk->xw, wk->tmp);
    for (int64_t i = 0; i < DR; i++) wk->tmp[i] = det_tanhf(wk->tmp[i]);
    mm(L->decay_w2, wk->tmp, wk->w);
    const float * td = f32data(L->att_time_decay);
    for (int64_t i = 0; i < D; i++) wk->w[i] = det_expf(-det_expf(wk->w[i] + td[i]));
    wkv6_token(heads, wk->r, wk->k, wk->v, f32data(L->att_time_faaaa), 1, wk->w, 1, wk->tmp2, H, S);
    group_norm(wk->tmp2, L->att_ln_x_w, L->att_ln_x_b, H, S, 64e-5f);
    for (int64_t i = 0; i < D; i++) wk->tmp2[i] *= wk->g[i];
    mm(L->att_output, wk->tmp2, wk->out);
    for (int64_t i = 0; i < D; i++) wk->x[i] += wk->out[i];
}

/* rwkv_wkv_v7_impl (rwkv_operators_wkv_v7.inc:37-107), one token. state[h][i][j]: i = value, j = key. */
static void wkv7_token(float * state, const float * r, const float * w, const float * k, const float * v,
                       const float * a, const float * b, float * out, int64_t H, int64_t S) {
    for (int64_t h = 0; h < H; h++) {
        float * s = state + h * S * S;
        const float * rh = r + h * S, * wh = w + h * S, * kh = k + h * S, * vh = v + h * S, * ah = a + h * S, * bh = b + h * S;
        for (int64_t i = 0; i < S; i++) {
            const float v_val = vh[i];
            float sa = 0.0f;
            for (int64_t j = 0; j < S; j++) sa += ah[j] * s[i * S + j];
            float res = 0.0f;
            for (int64_t j = 0; j < S; j++) {
                const float kv = v_val * kh[j];
                const float ns = s[i * S + j] * wh[j] + kv + sa * bh[j];
                s[i * S + j] = ns;
                res += ns * rh[j];
            }
            out[h * S + i] = res;
        }
    }
}

/* rwkv_att_v7 (:387-482) */
static void att_v7(orc_model * m, const orc_layer * L, orc_work * wk, float * st, int layer_idx) {
    const int64_t D = m->n_embed, H = m->head_count, S = m->head_size;
    float * att_xx = st + D, * heads = st + 2 * D;
    layer_norm(wk->x, L->ln1_w, L->ln1_b, wk->xn, D);
    const float * xm = f32data(L->x_rwkvag); /* ne (D,1,6): order r,w,k,v,a,g (:408-413) */
    float * dst[6] = { wk->xr, wk->xw, wk->xk, wk->xv, wk->xa, wk->xg };
    for (int64_t i = 0; i < D; i++) {
        const float sx = att_xx[i] - wk->xn[i];
        for (int f = 0; f < 6; f++) dst[f][i] = sx * xm[f * D + i] + wk->xn[i];
    }
    memcpy(att_xx, wk->xn, (size_t) D * 4);
    mm(L->att_receptance, wk->xr, wk->r);
    /* g = G2 * sigmoid(G1 * xg) (:416) */
    mm(L->g1, wk->xg, wk->tmp);
    for (int64_t i = 0; i < L->g1->ne[1]; i++) wk->tmp[i] = sigmoidf_(wk->tmp[i]);
    mm(L->g2, wk->tmp, wk->g);
    /* a = sigmoid(A2 * (A1 * xa) + a0) (:417-423) */
    mm(L->a1, wk->xa, wk->tmp);
    mm(L->a2, wk->tmp, wk->a);
    { const float * a0 = f32data(L->a0); for (int64_t i = 0; i < D; i++) wk->a[i] = sigmoidf_(wk->a[i] + a0[i]); }
    /* w = exp(-0.606531 * sigmoid(W2 * tanh(W1 * xw) + w0)) (:425-430) */
    mm(L->w1, wk->xw, wk->tmp);
    for (int64_t i = 0; i < L->w1->ne[1]; i++) wk->tmp[i] = det_tanhf(wk->tmp[i]);
    mm(L->w2, wk->tmp, wk->w);
    { const float * w0 = f32data(L->w0); for (int64_t i = 0; i < D; i++) wk->w[i] = det_expf(sigmoidf_(wk->w[i] + w0[i]) * -0.606531f); }
    /* k, kk = l2norm_head(k * k_k), k += a*ka - ka (:432-437); l2norm: rwkv_operators.inc:40-82 */
    mm(L->att_key, wk->xk, wk->k);
    const float * k_k = f32data(L->k_k), * k_a = f32data(L->k_a);
    for (int64_t h = 0; h < H; h++) {
        float ps[64];
        for (int i = 0; i < 64; i++) ps[i] = 0.0f;
        for (int64_t j = 0; j < S; j++) { const float t = wk->k[h * S + j] * k_k[h * S + j]; wk->kk[h * S + j] = t; ps[j & 63] += t * t; }
        const float scale = 1.0f / fmaxf(sqrtf(fold_f(ps, 64)), 1e-12f);
        for (int64_t j = 0; j < S; j++) wk->kk[h * S + j] *= scale;
    }
    for (int64_t i = 0; i < D; i++) { const float ka = wk->k[i] * k_a[i]; wk->k[i] = wk->k[i] + (wk->a[i] * ka - ka); }
    /* v (+ v_first residual for layer > 0) (:439-453) */
    mm(L->att_value, wk->xv, wk->v);
    if (layer_idx == 0) {
        memcpy(wk->v_first, wk->v, (size_t) D * 4);
    } else {
        mm(L->v1, wk->xv, wk->tmp);
        mm(L->v2, wk->tmp, wk->tmp2);
        const float * v0 = f32data(L->v0);
        for (int64_t i = 0; i < D; i++) wk->v[i] = wk->v[i] + (wk->v_first[i] - wk->v[i]) * sigmoidf_(wk->tmp2[i] + v0[i]);
    }
    /* wkv7(state, r, w, k, v, -kk, kk*a) (:460) */
    for (int64_t i = 0; i < D; i++) { wk->tmp[i] = -wk->kk[i]; wk->tmp2[i] = wk->kk[i] * wk->a[i]; }
    wkv7_token(heads, wk->r, wk->w, wk->k, wk->v, wk->tmp, wk->tmp2, wk->out, H, S);
    group_norm(wk->out, L->att_ln_x_w, L->att_ln_x_b, H, S, 64e-5f);
    /* + v * sum_head(k * r * r_k) (:472-477) */
    const float * r_k = f32data(L->r_k);
    for (int64_t h = 0; h < H; h++) {
        float ps[64];
        for (int i = 0; i < 64; i++) ps[i] = 0.0f;
        for (int64_t j = 0; j < S; j++) ps[j & 63] += (wk->k[h * S + j] * wk->r[h * S + j]) * r_k[h * S + j];
        const float sum = fold_f(ps, 64);
        for (int64_t j = 0; j < S; j++) wk->out[h * S + j] += wk->v[h * S + j] * sum;
    }
    for (int64_t i = 0; i < D; i++) wk->out[i] *= wk->g[i];
    mm(L->att_output, wk->out, wk->tmp);
    for (int64_t i = 0; i < D; i++) wk->x[i] += wk->tmp[i];
}

/* rwkv_ffn_v4_v5 (:484-511), rwkv_ffn_v6 (:513-531), rwkv_ffn_v7 (:533-543) */
static void ffn(orc_model * m, const orc_layer * L, orc_work * wk, float * st) {
    const int64_t D = m->n_embed, F = m->ffn_size;
    float * ffn_xx = st;
    layer_norm(wk->x, L->ln2_w, L->ln2_b, wk->xn, D);
    if (m->arch_major <= 5) {
        lerp_v4(wk->xn, ffn_xx, L->ffn_mix_k, wk->xk, D);
        lerp_v4(wk->xn, ffn_xx, L->ffn_mix_r, wk->xr, D);
    } else if (m->arch_major == 6) {
        const float * mk = f32data(L->ffn_maa_k), * mr = f32data(L->ffn_maa_r);
        for (int64_t i = 0; i < D; i++) { const float sx = ffn_xx[i] - wk->xn[i]; wk->xk[i] = sx * mk[i] + wk->xn[i]; wk->xr[i] = sx * mr[i] + wk->xn[i]; }
    } else {
        const float * mk = f32data(L->ffn_x_k);
        for (int64_t i = 0; i < D; i++) { const float sx = ffn_xx[i] - wk->xn[i]; wk->xk[i] = sx * mk[i] + wk->xn[i]; }
    }
    memcpy(ffn_xx, wk->xn, (size_t) D * 4);
    mm(L->ffn_key, wk->xk, wk->ffk);
    for (int64_t i = 0; i < F; i++) { const float t = wk->ffk[i] > 0.0f ? wk->ffk[i] : 0.0f; wk->ffk[i] = t * t; }
    mm(L->ffn_value, wk->ffk, wk->out);
    if (m->arch_major == 7) {
        for (int64_t i = 0; i < D; i++) wk->x[i] += wk->out[i];
    } else {
        mm(L->ffn_receptance, wk->xr, wk->r);
        for (int64_t i = 0; i < D; i++) wk->x[i] += sigmoidf_(wk->r[i]) * wk->out[i];
    }
}

/* One pipeline stage: layers [lb, le) of the serial graph (rwkv_graph.inc:611-720). The first stage starts from the
 * token (embedding + ln0), later stages from the residual stream handed over in xio (D floats, plus D floats of v_first
 * for v7); the last stage finishes with ln_out + head. `state` (full layout) is updated in place for the stage's layers. */
static int eval_stage_inplace(orc_model * m, uint32_t lb, uint32_t le, uint32_t token, float * xio, float * state, float * logits_out) {
    orc_work wk; get_work(m, &wk);
    const int64_t D = m->n_embed;
    if (lb == 0) {
        if (token >= m->n_vocab) return 1;
        /* ggml_get_rows (:655): F16 rows are converted to f32 */
        if (m->emb->type == ORC_F32 || m->emb->type == ORC_F16) {
            orc_dequantize_row(m->emb->type, m->emb->data + (size_t) token * (size_t) D * orc_type_size(m->emb->type), wk.x, D);
        } else {
            orc_dequantize_row(m->emb->type, m->emb->data + (size_t) token * (size_t)(D / QK) * orc_type_size(m->emb->type), wk.x, D);
        }
        layer_norm(wk.x, m->ln0_w, m->ln0_b, wk.xn, D);
        memcpy(wk.x, wk.xn, (size_t) D * 4);
    } else {
        memcpy(wk.x, xio, (size_t) D * 4);
        if (m->arch_major == 7) memcpy(wk.v_first, xio + D, (size_t) D * 4);
    }
    const size_t per_layer = (m->arch_major >= 5) ? (size_t) D * (2 + (size_t) m->head_size) : (size_t) D * 5;
    for (uint32_t i = lb; i < le; i++) {
        float * st = state + (size_t) i * per_layer;
        const orc_layer * L = &m->layers[i];
        switch (m->arch_major) {
            case 4: att_v4(m, L, &wk, st); break;
            case 5: att_v5(m, L, &wk, st); break;
            case 6: att_v6(m, L, &wk, st); break;
            case 7: att_v7(m, L, &wk, st, (int) i); break;
        }
        ffn(m, L, &wk, st);
    }
    if (le == m->n_layer) {
        if (logits_out) {
            layer_norm(wk.x, m->ln_out_w, m->ln_out_b, wk.xn, D);
            orc_mul_mat(m->head->type, m->head->data, D, m->n_vocab, wk.xn, 1, logits_out);
        }
    } else if (xio) {
        memcpy(xio, wk.x, (size_t) D * 4);
        if (m->arch_major == 7) memcpy(xio + D, wk.v_first, (size_t) D * 4);
    }
    return 0;
}

int orc_eval_stage(orc_model * m, uint32_t layer_begin, uint32_t layer_end, uint32_t token, float * xio, float * state, float * logits_out) {
    if (layer_begin >= layer_end || layer_end > m->n_layer) return 1;
    return eval_stage_inplace(m, layer_begin, layer_end, token, xio, state, logits_out);
}

static int eval_inplace(orc_model * m, uint32_t token, float * state, float * logits_out) {
    return eval_stage_inplace(m, 0, m->n_layer, token, NULL, state, logits_out);
}

/* rwkv_eval (rwkv_eval.inc:38-76) */
int orc_eval(orc_model * m, uint32_t token, const float * state_in, float * state_out, float * logits_out) {
    return orc_eval_sequence(m, &token, 1, state_in, state_out, logits_out);
}

/* rwkv_eval_sequence (rwkv_eval.inc:79-155). The reference's sequence graph is the same per-token mathematics batched
 * over T (rwkv_graph.inc:744-866); the oracle runs it token by token. Logits are produced for the last token only. */
int orc_eval_sequence(orc_model * m, const uint32_t * tokens, size_t n, const float * state_in, float * state_out, float * logits_out) {
    if (n == 0) return 1;
    for (size_t i = 0; i < n; i++) if (tokens[i] >= m->n_vocab) return 1;
    const size_t sl = orc_state_len(m);
    float * state = (float *) malloc(sl * sizeof(float));
    if (state_in) memcpy(state, state_in, sl * sizeof(float)); else orc_init_state(m, state);
    int rc = 0;
    for (size_t i = 0; i < n && rc == 0; i++) rc = eval_inplace(m, tokens[i], state, (i == n - 1) ? logits_out : NULL);
    if (rc == 0 && state_out) memcpy(state_out, state, sl * sizeof(float));
    free(state);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* File -> file quantiser (rwkv_quantize.inc:1-171)                                                    */
/* ------------------------------------------------------------------------------------------------ */

static int type_from_name(const char * s) {
    if (!strcmp(s, "Q4_0")) return ORC_Q4_0; if (!strcmp(s, "Q4_1")) return ORC_Q4_1;
    if (!strcmp(s, "Q5_0")) return ORC_Q5_0; if (!strcmp(s, "Q5_1")) return ORC_Q5_1;
    if (!strcmp(s, "Q8_0")) return ORC_Q8_0;
    return -1;
}

/* rwkv_quantize.inc:1-13 */
static int needs_quant(const char * name) {
    static const char * skip[] = { "att.v1", "att.v2", "att.g1", "att.g2", "att.a1", "att.a2", "att.w1", "att.w2", "att.r_k" };
    if (!strcmp(name, "emb.weight") || !strcmp(name, "head.weight")) return 0;
    for (size_t i = 0; i < sizeof skip / sizeof skip[0]; i++) if (strstr(name, skip[i])) return 0;
    return 1;
}

int orc_quantize_file(const char * in_path, const char * out_path, const char * format_name) {
    const int out_type = type_from_name(format_name);
    if (out_type < 0) return 1;
    orc_model * m = NULL;
    /* parse with the loader's tensor walker, but without requiring a complete parameter table */
    FILE * f = fopen(in_path, "rb");
    if (!f) return 2;
    struct stat st; fstat(fileno(f), &st);
    uint8_t * blob = (uint8_t *) malloc((size_t) st.st_size);
    if (fread(blob, 1, (size_t) st.st_size, f) != (size_t) st.st_size) { fclose(f); free(blob); return 2; }
    fclose(f);
    (void) m;
    const uint8_t * p = blob, * end = blob + st.st_size;
    if (st.st_size < 24 || rd32(p) != 0x67676d66u) { free(blob); return 3; }
    const uint32_t in_dt = rd32(p + 20);
    if (in_dt != ORC_F32 && in_dt != ORC_F16) { free(blob); return 4; }
    FILE * o = fopen(out_path, "wb");
    if (!o) { free(blob); return 2; }
    uint32_t hdr[6]; memcpy(hdr, p, 24);
    hdr[1] = 101; hdr[5] = (uint32_t) out_type; /* :51-54 */
    fwrite(hdr, 1, 24, o);
    p += 24;
    while (p < end) {
        const uint32_t dc = rd32(p), kl = rd32(p + 4), ty = rd32(p + 8);
        int64_t ne[3] = { 1, 1, 1 };
        for (uint32_t i = 0; i < dc; i++) ne[i] = rd32(p + 12 + 4 * i);
        const uint8_t * key = p + 12 + 4 * dc;
        char name[128]; memcpy(name, key, kl); name[kl] = 0;
        const uint8_t * data = key + kl;
        const int64_t n = ne[0] * ne[1] * ne[2];
        const size_t nbytes = orc_type_size((int) ty) * (size_t) n / (size_t) orc_block_size((int) ty);
        if ((ty == ORC_F32 || ty == ORC_F16) && dc == 2 && needs_quant(name)) { /* :137-140 */
            float * f32 = (float *) malloc((size_t) n * 4);
            orc_dequantize_row((int) ty, data, f32, n);
            const size_t ob = orc_type_size(out_type) * (size_t) n / QK;
            uint8_t * out = (uint8_t *) malloc(ob);
            for (int64_t r = 0; r < ne[1]; r++) orc_quantize_row(out_type, f32 + r * ne[0], out + (size_t) r * (size_t)(ne[0] / QK) * orc_type_size(out_type), ne[0]);
            uint32_t th[3] = { dc, kl, (uint32_t) out_type };
            fwrite(th, 4, 3, o); fwrite(p + 12, 4, dc, o); fwrite(key, 1, kl, o); fwrite(out, 1, ob, o);
            free(f32); free(out);
        } else {
            fwrite(p, 1, 12 + 4 * dc + kl + nbytes, o);
        }
        p = data + nbytes;
    }
    fclose(o); free(blob);
    return 0;
}
