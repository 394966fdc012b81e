// persist_v47.hip -- the RWKV-4 and RWKV-7 single-token (decode) step over all layers of a stage as ONE persistent launch
// (rwkv_att_v4, rwkv_graph.inc:84-197; rwkv_ffn_v4_v5, :484-511; rwkv_att_v7, :387-482; rwkv_wkv_v7_impl,
// rwkv_operators_wkv_v7.inc:37-107; rwkv_ffn_v7, rwkv_graph.inc:533-543).
//
// Why: the fused per-layer launches of fused_v7.hip (RWKV-7: five, RWKV-4: four per layer) spend ~5 us per dependent launch before a byte
// moves -- 59.9 us per 66 MB layer of the 2.9B, 450 us per token on the 169M whose weights stream in 20 us. Here the
// layer is a chain of tagged hand-overs inside one launch, as in mega_v6.hip, with the structure cut to these architectures:
//
//   RWKV-4 (four hand-overs per layer)     x -> [LN1, three lerps, K / V / R rows of the SAME channels on one wave, WKV-4 of those channels,
//                                          sigmoid(r) * wkv] -> y -> [quantise, output rows + residual] -> x -> [LN2, two lerps, key rows
//                                          (relu^2, quantised per 32) + receptance rows] -> kq -> [value rows, gate, residual] -> x
//   RWKV-7 (five)                          x -> [LN1, six lerps, R / K / V rows + first low-rank stages] -> r k v lr1 -> [per head on its own
//                                          workgroup: second stages, key normalisation, WKV-7, GroupNorm + bonus, gate, quantise] -> yq ->
//                                          [output rows + residual] -> x -> [LN2, lerp, key rows] -> kq -> [value rows + residual] -> x
//
// Geometry (compile time, from D; F = 4 D for both architectures): the F / 32 key groups set the number of ROW workgroups NR = F / 32 / GPB
// (GPB = 1, or 2 when that many workgroups + the heads do not fit the chip); a row workgroup has 8 worker waves, worker w of workgroup b
// owns rows {e0, e0 + 64 (GPB = 2)} of EVERY D-row matrix (so the residual of those rows lives in its registers for the whole token and
// RWKV-4's per-channel recurrence is wave-local) and four rows of each of its key groups; one more wave ("comm", wave 8) polls every
// hand-over into LDS, runs the LayerNorm statistics on what it polled (the x units are laid out so that lane l receives the elements
// l mod 64: exactly the partials of the specified reduction tree), quantises the key groups, and on RWKV-7 runs the
// workgroup's share of the first low-rank stages. RWKV-7's heads run on H further workgroups that own no rows: their comm wave polls
// lr1, then r / k / v, and runs the recurrence of kdev's k7_head statement for statement, their eight workers the second stages.
// Weights go global -> registers one phase ahead (no ring: a workgroup's share of a 169M layer is 55 KB, of a 2.9B layer 412 KB in four
// phases). Arithmetic and reduction orders are those of fused_v7.hip / kernels.hip and of the CPU oracle: bit-identical.
//
// Embedding + ln0 (first stage) and ln_out + head + argmax (last stage; every workgroup of the grid, the CUs the layers do not use
// included) run inside the launch: a token is one graph node. Layer ranges serve pipeline stages and the streamed rwkv_eval.
//
// Hand-overs (DESIGN.md 6.3c has the measurements behind each): a wave that waits WATCHES before it sweeps -- one tag word per group of
// eight lanes, eight producers' units, four reads in flight on short rows (persist.h watch4) -- and sweeps when all eight have turned;
// RWKV-4's y is swept and quantised by the eight workers (YPAR); RWKV-7's head workgroups gather lr1 first and r / k / v under their
// second low-rank stages (P47_HEAD_SPLIT); three of five steps of RWKV-7's value rows wait in LDS from the end of the time mixing
// (ESTAGE, LDS-DMA); the layer table is read through the scalar cache (KLayer); spare workgroups park a third pass of the head in LDS
// (PARK). Compile-time switches for same-box A/B builds (tools/build_variant.sh): P47_WATCH (0 none / 1 one read / 2 four deep),
// P47_WATCH_SPREAD (units watched: 8), P47_HEAD_SPLIT, P47_ESTAGE, P47_E_NOWAIT, P47_YPAR, P47_PARK -- on; P47_EARLY, P47_E_MID,
// P47_E_WITH_C, P47_PRO2_FIRST -- measured, off. Run time: RWKV_MI_P47_NOFOLD, RWKV_MI_P47_CALM.
//
// Residency and safety as mega_v6.hip: NR (+ H) <= CUs, polls are bounded by the abort word.
#include "persist.h"

#include <cstring>

namespace rwkvmi {

struct P47Layer {
    long long ln1_w, ln1_b, ln2_w, ln2_b;
    long long mix_a[6];            // inputs of R, K, V, then (v7) w, a, g: coefficient vectors (v4: time_mix_r / _k / _v; v7: rows of x_rwkvag)
    long long mix_f[2];            // channel mixing: key input, receptance input (v4); v7: ffn.x_k
    long long tf, td;              // v4: time_first, time_decay
    M6Off wr, wk, wv, wo, fk, fr, fv;
    long long lr1[4], lr2[4];      // v7: w, a, g, v first / second stages (F16), byte offsets
    int rank[4], lbase[4];         // ranks and their offsets in the concatenated lr1 vector (w, a, g, v)
    long long w0, a0, v0, k_k, k_a, r_k, lnx_w, lnx_b;
    int has_v, layer0;             // v1 / v2 / v0 exist (not on absolute layer 0); this IS absolute layer 0
    int lr_n;                      // lr1 values produced on this layer (lr_total, or lr_total - rank_v on layer 0)
    int pad_;
};

// A layer's table entry read through the scalar cache (constant address space: the table is written before the launch and a uniform address
// then makes s_load instructions). As plain global reads the fields are VECTOR reads -- the kernel writes memory, so nothing proves the
// table invariant -- and a field read between two batches of weights waits, in order, for every weight issued before it.
struct KLayer {
    unsigned long long a;
    __device__ __forceinline__ long long q(size_t off) const { return *(const __attribute__((address_space(4))) long long *) (a + off); }
    __device__ __forceinline__ M6Off o(size_t off) const { return M6Off{q(off), q(off + 8), q(off + 16)}; }
    __device__ __forceinline__ int i(size_t off) const { return *(const __attribute__((address_space(4))) int *) (a + off); }
    template <int N> __device__ __forceinline__ void qs(size_t off, long long (&out)[N]) const {
#pragma unroll
        for (int i = 0; i < N; i++) out[i] = q(off + 8 * (size_t) i);
    }
};
#define KQ(L, f) (L).q(offsetof(P47Layer, f))
#define KO(L, f) (L).o(offsetof(P47Layer, f))
#define KI(L, f) (L).i(offsetof(P47Layer, f))
#define KLAYER(p, li) KLayer{(unsigned long long) ((p).layers + __builtin_amdgcn_readfirstlane(li))}

struct P47 {
    const P47Layer * layers; int l0, l1;
    const unsigned char * arena;
    float * x; float * v_first;
    float * x_out;                                   // where the last layer of the launch leaves x (= x; a pipeline stage: the next stage's x)
    const float * sin; float * sout; long long state_stride;
    void * xch; unsigned xch_bytes;
    int u_a, u_lr1, u_y, u_xatt, u_kq, u_xffn;      // unit (16-byte) offsets of the tagged buffers in the exchange arena
    unsigned * ctl;                                  // [0] tag generation, [1] abort
    long long * trace; int trace_layer;
    // embedding + ln0 inside the launch (first stage, rwkv_graph.inc:655-658): tok != nullptr -> x is LN0(emb[*tok]) instead of the plain input
    const uint32_t * tok; const void * emb; int emb_f16; long long ln0_w, ln0_b;
    // ln_out + head + argmax inside the launch (last stage, rwkv_graph.inc:704-708): logits != nullptr; every workgroup of the grid takes rows
    float * logits; uint32_t * next_tok; const void * head; long long lnout_w, lnout_b; int V; int u_am; int n_spare;
    int calm;                                        // long waits watch ONE unit before the full-width poll -- bit 0: spare / head workgroups for the last layer
                                                     // (the polling waves of the layers: P47_WATCH, compile time)
};

#ifndef P47_WATCH_SL
#define P47_WATCH_SL 5
#endif
#ifndef P47_WATCH_SPREAD
#define P47_WATCH_SPREAD 8
#endif
#ifndef P47_HEAD_SPLIT
#define P47_HEAD_SPLIT 1
#endif
#ifndef P47_PRO2_FIRST
#define P47_PRO2_FIRST 0
#endif
#ifndef P47_E_WITH_C
#define P47_E_WITH_C 0
#endif
#ifndef P47_E_MID
#define P47_E_MID 0
#endif
#ifndef P47_E_NOWAIT
#define P47_E_NOWAIT 1
#endif
#ifndef P47_ESTAGE
#define P47_ESTAGE 1
#endif
#ifndef P47_YPAR
#define P47_YPAR 1
#endif
#ifndef P47_EARLY
#define P47_EARLY 0
#endif
#ifndef P47_PARK
#define P47_PARK 1
#endif
#ifndef P47_WATCH
#define P47_WATCH 2
#endif

enum { S47_A = 0, S47_Y = 1, S47_XATT = 2, S47_KQ = 3, S47_XFFN = 4, S47_AM = 5, S47_IN = 6 };

__device__ __forceinline__ unsigned lf_ld(const unsigned * f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lf_add(unsigned * f, unsigned v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) (void) __hip_atomic_fetch_add(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lf_wait(Poll & pl, const unsigned * f, unsigned want) {
    for (unsigned spin = 0;; spin++) {
        if ((int) (lf_ld(f) - want) >= 0 || pl.dead) break;
        if ((spin & 1023u) == 1023u) {
            if (__hip_atomic_load(pl.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) pl.dead = true;
            else if (spin > 40000000u) { __hip_atomic_store(pl.ctl + 1, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pl.dead = true; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// acc + f16(w) * x with the f16 operand converted inside the instruction (v_fma_mix_f32: exact conversion, one rounding -- the same value as
// v_cvt_f32_f16 + v_fma_f32, in half the issue slots; P47_FMA_MIX=0 builds the two-instruction form for A/B runs)
#ifndef P47_FMA_MIX
#define P47_FMA_MIX 1
#endif
__device__ __forceinline__ float fma_h_lo(unsigned wpair, float x, float acc) {
#if P47_FMA_MIX
    float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(wpair), "v"(x), "v"(acc)); return d;
#else
    return fmaf(h2f_bits((uint16_t) (wpair & 0xFFFFu)), x, acc);
#endif
}
__device__ __forceinline__ float fma_h_hi(unsigned wpair, float x, float acc) {
#if P47_FMA_MIX
    float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(wpair), "v"(x), "v"(acc)); return d;
#else
    return fmaf(h2f_bits((uint16_t) (wpair >> 16)), x, acc);
#endif
}

// R rows row0, row0 + rstride, ... of a quantised matrix with nbk blocks per row: every load of the batch in flight (fused_blocks.h's
// batch_issue with a row stride; rows are always valid here)
template <int FMT, int R, int U, int U0 = 0>
__device__ __forceinline__ void rows_issue(Batch<FMT, R, U> & bt, const WPl & w, int row0, int rstride, int nbk, int lane) {
#pragma unroll
    for (int u = U0; u < U; u++) {
        const int bb = u * WAVE + lane;
        unsigned b = (unsigned) (bb < nbk ? bb : nbk - 1);
        asm volatile("" : "+v"(b));
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int64_t row = row0 + r * rstride;
            const uint8_t * rq = w.qs + row * nbk * QF<FMT>::QS;
            RawBlk<FMT> & o = bt.raw[u][r];
            if constexpr (QF<FMT>::HM) o.sc = ldw4(reinterpret_cast<const uint32_t *>(w.sc) + row * nbk + b);
            else o.sc = ldw2(reinterpret_cast<const uint16_t *>(w.sc) + row * nbk + b);
            if constexpr (QF<FMT>::QH) o.qh = ldw4(w.qh + row * nbk + b);
            o.q[0] = ldw16(rq + b * QF<FMT>::QS);
            if constexpr (QF<FMT>::QS == 32) o.q[1] = ldw16(rq + b * QF<FMT>::QS + 16);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int FMT, int R, int U>
__device__ __forceinline__ void rows_sum(const Batch<FMT, R, U> & bt, int nbk, int lane, const QVec & a, float (&res)[R]) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0f;
    batch_consume<FMT, R, U>(bt, nbk, 0, lane, a, acc);
#pragma unroll
    for (int r = 0; r < R; r++) res[r] = wave_sum_f(acc[r]);
}

struct L47 { size_t x, sc, q, lr, yq, kq, out, fl, hx, am, lr1, ch, hv, st, park, stage, total; };
// rows of the head parked in LDS by the workgroups the layers do not use (K47::PARK): one pass (8 rows) per wave, lane-major like the register buffers
__host__ __device__ constexpr int l47_ch(int D) { return D / 32 <= 24 ? 24 : 16; }
__host__ __device__ constexpr size_t l47_park_bytes(int D) { return D <= 768 ? (size_t) 9 * l47_ch(D) * 64 * 8 : 0; }
// RWKV-7 with long rows (K47::ESTAGE): three of the five 64-block steps of every worker's value rows (two rows: 16 + 4 + 4 bytes per block)
// wait in LDS from the end of the time mixing on -- 8 waves x 3 x 2 x 1.5 KiB = 72 KiB, which is exactly what D = 2560 leaves of 160 KiB
__host__ __device__ constexpr int l47_nst(int D, bool v7) { return (v7 && D > 768) ? 3 : 0; }
__host__ __device__ constexpr size_t l47_stage_bytes(int D, bool v7) { return (size_t) 8 * l47_nst(D, v7) * 2 * 1536; }
__host__ __device__ inline L47 l47_lds(int D, bool v7) {
    L47 o; size_t p = 0;
    auto take = [&](size_t n) { const size_t r = p; p += m6_round16(n); return r; };
    o.x = take((size_t) D * 4); o.sc = take(64);
    o.q = take(3 * m6_round16(qvec_bytes(D)));
    o.lr = take(v7 ? (size_t) 4 * D * 4 : 16);
    o.yq = take(qvec_bytes(D)); o.kq = take(qvec_bytes(4 * (size_t) D)); o.out = take(64 * 4); o.fl = take(64);
    o.hx = take((size_t) D * 4); o.am = take(2 * 16 * 4);
    // head workgroups (RWKV-7) use their own carving of the same allocation
    size_t h = 0;
    auto takeh = [&](size_t n) { const size_t r = h; h += m6_round16(n); return r; };
    o.lr1 = takeh(2048 * 4); o.ch = takeh(4 * 64 * 4); o.hv = takeh(5 * 64 * 4); o.st = takeh(64 * 68 * 4);
    o.park = p > h ? p : h;
    o.stage = o.park + l47_park_bytes(D);
    o.total = o.stage + l47_stage_bytes(D, v7);
    return o;
}

#define TT47(K) do { if (p.trace && (threadIdx.x & 63) == 0) p.trace[((long long) blockIdx.x * 9 + (threadIdx.x >> 6)) * 16 + (K)] = (long long) __builtin_amdgcn_s_memrealtime(); } while (0)
#define T47(K) do { if (p.trace && li == p.trace_layer && (threadIdx.x & 63) == 0) p.trace[((long long) blockIdx.x * 9 + (threadIdx.x >> 6)) * 16 + (K)] = (long long) __builtin_amdgcn_s_memrealtime(); } while (0)

// ARCH 4 / 7; HUB = 32-column steps of the longest second low-rank stage (max rank / 32); NL1 = 64-unit poll slots of the lr1 vector;
// MAXJ = four-row jobs of the first low-rank stages per comm wave; HUB2 = steps of the longest of w2 / a2 / v2 (HUB then is g2's)
template <int ARCH, int FMT, int D, int HUB, int NL1, int MAXJ, int HUB2>
struct K47 {
    static constexpr bool V7 = ARCH == 7;
    static constexpr int S = 64, H = D / 64;
    static constexpr int F = 4 * D, nb = D / 32, nbF = F / 32, GK = nbF;
    static constexpr int GPB = (GK + (V7 ? H : 0) <= 256) ? 1 : 2;
    static constexpr int NR = GK / GPB;                     // row workgroups
    static constexpr int NBLK = NR + (V7 ? H : 0);
    static constexpr int UD = (nb + 63) / 64, UF = (nbF + 63) / 64;
    static constexpr int NU = D / (64 * GPB);               // x units per lane of a comm wave
    static constexpr int NG4 = D / 4, SL = (NG4 + 511) / 512;   // float4 groups of the prologues, slots per worker thread
    static constexpr int DU = (3 * nb + 63) / 64, KQU = (3 * nbF + 63) / 64;
    static constexpr int STEPS = D / 32;
    static constexpr int NIA = V7 ? 6 : 3, NIF = V7 ? 1 : 2;
    // Short rows (D <= 768: a layer's weights are 66 registers per lane at Q5_1): every batch of weights is issued a whole phase earlier than it
    // is needed -- behind B1 what the phases after the y hand-over read, behind B4 the value rows and the next layer's r / k / v rows -- so no
    // read of the workers is in flight in front of a hand-over's sweep on the same CU (the sweeps behind a batch took 1.7 - 2.0 us after the
    // last store, the one with nothing in front of it 1.06). Long rows keep the late issue: 168 registers hold one phase's batch, not two.
    static constexpr bool EARLY = P47_EARLY && D <= 768;
    // The value rows' weights (F = 4 D long: 123 KB per workgroup at D = 2560) cannot be issued before the key rows are done -- the registers
    // hold the key rows' -- so the layer's last phase was its weight stream: 19.7 MB behind the kq hand-over, 5.4 us. ESTAGE: NST of a
    // wave's UF steps go global -> LDS (LDS-DMA, no registers) when the time mixing's rows are done, and wait there through the head's phase;
    // behind the key rows only UF - NST steps are left to stream. (Formats with a 4-byte scale pair and 16 code bytes per block.)
    static constexpr bool ESTAGE = P47_ESTAGE && V7 && GPB == 2 && UF >= 4 && QF<FMT>::HM && QF<FMT>::QS == 16 && nbF % 64 == 0 && l47_nst(D, V7) > 0;
    static constexpr int NST = ESTAGE ? l47_nst(D, V7) : 0, ST_Q = 0, ST_H = NST * GPB * 1024, ST_S = ST_H + NST * GPB * 256, ST_W = NST * GPB * 1536;
    // short rows: the value rows' weights (12 registers per lane at D = 768) go in flight with the output and key rows', a hand-over earlier --
    // nothing of the workers is then in front of the kq sweep on the CU
    static constexpr bool E_WITH_C = P47_E_WITH_C && !EARLY && UF <= 2;
    static constexpr bool YPAR = P47_YPAR && !V7 && GPB == 1 && (D / 8) % 32 == 0;   // RWKV-4's y hand-over swept and quantised by the eight workers
    static_assert(D % 256 == 0 && GK % GPB == 0 && NR * 8 * GPB == D && NBLK <= 256 && NU <= 32 && KQU <= 32, "geometry");

    struct Lds {
        float * x; float * sc; unsigned char * q[3]; float * lr[4]; unsigned char * yq; unsigned char * kq; float * out; unsigned * fl;
        float * lr1; float * ch; float * hv; float * st; float * hx; int * am; unsigned char * park; unsigned char * stage;
    };
    static __device__ __forceinline__ Lds carve(unsigned char * smem) {
        const L47 lo = l47_lds(D, V7);
        Lds l;
        l.x = reinterpret_cast<float *>(smem + lo.x); l.sc = reinterpret_cast<float *>(smem + lo.sc);
        for (int i = 0; i < 3; i++) l.q[i] = smem + lo.q + i * m6_round16(qvec_bytes(D));
        for (int i = 0; i < 4; i++) l.lr[i] = reinterpret_cast<float *>(smem + lo.lr) + (V7 ? i * D : 0);
        l.yq = smem + lo.yq; l.kq = smem + lo.kq; l.out = reinterpret_cast<float *>(smem + lo.out); l.fl = reinterpret_cast<unsigned *>(smem + lo.fl);
        l.lr1 = reinterpret_cast<float *>(smem + lo.lr1); l.ch = reinterpret_cast<float *>(smem + lo.ch); l.hv = reinterpret_cast<float *>(smem + lo.hv); l.st = reinterpret_cast<float *>(smem + lo.st);
        l.hx = reinterpret_cast<float *>(smem + lo.hx); l.am = reinterpret_cast<int *>(smem + lo.am); l.park = smem + lo.park; l.stage = smem + lo.stage;
        return l;
    }

    // rows of the D-row matrices owned by worker `own` of row workgroup `blk`: unit u = 8 blk + own is polled by lane u % 64 in slot u / 64
    static __device__ __forceinline__ int unit_of(int blk, int own) { return blk * 8 + own; }
    // the unit a long wait watches: one of the row workgroup half the grid away (any unit does; not one of this workgroup's own)
    // P47_WATCH_SPREAD = n (a power of two, 1 .. 64): the watch reads n units of n row workgroups spread over the grid, one per group of 64 / n
    // lanes (n requests per read), and turns when ALL of them have -- the sweep behind it rarely comes back incomplete (an incomplete sweep
    // costs a memory round trip). 1 -> 8: +6.4 % at 169M, +5.1 % at 2.9B.
    static __device__ __forceinline__ int far_unit() {
        constexpr int NSP = P47_WATCH_SPREAD < 1 ? 1 : P47_WATCH_SPREAD;
        const int k = (int) (threadIdx.x & 63) / (64 / NSP);
        return (((int) blockIdx.x + (2 * k + 1) * NR / (2 * NSP)) % NR) * 8 + (k & 7);
    }
    static __device__ __forceinline__ int far_head() {
        constexpr int NSP = P47_WATCH_SPREAD < 1 ? 1 : P47_WATCH_SPREAD;
        const int k = (int) (threadIdx.x & 63) / (64 / NSP);
        return ((int) blockIdx.x + k * H / NSP) % H;
    }
    // in front of a hand-over's sweep on a row workgroup's polling wave (P47_WATCH: 0 none, 1 one read at a time, 2 four reads deep);
    // watch_done(w) behind the sweep
    static __device__ __forceinline__ void watch(Watch4 & w, const P47 & p, Poll & pl, xrsrc xr, int unit, unsigned tag) {
#if P47_WATCH == 2
        // (four deep where the layer is short: +1.4 % at 169M; at 2.9B, where the weight stream runs through two of the five waits, -0.6 %)
        if constexpr (D <= 768) watch4<P47_WATCH_SL>(w, pl, p.xch, p.xch_bytes, unit, tag);
        else { calm_wait<1>(pl, xr, unit, tag); w.v = 0u; }
#elif P47_WATCH == 1
        calm_wait<1>(pl, xr, unit, tag); w.v = 0u;
#else
        w.v = 0u;
#endif
    }
    static __device__ __forceinline__ int row0_of(int u) { return GPB == 1 ? u : (u & 63) + 128 * (u >> 6); }

    // -----------------------------------------------------------------------------------------------------------
    // comm wave: an x-like vector (units in the layout above) polled into registers
    // -----------------------------------------------------------------------------------------------------------
    static __device__ __forceinline__ void poll_x(Poll & pl, xrsrc xr, int src, unsigned tag, int lane, float (&xs)[NU][GPB]) {
        // every slot of every lane carries a unit (NR * 8 = 64 NU): no per-slot validity -- a conditionally written xs[j] would be a
        // phi(undef, value) the register allocator keeps alive around the whole layer loop
        v4u v[NU];
        const int mine = src + lane;
        for (unsigned spin = 0;; spin++) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < NU; u++) v[u] = tg_load(xr, mine + u * 64);
            bool ok = true;
#pragma unroll
            for (int u = 0; u < NU; u++) ok = ok && tg_ok(v[u], tag);
            if (__all(ok) || pl.dead) break;
            if (poll_backoff(pl, spin)) break;
        }
#pragma unroll
        for (int j = 0; j < NU; j++) {
            xs[j][0] = __uint_as_float(v[j].x);
            if constexpr (GPB == 2) xs[j][1] = __uint_as_float(v[j].y);
        }
    }
    // LayerNorm statistics of the vector in xs (lane l holds the elements l + 64 s, s = GPB j + r): thread t < 256 of the specified
    // reduction owns the partial over t, t + 256, ... -- lane l runs the four partials l, l + 64, l + 128, l + 192 itself, then the tree.
    // Leaves x - mean in l.x and the scale in l.sc[0].
    static __device__ __forceinline__ void ln_meanvar(float (&xs)[NU][GPB], float & mean_out, float & scale_out) {   // xs <- xs - mean
        double pp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < NU; j++) {
            if ((GPB * j) % 8 == 0) __builtin_amdgcn_sched_barrier(0);   // (eight elements at a time: unpinned, the scheduler converts all NU * GPB values at once and spills)
#pragma unroll
            for (int r = 0; r < GPB; r++) pp[(GPB * j + r) & 3] += (double) xs[j][r];
        }
        const float mean = (float) (wave_sum_d((pp[0] + pp[2]) + (pp[1] + pp[3])) / (double) D);
        double qq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < NU; j++) {
            if ((GPB * j) % 8 == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < GPB; r++) {
                const float d = xs[j][r] - mean;
                xs[j][r] = d;
                qq[(GPB * j + r) & 3] += (double) (d * d);
            }
        }
        const float var = (float) (wave_sum_d((qq[0] + qq[2]) + (qq[1] + qq[3])) / (double) D);
        mean_out = mean;
        scale_out = 1.0f / sqrtf(var + 1e-5f);
    }
    static __device__ __forceinline__ void ln_stats(const Lds & l, int lane, float (&xs)[NU][GPB]) {
        float mean, scale;
        ln_meanvar(xs, mean, scale);
#pragma unroll
        for (int j = 0; j < NU; j++)
#pragma unroll
            for (int r = 0; r < GPB; r++) l.x[lane + 64 * (GPB * j + r)] = xs[j][r];
        if (lane == 0) l.sc[0] = scale;
    }
    // embedding row of the token + ln0 (rwkv_graph.inc:655-658; k_embed_ln0's statements) in the polled layout
    static __device__ __forceinline__ float emb_at(const P47 & p, long long row, int e) {
        if (p.emb_f16) return __half2float(reinterpret_cast<const __half *>(p.emb)[row * D + e]);
        return reinterpret_cast<const float *>(p.emb)[row * D + e];
    }
    static __device__ __forceinline__ long long emb_row(const P47 & p) { const unsigned tk = p.tok[0]; return tk < (unsigned) p.V ? (long long) tk : 0ll; }
    static __device__ __forceinline__ void embed_ln0(const P47 & p, const Lds & l, int lane, float (&xs)[NU][GPB]) {
        const M6Arena ar{p.arena};
        const long long row = emb_row(p);
        float w0[NU][GPB], b0[NU][GPB];
#pragma unroll
        for (int j = 0; j < NU; j++)
#pragma unroll
            for (int r = 0; r < GPB; r++) {
                const int e = lane + 64 * (GPB * j + r);
                xs[j][r] = emb_at(p, row, e); w0[j][r] = ar.f(p.ln0_w)[e]; b0[j][r] = ar.f(p.ln0_b)[e];
            }
        float mean, scale;
        ln_meanvar(xs, mean, scale);
#pragma unroll
        for (int j = 0; j < NU; j++)
#pragma unroll
            for (int r = 0; r < GPB; r++) { const float y = xs[j][r] * scale; const float yw = y * w0[j][r]; xs[j][r] = yw + b0[j][r]; }
        // the residual stream of layer 0: the workers pick their own rows out of l.hx behind B1 (the buffer is the head phase's otherwise)
#pragma unroll
        for (int j = 0; j < NU; j++)
#pragma unroll
            for (int r = 0; r < GPB; r++) l.hx[lane + 64 * (GPB * j + r)] = xs[j][r];
    }

    // -----------------------------------------------------------------------------------------------------------
    // worker waves: elementwise part of a prologue (affine, token-shift lerps, quantisation) on the 512 worker threads
    // -----------------------------------------------------------------------------------------------------------
    // Slot 0 of every thread (groups tid) is loaded a phase ahead; the groups past 512 (D = 2560: 128 of them, waves 0 and 1) load theirs
    // when they run -- two slots of RWKV-7's six coefficient vectors in registers next to the R / K / V rows in flight do not fit a wave.
    template <int NI> struct Pro { float4 lw, lb, pv, cf[NI]; };
    template <int NI> struct ProSrc { const float * lw; const float * lb; const float * pv; const float * cf[NI]; };

    template <int NI>
    static __device__ __forceinline__ void pro_load(Pro<NI> & pr, const ProSrc<NI> & src, int g) {
        const int i = 4 * g;
        pr.lw = *reinterpret_cast<const float4 *>(src.lw + i); pr.lb = *reinterpret_cast<const float4 *>(src.lb + i);
        pr.pv = *reinterpret_cast<const float4 *>(src.pv + i);
#pragma unroll
        for (int c = 0; c < NI; c++) pr.cf[c] = *reinterpret_cast<const float4 *>(src.cf[c] + i);
    }
    template <int NI>
    static __device__ __forceinline__ ProSrc<NI> pro_src(const M6Arena & ar, long long lw, long long lb, const long long * cf, const float * prev) {
        ProSrc<NI> s;
        s.lw = ar.f(lw); s.lb = ar.f(lb); s.pv = prev;
#pragma unroll
        for (int c = 0; c < NI; c++) s.cf[c] = ar.f(cf[c]);
        return s;
    }
    // one group of four elements: affine, lerps, quantised images c < NQ; RWKV-7's time mixing also stores mixes 3, 4, 5 and 2 fp16-rounded in l.lr[0..3]
    template <int NI, int NQ, bool LR>
    static __device__ __forceinline__ void pro_group(const Lds & l, const Pro<NI> & pr, float scale, float * carry_out, bool write_state, int g, bool ok) {
        const int i = 4 * g;
        const float4 xc = *reinterpret_cast<const float4 *>(l.x + i);
        const float xs[4] = {xc.x, xc.y, xc.z, xc.w};
        const float lw[4] = {pr.lw.x, pr.lw.y, pr.lw.z, pr.lw.w}, lb[4] = {pr.lb.x, pr.lb.y, pr.lb.z, pr.lb.w};
        const float pv[4] = {pr.pv.x, pr.pv.y, pr.pv.z, pr.pv.w};
        float xn[4], mx[NI][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float y = xs[j] * scale;
            const float yw = y * lw[j];
            xn[j] = yw + lb[j];
        }
#pragma unroll
        for (int c = 0; c < NI; c++) {
            const float cf[4] = {pr.cf[c].x, pr.cf[c].y, pr.cf[c].z, pr.cf[c].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if constexpr (V7) {
                    const float sx = pv[j] - xn[j];
                    const float sm = sx * cf[j];
                    mx[c][j] = sm + xn[j];
                } else {
                    const float xcf = xn[j] * cf[j], pc = pv[j] * cf[j];
                    mx[c][j] = xcf + (pv[j] - pc);
                }
            }
        }
        if (write_state && ok) *reinterpret_cast<float4 *>(carry_out + i) = make_float4(xn[0], xn[1], xn[2], xn[3]);
#pragma unroll
        for (int c = 0; c < NQ; c++) {
            unsigned packed; float d16, s16; int isum;
            quant_vec4(mx[c], packed, d16, s16, isum);
            if (ok) qvec_store4(qvec_at(l.q[c], D), nb, i, packed, d16, s16, isum);
        }
        if constexpr (LR) {
            if (ok) {
                constexpr int src[4] = {3, 4, 5, 2};
#pragma unroll
                for (int m = 0; m < 4; m++)
                    *reinterpret_cast<float4 *>(l.lr[m] + i) = make_float4(round_f16(mx[src[m]][0]), round_f16(mx[src[m]][1]), round_f16(mx[src[m]][2]), round_f16(mx[src[m]][3]));
            }
        }
    }
    template <int NI, int NQ, bool LR>
    static __device__ __forceinline__ void pro_run(const Lds & l, const Pro<NI> & pr, const ProSrc<NI> & src, float * carry_out, bool write_state, int tid) {
        const float scale = l.sc[0];
        if constexpr (SL == 2 && P47_PRO2_FIRST) {
            // (the second slot's parameters go in flight BEFORE the first slot's arithmetic: loaded behind it, waves 0 and 1 finished the
            //  time-mixing prologue 1.45 us after the others -- a memory round trip -- and B2 waited for them)
            const int wave0 = __builtin_amdgcn_readfirstlane(tid) & ~63;
            if (wave0 + 512 < NG4) {           // whole waves (NG4 % 64 == 0)
                Pro<NI> p2;
                asm volatile("" : : "v"(pr.cf[NI - 1].w) : "memory");   // (the first slot's parameters have landed HERE: with p2 in flight the compiler's wait for them would be a wait for p2 as well)
                pro_load<NI>(p2, src, tid + 512);
                __builtin_amdgcn_sched_barrier(0);
                pro_group<NI, NQ, LR>(l, pr, scale, carry_out, write_state, tid, true);
                pro_group<NI, NQ, LR>(l, p2, scale, carry_out, write_state, tid + 512, true);
            } else {
                pro_group<NI, NQ, LR>(l, pr, scale, carry_out, write_state, tid < NG4 ? tid : 0, tid < NG4);
            }
            return;
        }
        pro_group<NI, NQ, LR>(l, pr, scale, carry_out, write_state, tid < NG4 ? tid : 0, tid < NG4);
#pragma unroll
        for (int u = 1; u < SL; u++) {
            const int wave0 = __builtin_amdgcn_readfirstlane(tid) & ~63;
            if (wave0 + 512 * u < NG4) {       // whole waves (NG4 % 64 == 0)
                Pro<NI> p2;
                pro_load<NI>(p2, src, tid + 512 * u);
                pro_group<NI, NQ, LR>(l, p2, scale, carry_out, write_state, tid + 512 * u, true);
            }
        }
    }

    // -----------------------------------------------------------------------------------------------------------
    // first low-rank stages (RWKV-7), four rows per job on a comm wave: lane = 16 row + q, lane q keeps partials 2q, 2q + 1 of ggml's 32
    // -----------------------------------------------------------------------------------------------------------
    struct Job { unsigned w[STEPS]; };
    static __device__ __forceinline__ void job_issue(Job & jb, const unsigned char * W, int row, int lane) {
        const uint32_t * base = reinterpret_cast<const uint32_t *>(W) + (long long) row * (D / 2) + (lane & 15);
#pragma unroll
        for (int s = 0; s < STEPS; s++) jb.w[s] = ldw4(base + 16 * s);
        __builtin_amdgcn_sched_barrier(0);
    }
    static __device__ __forceinline__ float job_row(const Job & jb, const float * l_x, int lane) {
        const int q = lane & 15;
        float a0 = 0.0f, a1 = 0.0f;
        // The activation reads go out sixteen steps at a time, the next batch under the current one's FMAs, pinned by scheduling barriers:
        // left alone the compiler pairs every read with a wait in front of its FMA -- 40 LDS round trips in a row (5.3 us per job at D = 2560).
        constexpr int BS = 16, NBAT = (STEPS + BS - 1) / BS;
        float2 xa[BS], xb[BS];
        auto rd = [&](float2 (&dst)[BS], int bi) {
#pragma unroll
            for (int t = 0; t < BS; t++) { const int s = bi * BS + t; if (s < STEPS) dst[t] = *reinterpret_cast<const float2 *>(l_x + 32 * s + 2 * q); }
        };
        auto fm = [&](const float2 (&src)[BS], int bi) {
#pragma unroll
            for (int t = 0; t < BS; t++) {
                const int s = bi * BS + t;
                if (s < STEPS) {
                    a0 = fma_h_lo(jb.w[s], src[t].x, a0);
                    a1 = fma_h_hi(jb.w[s], src[t].y, a1);
                }
            }
        };
        rd(xa, 0);
#pragma unroll
        for (int bi = 0; bi < NBAT; bi += 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (bi + 1 < NBAT) rd(xb, bi + 1);
            __builtin_amdgcn_sched_barrier(0);
            fm(xa, bi);
            __builtin_amdgcn_sched_barrier(0);
            if (bi + 2 < NBAT) rd(xa, bi + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (bi + 1 < NBAT) fm(xb, bi + 1);
        }
        // ggml's fold: ps[i] += ps[i + 16], ps[i] += ps[i + 8], ps[i] += ps[i + 4], (ps0 + ps1) + (ps2 + ps3)
        a0 = a0 + __int_as_float(lane_xor8_i(__float_as_int(a0))); a1 = a1 + __int_as_float(lane_xor8_i(__float_as_int(a1)));
        a0 = a0 + __int_as_float(lane_xor4_i(__float_as_int(a0))); a1 = a1 + __int_as_float(lane_xor4_i(__float_as_int(a1)));
        a0 = a0 + __int_as_float(lane_xor2_i(__float_as_int(a0))); a1 = a1 + __int_as_float(lane_xor2_i(__float_as_int(a1)));
        const float s01 = a0 + a1;
        return s01 + __int_as_float(lane_xor1_i(__float_as_int(s01)));
    }

    // -----------------------------------------------------------------------------------------------------------
    // row workgroup, comm wave
    // -----------------------------------------------------------------------------------------------------------
    static __device__ __forceinline__ void row_comm(const P47 & p, const Lds & l, int lane0, unsigned base) {
        const int blk = blockIdx.x;
        Poll pl{p.ctl, false};
        const xrsrc xr = make_xrsrc(p.xch, p.xch_bytes);
        unsigned keys_done = 0;
        __builtin_amdgcn_s_setprio(2);   // every hand-over of the workgroup goes through this wave: it issues ahead of the two workers on its SIMD
        for (int li = p.l0; li < p.l1; li++) {
            const KLayer L = KLAYER(p, li);
            const unsigned tagL = base + (unsigned) (li - p.l0) * 8u;
            T47(0);
            // (per-phase opaque copies of the lane id: poll and LDS addresses derived from it are recomputed where they are used instead of
            //  being hoisted out of the layer loop as ~100 loop-invariant registers -- and spilled)
            // ---- A: x of the previous layer (or of the launch's input: the workers published it under the tag before this launch's first) ----
            {
                const int lane = opq(lane0);
                float xs[NU][GPB];
                if (li == p.l0 && p.tok) embed_ln0(p, l, lane, xs);
                else {
                    Watch4 wt;
                    watch(wt, p, pl, xr, p.u_xffn + far_unit(), tagL - 8u + (li == p.l0 ? S47_IN : S47_XFFN));
                    poll_x(pl, xr, p.u_xffn, tagL - 8u + (li == p.l0 ? S47_IN : S47_XFFN), lane, xs);
                    watch_done(wt);
                }
                T47(1);
                ln_stats(l, lane, xs);
            }
            T47(2);
            __syncthreads();   // B1: x - mean and the scale are in LDS
            // (the first low-rank stages' rows go in flight behind B1: issuing eighty loads in front of it cost the workers a microsecond of x)
            Job jb[V7 ? MAXJ : 1];
            int jm[MAXJ], jrow[MAXJ]; bool jhas[MAXJ];
            if constexpr (V7) {
                __builtin_amdgcn_sched_barrier(0);   // (the job's loads stay behind the statistics: hoisted above them they cost the registers the poll needs)
                const int lane = opq(lane0);
#pragma unroll
                for (int k = 0; k < MAXJ; k++) {
                    const int r0 = 4 * (blk + NR * k);                       // first row of the job in the concatenated lr1 vector
                    jhas[k] = r0 < KI(L, lr_n);
                    int m = 0, mb = 0;
#pragma unroll
                    for (int t = 1; t < 4; t++) { const int lb = L.i(offsetof(P47Layer, lbase) + 4 * t); if (r0 >= lb) { m = t; mb = lb; } }
                    jm[k] = jhas[k] ? m : 0;
                    jrow[k] = jhas[k] ? r0 - mb + (lane >> 4) : 0;
                    job_issue(jb[k], p.arena + L.q(offsetof(P47Layer, lr1) + 8 * (size_t) __builtin_amdgcn_readfirstlane(jm[k])), jrow[k], lane);
                }
            }
            __syncthreads();   // B2: the workers' images
            if constexpr (V7) {
                const int lane = opq(lane0);
                __builtin_amdgcn_s_setprio(3);   // (two worker waves of this SIMD run their row phase beside it: the heads wait for THIS wave)
#pragma unroll
                for (int k = 0; k < MAXJ; k++) {
                    float v = job_row(jb[k], l.lr[0] + jm[k] * D, lane);   // (one base + offset: an indexed pointer array would live in scratch as generic pointers)
                    if (jm[k] == 0) v = det_tanhf(v);
                    else if (jm[k] == 2) v = sigmoid_f(v);
                    if (jhas[k] && (lane & 15) == 0) tg_store(xr, p.u_lr1 + L.i(offsetof(P47Layer, lbase) + 4 * (size_t) __builtin_amdgcn_readfirstlane(jm[k])) + jrow[k], __float_as_uint(v), 0u, 0u, 0u, tagL + S47_A);
                }
                __builtin_amdgcn_s_setprio(0);
            }
            T47(3);
            // ---- C: y ----
            if constexpr (V7) {
                Watch4 wt;
                watch(wt, p, pl, xr, p.u_y + 6 * far_head(), tagL + S47_Y);   // (a head = two 32-blocks = six units)
                stage_qvec<DU, 64>(pl, xr, p.u_y, D, tagL + S47_Y, l.yq, opq(lane0));
                watch_done(wt);
            } else if constexpr (YPAR) {
                // RWKV-4: y arrives as floats and every workgroup quantises all of it. This wave only watches; the eight workers -- idle at B3
                // otherwise -- sweep an eighth of the units each and quantise their blocks (one wave doing all of it: 12 reads per lane, an LDS
                // transposition and three quantiser passes, ~1.4 us behind the watch)
                Watch4 wt;
                watch(wt, p, pl, xr, p.u_y + far_unit(), tagL + S47_Y);
                lf_add(l.fl + 2, 1u);
                watch_drain(wt);   // (no sweep of this wave behind the watch: its last reads land here, off the critical path)
            } else {
                const int lane = opq(lane0);
                float ys[NU][GPB];
                Watch4 wt;
                watch(wt, p, pl, xr, p.u_y + far_unit(), tagL + S47_Y);
                poll_x(pl, xr, p.u_y, tagL + S47_Y, lane, ys);
                watch_done(wt);
                // Polled layout: lane l of slot s holds element l + 64 s. Quantising in that layout (one 32-block per half-wave and slot) costs
                // two f32 divisions + a rounding per element on this one wave (2.3 us of the y hand-over at D = 768); through LDS (l.x is free
                // between the time-mixing prologue and the next statistics) the vector comes back four consecutive elements per lane, eight
                // lanes per block: the prologues' quantiser, a quarter of the instructions. Max and integer sum are order-free: same codes.
#pragma unroll
                for (int j = 0; j < NU; j++)
#pragma unroll
                    for (int r = 0; r < GPB; r++) l.x[lane + 64 * (GPB * j + r)] = ys[j][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const QVec lq = qvec_at(l.yq, D);
#pragma unroll
                for (int pz = 0; pz < D / 256; pz++) {
                    const int i = 4 * (lane + 64 * pz);
                    const float4 y4 = *reinterpret_cast<const float4 *>(l.x + i);
                    const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
                    unsigned packed; float d16, s16; int isum;
                    quant_vec4(yv, packed, d16, s16, isum);
                    qvec_store4(lq, nb, i, packed, d16, s16, isum);
                }
            }
            T47(4);
            __syncthreads();   // B3: yq
            // ---- D: x after the time mixing ----
            {
                const int lane = opq(lane0);
                float xs[NU][GPB];
                Watch4 wt;
                watch(wt, p, pl, xr, p.u_xatt + far_unit(), tagL + S47_XATT);
                poll_x(pl, xr, p.u_xatt, tagL + S47_XATT, lane, xs);
                watch_done(wt);
                T47(5);
                ln_stats(l, lane, xs);
            }
            T47(6);
            __syncthreads();   // B4
            __syncthreads();   // B5
            // key groups of this workgroup: the workers' relu^2 outputs -> quantised 32-blocks (half-wave = group)
            keys_done += 8u;
            lf_wait(pl, l.fl, keys_done);
            T47(7);
            {
                const int lane = opq(lane0);
                const int gi = lane >> 5;
                const bool valid = gi < GPB;
                const float v = valid ? l.out[gi * 32 + (lane & 31)] : 0.0f;
                int qi, isum; float d16, s16;
                quant_block32(v, qi, d16, s16, isum);
                tq_store_block(xr, p.u_kq, valid ? blk * GPB + gi : 0, lane & 31, qi, d16, s16, isum, tagL + S47_KQ, valid);
                if constexpr (UF >= 4) lf_add(l.fl + 1, 1u);
            }
            T47(9);
            // ---- E: kq ----
            Watch4 wt;
            watch(wt, p, pl, xr, p.u_kq + 3 * GPB * (far_unit() >> 3), tagL + S47_KQ);
            stage_qvec<KQU, 64>(pl, xr, p.u_kq, F, tagL + S47_KQ, l.kq, opq(lane0));
            watch_done(wt);
            T47(8);
            __syncthreads();   // B6
        }
    }

    // -----------------------------------------------------------------------------------------------------------
    // row workgroup, worker waves
    // -----------------------------------------------------------------------------------------------------------
    static __device__ __forceinline__ void row_worker(const P47 & p, const Lds & l, int tid0, int own, unsigned base) {
        const int blk = blockIdx.x;
        const M6Arena ar{p.arena};
        const xrsrc xr = make_xrsrc(p.xch, p.xch_bytes);
        const int u = unit_of(blk, own), e0 = row0_of(u);
        auto myrow_of = [&](int lane) { return e0 + 64 * (lane < GPB ? lane : 0); };   // lane r < GPB finishes row r
        float xown[GPB];
        auto x_store = [&](int buf, unsigned tag) {
            if ((threadIdx.x & 63) == 0) tg_store(xr, buf + u, __float_as_uint(xown[0]), __float_as_uint(xown[GPB - 1]), 0u, 0u, tag);
        };
        if (p.tok) {   // first stage: the residual of this wave's rows is LN0 of the embedding row, left in LDS by the comm wave (read behind B1)
#pragma unroll
            for (int r = 0; r < GPB; r++) xown[r] = 0.0f;
        } else {
#pragma unroll
            for (int r = 0; r < GPB; r++) xown[r] = p.x[e0 + 64 * r];
            // the launch's input, as if a layer before the first had produced it -- under a slot of its own: with the head folded the last layer
            // of the PREVIOUS launch published its x under (its tag) + S47_XFFN, which is exactly this launch's (base - 8) + S47_XFFN
            x_store(p.u_xffn, base - 8u + S47_IN);
        }

        Pro<NIA> pa; Pro<NIF> pf;
        ProSrc<NIA> sa; ProSrc<NIF> sf;
        Batch<FMT, GPB, UD> wA[3], wC, wFr;
        Batch<FMT, 4, UD> wK[GPB];
        Batch<FMT, GPB, UF> wE;
        float st4[5];                                                         // v4: aa, bb, pp, time_first, time_decay of this lane's channel
        Poll plw{p.ctl, false};
        unsigned kq_seen = 0, y_seen = 0;

        // has = false (behind the last layer): every lane loads block 0 of row 0 / group 0 -- one request per instruction, and the issue stays
        // straight-line code (a branch around it leaves the buffers conditionally defined: they would live, and spill, around the whole loop)
        auto issue_A = [&](int li, bool has) {
            __builtin_amdgcn_sched_barrier(0);
            const int tid = opq(tid0), lane = has ? (tid & 63) : 0, myrow = myrow_of(tid & 63);
            const KLayer L{(unsigned long long) (p.layers + __builtin_amdgcn_readfirstlane(li))};
            const float * sin_l = p.sin + (long long) (li - p.l0) * p.state_stride;
            long long mix[NIA];
            L.qs<NIA>(offsetof(P47Layer, mix_a), mix);
            sa = pro_src<NIA>(ar, KQ(L, ln1_w), KQ(L, ln1_b), mix, sin_l + D);
            pro_load<NIA>(pa, sa, (has && tid < NG4) ? tid : 0);
            if constexpr (!V7) {
                st4[0] = sin_l[2 * D + myrow]; st4[1] = sin_l[3 * D + myrow]; st4[2] = sin_l[4 * D + myrow];
                st4[3] = ar.f(KQ(L, tf))[myrow]; st4[4] = ar.f(KQ(L, td))[myrow];
            }
            rows_issue<FMT, GPB, UD>(wA[0], ar.w(KO(L, wr)), has ? e0 : 0, has ? 64 : 0, has ? nb : 1, lane);
            rows_issue<FMT, GPB, UD>(wA[1], ar.w(KO(L, wk)), has ? e0 : 0, has ? 64 : 0, has ? nb : 1, lane);
            rows_issue<FMT, GPB, UD>(wA[2], ar.w(KO(L, wv)), has ? e0 : 0, has ? 64 : 0, has ? nb : 1, lane);
        };
        // EARLY: the youngest read of the previous batch as an operand of an (empty) instruction HERE. The compiler counts reads and writes of
        // vector memory in one counter and, with a write outstanding (a hand-over store, a state row), waits for the counter to reach zero
        // whatever it needs: a prologue that finds a fresh batch in flight then waits for the whole batch (measured: 1.05 -> 2.7 us). With the
        // wait taken before the new batch is issued, the phases that follow find their operands landed and wait for nothing.
        auto landed = [&](auto & bt) {
            constexpr int UU = (int) (sizeof(bt.raw) / sizeof(bt.raw[0])), RR = (int) (sizeof(bt.raw[0]) / sizeof(bt.raw[0][0]));
            if constexpr (QF<FMT>::QS == 32) asm volatile("" : : "v"(bt.raw[UU - 1][RR - 1].q[1].w) : "memory");
            else asm volatile("" : : "v"(bt.raw[UU - 1][RR - 1].q[0].w) : "memory");
        };
        // the channel-mixing prologue's parameters, the output rows and the key (+ receptance) rows
        auto issue_C = [&](int li) {
            __builtin_amdgcn_sched_barrier(0);
            const int tid = opq(tid0), lane = tid & 63;
            const KLayer L{(unsigned long long) (p.layers + __builtin_amdgcn_readfirstlane(li))};
            const float * sin_l = p.sin + (long long) (li - p.l0) * p.state_stride;
            long long mix[NIF];
            L.qs<NIF>(offsetof(P47Layer, mix_f), mix);
            sf = pro_src<NIF>(ar, KQ(L, ln2_w), KQ(L, ln2_b), mix, sin_l);
            pro_load<NIF>(pf, sf, tid < NG4 ? tid : 0);
            rows_issue<FMT, GPB, UD>(wC, ar.w(KO(L, wo)), e0, 64, nb, lane);
            const WPl fkp = ar.w(KO(L, fk));
#pragma unroll
            for (int g = 0; g < GPB; g++) rows_issue<FMT, 4, UD>(wK[g], fkp, 32 * (blk * GPB + g) + 4 * own, 1, nb, lane);
            if constexpr (!V7) rows_issue<FMT, GPB, UD>(wFr, ar.w(KO(L, fr)), e0, 64, nb, lane);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto issue_E = [&](int li) {
            __builtin_amdgcn_sched_barrier(0);
            const KLayer L{(unsigned long long) (p.layers + __builtin_amdgcn_readfirstlane(li))};
            rows_issue<FMT, GPB, UF, NST>(wE, ar.w(KO(L, fv)), e0, 64, nbF, opq(tid0) & 63);
        };
        // ESTAGE: steps 0 .. NST - 1 of this wave's value rows, global -> LDS: per (step, row) 1 KiB of codes, 256 B of fifth bits, 256 B of
        // scale pairs, lane-major (lane l's block lands at l x 16 / l x 4 of its slice). M0 carries the LDS address of an LDS-DMA
        // instruction and is not preserved around a statement: saved and restored inside it.
        auto stage_E = [&](int li) {
            if constexpr (ESTAGE) {
                __builtin_amdgcn_sched_barrier(0);
                const KLayer L{(unsigned long long) (p.layers + __builtin_amdgcn_readfirstlane(li))};
                const WPl fv = ar.w(KO(L, fv));
                const unsigned lane = (unsigned) (opq(tid0) & 63);
                const unsigned m0 = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) (l.stage + own * ST_W));
                const unsigned long long bq = (unsigned long long) fv.qs, bh = (unsigned long long) fv.qh, bs = (unsigned long long) fv.sc;
#pragma unroll
                for (int u = 0; u < NST; u++)
#pragma unroll
                    for (int r = 0; r < GPB; r++) {
                        const unsigned bi = (unsigned) (e0 + 64 * r) * (unsigned) nbF + (unsigned) u * 64u + lane;
                        const unsigned vq = bi * 16u, vh = bi * 4u;
                        const unsigned dq = m0 + ST_Q + (u * GPB + r) * 1024, dh = m0 + ST_H + (u * GPB + r) * 256, ds = m0 + ST_S + (u * GPB + r) * 256;
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(vq), "s"(bq), "s"(dq) : "memory");
                        if constexpr (QF<FMT>::QH) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(vh), "s"(bh), "s"(dh) : "memory");
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(vh), "s"(bs), "s"(ds) : "memory");
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ... and back into the batch's registers in front of the value rows (every read of this wave has landed: one full wait)
        auto unstage_E = [&]() {
            if constexpr (ESTAGE) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int lane = opq(tid0) & 63;
                const unsigned char * sb = l.stage + own * ST_W;
#pragma unroll
                for (int u = 0; u < NST; u++)
#pragma unroll
                    for (int r = 0; r < GPB; r++) {
                        RawBlk<FMT> & o = wE.raw[u][r];
                        o.q[0] = *reinterpret_cast<const int4 *>(sb + ST_Q + (u * GPB + r) * 1024 + lane * 16);
                        if constexpr (QF<FMT>::QH) o.qh = *reinterpret_cast<const unsigned *>(sb + ST_H + (u * GPB + r) * 256 + lane * 4);
                        o.sc = *reinterpret_cast<const unsigned *>(sb + ST_S + (u * GPB + r) * 256 + lane * 4);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        issue_A(p.l0, true);

        for (int li = p.l0; li < p.l1; li++) {
            float * sout_l = p.sout + (long long) (li - p.l0) * p.state_stride;
            const unsigned tagL = base + (unsigned) (li - p.l0) * 8u;
            const bool last = li + 1 == p.l1;
            T47(0);
            __syncthreads();   // B1
            T47(1);
            if (li == p.l0 && p.tok) {
#pragma unroll
                for (int r = 0; r < GPB; r++) xown[r] = l.hx[e0 + 64 * r];
            }
            if constexpr (EARLY) { landed(wA[2]); issue_C(li); }
            pro_run<NIA, 3, V7>(l, pa, sa, sout_l + D, blk == 0, opq(tid0));
            T47(2);
            __syncthreads();   // B2
            // ---- A rows ----
            {
                const int lane = opq(tid0) & 63, myrow = myrow_of(lane);
                float rr[GPB], kk[GPB], vv[GPB];
                rows_sum<FMT, GPB, UD>(wA[0], nb, lane, qvec_at(l.q[0], D), rr);
                rows_sum<FMT, GPB, UD>(wA[1], nb, lane, qvec_at(l.q[1], D), kk);
                rows_sum<FMT, GPB, UD>(wA[2], nb, lane, qvec_at(l.q[2], D), vv);
                if constexpr (V7) {
                    const float rv = pick_lane<GPB>(rr, lane), kv = pick_lane<GPB>(kk, lane), vvv = pick_lane<GPB>(vv, lane);
                    if (lane < GPB) tg_store(xr, p.u_a + myrow, __float_as_uint(rv), __float_as_uint(kv), __float_as_uint(vvv), 0u, tagL + S47_A);
                } else {
                    // WKV-4 of this wave's channel(s) (k_wkv4's statements, rwkv_graph.inc:119-161,178-195), then r * wkv. Every operand is
                    // wave-uniform, so the five exponentials of a channel (two per softmax pair + the sigmoid's) run as ONE call with a
                    // different argument per lane and come back through readlane: five double-precision polynomials in a row were a quarter
                    // of this phase.
                    float yout[GPB];
#pragma unroll
                    for (int r = 0; r < GPB; r++) {
                        auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), r)); };
                        const float aa = bc(st4[0]), bb = bc(st4[1]), pp = bc(st4[2]), uu = bc(st4[3]), w = bc(st4[4]);
                        const float kv = kk[r], vvv = vv[r], rv = rr[r];
                        const float ww1 = uu + kv;
                        const float qq1 = fmaxf(pp, ww1);
                        const float ww2 = pp + w;
                        const float qq2 = fmaxf(ww2, kv);
                        const float arg = lane == 0 ? pp - qq1 : (lane == 1 ? ww1 - qq1 : (lane == 2 ? ww2 - qq2 : (lane == 3 ? kv - qq2 : -rv)));
                        const float ex = det_expf(arg);
                        auto rl = [&](int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex), i)); };
                        const float e1 = rl(0), e2 = rl(1), f1 = rl(2), f2 = rl(3), er = rl(4);
                        const float rs = 1.0f / (1.0f + er);
                        const float a = e1 * aa + e2 * vvv;
                        const float b = e1 * bb + e2;
                        if (lane == 0) { sout_l[2 * D + e0 + 64 * r] = f1 * aa + f2 * vvv; sout_l[3 * D + e0 + 64 * r] = f1 * bb + f2; sout_l[4 * D + e0 + 64 * r] = qq2; }
                        yout[r] = rs * (a / b);
                    }
                    if (lane == 0) tg_store(xr, p.u_y + u, __float_as_uint(yout[0]), __float_as_uint(yout[GPB - 1]), 0u, 0u, tagL + S47_Y);
                }
            }
            T47(3);
            if constexpr (!EARLY) { issue_C(li); stage_E(li); if constexpr (E_WITH_C) issue_E(li); }   // (they stream through the y hand-over)
            if constexpr (YPAR) {
                // this wave's eighth of y: elements EPW own .. EPW own + EPW - 1 (units of the same index), a 32-block per half-wave and pass
                constexpr int EPW = D / 8, NPASS = (EPW + 63) / 64;
                y_seen += 1u;
                lf_wait(plw, l.fl + 2, y_seen);
                const int lane = opq(tid0) & 63;
                int idx[NPASS]; bool val[NPASS]; v4u yv[NPASS];
#pragma unroll
                for (int ps = 0; ps < NPASS; ps++) { const int i = 64 * ps + lane; val[ps] = i < EPW; idx[ps] = EPW * own + (val[ps] ? i : i - 32); }
                for (unsigned spin = 0;; spin++) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int ps = 0; ps < NPASS; ps++) yv[ps] = tg_load(xr, p.u_y + idx[ps]);
                    bool ok = true;
#pragma unroll
                    for (int ps = 0; ps < NPASS; ps++) ok = ok && tg_ok(yv[ps], tagL + S47_Y);
                    if (__all(ok) || plw.dead) break;
                    if (poll_backoff(plw, spin)) break;
                }
                const QVec lq = qvec_at(l.yq, D);
#pragma unroll
                for (int ps = 0; ps < NPASS; ps++) {
                    int qi, isum; float d16, s16;
                    quant_block32(__uint_as_float(yv[ps].x), qi, d16, s16, isum);
                    const int bk = idx[ps] >> 5, e = idx[ps] & 31;
                    if (val[ps]) {
                        lq.q[(e < 16 ? 0 : nb * 16) + bk * 16 + (e & 15)] = (int8_t) qi;
                        if (e == 0) { lq.d[bk] = d16; lq.s[bk] = s16; lq.isum[bk] = isum; }
                    }
                }
            }
            __syncthreads();   // B3: yq
            T47(4);
            {
                const int lane = opq(tid0) & 63;
                float res[GPB];
                rows_sum<FMT, GPB, UD>(wC, nb, lane, qvec_at(l.yq, D), res);
#pragma unroll
                for (int r = 0; r < GPB; r++) xown[r] = xown[r] + res[r];
                x_store(p.u_xatt, tagL + S47_XATT);
            }
            T47(5);
            __syncthreads();   // B4
            T47(6);
            if constexpr (EARLY) { if constexpr (V7) landed(wK[GPB - 1]); else landed(wFr); issue_E(li); issue_A(last ? li : li + 1, !last); }
            pro_run<NIF, NIF, false>(l, pf, sf, sout_l, blk == 0, opq(tid0));
            __syncthreads();   // B5
            T47(7);
            float rgate[GPB];
            {
                const int lane = opq(tid0) & 63;
#pragma unroll
                for (int g = 0; g < GPB; g++) {
                    float res[4];
                    rows_sum<FMT, 4, UD>(wK[g], nb, lane, qvec_at(l.q[0], D), res);
                    const float v = pick_lane<4>(res, lane);
                    const float t = v > 0.0f ? v : 0.0f;
                    if (lane < 4) l.out[32 * g + 4 * own + lane] = t * t;
                    if constexpr (ESTAGE && P47_E_MID && GPB == 2) { if (g == 0) issue_E(li); }   // (the first key group's registers are free: the value rows' last steps go in flight under the second)
                }
                lf_add(l.fl, 1u);
                if constexpr (!V7) rows_sum<FMT, GPB, UD>(wFr, nb, lane, qvec_at(l.q[1], D), rgate);
            }
            T47(8);
            if constexpr (UF >= 4 && !(ESTAGE && P47_E_NOWAIT)) {
                // (long rows: 77 KB per workgroup at 2.9B. Issued before the comm wave has stored this workgroup's key groups they sit in the
                //  CU's memory pipe in front of that store -- and 159 other workgroups wait for it: measured 1.6 us on the slowest)
                kq_seen += 1u;
                lf_wait(plw, l.fl + 1, kq_seen);
            }
            if constexpr (!EARLY && !E_WITH_C && !(ESTAGE && P47_E_MID && GPB == 2)) issue_E(li);
            __syncthreads();   // B6: kq
            T47(9);
            unstage_E();
            {
                const int lane = opq(tid0) & 63, myrow = myrow_of(lane);
                float res[GPB];
                rows_sum<FMT, GPB, UF>(wE, nbF, lane, qvec_at(l.kq, F), res);
#pragma unroll
                for (int r = 0; r < GPB; r++) {
                    if constexpr (V7) xown[r] = xown[r] + res[r];
                    else { const float gte = sigmoid_f(rgate[r]) * res[r]; xown[r] = xown[r] + gte; }
                }
                if (last && !p.logits) { const float xv = pick_lane<GPB>(xown, lane); if (lane < GPB) p.x_out[myrow] = xv; }
                else x_store(p.u_xffn, tagL + S47_XFFN);        // (the last layer's x goes to every workgroup's ln_out when the head follows in this launch)
            }
            T47(10);
            if constexpr (!EARLY) issue_A(last ? li : li + 1, !last);
        }
    }

    // -----------------------------------------------------------------------------------------------------------
    // head workgroup (RWKV-7): comm wave = the head's recurrence (lane = channel), workers = second low-rank stages
    // -----------------------------------------------------------------------------------------------------------
    // 16 rows per pass: lane = 4 row + q, lane q keeps partials 8q .. 8q + 7; one 16-byte load per 32 columns, NS steps at most
    template <int NS> struct HB { int4 r[NS]; };
    template <int NS>
    static __device__ __forceinline__ void hb_issue(HB<NS> & b, const unsigned char * W, long long row, int K, int lane) {
        const int nsteps = K / 32;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int sidx = s < nsteps ? s : nsteps - 1;
            b.r[s] = ldw16(reinterpret_cast<const uint16_t *>(W) + row * K + 32 * sidx + 8 * (lane & 3));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int NS>
    static __device__ __forceinline__ float hb_row(const HB<NS> & b, int K, const float * l_x, int lane) {
        const int nsteps = K / 32, q = lane & 3;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = 0.0f;
        // activation reads in pinned batches of three steps (six 16-byte reads), the next batch under the current one's FMAs
        constexpr int BS = 3, NBAT = (NS + BS - 1) / BS;
        float4 xa[BS][2], xb[BS][2];
        auto rd = [&](float4 (&dst)[BS][2], int bi) {
#pragma unroll
            for (int t = 0; t < BS; t++) {
                const int s = bi * BS + t;
                if (s < NS) {
                    const int sc = s < nsteps ? s : 0;
                    dst[t][0] = *reinterpret_cast<const float4 *>(l_x + 32 * sc + 8 * q);
                    dst[t][1] = *reinterpret_cast<const float4 *>(l_x + 32 * sc + 8 * q + 4);
                }
            }
        };
        auto fm = [&](const float4 (&src)[BS][2], int bi) {
#pragma unroll
            for (int t = 0; t < BS; t++) {
                const int s = bi * BS + t;
                if (s < NS) {
                    if (s < nsteps) {
                        const unsigned uu[4] = {(unsigned) b.r[s].x, (unsigned) b.r[s].y, (unsigned) b.r[s].z, (unsigned) b.r[s].w};
                        const float4 xa4 = src[t][0], xb4 = src[t][1];
                        acc[0] = fma_h_lo(uu[0], xa4.x, acc[0]); acc[1] = fma_h_hi(uu[0], xa4.y, acc[1]); acc[2] = fma_h_lo(uu[1], xa4.z, acc[2]); acc[3] = fma_h_hi(uu[1], xa4.w, acc[3]);
                        acc[4] = fma_h_lo(uu[2], xb4.x, acc[4]); acc[5] = fma_h_hi(uu[2], xb4.y, acc[5]); acc[6] = fma_h_lo(uu[3], xb4.z, acc[6]); acc[7] = fma_h_hi(uu[3], xb4.w, acc[7]);
                    }
                }
            }
        };
        rd(xa, 0);
#pragma unroll
        for (int bi = 0; bi < NBAT; bi += 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (bi + 1 < NBAT) rd(xb, bi + 1);
            __builtin_amdgcn_sched_barrier(0);
            fm(xa, bi);
            __builtin_amdgcn_sched_barrier(0);
            if (bi + 2 < NBAT) rd(xa, bi + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (bi + 1 < NBAT) fm(xb, bi + 1);
        }
        float ps[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float v = acc[e];
            v = v + __int_as_float(lane_xor2_i(__float_as_int(v)));   // ps[i] += ps[i + 16]
            v = v + __int_as_float(lane_xor1_i(__float_as_int(v)));   // ps[i] += ps[i + 8]
            ps[e] = v;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) ps[e] += ps[e + 4];
        return (ps[0] + ps[1]) + (ps[2] + ps[3]);
    }

    // 2 NB columns of the WKV-7 update: ba holds {k, w, b, r} of columns J0 .. J0 + NB - 1 (read by the caller / the previous pair), bb is
    // read here for the next NB under the first NB columns' arithmetic, ba is refilled for the next pair under the second NB
    static constexpr int NBW = 4;
    template <int J0>
    static __device__ __forceinline__ void wkv_pair(float (&s)[S], const float4 * bc, float4 (&ba)[NBW], float4 (&bb)[NBW], float vv, float sa, float & res) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NBW; t++) bb[t] = bc[J0 + NBW + t];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NBW; t++) {
            const float kvj = vv * ba[t].x;
            const float ns = (s[J0 + t] * ba[t].y + kvj) + sa * ba[t].z;
            s[J0 + t] = ns;
            res += ns * ba[t].w;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (J0 + 2 * NBW < S) {
#pragma unroll
            for (int t = 0; t < NBW; t++) ba[t] = bc[J0 + 2 * NBW + t];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NBW; t++) {
            const float kvj = vv * bb[t].x;
            const float ns = (s[J0 + NBW + t] * bb[t].y + kvj) + sa * bb[t].z;
            s[J0 + NBW + t] = ns;
            res += ns * bb[t].w;
        }
    }

    static __device__ __forceinline__ void head_comm(const P47 & p, const Lds & l, int lane0, unsigned base) {
        const int hb = (int) blockIdx.x - NR;
        Poll pl{p.ctl, false};
        const M6Arena ar{p.arena};
        const xrsrc xr = make_xrsrc(p.xch, p.xch_bytes);
        float vf = KI(KLAYER(p, p.l0), layer0) ? 0.0f : p.v_first[hb * S + lane0];
        float s[S], cp[5];
        // per-channel parameters a layer ahead; the state row (64 registers) only after the poll -- it lands under the workers' second stages
        auto issue_cp = [&](int li) {
            const int lane = opq(lane0), c = hb * S + lane;
            const KLayer L = KLAYER(p, li);
            cp[0] = ar.f(KQ(L, k_k))[c]; cp[1] = ar.f(KQ(L, k_a))[c]; cp[2] = ar.f(KQ(L, r_k))[c]; cp[3] = ar.f(KQ(L, lnx_w))[c]; cp[4] = ar.f(KQ(L, lnx_b))[c];
            __builtin_amdgcn_sched_barrier(0);
        };
        // The head's state (64 x 64 floats, contiguous) goes HBM -> LDS at the layer top, long before r / k / v arrive: sixteen coalesced
        // 1 KiB loads (a row per lane straight into registers is 64 lanes x 16 B on 64 different lines per instruction -- measured: the
        // recurrence waited ~4 us for it), parked in a tile with a 68-float row pitch (16-byte reads of a row per lane are conflict-free),
        // so the registers are free for the poll. After H2 lane i reads row i back.
        auto stage_state = [&](int li) {
            __builtin_amdgcn_sched_barrier(0);
            const int lane = opq(lane0);
            const float * st = p.sin + (long long) (li - p.l0) * p.state_stride + 2 * D + (long long) hb * S * S;
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f ta[8], tb[8];   // (two arrays of eight native vectors: one array of sixteen float4 structs was left in scratch)
#pragma unroll
            for (int k = 0; k < 8; k++) { ta[k] = *reinterpret_cast<const v4f *>(st + 256 * k + 4 * lane); tb[k] = *reinterpret_cast<const v4f *>(st + 256 * (8 + k) + 4 * lane); }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                *reinterpret_cast<v4f *>(l.st + (4 * k + (lane >> 4)) * 68 + 4 * (lane & 15)) = ta[k];
                *reinterpret_cast<v4f *>(l.st + (4 * (8 + k) + (lane >> 4)) * 68 + 4 * (lane & 15)) = tb[k];
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        issue_cp(p.l0);
        for (int li = p.l0; li < p.l1; li++) {
            const KLayer L = KLAYER(p, li);
            float * sout_l = p.sout + (long long) (li - p.l0) * p.state_stride;
            const unsigned tagL = base + (unsigned) (li - p.l0) * 8u;
            T47(0);
            stage_state(li);
            float rv, kv0, vv;
#if P47_HEAD_SPLIT
            // The lr1 vector first (fp16-rounded into LDS: what ggml feeds an F16 matrix): the polling waves of the row workgroups publish it a
            // row phase before r / k / v (their jobs run beside the workers' R / K / V rows), and the second low-rank stages need nothing else
            // -- they run on the eight workers while this wave waits for r, k, v of its channels (one unit per lane). Gathered together the
            // second stages started behind the LAST r / k / v row: 2.2 us on the layer's longest chain.
            {
                const int lane = opq(lane0);
                int ptr[NL1]; bool valid[NL1]; v4u dv[NL1];
#pragma unroll
                for (int k = 0; k < NL1; k++) { ptr[k] = p.u_lr1 + lane + 64 * k; valid[k] = lane + 64 * k < KI(L, lr_n); }
                Watch4 wt;
                watch(wt, p, pl, xr, p.u_lr1 + 72 * (lane >> 3) + 5 < p.u_lr1 + KI(L, lr_n) ? p.u_lr1 + 72 * (lane >> 3) + 5 : p.u_lr1, tagL + S47_A);   // (~25 us per layer: eight units of eight jobs first, then the sweep)
                poll_ptrs<NL1>(pl, xr, ptr, valid, tagL + S47_A, dv);
                watch_done(wt);
#pragma unroll
                for (int k = 0; k < NL1; k++) l.lr1[lane + 64 * k] = round_f16(__uint_as_float(dv[k].x));
            }
            T47(1);
            __syncthreads();   // H1
            {
                const int lane = opq(lane0), c = hb * S + lane;
                int ptr[1] = {p.u_a + c}; bool valid[1] = {true}; v4u dv[1];
                Watch4 wt;
                watch(wt, p, pl, xr, p.u_a + hb * S + (lane & ~7) + 3, tagL + S47_A);
                poll_ptrs<1>(pl, xr, ptr, valid, tagL + S47_A, dv);
                watch_done(wt);
                rv = __uint_as_float(dv[0].x); kv0 = __uint_as_float(dv[0].y); vv = __uint_as_float(dv[0].z);
            }
#else
            // r, k, v of this lane's channel (one unit) and the lr1 vector (fp16-rounded into LDS: what ggml feeds an F16 matrix)
            {
                const int lane = opq(lane0), c = hb * S + lane;
                int ptr[NL1 + 1]; bool valid[NL1 + 1]; v4u dv[NL1 + 1];
                ptr[0] = p.u_a + c; valid[0] = true;
#pragma unroll
                for (int k = 0; k < NL1; k++) { ptr[k + 1] = p.u_lr1 + lane + 64 * k; valid[k + 1] = lane + 64 * k < KI(L, lr_n); }
                Watch4 wt;
                watch(wt, p, pl, xr, p.u_a + hb * S + 32, tagL + S47_A);   // (~30 us per layer: r / k / v of one of the head's channels first, then the sweep)
                poll_ptrs<NL1 + 1>(pl, xr, ptr, valid, tagL + S47_A, dv);
                watch_done(wt);
                rv = __uint_as_float(dv[0].x); kv0 = __uint_as_float(dv[0].y); vv = __uint_as_float(dv[0].z);
#pragma unroll
                for (int k = 0; k < NL1; k++) l.lr1[lane + 64 * k] = round_f16(__uint_as_float(dv[k + 1].x));
            }
            T47(1);
            __syncthreads();   // H1
#endif
            __syncthreads();   // H2: the second stages' results are in l.ch
            T47(2);
            const int lane = opq(lane0), c = hb * S + lane;
#pragma unroll
            for (int j = 0; j < S; j += 4) { const float4 q4 = *reinterpret_cast<const float4 *>(l.st + lane * 68 + j); s[j] = q4.x; s[j + 1] = q4.y; s[j + 2] = q4.z; s[j + 3] = q4.w; }
            const float c_kk = cp[0], c_ka = cp[1], c_rk = cp[2], c_lw = cp[3], c_lb = cp[4];
            const float wv = l.ch[lane], av = l.ch[64 + lane], gv = l.ch[128 + lane];
            // ---- key path, value residual (rwkv_graph.inc:432-453) ----
            const float kkr = kv0 * c_kk;
            const float ssum = wave_sum_f(kkr * kkr);
            const float kscale = 1.0f / fmaxf(sqrtf(ssum), 1e-12f);
            const float kk = kkr * kscale;
            const float ka = kv0 * c_ka;
            const float aka = av * ka;
            const float kn = kv0 + (aka - ka);
            if (KI(L, layer0)) { p.v_first[c] = vv; vf = vv; }
            else { const float dv = (vf - vv) * l.ch[192 + lane]; vv = vv + dv; }
            // {k, w, b, r}_j as one 16-byte broadcast read per step, l_a as sixteen; eight reads in flight at a time
            float4 * bc = reinterpret_cast<float4 *>(l.hv);
            float * l_a = l.hv + 256;
            bc[lane] = make_float4(kn, wv, kk * av, rv);
            l_a[lane] = -kk;
            __builtin_amdgcn_wave_barrier();
            // ---- WKV7 (rwkv_operators_wkv_v7.inc:37-107): lane i = value row i of state[h][i][:] ----
            // broadcast reads double-buffered and pinned: a batch is in flight under the arithmetic of the one before it
            float sa = 0.0f;
            {
                const float4 * la4 = reinterpret_cast<const float4 *>(l_a);
                float4 a0[8], a1[8];
#pragma unroll
                for (int t = 0; t < 8; t++) a0[t] = la4[t];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 8; t++) a1[t] = la4[8 + t];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 8; t++) { sa += a0[t].x * s[4 * t]; sa += a0[t].y * s[4 * t + 1]; sa += a0[t].z * s[4 * t + 2]; sa += a0[t].w * s[4 * t + 3]; }
#pragma unroll
                for (int t = 0; t < 8; t++) { sa += a1[t].x * s[32 + 4 * t]; sa += a1[t].y * s[33 + 4 * t]; sa += a1[t].z * s[34 + 4 * t]; sa += a1[t].w * s[35 + 4 * t]; }
            }
            float res = 0.0f;
            {
                float4 ba[NBW], bb[NBW];
#pragma unroll
                for (int t = 0; t < NBW; t++) ba[t] = bc[t];
                wkv_pair<0>(s, bc, ba, bb, vv, sa, res); wkv_pair<8>(s, bc, ba, bb, vv, sa, res); wkv_pair<16>(s, bc, ba, bb, vv, sa, res); wkv_pair<24>(s, bc, ba, bb, vv, sa, res);
                wkv_pair<32>(s, bc, ba, bb, vv, sa, res); wkv_pair<40>(s, bc, ba, bb, vv, sa, res); wkv_pair<48>(s, bc, ba, bb, vv, sa, res); wkv_pair<56>(s, bc, ba, bb, vv, sa, res);
            }
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
            tq_store_block(xr, p.u_y, 2 * hb + (lane >> 5), lane & 31, qi, d16, s16, isum, tagL + S47_Y);
            T47(3);
            {   // the new state, behind the hand-over: back through the tile, sixteen coalesced 1 KiB stores
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < S; j += 4) *reinterpret_cast<float4 *>(l.st + lane * 68 + j) = make_float4(s[j], s[j + 1], s[j + 2], s[j + 3]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                float * so = sout_l + 2 * D + (long long) hb * S * S;
#pragma unroll
                for (int k = 0; k < 16; k++) *reinterpret_cast<float4 *>(so + 256 * k + 4 * lane) = *reinterpret_cast<const float4 *>(l.st + (4 * k + (lane >> 4)) * 68 + 4 * (lane & 15));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the tile is rewritten at the next layer's top)
            }
            issue_cp(li + 1 < p.l1 ? li + 1 : li);
        }
    }

    // Second low-rank stages of one head: 64 rows of w2 / a2 / v2 (K = their ranks, at most 32 HUB2) and of g2 (K up to 32 HUB: 320 at 2.9B).
    // Row-steps, not rows, are what a wave pays: waves 0..3 take sixteen rows of g2 each (one pass), waves 4..7 three passes of sixteen rows
    // out of the twelve of [w2, a2, v2] -- 10 against 9 steps per wave at 2.9B where two waves per matrix ran 20 against 6.
    static __device__ __forceinline__ void head_worker(const P47 & p, const Lds & l, int tid0, int wave) {
        const int hb = (int) blockIdx.x - NR;
        const M6Arena ar{p.arena};
        const bool gw = wave < 4;
        HB<HUB> bg; HB<HUB2> bs[3];
        float e0[3];
        // pass j of a wave of the second kind: matrix (0 w, 1 a, 3 v) and first row
        auto pmtx = [&](int j) { const int pp = 3 * (wave - 4) + j; const int t = pp >> 2; return t == 2 ? 3 : t; };
        auto prow = [&](int j) { const int pp = 3 * (wave - 4) + j; return 16 * (pp & 3); };
        auto issue = [&](int li) {
            const int lane = opq(tid0) & 63;
            const KLayer L = KLAYER(p, li);
            const int has_v = KI(L, has_v);
            auto lr2_of = [&](int m) { return L.q(offsetof(P47Layer, lr2) + 8 * (size_t) __builtin_amdgcn_readfirstlane(m)); };
            auto rank_of = [&](int m) { return L.i(offsetof(P47Layer, rank) + 4 * (size_t) __builtin_amdgcn_readfirstlane(m)); };
            {   // g2 rows (every wave issues: the absent kind loads row 0 of w2 -- one request -- so the issue stays straight-line)
                const long long row = gw ? (long long) hb * S + 16 * wave + (lane >> 2) : 0;
                hb_issue<HUB>(bg, p.arena + lr2_of(gw ? 2 : 0), row, gw ? rank_of(2) : 32, gw ? lane : 0);
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int m = gw ? 0 : pmtx(j);
                const bool has = !gw && (m != 3 || has_v);
                const long long row = has ? (long long) hb * S + prow(j) + (lane >> 2) : 0;
                hb_issue<HUB2>(bs[j], p.arena + lr2_of(has ? m : 0), row, has ? rank_of(m) : 32, has ? lane : 0);
                const long long ci = (long long) hb * S + (gw ? 0 : prow(j)) + (lane >> 2);
                const long long off = (has && m == 1) ? KQ(L, a0) : ((has && m == 3) ? KQ(L, v0) : KQ(L, w0));
                e0[j] = ar.f(off)[ci];
            }
        };
        issue(p.l0);
        for (int li = p.l0; li < p.l1; li++) {
            const KLayer L = KLAYER(p, li);
            const int has_v = KI(L, has_v);
            auto rank_of = [&](int m) { return L.i(offsetof(P47Layer, rank) + 4 * (size_t) __builtin_amdgcn_readfirstlane(m)); };
            auto lbase_of = [&](int m) { return L.i(offsetof(P47Layer, lbase) + 4 * (size_t) __builtin_amdgcn_readfirstlane(m)); };
            __syncthreads();   // H1: lr1 staged
            const int lane = opq(tid0) & 63;
            if (gw) {
                const float v = hb_row<HUB>(bg, rank_of(2), l.lr1 + lbase_of(2), lane);
                if ((lane & 3) == 0) l.ch[2 * 64 + 16 * wave + (lane >> 2)] = v;
            } else {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int m = pmtx(j);
                    if (m != 3 || has_v) {
                        float v = hb_row<HUB2>(bs[j], rank_of(m), l.lr1 + lbase_of(m), lane);
                        if (m == 0) v = det_expf(sigmoid_f(v + e0[j]) * -0.606531f);
                        else v = sigmoid_f(v + e0[j]);
                        if ((lane & 3) == 0) l.ch[m * 64 + prow(j) + (lane >> 2)] = v;
                    }
                }
            }
            __syncthreads();   // H2
            issue(li + 1 < p.l1 ? li + 1 : li);
        }
    }

    // -----------------------------------------------------------------------------------------------------------
    // ln_out + head + argmax inside the launch (rwkv_graph.inc:704-708; k_mvf's order: 32 partials per row, each a chain of FMAs in
    // increasing k on fp16-rounded activations, ggml's fold). Every wave of the grid takes "passes" of eight rows (eight lanes per row,
    // lane q keeps partials 4q .. 4q + 3) in chunks of CH 32-column steps through NHB register buffers. The head does not depend on the
    // token: spare workgroups (CUs the layers do not use: 160 of 256 at D = 768) put their first chunks in flight when the launch starts,
    // the others when their last layer is done -- ESP (+ LPW parked) passes per spare wave are reserved for that (the "spare region" of the
    // pass numbers), the rest is dealt evenly over every wave of the grid.
    // -----------------------------------------------------------------------------------------------------------
    static constexpr int CH = STEPS <= 24 ? 24 : 16, CPP = (STEPS + CH - 1) / CH, NHB = 2;
    static constexpr int ESP = CPP >= NHB ? 1 : NHB / CPP;
    // Short rows (a pass is one chunk, 12 KB at D = 768): a spare wave also parks LPW passes in LDS at the start of the launch -- the head of
    // the 169M model is 77 MB, the registers of 256 workgroups hold 55 MB of it and the row workgroups can only fill theirs when their last
    // layer is done (measured: their 2 x 12 KB per wave took 8.8 us to ISSUE, 96 CUs pulling 18.8 MB at the pace of their outstanding
    // requests). With 3 passes per spare wave settled before the token is known, a row wave is left with one pass and the tail with ~3 MB.
    static constexpr bool PARK = P47_PARK && CPP == 1 && NHB == 2 && D <= 768;
    static constexpr int LPW = PARK ? 1 : 0;
    static_assert(CH == (STEPS <= 24 ? 24 : 16) && (!PARK || l47_park_bytes(D) == (size_t) 9 * LPW * CH * 512), "park area");
    static constexpr int HXB = NHB > 2 ? 4 : 8;   // activation reads per pinned batch
    struct HJ {
        int2 buf[NHB][CH];
        int c1, c2, sw, gw, ns, nwt, p1;      // passes of the spare region (register ones) / the regular region owned by this wave, and where they start
        int nitems;
        int cp;                               // passes of the spare region this wave parked in LDS (they follow its register ones: sw + (ESP + j) ns)
    };
    static __device__ __forceinline__ int hj_pass(const HJ & h, int k) { return k < h.c1 ? h.sw + k * h.ns : h.p1 + h.gw + (k - h.c1) * h.nwt; }
    static __device__ __forceinline__ int hj_parked(const HJ & h, int j) { return h.sw + (ESP + j) * h.ns; }
    static __device__ __forceinline__ void hj_init(HJ & h, const P47 & p, int wave) {
        const int np = (p.V + 7) / 8, blk = blockIdx.x;
        h.ns = p.n_spare * 9; h.nwt = (int) gridDim.x * 9;
        h.gw = blk * 9 + wave;
        h.sw = blk >= NBLK ? (blk - NBLK) * 9 + wave : -1;
        h.p1 = np < (ESP + LPW) * h.ns ? np : (ESP + LPW) * h.ns;
        const int call = (h.sw >= 0 && h.sw < h.p1) ? (h.p1 - h.sw + h.ns - 1) / h.ns : 0;   // this wave's passes of the spare region: the first ESP in registers
        h.c1 = call < ESP ? call : ESP;
        h.cp = call - h.c1;
        h.c2 = h.p1 + h.gw < np ? (np - h.p1 - h.gw + h.nwt - 1) / h.nwt : 0;
        h.nitems = (h.c1 + h.c2) * CPP;
    }
    // chunk `it` of this wave's work into buffer b; past the end: row 0, in range (no control flow around the loads: the waits stay counted)
    template <int B>
    static __device__ __forceinline__ void hj_issue(HJ & h, const P47 & p, int it, int lane) {
        const bool has = it < h.nitems;
        const int k = it / CPP, c = it - k * CPP;
        long long row = has ? (long long) hj_pass(h, k) * 8 + (lane >> 3) : 0;
        if (row >= p.V) row = p.V - 1;
        const uint16_t * base = reinterpret_cast<const uint16_t *>(p.head) + row * D + 4 * (lane & 7);
#pragma unroll
        for (int u = 0; u < CH; u++) {
            int st = c * CH + u; if (st >= STEPS) st = STEPS - 1;
            { typedef int wv2i __attribute__((ext_vector_type(2))); const wv2i t = __builtin_nontemporal_load(reinterpret_cast<const wv2i *>(base + 32 * st)); h.buf[B][u] = make_int2(t.x, t.y); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int B>
    static __device__ __forceinline__ void hj_consume(HJ & h, const P47 & p, const Lds & l, int it, int lane, float (&acc)[4], float & best, int & bi) {
        const int k = it / CPP;
        hj_consume_pass<B>(h, p, l, it < h.nitems, hj_pass(h, k), it - k * CPP, lane, acc, best, bi);
    }
    // chunk c of pass `pass` from register buffer B
    template <int B>
    static __device__ __forceinline__ void hj_consume_pass(HJ & h, const P47 & p, const Lds & l, bool has, int pass, int c, int lane, float (&acc)[4], float & best, int & bi) {
        const int q = lane & 7;
        if (c == 0) { acc[0] = 0.0f; acc[1] = 0.0f; acc[2] = 0.0f; acc[3] = 0.0f; }
#pragma unroll
        for (int u0 = 0; u0 < CH; u0 += HXB) {
            float4 xv[HXB];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < HXB; t++) { int st = c * CH + u0 + t; if (st >= STEPS) st = STEPS - 1; xv[t] = *reinterpret_cast<const float4 *>(l.hx + 32 * st + 4 * q); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < HXB; t++) {
                const unsigned w0 = (unsigned) h.buf[B][u0 + t].x, w1 = (unsigned) h.buf[B][u0 + t].y;
                const float a0 = fma_h_lo(w0, xv[t].x, acc[0]), a1 = fma_h_hi(w0, xv[t].y, acc[1]), a2 = fma_h_lo(w1, xv[t].z, acc[2]), a3 = fma_h_hi(w1, xv[t].w, acc[3]);
                if constexpr (CH * CPP == STEPS) { acc[0] = a0; acc[1] = a1; acc[2] = a2; acc[3] = a3; }     // (every step of every chunk is a step of the row)
                else { const bool on = c * CH + u0 + t < STEPS; acc[0] = on ? a0 : acc[0]; acc[1] = on ? a1 : acc[1]; acc[2] = on ? a2 : acc[2]; acc[3] = on ? a3 : acc[3]; }
            }
        }
        if (c == CPP - 1) {
            // ggml's fold of the 32 partials: ps[i] += ps[i + 16], ps[i] += ps[i + 8], ps[i] += ps[i + 4], (ps0 + ps1) + (ps2 + ps3)
            float ps[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v = acc[e];
                v = v + __int_as_float(lane_xor4_i(__float_as_int(v)));
                v = v + __int_as_float(lane_xor2_i(__float_as_int(v)));
                v = v + __int_as_float(lane_xor1_i(__float_as_int(v)));
                ps[e] = v;
            }
            const float sum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
            const long long row = has ? (long long) pass * 8 + (lane >> 3) : p.V;
            if (q == 0 && row < p.V) {
                p.logits[row] = sum;
                if (sum > best || (sum == best && (int) row < bi)) { best = sum; bi = (int) row; }
            }
        }
    }
    static __device__ __forceinline__ void hj_prefetch(HJ & h, const P47 & p, int lane) {
        hj_issue<0>(h, p, 0, lane);
        if constexpr (NHB > 1) hj_issue<1>(h, p, 1, lane);
        if constexpr (NHB > 2) hj_issue<2>(h, p, 2, lane);
    }
    // spare waves, start of the launch: pass j of the parked ones through register buffer 0 into the wave's slice of the park area
    static __device__ __forceinline__ void hj_park(HJ & h, const P47 & p, const Lds & l, int wave, int lane) {
        if constexpr (PARK) {
#pragma unroll
            for (int j = 0; j < LPW; j++) {
                const bool has = j < h.cp;
                long long row = has ? (long long) hj_parked(h, j) * 8 + (lane >> 3) : 0;
                if (row >= p.V) row = p.V - 1;
                const uint16_t * base = reinterpret_cast<const uint16_t *>(p.head) + row * D + 4 * (lane & 7);
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    int st = u; if (st >= STEPS) st = STEPS - 1;
                    { typedef int wv2i __attribute__((ext_vector_type(2))); const wv2i t = __builtin_nontemporal_load(reinterpret_cast<const wv2i *>(base + 32 * st)); h.buf[0][u] = make_int2(t.x, t.y); }
                }
                __builtin_amdgcn_sched_barrier(0);
                int2 * dst = reinterpret_cast<int2 *>(l.park) + ((wave * LPW + j) * CH) * 64 + lane;
#pragma unroll
                for (int u = 0; u < CH; u++) dst[u * 64] = h.buf[0][u];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    template <int B>
    static __device__ __forceinline__ void hj_unpark(HJ & h, const Lds & l, int wave, int j, int lane) {
        const int2 * src = reinterpret_cast<const int2 *>(l.park) + ((wave * LPW + j) * CH) * 64 + lane;
#pragma unroll
        for (int u = 0; u < CH; u++) h.buf[B][u] = src[u * 64];
        __builtin_amdgcn_sched_barrier(0);
    }
    static __device__ __forceinline__ void am_merge(float & best, int & bi, float ov, int oi) { if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; } }
    static __device__ __forceinline__ void am_wave(float & best, int & bi) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(best, o, WAVE); const int oi = __shfl_xor(bi, o, WAVE); am_merge(best, bi, ov, oi); }
    }

    // every workgroup of the grid, after its layers (spare workgroups: straight away); prefetched = hj_prefetch ran already
    static __device__ __forceinline__ void tail(const P47 & p, const Lds & l, HJ & h, int tid0, int wave, unsigned base) {
        const unsigned tagT = base + (unsigned) (p.l1 - p.l0 - 1) * 8u;
        TT47(11);
        Poll pl{p.ctl, false};
        const M6Arena ar{p.arena};
        const xrsrc xr = make_xrsrc(p.xch, p.xch_bytes);
        if (wave == 8) {   // ln_out statistics on what the last layer published
            const int lane = opq(tid0) & 63;
            float xs[NU][GPB];
            if ((p.calm & 1) && (int) blockIdx.x >= NR) calm_wait<8>(pl, xr, p.u_xffn + (int) (blockIdx.x % (unsigned) (NR * 8)), tagT + S47_XFFN);
            poll_x(pl, xr, p.u_xffn, tagT + S47_XFFN, lane, xs);
            ln_stats(l, lane, xs);
            if (!PARK || (int) blockIdx.x < NBLK) { hj_init(h, p, wave); hj_prefetch(h, p, lane); }   // (PARK: a spare workgroup's polling wave filled its buffers when the launch started)
        }
        __syncthreads();   // T1
        TT47(12);
        {
            const int tid = opq(tid0);
            const float scale = l.sc[0];
            for (int g = tid; g < NG4; g += 576) {
                const float4 xc = *reinterpret_cast<const float4 *>(l.x + 4 * g);
                const float4 w4 = *reinterpret_cast<const float4 *>(ar.f(p.lnout_w) + 4 * g), b4 = *reinterpret_cast<const float4 *>(ar.f(p.lnout_b) + 4 * g);
                const float xs4[4] = {xc.x, xc.y, xc.z, xc.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { const float y = xs4[j] * scale; const float yw = y * ww[j]; o[j] = round_f16(yw + bb[j]); }   // (fp16-rounded: what ggml feeds an F16 matrix)
                *reinterpret_cast<float4 *>(l.hx + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();   // T2
        TT47(13);
        float best = -INFINITY; int bi = 0x7fffffff;
        {
            const int lane = opq(tid0) & 63;
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (PARK) {
                // register passes first (the streamed pass goes in flight behind the first), then the parked ones out of LDS through the
                // buffer the second left free, then whatever was streamed
                hj_consume<0>(h, p, l, 0, lane, acc, best, bi);
                hj_issue<0>(h, p, 2, lane);
                hj_consume<1>(h, p, l, 1, lane, acc, best, bi);
#pragma unroll
                for (int j = 0; j < LPW; j++) {
                    if (j < h.cp) {   // (wave-uniform; the buffer is refilled whole by the issue below)
                        hj_unpark<1>(h, l, wave, j, lane);
                        hj_consume_pass<1>(h, p, l, true, hj_parked(h, j), 0, lane, acc, best, bi);
                    }
                }
                hj_issue<1>(h, p, 3, lane);
                for (int it = 2; it < h.nitems; it += 2) {
                    hj_consume<0>(h, p, l, it, lane, acc, best, bi);
                    hj_issue<0>(h, p, it + 2, lane);
                    hj_consume<1>(h, p, l, it + 1, lane, acc, best, bi);
                    hj_issue<1>(h, p, it + 3, lane);
                }
            } else
            for (int it = 0; it < h.nitems; it += NHB) {
                hj_consume<0>(h, p, l, it, lane, acc, best, bi);
                hj_issue<0>(h, p, it + NHB, lane);
                if constexpr (NHB > 1) {
                    hj_consume<1>(h, p, l, it + 1, lane, acc, best, bi);
                    hj_issue<1>(h, p, it + 1 + NHB, lane);
                }
                if constexpr (NHB > 2) {
                    hj_consume<2>(h, p, l, it + 2, lane, acc, best, bi);
                    hj_issue<2>(h, p, it + 2 + NHB, lane);
                }
            }
        }
        TT47(14);
        // ---- argmax: lanes -> wave -> workgroup -> one tagged unit per workgroup -> workgroup 0 (k_argmax's rule: greatest, then smallest index) ----
        am_wave(best, bi);
        if ((tid0 & 63) == 0) { l.am[wave] = __float_as_int(best); l.am[16 + wave] = bi; }
        __syncthreads();   // T3
        if (wave == 0) {
            const int lane = opq(tid0) & 63;
            float b2 = lane < 9 ? __int_as_float(l.am[lane < 9 ? lane : 0]) : -INFINITY;
            int i2 = lane < 9 ? l.am[16 + (lane < 9 ? lane : 0)] : 0x7fffffff;
            am_wave(b2, i2);
            if (lane == 0) tg_store(xr, p.u_am + (int) blockIdx.x, __float_as_uint(b2), (unsigned) i2, 0u, 0u, tagT + S47_AM);
            if (blockIdx.x == 0) {
                const int G = (int) gridDim.x;
                int ptr[4]; bool valid[4]; v4u dv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { ptr[k] = p.u_am + lane + 64 * k; valid[k] = lane + 64 * k < G; }
                poll_ptrs<4>(pl, xr, ptr, valid, tagT + S47_AM, dv);
                float b3 = -INFINITY; int i3 = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 4; k++) if (valid[k]) am_merge(b3, i3, __uint_as_float(dv[k].x), (int) dv[k].y);
                am_wave(b3, i3);
                // (no element compared greater than -inf: every logit is NaN or -inf -- the token feeds the next embedding lookup and must stay a row of the table)
                if (lane == 0) {
                    const unsigned tokn = i3 == 0x7fffffff ? 0u : (unsigned) i3;
                    if (p.next_tok) p.next_tok[0] = tokn;
                    // greedy loops park a history pointer in the control words (ctl[4..5], position ctl[3]): no copy node per token on the stream
                    TT47(15);
                    if (p.ctl[2] != 0u) {
                        unsigned * hist = reinterpret_cast<unsigned *>((unsigned long long) p.ctl[4] | ((unsigned long long) p.ctl[5] << 32));
                        const unsigned pos = p.ctl[3];
                        if (pos < p.ctl[6]) hist[pos] = tokn;   // (ctl[6]: the history's capacity)
                        p.ctl[3] = pos + 1u;
                    }
                }
            }
        }
    }
};

template <int ARCH, int FMT, int D, int HUB, int NL1, int MAXJ, int HUB2>
__global__ __launch_bounds__(576) void k47_persist(P47 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef K47<ARCH, FMT, D, HUB, NL1, MAXJ, HUB2> K;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const typename K::Lds l = K::carve(smem);
    const unsigned base = p.ctl[0];
    if (tid == 0) { l.fl[0] = 0u; l.fl[1] = 0u; l.fl[2] = 0u; }
    __syncthreads();
    typename K::HJ hj;
    const bool fold_head = p.logits != nullptr;
    if ((int) blockIdx.x < K::NR) {
#ifndef P47_X_NO_ROW_COMM
        if (wave == 8) K::row_comm(p, l, tid & 63, base);
#endif
#ifndef P47_X_NO_ROW_WORKER
        if (wave != 8) { K::row_worker(p, l, tid, wave, base); if (fold_head) { K::hj_init(hj, p, wave); K::hj_prefetch(hj, p, tid & 63); } }
#endif
    } else if ((int) blockIdx.x < K::NBLK) {
        if constexpr (ARCH == 7) {
#ifndef P47_X_NO_HEAD_COMM
            if (wave == 8) K::head_comm(p, l, tid & 63, base);
#endif
#ifndef P47_X_NO_HEAD_WORKER
            if (wave != 8) { K::head_worker(p, l, tid, wave); if (fold_head) { K::hj_init(hj, p, wave); K::hj_prefetch(hj, p, tid & 63); } }
#endif
        }
    } else {
        // spare workgroup: only rows of the head; its waves that do not poll start their stream now
        if (fold_head && (K::PARK || wave != 8)) { K::hj_init(hj, p, wave); K::hj_park(hj, p, l, wave, tid & 63); K::hj_prefetch(hj, p, tid & 63); }
    }
#ifndef P47_X_NO_TAIL
    if (fold_head) {
        // (the polling waves issue theirs behind the poll inside tail(): results return in order per wave)
        K::tail(p, l, hj, tid, wave, base);
    }
#endif
    if (blockIdx.x == 0 && tid == 0) p.ctl[0] = base + (unsigned) (p.l1 - p.l0) * 8u;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

#ifndef P47_CALM_DEFAULT
#define P47_CALM_DEFAULT 1
#endif

struct P47Handle {
    int kind = 3;                 // (mega_v6.hip dispatches on the first member: 1 register prefetch v6, 2 ring v6, 3 this file)
    P47Layer * d_layers = nullptr;
    void * xch = nullptr;
    unsigned * ctl = nullptr;
    unsigned * h_ctl = nullptr;
    P47 proto{};
    long long * trace = nullptr;
    int variant = -1, n_blocks = 0, n_layers = 0, n_cu = 0;
    size_t lds = 0;
    std::vector<uint64_t> layer_bytes;   // algorithmic bytes per layer: every tensor once + the recurrent state read and written
    bool fold_embed = false, fold_head = false;
    uint64_t embed_bytes = 0, head_bytes = 0;
    float * x_out = nullptr;      // p47_set_x_out (pipeline stages, runner.cpp)
};

typedef void (*P47Kernel)(P47);
struct P47Variant { int arch, fmt, D, hub, nl1, maxj, hub2, nblk; P47Kernel fn; };

#define P47_V4(FMT, DD) {4, FMT, DD, 1, 1, 1, 1, K47<4, FMT, DD, 1, 1, 1, 1>::NBLK, k47_persist<4, FMT, DD, 1, 1, 1, 1>}
#define P47_V7(FMT, DD, HUB, NL1, MAXJ, HUB2) {7, FMT, DD, HUB, NL1, MAXJ, HUB2, K47<7, FMT, DD, HUB, NL1, MAXJ, HUB2>::NBLK, k47_persist<7, FMT, DD, HUB, NL1, MAXJ, HUB2>}
#ifndef P47_ONLY
#define P47_ALL(FMT) \
    P47_V4(FMT, 256), P47_V4(FMT, 768), \
    P47_V7(FMT, 256, 4, 4, 2, 2), P47_V7(FMT, 2560, 10, 9, 1, 3)
// (Q8_0 at D = 2560: eight 34-byte-block key rows per wave next to the output rows do not fit 168 registers -- that file keeps the fused launches)
static const P47Variant g_p47[] = {P47_ALL(T_Q4_0), P47_ALL(T_Q4_1), P47_ALL(T_Q5_0), P47_ALL(T_Q5_1),
                                   P47_V4(T_Q8_0, 256), P47_V4(T_Q8_0, 768), P47_V7(T_Q8_0, 256, 4, 4, 2, 2)};
#else    // (register-budget experiments: one instantiation)
static const P47Variant g_p47[] = {P47_ONLY};
#endif

static int p47_variant(const Model & m, int n_cu) {
    if ((m.arch_major != 4 && m.arch_major != 7) || m.layer_end <= m.layer_begin) return -1;
    if (m.arch_major == 7 ? !fused_v7_supported(m) : !fused_v4_supported(m)) return -1;
    const int64_t D = m.n_embed();
    const int fmt = (int) m.header.data_type;
    int lr_total = 0, max_rank = 0, max_rank_wav = 0;
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        if (L.ffn_key->ne[1] != 4 * D) return -1;
        if (m.arch_major != 7) continue;
        if (L.att_w1->type != T_F16) return -1;                       // (the low-rank stages of a file quantised from FP32 stay F32: fused path)
        const DevTensor * l1[4] = {L.att_w1, L.att_a1, L.att_g1, L.att_v1};
        int tot = 0;
        for (int k = 0; k < 4; k++) {
            const int rk = l1[k] ? (int) l1[k]->ne[1] : 0;
            tot += rk;
            if (rk > max_rank) max_rank = rk;
            if (k != 2 && rk > max_rank_wav) max_rank_wav = rk;
        }
        if (tot > lr_total) lr_total = tot;                           // (layer 0 has no v1: the widest layer sizes the buffers)
    }
    for (size_t v = 0; v < sizeof(g_p47) / sizeof(g_p47[0]); v++) {
        const P47Variant & pv = g_p47[v];
        if (pv.arch != m.arch_major || pv.fmt != fmt || pv.D != D || pv.nblk > n_cu) continue;
        if (m.arch_major == 7) {
            const int NR = (int) (D / 8) / ((D / 8 + D / 64 <= 256) ? 1 : 2);
            if (max_rank > 32 * pv.hub || max_rank_wav > 32 * pv.hub2 || lr_total > 64 * pv.nl1 || lr_total > 2048 || lr_total / 4 > pv.maxj * NR) continue;
        }
        return (int) v;
    }
    return -1;
}

void p47_destroy(void * h) {
    P47Handle * g = (P47Handle *) h;
    if (!g) return;
    if (g->d_layers) (void) hipFree(g->d_layers);
    if (g->xch) (void) hipFree(g->xch);
    if (g->ctl) (void) hipFree(g->ctl);
    if (g->h_ctl) (void) hipHostFree(g->h_ctl);
    if (g->trace) (void) hipFree(g->trace);
    delete g;
}

// Returns nullptr when the model / device does not qualify (the caller keeps the fused per-layer launches).
void * p47_create(const Model & m) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m.device) != hipSuccess) return nullptr;
    const int v = p47_variant(m, prop.multiProcessorCount);
    if (v < 0) return nullptr;
    const P47Variant & pv = g_p47[v];
    const int64_t D = m.n_embed(), F = 4 * D;
    const bool v7 = m.arch_major == 7;
    P47Handle * g = new P47Handle();
    g->variant = v; g->n_blocks = pv.nblk; g->n_cu = prop.multiProcessorCount;
    g->lds = l47_lds((int) D, v7).total;
    if (hipFuncSetAttribute((const void *) pv.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) g->lds) != hipSuccess) { delete g; return nullptr; }
    std::vector<P47Layer> hl;
    const unsigned char * abase = (const unsigned char *) m.arena;
    bool in_arena = true;
    auto off = [&](const void * ptr) -> long long {
        const long long o = (const unsigned char *) ptr - abase;
        if (!ptr || o < 0 || (uint64_t) o >= m.arena_bytes) in_arena = false;
        return o;
    };
    auto f = [&](const DevTensor * t) -> long long { return t ? off(t->data) : 0; };
    auto pl3 = [&](const DevTensor * t) { M6Off o; o.qs = off(t->qs); o.qh = t->qh ? off(t->qh) : 0; o.sc = off(t->sc); return o; };
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        P47Layer d{};
        d.ln1_w = f(L.ln1_w); d.ln1_b = f(L.ln1_b); d.ln2_w = f(L.ln2_w); d.ln2_b = f(L.ln2_b);
        d.wr = pl3(L.att_receptance); d.wk = pl3(L.att_key); d.wv = pl3(L.att_value); d.wo = pl3(L.att_output);
        d.fk = pl3(L.ffn_key); d.fv = pl3(L.ffn_value);
        uint64_t bytes = 0;
        std::vector<const DevTensor *> all = {L.ln1_w, L.ln1_b, L.ln2_w, L.ln2_b, L.att_receptance, L.att_key, L.att_value, L.att_output, L.ffn_key, L.ffn_value};
        if (!v7) {
            d.mix_a[0] = f(L.att_time_mix_r); d.mix_a[1] = f(L.att_time_mix_k); d.mix_a[2] = f(L.att_time_mix_v);
            d.mix_f[0] = f(L.ffn_time_mix_k); d.mix_f[1] = f(L.ffn_time_mix_r);
            d.tf = f(L.att_time_first); d.td = f(L.att_time_decay);
            d.fr = pl3(L.ffn_receptance);
            for (const DevTensor * t : {L.att_time_mix_r, L.att_time_mix_k, L.att_time_mix_v, L.ffn_time_mix_k, L.ffn_time_mix_r, L.att_time_first, L.att_time_decay, L.ffn_receptance}) all.push_back(t);
        } else {
            const long long xm = f(L.att_x_rwkvag);   // rows r, w, k, v, a, g
            const int order[6] = {0, 2, 3, 1, 4, 5};
            for (int k = 0; k < 6; k++) d.mix_a[k] = xm + (long long) order[k] * D * 4;
            d.mix_f[0] = f(L.ffn_x_k); d.mix_f[1] = d.mix_f[0];
            d.fr = d.fk;
            const DevTensor * l1[4] = {L.att_w1, L.att_a1, L.att_g1, L.att_v1}, * l2[4] = {L.att_w2, L.att_a2, L.att_g2, L.att_v2};
            int basev = 0;
            for (int k = 0; k < 4; k++) {
                d.lr1[k] = l1[k] ? f(l1[k]) : f(l1[0]); d.lr2[k] = l2[k] ? f(l2[k]) : f(l2[0]);
                d.rank[k] = l1[k] ? (int) l1[k]->ne[1] : 0; d.lbase[k] = basev; basev += d.rank[k];
                if (d.rank[k] % 4 != 0) in_arena = false;
                if (l1[k]) all.push_back(l1[k]);
                if (l2[k]) all.push_back(l2[k]);
            }
            d.has_v = L.att_v1 ? 1 : 0; d.layer0 = i == 0 ? 1 : 0; d.lr_n = basev;
            if (!d.has_v) d.lbase[3] = basev;
            d.w0 = f(L.att_w0); d.a0 = f(L.att_a0); d.v0 = L.att_v0 ? f(L.att_v0) : d.w0;
            d.k_k = f(L.att_k_k); d.k_a = f(L.att_k_a); d.r_k = f(L.att_r_k); d.lnx_w = f(L.att_ln_x_w); d.lnx_b = f(L.att_ln_x_b);
            for (const DevTensor * t : {L.att_x_rwkvag, L.ffn_x_k, L.att_w0, L.att_a0, L.att_v0, L.att_k_k, L.att_k_a, L.att_r_k, L.att_ln_x_w, L.att_ln_x_b}) all.push_back(t);
        }
        for (const DevTensor * t : all) if (t) bytes += t->nbytes;
        bytes += 2 * (uint64_t) m.state_per_layer() * sizeof(float);
        g->layer_bytes.push_back(bytes);
        hl.push_back(d);
    }
    if (!in_arena) { delete g; return nullptr; }
    g->n_layers = (int) hl.size();
    const int64_t nbD = D / 32, nbF = F / 32;
    const int64_t PAD = 2048;   // polls read whole 64-lane rounds: keep every buffer readable past its end
    auto up = [](int64_t x) { return (x + 63) / 64 * 64; };
    const int64_t sizes[7] = {up(D) + PAD, 2048 + PAD, up(3 * nbD > D ? 3 * nbD : D) + PAD, up(D) + PAD, up(3 * nbF) + PAD, up(D) + PAD, 256 + PAD};
    int64_t units = 0;
    for (int64_t z : sizes) units += z;
    bool ok = hipMalloc((void **) &g->d_layers, hl.size() * sizeof(P47Layer)) == hipSuccess
           && hipMemcpy(g->d_layers, hl.data(), hl.size() * sizeof(P47Layer), hipMemcpyHostToDevice) == hipSuccess
           && hipMalloc(&g->xch, (size_t) units * 16) == hipSuccess && hipMemset(g->xch, 0, (size_t) units * 16) == hipSuccess
           && hipMalloc((void **) &g->ctl, 256) == hipSuccess && hipMemset(g->ctl, 0, 256) == hipSuccess   // (ctl[2..5]: the greedy history words)
           && hipHostMalloc((void **) &g->h_ctl, 64, hipHostMallocDefault) == hipSuccess;
    if (ok) { g->h_ctl[0] = 16u; g->h_ctl[1] = 0u; }
    const unsigned init[2] = {16u, 0u};
    ok = ok && hipMemcpy(g->ctl, init, sizeof(init), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { p47_destroy(g); return nullptr; }
    P47 & q = g->proto;
    q.layers = g->d_layers; q.l0 = 0; q.l1 = g->n_layers;
    q.arena = abase;
    q.state_stride = m.state_per_layer();
    q.xch = g->xch; q.xch_bytes = (unsigned) (units * 16);
    int u = 0;
    int * slots[7] = {&q.u_a, &q.u_lr1, &q.u_y, &q.u_xatt, &q.u_kq, &q.u_xffn, &q.u_am};
    for (int i = 0; i < 7; i++) { *slots[i] = u; u += (int) sizes[i]; }
    q.ctl = g->ctl;
    q.V = (int) m.n_vocab();
    // embedding + ln0 and ln_out + head + argmax inside the launch where the stage has them in a dtype the kernel reads (RWKV_MI_P47_NOFOLD=1: measurement aid)
    const char * nofold = getenv("RWKV_MI_P47_NOFOLD");
    const bool fold = !(nofold && nofold[0] == '1');
    if (fold && m.has_embed && m.emb && m.ln0_w && m.ln0_b && (m.emb->type == T_F16 || m.emb->type == T_F32)) {
        g->fold_embed = true;
        q.emb = m.emb->data; q.emb_f16 = m.emb->type == T_F16 ? 1 : 0; q.ln0_w = f(m.ln0_w); q.ln0_b = f(m.ln0_b);
        g->embed_bytes = (uint64_t) D * (m.emb->type == T_F16 ? 2 : 4) + m.ln0_w->nbytes + m.ln0_b->nbytes;
    }
    if (fold && m.has_head && m.head && m.ln_out_w && m.ln_out_b && m.head->type == T_F16 && g->n_cu <= 256) {
        g->fold_head = true;
        q.head = m.head->data; q.lnout_w = f(m.ln_out_w); q.lnout_b = f(m.ln_out_b);
        q.n_spare = g->n_cu - g->n_blocks;
        g->head_bytes = m.head->nbytes + m.ln_out_w->nbytes + m.ln_out_b->nbytes + (uint64_t) m.n_vocab() * 4;
    }
    // long waits (spare and head workgroups) on one unit instead of the full-width poll; RWKV_MI_P47_CALM=0..3 selects which (measurement aid)
    const char * calm = getenv("RWKV_MI_P47_CALM");
    q.calm = calm && calm[0] >= '0' && calm[0] <= '9' ? atoi(calm) : P47_CALM_DEFAULT;
    if (!in_arena) { p47_destroy(g); return nullptr; }
    return g;
}

uint64_t p47_bytes(void * h) { P47Handle * g = (P47Handle *) h; uint64_t s = g->embed_bytes + g->head_bytes; for (uint64_t b : g->layer_bytes) s += b; return s; }

// layers [l0, l1) of the stage (indices into the stage's own layer table); sin / sout: state of layer l0. tok (device): the launch starts
// from LN0(emb[*tok]) instead of x when it begins at the stage's first layer and the handle folds the embedding; logits (device): ln_out +
// head + argmax (into next_tok) run inside the launch when it ends at the stage's last layer and the handle folds the head.
bool p47_folds_embed(void * h) { return ((P47Handle *) h)->fold_embed; }
bool p47_folds_head(void * h) { return ((P47Handle *) h)->fold_head; }
void p47_forward_range(void * h, float * x, float * v_first, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf, int l0, int l1,
                       float * logits, const uint32_t * tok, uint32_t * next_tok) {
    P47Handle * g = (P47Handle *) h;
    P47 q = g->proto;
    q.x = x; q.v_first = v_first; q.sin = sin; q.sout = sout; q.l0 = l0; q.l1 = l1;
    q.x_out = (g->x_out && l1 == g->n_layers) ? g->x_out : x;
    q.tok = (g->fold_embed && l0 == 0) ? tok : nullptr;
    q.logits = (g->fold_head && l1 == g->n_layers) ? logits : nullptr;
    q.next_tok = next_tok;
    const unsigned grid = (unsigned) (q.logits ? g->n_cu : g->n_blocks);
    const P47Kernel fn = g_p47[g->variant].fn;
    if (pf && pf->on) {
        if (pf->used * 2 + 2 > pf->events.size()) {
            hipEvent_t a = nullptr, c = nullptr;
            (void) hipEventCreate(&a); (void) hipEventCreate(&c);
            pf->events.push_back(a); pf->events.push_back(c); pf->bytes.push_back(0);
        }
        uint64_t bytes = (q.tok ? g->embed_bytes : 0) + (q.logits ? g->head_bytes : 0);
        for (int i = l0; i < l1; i++) bytes += g->layer_bytes[(size_t) i];
        pf->bytes[pf->used] = bytes;
        hipExtLaunchKernelGGL(fn, dim3(grid), dim3(576), (uint32_t) g->lds, st, pf->events[pf->used * 2], pf->events[pf->used * 2 + 1], 0, q);
        pf->used++;
    } else {
        hipLaunchKernelGGL(fn, dim3(grid), dim3(576), g->lds, st, q);
    }
}
// greedy loops: the kernel appends every token it picks to hist (device memory, n entries) from position 0; nullptr switches it off
bool p47_set_history(void * h, uint32_t * hist, size_t n, hipStream_t st) {
    P47Handle * g = (P47Handle *) h;
    const unsigned long long a = (unsigned long long) hist;
    const unsigned w[5] = {hist ? 1u : 0u, 0u, (unsigned) (a & 0xFFFFFFFFull), (unsigned) (a >> 32), hist ? (unsigned) (n > 0xFFFFFFFFull ? 0xFFFFFFFFull : n) : 0u};   // (ctl[6]: capacity)
    return hipMemcpyAsync(g->ctl + 2, w, sizeof(w), hipMemcpyHostToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
}
int p47_layers(void * h) { return ((P47Handle *) h)->n_layers; }

void p47_set_x_out(void * h, float * x_out) { ((P47Handle *) h)->x_out = x_out; }

bool p47_ctl_fetch(void * h, hipStream_t st) {
    P47Handle * g = (P47Handle *) h;
    return hipMemcpyAsync(g->h_ctl, g->ctl, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st) == hipSuccess;
}
unsigned * p47_ctl(void * h) { return ((P47Handle *) h)->ctl; }
bool p47_aborted_cached(void * h) { return ((P47Handle *) h)->h_ctl[1] != 0; }
unsigned p47_generation_cached(void * h) { return ((P47Handle *) h)->h_ctl[0]; }
bool p47_clear_abort(void * h, hipStream_t st) {
    P47Handle * g = (P47Handle *) h;
    g->h_ctl[1] = 0u;
    return hipMemsetAsync(g->ctl + 1, 0, sizeof(unsigned), st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
}
bool p47_set_tag(void * h, unsigned base, hipStream_t st) {
    P47Handle * g = (P47Handle *) h;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    return hipMemcpy(g->ctl, &base, sizeof(unsigned), hipMemcpyHostToDevice) == hipSuccess;
}
// real-time-counter stamps of one layer (16 per wave, 9 waves per workgroup) for the next launches; out holds n_blocks * 9 * 16 values
bool p47_trace(void * h, int layer, long long * out, bool fetch) {
    P47Handle * g = (P47Handle *) h;
    const size_t n = (size_t) 256 * 9 * 16;
    if (!g->trace) { if (hipMalloc((void **) &g->trace, n * 8) != hipSuccess) return false; (void) hipMemset(g->trace, 0, n * 8); }
    g->proto.trace = g->trace; g->proto.trace_layer = layer;
    if (fetch) return hipMemcpy(out, g->trace, n * 8, hipMemcpyDeviceToHost) == hipSuccess;
    return true;
}

}  // namespace rwkvmi
