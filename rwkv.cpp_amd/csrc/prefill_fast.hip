// prefill_fast.hip -- the quantised projections of sequence mode (every ggml_mul_mat of rwkv_build_sequential_graph, rwkv_graph.inc:744-866,
// with T >= 64 columns) as an int8 GEMM on v_mfma_i32_32x32x32_i8 whose block sums are accumulated in PLAIN K ORDER, one f32 accumulator
// per output: the opt-in arm of sequence mode for quantised matrices (RWKV_MI_SEQ_Q=fast; the default keeps k_mmq_mfma of prefill.hip, bit-identical
// to the serial path -- round 5 shipped this arm as the default, round 6 turned that around: ADVICE.md, `chunking must not change results`).
//
// What is kept of ggml's product: the quantised operands (Q8_0 / Q8_1 activations per 32-block, the weight codes), the exact integer block
// sums, f32 accumulation of d_w d_x isum (+ m_w s_x). What is not: the ORDER of the f32 additions (ggml / the single-token kernel: 64
// partial sums b mod 64 and a halving tree; here: increasing b) and the association of the scale product ((d_w isum) d_x instead of
// (d_w d_x) isum: one rounding each way). Per product that is inside 2e-6 sum_b |term_b| of the float64 sum, like the oracle itself
// (tests/test_gpu_prefill_fast.py). The contract is the reference's own: it promises serial == sequence for FP32 files
// (tests/test_eval_sequence_in_chunks.c:54 runs on one) and validates quantised formats against recorded thresholds
// (tests/test_tiny_rwkv.c:70-134) -- the bit-exact walk of prefill.hip paid every launch for an invariant nobody asked for (67 VALU
// instructions per MFMA, matrix pipe 4.7 % busy: DESIGN.md 6.5b).
//
// Shape: 256-thread workgroup (4 waves, one per SIMD; two workgroups fit a CU) = 128 rows x 64 tokens; wave (rg, tg) = rows [64 rg, +64)
// x tokens [32 tg, +32): TWO MFMAs per block share the token operand, its scales and the chunk protocol. Per block and wave: 2 x (8
// unpack + 8 packed fma  g = d_w (isum' - magic)  + 8 packed fma  acc += g d_x) -- the integer -> float conversion rides on the first
// fma: the MFMA accumulates onto the bit pattern of 1.5 * 2^23, so its output read as a float IS magic + isum, and d_w * magic is exact
// (11-bit x 2-bit mantissas). Staging as in prefill.hip: chunks of 8 blocks, LDS-DMA (global_load_lds_dwordx4, 1 KiB rows) into two
// buffers, one workgroup barrier per chunk; the images are prefill.hip's (tile-major weights, MFMA-operand-order activation tiles).
#include "prefill_mm.h"

#include <cstdlib>
#include <mutex>

namespace rwkvmi {

template <int FMT> struct MG {
    typedef MF<FMT> M;                       // the per-step slot layout (128 rows, two token tiles) is the exact kernel's
// (measured, 1.6B Q4_0, 1024-token pass, same box: two buffers of 8 blocks = 78 KB, two workgroups per CU: 86.1 k tokens/s; three buffers,
//  one workgroup per CU: 76.0 k -- the second workgroup hides more than the deeper prefetch; the exact kernel: 73.2 k)
#ifndef PFF_NBUF
#define PFF_NBUF 2
#endif
    static constexpr int NT = 256, CH = 8;
    static constexpr int NBUF = PFF_NBUF * CH * M::SLOT <= 160 * 1024 ? PFF_NBUF : 2;   // chunk buffers: chunk k + NBUF - 1 is issued when chunk k starts (Q8_0: two fit)
    static constexpr int DMA_PER_CHUNK = 2 * M::n_dma(0);                   // DMA instructions of one wave per chunk (its two steps): the vmcnt share
    static constexpr int LDS_BYTES = NBUF * CH * M::SLOT;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");   // (Q4_0 / Q4_1: 78 - 82 KB, two workgroups per CU; Q5 / Q8_0: one)
};

struct FastArgs {
    PfW w[MMQ_BATCH]; PfX x[MMQ_BATCH]; float * y[MMQ_BATCH]; Epi epi[MMQ_BATCH];
    int64_t N, T, ldy;
    int nb, RT, C;
    int dbg;   // RWKV_MI_PFF_DBG (timing-only experiments, results INVALID): 1 no MFMAs / folds, 2 no staging behind the first chunks, 4 no chunk barrier
};

template <int FMT>
__global__ __launch_bounds__(256, 2) void k_mmq_fast(FastArgs A) {
    typedef MF<FMT> M;
    typedef MG<FMT> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int64_t N = A.N, T = A.T, ldy = A.ldy;
    const int nb = A.nb, RT = A.RT, C = A.C;
    // XCD-aware tile map (block id % 8 = XCD): the token tiles of one 128-row panel share an XCD's L2; dense map below 8 panels
    const int RP = (RT + 3) / 4;
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int rp = RP < 8 ? (int) blockIdx.x % RP : (kk / C) * 8 + xcd;
    const int ct = RP < 8 ? (int) blockIdx.x / RP : kk % C;
    if (rp * 4 >= RT) return;
    const int bz = blockIdx.y;
    const PfW w = A.w[bz];
    const PfX x = A.x[bz];
    float * __restrict__ const y = A.y[bz];
    const Epi epi = A.epi[bz];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave & 1, tg = wave >> 1;
    const int nn = lane & 31, h = lane >> 5;
    const int rt0 = rp * 4, tt0 = ct * 2;

    // ---- staging (prefill.hip's, with the blocks in plain order): wave w stages steps w and w + 4 of a chunk ----
    const unsigned lds_base = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) lds;
    auto dma_s = [&](const unsigned char * sbase, unsigned voff, unsigned dst) {
        const unsigned long long sb = (unsigned long long) sbase;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) sb), hi = __builtin_amdgcn_readfirstlane((unsigned) (sb >> 32));
        const unsigned long long sbu = ((unsigned long long) hi << 32) | lo;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbu), "s"(dst) : "memory");
    };
    auto dma_v = [&](const unsigned char * src, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto stage_step = [&](int k, int st) {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int64_t b = (int64_t) k * G::CH + st;       // (nb is a multiple of CH: every step of every chunk is a block of the row)
        const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned) ((G::CH * (k % G::NBUF) + st) * M::SLOT));
#pragma unroll
        for (int r = 0; r < M::NR; r++) {
            const unsigned dst = dst0 + r * 1024;
            if (r < M::NW) {
                if constexpr (M::Q8) {
                    int rt = rt0 + r; rt = rt < RT ? rt : RT - 1;
                    dma_s(w.qs + ((int64_t) rt * nb + b) * 1024, (unsigned) lane_o * 16, dst);
                } else {
                    int rta = rt0 + 2 * r, rtb = rta + 1;
                    rta = rta < RT ? rta : RT - 1; rtb = rtb < RT ? rtb : RT - 1;
                    const unsigned up = (unsigned) (rtb - rta) * (unsigned) nb * 512u;
                    dma_s(w.qs + ((int64_t) rta * nb + b) * 512, (unsigned) (lane_o & 31) * 16 + ((lane_o >> 5) ? up : 0u), dst);
                }
            } else if (r < M::NW + 2) {
                const int t2 = r - M::NW;
                dma_s(reinterpret_cast<const unsigned char *>(x.q) + ((int64_t) (tt0 + t2) * nb + b) * 1024, (unsigned) lane_o * 16, dst);
            } else {
                const int q = lane_o + 64 * (r - M::NW - 2);
                const int part = q & 7, lo = 64 * (r - M::NW - 2), hi = lo + 64;
                auto seg = [&](int first, int count, const void * base, bool weights) {
                    if (count == 0 || first >= hi || first + count <= lo) return;
                    if (q >= first && q < first + count) {
                        const int e = (q - first) >> 3;
                        int64_t tile;
                        if (weights) { int rt = rt0 + e; rt = rt < RT ? rt : RT - 1; tile = rt; } else tile = tt0 + e;
                        dma_v(reinterpret_cast<const unsigned char *>(base) + (tile * nb + b) * 128 + part * 16, dst);
                    }
                };
                seg(0, M::P_WSC, w.sc, true);
                seg(M::P_WSC, M::P_WQH, w.qh, true);
                seg(M::P_WSC + M::P_WQH, M::P_XD, x.d, false);
                seg(M::P_WSC + M::P_WQH + M::P_XD, M::P_XS, x.s, false);
                seg(M::P_WSC + M::P_WQH + M::P_XD + M::P_XS, M::P_XO, x.o, false);
            }
        }
    };
    auto issue = [&](int k) { stage_step(k, wave); stage_step(k, wave + 4); };

    constexpr int   MAGIC_I = 0x4B400000;
    constexpr float MAGIC_F = 12582912.0f;
    v16i magic;
#pragma unroll
    for (int r = 0; r < 16; r++) magic[r] = MAGIC_I;
    asm volatile("" : "+v"(magic));
    int nib_sh = h ? 0 : 4;
    asm volatile("" : "+v"(nib_sh));

    // per-lane LDS address parts of this wave's operands inside a slot
    unsigned a_b[2], a_sc[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int rtl = 2 * rg + i;                                  // row tile of the workgroup's four
        a_b[i] = (unsigned) (M::OFF_WC + (M::Q8 ? (rtl * 64 + h * 32 + nn) : (rtl * 32 + nn)) * 16);
        a_sc[i] = (unsigned) (M::OFF_WSC + (rtl * 32 + nn) * 4);
    }
    const unsigned a_aop = (unsigned) (M::OFF_XQ + (tg * 64 + lane) * 16);
    const unsigned a_xd = (unsigned) (M::OFF_XD + (tg * 32 + 4 * h) * 4);

    v2f cur[2][8];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) cur[i][j] = (v2f){0.0f, 0.0f};

    auto unpack = [&](const int4 & braw, unsigned qhw) -> v4i {
        v4i bop;
        const int raw[4] = {braw.x, braw.y, braw.z, braw.w};
        if constexpr (FMT == T_Q8_0) {
            bop[0] = raw[0]; bop[1] = raw[1]; bop[2] = raw[2]; bop[3] = raw[3];
        } else if constexpr (FMT == T_Q4_0) {
            // signed (q - 8) in the HIGH nibble: the MFMA returns 16 x the block sum, the token scales carry 1 / 16 (both exact)
#pragma unroll
            for (int i = 0; i < 4; i++) { const int t = raw[i] << nib_sh; bop[i] = (t & (int) 0xF0F0F0F0) ^ (int) 0x80808080; }
        } else if constexpr (FMT == T_Q4_1) {
#pragma unroll
            for (int i = 0; i < 4; i++) bop[i] = (raw[i] >> (4 * h)) & 0x0F0F0F0F;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned nl = (qhw >> (16 * h + 4 * i)) & 0xFu;
                bop[i] = ((raw[i] >> (4 * h)) & 0x0F0F0F0F) | (int) (((nl * 0x00204081u) & 0x01010101u) << 4);
            }
        }
        return bop;
    };

    // Software pipeline over the blocks of a chunk, three stages deep: while the fold of block s runs on the VALU (32 packed FMAs), the two
    // MFMAs of block s + 1 run on the matrix pipe and the LDS reads of block s + 2 (operands) and s + 1 (token scales) are in flight -- every
    // LDS read is issued a stage before its use (with one wave per SIMD a read that is waited for right away costs its whole latency: the
    // first version of this kernel spent 1100 cycles per block on 50 instructions). A block in flight = its integer sums (two 32 x 32 tiles)
    // and the two weight scales.
    struct Ops { v4i aop; int4 braw[2]; unsigned scw[2], qhw[2]; };
    struct Blk { v16i acc[2]; float dw[2], mw[2]; };
    struct Sc { v2f dd[8], aux[8]; };
    auto ld_ops = [&](Ops & o, unsigned off) {
        const unsigned char * S = lds + off;
        o.aop = *reinterpret_cast<const v4i *>(S + a_aop);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            o.braw[i] = *reinterpret_cast<const int4 *>(S + a_b[i]);
            o.scw[i] = *reinterpret_cast<const unsigned *>(S + a_sc[i]);
            if constexpr (M::QH) o.qhw[i] = *reinterpret_cast<const unsigned *>(S + a_sc[i] + (M::OFF_WQH - M::OFF_WSC)); else o.qhw[i] = 0u;
        }
    };
    auto ld_sc = [&](Sc & c, unsigned off) {
        const unsigned char * S = lds + off;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 dx4 = *reinterpret_cast<const float4 *>(S + a_xd + 32 * g);
            c.dd[2 * g] = (v2f){dx4.x, dx4.y}; c.dd[2 * g + 1] = (v2f){dx4.z, dx4.w};
            if constexpr (M::HM || M::XO) {
                const float4 a4 = *reinterpret_cast<const float4 *>(S + a_xd + ((M::HM ? M::OFF_XS : M::OFF_XO) - M::OFF_XD) + 32 * g);
                c.aux[2 * g] = (v2f){a4.x, a4.y}; c.aux[2 * g + 1] = (v2f){a4.z, a4.w};
            }
        }
    };
    auto launch = [&](Blk & q, const Ops & o) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            q.dw[i] = h2f_bits((uint16_t) (o.scw[i] & 0xFFFFu));
            q.mw[i] = M::HM ? h2f_bits((uint16_t) (o.scw[i] >> 16)) : 0.0f;
            q.acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.aop, unpack(o.braw[i], o.qhw[i]), magic, 0, 0, 0);
        }
    };
    auto fold = [&](const Blk & q, const Sc & c) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const v2f dw2 = {q.dw[i], q.dw[i]};
            const float cdw = -MAGIC_F * q.dw[i];                    // exact: an fp16 value times 1.5 * 2^23
            const v2f cdw2 = {cdw, cdw};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const v2f raw = {__int_as_float(q.acc[i][2 * j]), __int_as_float(q.acc[i][2 * j + 1])};
                v2f g;
                if constexpr (M::XO) { const v2f sf = (raw - (v2f){MAGIC_F, MAGIC_F}) - c.aux[j]; g = sf * dw2; }    // (exact integers below 2^24, then one rounding)
                else g = __builtin_elementwise_fma(raw, dw2, cdw2);                                                  // = d_w * isum', one rounding
                cur[i][j] = __builtin_elementwise_fma(g, c.dd[j], cur[i][j]);
                if constexpr (M::HM) cur[i][j] = __builtin_elementwise_fma((v2f){q.mw[i], q.mw[i]}, c.aux[j], cur[i][j]);
            }
        }
    };

    const int n_chunks = nb / G::CH;
#pragma unroll
    for (int k0 = 0; k0 < G::NBUF - 1; k0++) if (k0 < n_chunks) issue(k0);
    Blk qa, qb;
    Ops oa, ob;
    Sc sa, sb;
#define PF_PIN() __builtin_amdgcn_sched_barrier(0)
#pragma unroll 1
    for (int k = 0; k < n_chunks; k++) {
        // this wave's share of chunk k has landed when only the younger chunks' DMAs are outstanding (vmcnt counts this wave's, in order) ...
        if (G::NBUF > 2 && k + G::NBUF - 2 < n_chunks) wait_vm<G::DMA_PER_CHUNK * (G::NBUF - 2)>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(A.dbg & 4)) __syncthreads();                           // ... everybody's has, and nobody still reads the buffer the next issue goes into
        if (k + G::NBUF - 1 < n_chunks && !((A.dbg & 2) && k >= 1)) issue(k + G::NBUF - 1);
        if (A.dbg & 1) continue;
        const unsigned cb = (unsigned) (G::CH * (k % G::NBUF)) * M::SLOT;
        ld_ops(oa, cb); ld_sc(sa, cb); ld_ops(ob, cb + M::SLOT);
        PF_PIN();
        launch(qa, oa);
        PF_PIN();
#pragma unroll
        for (int s = 0; s < G::CH; s += 2) {
            // block s: sums in qa, scales in sa; block s + 1: operands in ob
            launch(qb, ob);
            PF_PIN();
            if (s + 2 < G::CH) ld_ops(oa, cb + (unsigned) (s + 2) * M::SLOT);
            ld_sc(sb, cb + (unsigned) (s + 1) * M::SLOT);
            PF_PIN();
            fold(qa, sa);
            PF_PIN();
            // block s + 1: sums in qb, scales in sb; block s + 2: operands in oa
            if (s + 2 < G::CH) launch(qa, oa);
            PF_PIN();
            if (s + 3 < G::CH) ld_ops(ob, cb + (unsigned) (s + 3) * M::SLOT);
            if (s + 2 < G::CH) ld_sc(sa, cb + (unsigned) (s + 2) * M::SLOT);
            PF_PIN();
            fold(qb, sb);
            PF_PIN();
        }
    }
#undef PF_PIN

    // ---- epilogue: row n = column of the tile (lane), tokens along the registers ----
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int64_t n = (int64_t) (rt0 + 2 * rg + i) * 32 + nn;
        if (n < N) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t t = (int64_t) (tt0 + tg) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (t < T) y[t * ldy + n] = apply_epi(epi, cur[i][r >> 1][r & 1], t, n, ldy);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

std::atomic<unsigned long long> g_mmq_fast_launches{0};   // launches of k_mmq_fast by this process (tests assert the arm they mean ran)

// RWKV_MI_SEQ_Q = exact (default since round 6) | fast | force (tests: the plain-order kernel also on shapes with too few tiles to be worth
// it). Read per call: the test suite runs the arms in one process. The default is the arm that keeps rwkv_eval_sequence bit-identical to
// repeated rwkv_eval whatever the chunking (ggml's mul_mat runs the same vec_dot per column for every T, so the reference has that property
// for quantised files too); the plain-order kernel is an opt-in for callers that take its stated tolerance for +15 % prefill throughput.
int seq_q_arm() {
    const char * e = getenv("RWKV_MI_SEQ_Q");
    if (e && e[0] == 'f' && e[1] == 'a') return 1;
    if (e && e[0] == 'f' && e[1] == 'o') return 2;
    return 0;
}

static std::mutex g_fast_mu;
void mmq_fast_prepare_current_device() {
    static uint64_t ready[4] = {0, 0, 0, 0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) return;
    std::lock_guard<std::mutex> lk(g_fast_mu);
    if (ready[dev >> 6] & (1ull << (dev & 63))) return;
    ready[dev >> 6] |= 1ull << (dev & 63);
    (void) hipFuncSetAttribute((const void *) k_mmq_fast<T_Q4_0>, hipFuncAttributeMaxDynamicSharedMemorySize, MG<T_Q4_0>::LDS_BYTES);
    (void) hipFuncSetAttribute((const void *) k_mmq_fast<T_Q4_1>, hipFuncAttributeMaxDynamicSharedMemorySize, MG<T_Q4_1>::LDS_BYTES);
    (void) hipFuncSetAttribute((const void *) k_mmq_fast<T_Q5_0>, hipFuncAttributeMaxDynamicSharedMemorySize, MG<T_Q5_0>::LDS_BYTES);
    (void) hipFuncSetAttribute((const void *) k_mmq_fast<T_Q5_1>, hipFuncAttributeMaxDynamicSharedMemorySize, MG<T_Q5_1>::LDS_BYTES);
    (void) hipFuncSetAttribute((const void *) k_mmq_fast<T_Q8_0>, hipFuncAttributeMaxDynamicSharedMemorySize, MG<T_Q8_0>::LDS_BYTES);
    (void) hipGetLastError();
}

template <int FMT>
static bool launch_fast_t(int n, const DevTensor * const * Ws, const TileAct * xs, float * const * ys, const Epi * epis, int64_t T, int64_t ldy, hipStream_t st) {
    const DevTensor & W0 = *Ws[0];
    const int64_t N = W0.rows();
    const int nb = (int) (W0.cols() / 32);
    for (int i = 0; i < n; i++) if (!Ws[i]->pf_qs && !ensure_pf(*Ws[i], st)) return false;
    FastArgs A;
    for (int i = 0; i < MMQ_BATCH; i++) {
        const int j = i < n ? i : 0;
        A.w[i] = PfW{Ws[j]->pf_qs, Ws[j]->pf_sc, Ws[j]->pf_qh};
        A.x[i] = PfX{xs[j].q, xs[j].d, xs[j].s, xs[j].o};
        A.y[i] = ys[j];
        A.epi[i] = epis[j];
    }
    const int RT = (int) ((N + 31) / 32), RP = (RT + 3) / 4, C = (int) ((T + 63) / 64);
    A.N = N; A.T = T; A.ldy = ldy; A.nb = nb; A.RT = RT; A.C = C;
    { const char * e = getenv("RWKV_MI_PFF_DBG"); A.dbg = e ? atoi(e) : 0; }
    mmq_fast_prepare_current_device();
    const dim3 grid((unsigned) (RP < 8 ? RP * C : ((RP + 7) / 8) * 8 * C), (unsigned) n, 1);
    hipLaunchKernelGGL((k_mmq_fast<FMT>), grid, dim3(MG<FMT>::NT), (size_t) MG<FMT>::LDS_BYTES, st, A);
    g_mmq_fast_launches.fetch_add(1, std::memory_order_relaxed);
    return true;
}

// false: not this kernel's shape (rows of fewer than 8 blocks or not a multiple of 8; too few output tiles to fill the chip without
// cutting K -- the exact kernel cuts its walk for those) or the exact arm is asked for
bool launch_mmq_fast(int n, const DevTensor * const * Ws, const TileAct * xs, float * const * ys, const Epi * epis, int64_t T, int64_t ldy, hipStream_t st) {
    const int arm = seq_q_arm();
    if (arm == 0 || n < 1 || n > MMQ_BATCH) return false;
    const int64_t N = Ws[0]->rows(), K = Ws[0]->cols();
    const int64_t nb = K / 32, tiles = (int64_t) n * ((N + 127) / 128) * ((T + 63) / 64);
    if (K % 256 != 0 || nb < 16 || N < 128 || (tiles < 128 && arm != 2)) return false;
    switch (Ws[0]->type) {
        case T_Q4_0: return launch_fast_t<T_Q4_0>(n, Ws, xs, ys, epis, T, ldy, st);
        case T_Q4_1: return launch_fast_t<T_Q4_1>(n, Ws, xs, ys, epis, T, ldy, st);
        case T_Q5_0: return launch_fast_t<T_Q5_0>(n, Ws, xs, ys, epis, T, ldy, st);
        case T_Q5_1: return launch_fast_t<T_Q5_1>(n, Ws, xs, ys, epis, T, ldy, st);
        case T_Q8_0: return launch_fast_t<T_Q8_0>(n, Ws, xs, ys, epis, T, ldy, st);
        default: return false;
    }
}

}  // namespace rwkvmi
