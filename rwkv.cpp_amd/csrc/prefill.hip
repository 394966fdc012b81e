// prefill.hip -- sequence mode (rwkv_eval_sequence, T tokens at once) on the matrix cores:
//
//   k_mmq_mfma<FMT>    every quantised projection of the sequence graph (rwkv_build_sequential_graph, rwkv_graph.inc:744-866:
//                      each ggml_mul_mat with T columns) as a dense int8 GEMM on v_mfma_i32_32x32x32_i8. K = 32 of the
//                      instruction is exactly one quantisation block, so an MFMA returns the EXACT integer block sums of a
//                      32-token x 32-row tile; the block is then folded into the f32 accumulators with the same statement
//                      as the single-token kernel (kdev.h blk_fma):  P = fma(d_w * d_x, isum, P) [+ fma(m_w, s_x, P)].
//   k_mmq_combine      second half of a GEMM whose walk was cut into parts (too few output tiles for 256 CUs): adds the parts in
//                      tree order and applies the epilogue.
//   k_quant_act_tiles  the activation quantiser of ggml's mul_mat (Q8_0 / Q8_1) writing the tile-major image the GEMM reads (up to
//                      five inputs per launch); k_v6_mix2_seq / k_mix_seq_q: the RWKV-6 mixes with that quantiser fused in.
//   k_pf_repack        load-/first-use-time copy of a quantised matrix into the tile-major image the GEMM reads.
//   k_wkv6_seq         the WKV-5/6 recurrence with the T loop pipelined across the 64 lanes of a wave (one wave = two value
//                      columns in packed-f32 registers instead of one wave per head).
//
// Bit-exactness with the single-token path (the reference guarantees serial == sequence and tests it with memcmp,
// tests/test_eval_sequence_in_chunks.c:54; here it holds for every format). A row sum of k_mvq_t1 is a fixed expression:
// 64 partials P[l], l = b mod 64, each a chain over its blocks in increasing b, then the xor-butterfly 32, 16, 8, 4, 2, 1 --
// a binary tree whose root splits the leaves by bit 0 of l, the next level by bit 1, ... A depth-first walk of that tree
// visits the leaves in bit-reversed order l = 0, 32, 16, 48, 8, ..., so the GEMM walks the K dimension in that order and
// needs a stack of only six partial sums per output (plus the running leaf): after leaf number c, as many merges as c has
// trailing one bits. Every output element therefore sees exactly the additions of the single-token kernel, in the same order.
// Leaves without blocks (l >= K/32) are +0.0 there (acc + 0 is the identity on these values) and +0.0 here.
//
// Memory. A depth-first walk touches blocks 0, 64, ..., 32, 96, ...: in the row-major planes of the decode path that is one
// 16-byte piece per 128-byte line and eight fetches of every line. The GEMM reads a second image of the matrix instead
// ("pf": per 32-row tile and block one contiguous 512-byte run of codes, 128 bytes of scales), built on the GPU the first
// time a matrix is used with T >= 32; 288 GB of HBM make the second copy of the quantised matrices a non-issue, and the
// decode path keeps its own layout. Activations are quantised straight into the matching image (per 32-token tile and
// block 1 KiB of codes in MFMA operand order + 128 bytes of scales).
//
// Work decomposition: a 512-thread workgroup (8 waves, two per SIMD) owns a 128-row x 64-token output tile; wave (rg, tg)
// owns rows [32 rg, +32) x tokens [32 tg, +32). The K walk is cut into chunks of 8 steps (blocks); wave w stages step w of a chunk
// with LDS-DMA (scalar base + per-lane offset per 1-KiB row) into one of two LDS buffers while the previous chunk is computed, one
// workgroup barrier per chunk. Per step and wave: the sums of the previous MFMA leave the accumulator as floats (the MFMA
// accumulates onto 1.5 * 2^23: one packed subtract per pair), the next MFMA is issued under the fold of the current block, LDS
// reads run a stage ahead: ~52 VALU + ~18 SALU + 7 LDS instructions per MFMA -- the kernel is bound by instruction issue (the f32
// fold per block is what ggml's arithmetic prescribes). DESIGN.md section 6.5 has the measurements and what was tried.
#include "prefill_mm.h"

#include <mutex>
#include <type_traits>

namespace rwkvmi {

// ---------------------------------------------------------------------------------------------------------------
// tile-major images
// ---------------------------------------------------------------------------------------------------------------

// weights: row tile rt (32 rows), block b:
//   qs  Q4/Q5: [rt][b][n][16 B]        Q8_0: [rt][b][half][n][16 B]      (n = row within the tile)
//   sc  [rt][b][n] u32 = fp16 d | fp16 m << 16 (m = 0 for the formats without one)
//   qh  [rt][b][n] u32 (Q5 only)
__global__ __launch_bounds__(256) void k_pf_repack(int type, const uint8_t * __restrict__ qs, const uint32_t * __restrict__ qh, const void * __restrict__ sc,
                                                   int64_t N, int nb, uint8_t * __restrict__ pq, uint32_t * __restrict__ psc, uint32_t * __restrict__ pqh) {
    const int64_t RT = (N + 31) / 32;
    const int64_t idx = (int64_t) blockIdx.x * 256 + threadIdx.x;   // over (rt, b, n)
    if (idx >= RT * nb * 32) return;
    const int n = (int) (idx & 31);
    const int64_t rb = idx >> 5;
    const int b = (int) (rb % nb);
    const int64_t rt = rb / nb;
    const int64_t row = rt * 32 + n;
    const bool valid = row < N;
    const int64_t src = row * nb + b;
    const bool q8 = type == T_Q8_0;
    const bool hm = type == T_Q4_1 || type == T_Q5_1;
    int4 a = make_int4(0, 0, 0, 0), c = make_int4(0, 0, 0, 0);
    uint32_t s = 0, h = 0;
    if (valid) {
        if (q8) { a = *reinterpret_cast<const int4 *>(qs + src * 32); c = *reinterpret_cast<const int4 *>(qs + src * 32 + 16); }
        else a = *reinterpret_cast<const int4 *>(qs + src * 16);
        s = hm ? reinterpret_cast<const uint32_t *>(sc)[src] : (uint32_t) reinterpret_cast<const uint16_t *>(sc)[src];
        if (qh) h = qh[src];
    }
    if (q8) {
        *reinterpret_cast<int4 *>(pq + ((rb * 2 + 0) * 32 + n) * 16) = a;
        *reinterpret_cast<int4 *>(pq + ((rb * 2 + 1) * 32 + n) * 16) = c;
    } else {
        *reinterpret_cast<int4 *>(pq + (rb * 32 + n) * 16) = a;
    }
    psc[rb * 32 + n] = s;
    if (pqh) pqh[rb * 32 + n] = h;
}

// activations, token tile tt (32 tokens), block b:
//   q [tt][b][half][i][16 B]   (i = token within the tile; half 0 = elements 0..15: the MFMA A-operand image, lane = half * 32 + i)
//   d [tt][b][i] f32 = fp16-rounded amax / 127, times dscale       s [tt][b][i] f32 = fp16(d * sum q)       o [tt][b][i] f32 = off * sum q
// Tokens t >= T of the last tiles are written as zeros (their products are never stored).
// One wave = one (token tile, block): lane = half * 32 + i owns the 16 elements of its half of token i's block -- exactly its
// 16 bytes of the A-operand image, so the wave reads 32 full 128-byte lines and writes one contiguous KiB. The block's amax and
// code sum combine the two halves through one permlane32 swap (max and integer sum are order-free: same result as k_quant_act).
// The quantiser of one wave = one (token tile, block): lane = half * 32 + i holds the 16 elements v of its half of token i's block.
// Writes the wave's KiB of codes and (lanes of the first half) the token's scales at index tb = token tile * nb + block.
static __device__ __forceinline__ void quantize_tile_unit(const float (&v)[16], int lane, int64_t tb, float dscale, float off,
                                                          int8_t * __restrict__ q, float * __restrict__ d, float * __restrict__ s, float * __restrict__ o) {
    const int i = lane & 31, h = lane >> 5;
    float amax = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; j++) amax = fmaxf(amax, fabsf(v[j]));
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
        amax = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    const float dd = amax / 127.0f;
    const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
    int sum = 0;
    int w4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int w = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) { const int qi = (int) roundf(v[4 * j + e] * id); sum += qi; w |= (qi & 0xFF) << (8 * e); }
        w4[j] = w;
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned) sum, (unsigned) sum, false, false);
        sum = (int) r[0] + (int) r[1];
    }
    *reinterpret_cast<int4 *>(q + tb * 1024 + lane * 16) = make_int4(w4[0], w4[1], w4[2], w4[3]);
    if (h == 0) {
        const float d16 = round_f16(dd);
        d[tb * 32 + i] = d16 * dscale;   // a power of two: exact
        if (s) s[tb * 32 + i] = round_f16((float) sum * dd);
        if (o) o[tb * 32 + i] = off * (float) sum;
    }
}

// The same block in ONE thread (the 32 elements v of token i's block in its registers): no lane exchange, no LDS -- for producers whose
// thread owns a whole block of a token (k_groupnorm_seq_q, k_mix_seq_q). Same statements as above: amax and the code sum are order-free,
// everything else is per element.
static __device__ __forceinline__ void quantize_block_thread(const float (&v)[32], int i, int64_t tb, float dscale, float off,
                                                             int8_t * __restrict__ q, float * __restrict__ d, float * __restrict__ s, float * __restrict__ o) {
    float amax = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; j++) amax = fmaxf(amax, fabsf(v[j]));
    const float dd = amax / 127.0f;
    const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
    int sum = 0;
    int w8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        int w = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) { const int qi = (int) roundf(v[4 * j + e] * id); sum += qi; w |= (qi & 0xFF) << (8 * e); }
        w8[j] = w;
    }
    *reinterpret_cast<int4 *>(q + tb * 1024 + i * 16) = make_int4(w8[0], w8[1], w8[2], w8[3]);            // half 0: lane i of the unit
    *reinterpret_cast<int4 *>(q + tb * 1024 + (32 + i) * 16) = make_int4(w8[4], w8[5], w8[6], w8[7]);     // half 1: lane 32 + i
    const float d16 = round_f16(dd);
    d[tb * 32 + i] = d16 * dscale;   // a power of two: exact
    if (s) s[tb * 32 + i] = round_f16((float) sum * dd);
    if (o) o[tb * 32 + i] = off * (float) sum;
}

// The same from an f32 tile of 32 tokens x 256 channels staged in LDS (row stride QT_LD floats: 16-byte reads of 32 different rows
// spread over all banks): the 256-thread workgroup that produced the tile quantises its 8 blocks, two (token tile, block) units per
// wave. Used by the producers whose only consumers are quantised products (the sequence-mode mixes): the f32 activations never
// travel to HBM and back.
constexpr int QT_LD = 260;
static __device__ __forceinline__ void quantize_lds_tile(const float * __restrict__ l_tile, int tid, int64_t tt, int b0, int nb, float dscale, float off,
                                                         int8_t * __restrict__ q, float * __restrict__ d, float * __restrict__ s, float * __restrict__ o) {
    const int lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int bl = wave + 4 * k;
        float v[16];
        const float4 * src = reinterpret_cast<const float4 *>(l_tile + i * QT_LD + bl * 32 + h * 16);
#pragma unroll
        for (int j = 0; j < 4; j++) { const float4 f = src[j]; v[4 * j] = f.x; v[4 * j + 1] = f.y; v[4 * j + 2] = f.z; v[4 * j + 3] = f.w; }
        quantize_tile_unit(v, lane, tt * nb + b0 + bl, dscale, off, q, d, s, o);
    }
}

struct TileOut { int8_t * q[6]; float * d[6]; float * s[6]; float * o[6]; float dscale, off; int nb, on; };
constexpr int QUANT_BATCH = 5;                // inputs per launch (blockIdx.y): e.g. the five mixed inputs of an RWKV-6 time-mixing block
struct QuantBatch { const float * x[QUANT_BATCH]; int8_t * q[QUANT_BATCH]; float * d[QUANT_BATCH]; float * s[QUANT_BATCH]; float * o[QUANT_BATCH]; };
__global__ __launch_bounds__(256) void k_quant_act_tiles(QuantBatch qb, int64_t T, int64_t T_pad, int nb, float dscale, float off) {
    const float * __restrict__ const x = qb.x[blockIdx.y];
    int8_t * __restrict__ const q = qb.q[blockIdx.y];
    float * __restrict__ const d = qb.d[blockIdx.y];
    float * __restrict__ const s = qb.s[blockIdx.y];
    float * __restrict__ const o = qb.o[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int64_t wv = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);   // (tt, b)
    const int64_t n_tiles = (T_pad >> 5) * nb;
    if (wv >= n_tiles) return;                                          // whole waves
    const int64_t tt = wv / nb;
    const int b = (int) (wv - tt * nb);
    const int i = lane & 31, h = lane >> 5;
    const int64_t t = tt * 32 + i;
    const bool live = t < T;
    float v[16];
    const float4 * src = reinterpret_cast<const float4 *>(x + ((live ? t : 0) * nb + b) * 32 + h * 16);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float4 f = live ? src[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        v[4 * j] = f.x; v[4 * j + 1] = f.y; v[4 * j + 2] = f.z; v[4 * j + 3] = f.w;
    }
    quantize_tile_unit(v, lane, tt * nb + b, dscale, off, q, d, s, o);
}

// ---------------------------------------------------------------------------------------------------------------
// the GEMM
// ---------------------------------------------------------------------------------------------------------------

template <int FMT>
__global__ __launch_bounds__(MF<FMT>::NT, 2) void k_mmq_mfma(MmqArgs A) {
    typedef MF<FMT> M;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef PF_EXP_STAMP
    // timing-only build: workgroup 0's thread 0 overwrites y[0][0 .. 5] with its own timeline in microseconds (100 MHz wall clock):
    // prologue issued, first chunk landed, walk done, epilogue done; y[0][5] = the wall clock at entry (low bits), for launch gaps
    const unsigned long long st_t0 = wall_clock64();
    unsigned long long st_t1 = 0, st_t2 = 0, st_t3 = 0;
#endif
    const int64_t N = A.N, T = A.T, ldy = A.ldy;
    const int nb = A.nb, RT = A.RT, C = A.C;
    // XCD-aware tile map: block id -> XCD id % 8; the token tiles of one 128-row panel run on the same XCD and share its L2
    // (with fewer than 8 row panels that map would leave whole XCDs idle: dense map, consecutive blocks = different XCDs)
    const int RP = (RT + M::RGN - 1) / M::RGN;
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int rp = RP < 8 ? (int) blockIdx.x % RP : (kk / C) * 8 + xcd;          // row panel (RGN row tiles)
    const int ct = RP < 8 ? (int) blockIdx.x / RP : kk % C;                       // token tile of 64
    if (rp * M::RGN >= RT) return;               // (whole workgroup: before any barrier)
    const int bz = blockIdx.y;
    const PfW w = A.w[bz];
    const PfX x = A.x[bz];
    float * __restrict__ const y = A.y[bz];
    const Epi epi = A.epi[bz];
    const int a0 = (int) blockIdx.z * (8 / A.split), a1 = a0 + 8 / A.split;
    const int s_base = A.a_start[a0], n_steps = A.a_start[a1] - s_base;            // this part's steps of the walk
    // (constant address space: the block numbers are read with scalar loads -- a vector load here would share vmcnt with the DMAs)
    typedef const int __attribute__((address_space(4))) * cint_p;
    const cint_p order = (cint_p) (uintptr_t) A.order + s_base;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave % M::RGN, tg = wave / M::RGN;
    const int nn = lane & 31, h = lane >> 5;
    const int rt0 = rp * M::RGN, tt0 = ct * 2;

    // ---- staging: chunk k = steps [CH k, CH k + CH) goes straight from global memory into LDS buffer k & 1 (LDS-DMA: 16 bytes per
    //      lane, 1 KiB per wave-instruction, no staging registers). Wave w stages the rows r = w / CH, + 8 / CH, ... of step CH k + w % CH:
    //      the block number of its step is wave-uniform, so a row's address is a scalar base (SALU) plus a per-lane offset. ----
    const unsigned lds_base = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) lds;
    const int st_w = wave % M::CH, sub_w = wave / M::CH;
    constexpr int NSUB = 8 / M::CH;
    auto blk_of = [&](int k) { int step = M::CH * k + st_w; step = step < n_steps ? step : n_steps - 1; return order[step]; };   // (tail: harmless duplicates)
    int b_nx = blk_of(0);                        // block of this wave's step in the chunk issued next (read one chunk ahead)
    auto dma_s = [&](const unsigned char * sbase, unsigned voff, unsigned dst) {
        // Issued through inline asm on purpose: the compiler cannot prove that an LDS-DMA does not alias the operand reads of the
        // steps that follow and would put `s_waitcnt vmcnt(0)` in front of every one of them. Completion is waited for by hand.
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
    auto issue = [&](int k) {
        // (the lane index goes through an opaque copy: the offsets below are recomputed at every chunk boundary instead of being kept
        //  in registers across the walk, where every register counts)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int64_t b = b_nx;
        const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned) ((M::CH * (k % M::NBUF) + st_w) * M::SLOT));
#pragma unroll
        for (int r = 0; r < M::NR; r++) {
            if (r % NSUB != sub_w) continue;                         // (wave-uniform)
            const unsigned dst = dst0 + r * 1024;
            if (r < M::NW) {
                if constexpr (M::Q8) {
                    int rt = rt0 + r; rt = rt < RT ? rt : RT - 1;
                    dma_s(w.qs + ((int64_t) rt * nb + b) * 1024, (unsigned) lane_o * 16, dst);
                } else {
                    int rta = rt0 + 2 * r, rtb = rta + 1;
                    rta = rta < RT ? rta : RT - 1; rtb = rtb < RT ? rtb : RT - 1;
                    const unsigned up = (unsigned) (rtb - rta) * (unsigned) nb * 512u;      // 0 when the upper tile is clamped onto the lower one
                    dma_s(w.qs + ((int64_t) rta * nb + b) * 512, (unsigned) (lane_o & 31) * 16 + ((lane_o >> 5) ? up : 0u), dst);
                }
            } else if (r < M::NW + 2) {
                const int t2 = r - M::NW;
                dma_s(reinterpret_cast<const unsigned char *>(x.q) + ((int64_t) (tt0 + t2) * nb + b) * 1024, (unsigned) lane_o * 16, dst);
            } else {
                // tail pieces: one masked DMA per array that has pieces in this row (no pointer selects: they become stack objects)
                const int q = lane_o + 64 * (r - M::NW - 2);
                const int part = q & 7, lo = 64 * (r - M::NW - 2), hi = lo + 64;
                auto seg = [&](int first, int count, const void * base, bool weights) {
                    if (count == 0 || first >= hi || first + count <= lo) return;          // (compile-time after unrolling)
                    if (q >= first && q < first + count) {
                        const int e = (q - first) >> 3;                                    // row tile / token tile within the step
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
        b_nx = blk_of(k + 1);
    };

    const int n_chunks = (n_steps + M::CH - 1) / M::CH;
#pragma unroll
    for (int k0 = 0; k0 < M::NBUF - 1; k0++) if (k0 < n_chunks) issue(k0);

    // ---- software pipeline. Step sigma = one quantisation block of the walk. While the f32 fold of step sigma runs on the VALU,
    //      the MFMA of step sigma + 1 runs on the matrix pipe and the LDS reads of step sigma + 2 are in flight:
    //        n_*   operands of the step after the one in `acc` (LDS -> registers, read by load_ops)
    //        acc   integer block sums of the current step (written by launch)
    //        dd    d_w * d_x of the current step (16 tokens of this lane's row), aux: its s / o values
    int4 n_braw = make_int4(0, 0, 0, 0);
    v4i n_aop = {0, 0, 0, 0};
    unsigned n_scw = 0, n_qhw = 0;
    v16i acc;
    v2f dd[8], aux[8];
    float dw_l = 0.0f, mw_l = 0.0f;            // scales of the step in `acc`
    int sg_l = 0;                                // next step to read from LDS
    int to_bnd = 0;                              // steps until the next chunk boundary
    unsigned off_l = 0, off_s = 0;               // LDS byte offset of the slot of step sg_l / of the slot before it
    // per-lane address parts (opaque: kept as one VGPR each, the wave-uniform terms are not split off into scalar adds per step)
    unsigned a_braw = (unsigned) (M::OFF_WC + (M::Q8 ? (rg * 64 + h * 32 + nn) : (rg * 32 + nn)) * 16);
    unsigned a_scw = (unsigned) (M::OFF_WSC + (rg * 32 + nn) * 4);
    unsigned a_aop = (unsigned) (M::OFF_XQ + (tg * 64 + lane) * 16);
    unsigned a_xd = (unsigned) (M::OFF_XD + (tg * 32 + 4 * h) * 4);
    asm volatile("" : "+v"(a_braw), "+v"(a_scw), "+v"(a_aop), "+v"(a_xd));
    auto load_ops = [&]() {
        if (to_bnd == 0) {
            // chunk boundary: this wave's part of chunk k has landed (issued one chunk ago); after the barrier everybody's has, and
            // every wave has finished reading the buffer the next chunk goes into (its last reads are two stages back)
            const int k = sg_l / M::CH;
            // chunk k has landed when only the DMAs of the younger chunks k + 1 .. k + NBUF - 2 are outstanding (vmcnt counts this wave's)
#ifdef PF_EXP_NOSYNC
            if (false) { } else if (true) { } else
#endif
            if (M::NBUF > 2 && k + M::NBUF - 2 < n_chunks) {
                if (NSUB == 1 || sub_w == 0) wait_vm<M::n_dma(0) * (M::NBUF - 2)>(); else wait_vm<M::n_dma(NSUB - 1) * (M::NBUF - 2)>();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#ifndef PF_EXP_NOSYNC
            __syncthreads();
#endif
#ifdef PF_EXP_NODMA
            if (k + M::NBUF - 1 < n_chunks && k < 1) issue(k + M::NBUF - 1);
#else
            if (k + M::NBUF - 1 < n_chunks) issue(k + M::NBUF - 1);
#endif
            to_bnd = k + 1 < n_chunks ? M::CH : 0x40000000;      // (no boundary behind the last chunk)
        }
        to_bnd--;
#ifdef PF_EXP_NOOPS
        if (sg_l <= 2)
#endif
        n_braw = *reinterpret_cast<const int4 *>(lds + (off_l + a_braw));
        n_scw = *reinterpret_cast<const unsigned *>(lds + (off_l + a_scw));
#ifdef PF_EXP_NOOPS
        if (sg_l <= 2)
#endif
        n_aop = *reinterpret_cast<const v4i *>(lds + (off_l + a_aop));
        if constexpr (M::QH) n_qhw = *reinterpret_cast<const unsigned *>(lds + (off_l + a_scw + (M::OFF_WQH - M::OFF_WSC)));
        off_s = off_l;
        off_l = off_l == (M::NBUF * M::CH - 1) * M::SLOT ? 0u : off_l + M::SLOT;
        sg_l++;
    };
    // The MFMA accumulates onto the bit pattern of 1.5 * 2^23: for |sum| < 2^22 the result, read as a float, IS 12582912 + sum exactly,
    // and one packed subtraction per register pair (exact) replaces two v_cvt_f32_i32. |sum| <= 32 * 128 * 127 < 2^19 for every format.
    constexpr int   MAGIC_I = 0x4B400000;
    constexpr float MAGIC_F = 12582912.0f;
    v16i magic;
#pragma unroll
    for (int r = 0; r < 16; r++) magic[r] = MAGIC_I;
    asm volatile("" : "+v"(magic));              // (kept in 16 registers: not rematerialised in front of every MFMA)
    int nib_sh = h ? 0 : 4;                      // Q4_0: the lanes of the first half move the low nibbles into the high ones
    asm volatile("" : "+v"(nib_sh));
    auto launch = [&]() {
        v4i bop;
        const int raw[4] = {n_braw.x, n_braw.y, n_braw.z, n_braw.w};
        if constexpr (FMT == T_Q8_0) {
            bop[0] = raw[0]; bop[1] = raw[1]; bop[2] = raw[2]; bop[3] = raw[3];
        } else if constexpr (FMT == T_Q4_0) {
            // signed (q - 8) placed in the HIGH nibble: the MFMA returns 16 x the block sum, the token scales carry 1/16
#pragma unroll
            for (int i = 0; i < 4; i++) { const int t = raw[i] << nib_sh; bop[i] = (t & (int) 0xF0F0F0F0) ^ (int) 0x80808080; }
        } else if constexpr (FMT == T_Q4_1) {
#pragma unroll
            for (int i = 0; i < 4; i++) bop[i] = (raw[i] >> (4 * h)) & 0x0F0F0F0F;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned nl = (n_qhw >> (16 * h + 4 * i)) & 0xFu;
                bop[i] = ((raw[i] >> (4 * h)) & 0x0F0F0F0F) | (int) (((nl * 0x00204081u) & 0x01010101u) << 4);
            }
        }
        dw_l = h2f_bits((uint16_t) (n_scw & 0xFFFFu));
        if constexpr (M::HM) mw_l = h2f_bits((uint16_t) (n_scw >> 16));
        asm volatile("" : "+v"(dw_l), "+v"(mw_l));     // (converted here, while n_scw is live: not sunk to the next step's fold)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(n_aop, bop, magic, 0, 0, 0);
    };
    // token scales of step sg -> dd (raw d_x: multiplied by d_w in place just before the fold), aux = s (HM) / o (XO).
    // Tokens of register r: (r & 3) + 8 (r >> 2) + 4 h.
    auto read_scales = [&]() {                   // of the step whose codes load_ops read last
        const unsigned char * S = lds + (off_s + a_xd);
#ifdef PF_EXP_NODX
        if (sg_l > 2) return;
#endif
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 dx4 = *reinterpret_cast<const float4 *>(S + 32 * g);
            dd[2 * g] = (v2f){dx4.x, dx4.y};
            dd[2 * g + 1] = (v2f){dx4.z, dx4.w};
            if constexpr (M::HM || M::XO) {
                const float4 a4 = *reinterpret_cast<const float4 *>(S + ((M::HM ? M::OFF_XS : M::OFF_XO) - M::OFF_XD) + 32 * g);
                aux[2 * g] = (v2f){a4.x, a4.y}; aux[2 * g + 1] = (v2f){a4.z, a4.w};
            }
        }
    };
    [[maybe_unused]] int sigma = 0;          // (read by the timing-only builds)
    // One step: fold block sigma into cur (FIRST: the leaf's first block, cur = fma(.., .., +0)), start block sigma + 1, read block
    // sigma + 2. Every LDS read is issued a stage before its use: the eight waves of the workgroup run in lock-step between the chunk
    // barriers, so a read that is waited for right away queues behind everybody else's (measured: 46 % of the wave cycles in waits).
    auto step = [&](v2f (&cur)[8], auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        v2f sf[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            sf[j] = (v2f){__int_as_float(acc[2 * j]), __int_as_float(acc[2 * j + 1])} - (v2f){MAGIC_F, MAGIC_F};
            if constexpr (M::XO) sf[j] = sf[j] - aux[j];            // exact: integers below 2^24
        }
        const float mw_c = mw_l, dw_c = dw_l;
        // (opaque uses pin the order: the sums leave `acc` before the next MFMA is issued into the same registers, and the fold
        //  releases dd / aux before the next step's scales are read into them -- otherwise the compiler pipelines by rotating
        //  16 + 16 registers with copies at the end of every step)
#define PIN8(A) asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]))
        PIN8(sf);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (M::OVERLAP) launch();
#ifdef PF_EXP_NOFOLD
        if (sigma == 0)
#endif
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const v2f zero = {0.0f, 0.0f};
            const v2f ddj = (v2f){dw_c, dw_c} * dd[j];
            cur[j] = __builtin_elementwise_fma(ddj, sf[j], FIRST ? zero : cur[j]);
            if constexpr (M::HM) cur[j] = __builtin_elementwise_fma((v2f){mw_c, mw_c}, aux[j], cur[j]);
        }
        PIN8(cur);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!M::OVERLAP) launch();
        read_scales();                          // block sigma + 1, into the registers the fold has just released
        load_ops();
        __builtin_amdgcn_sched_barrier(0);
#undef PIN8
        sigma++;
    };
#ifdef PF_EXP_STAMP
    st_t1 = wall_clock64();
#endif
    load_ops();          // operands of step 0
#ifdef PF_EXP_STAMP
    st_t2 = wall_clock64();
#endif
    launch();            // block sums of step 0 under way
    read_scales();
    load_ops();          // operands of step 1

    // ---- the walk: leaves in bit-reversed order, 8 per iteration of the outer loop (the merges are compile-time code) ----
    v2f s0[8], s1[8], s2[8], S3[8], S4[8], S5[8], V[8];
    v2f * stk = reinterpret_cast<v2f *>(lds + M::NBUF * M::CH * M::SLOT);   // [level][j][thread]
    constexpr int REV3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    const int rng = a1 - a0;                     // 8, or 4 / 2 / 1 for a part of a split walk: the merges stop at the part's own root
#pragma unroll 1
    for (int a = a0; a < a1; a++) {
        const int ra = ((a & 1) << 2) | (a & 2) | (a >> 2);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int l = 8 * REV3[u] + ra;
            // A leaf whose value starts a new subtree (even u) accumulates straight into s0: no copy. Other leaves use `tmp`.
            v2f tmp[8];
            v2f (&cur)[8] = (u & 1) ? tmp : s0;
            if (l < nb) {
                step(cur, std::true_type{});
                for (int b = l + 64; b < nb; b += 64) step(cur, std::false_type{});
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) cur[j] = (v2f){0.0f, 0.0f};
            }
            // merges after leaf c = 8 a + u: one per trailing one bit of c
#ifdef PF_EXP_NOMERGE
            if (u == 7) { for (int j = 0; j < 8; j++) V[j] = s0[j] + cur[j]; } else if (false) {
#else
            if (u == 1 || u == 5) {
#endif
#pragma unroll
                for (int j = 0; j < 8; j++) s1[j] = s0[j] + cur[j];
            } else if (u == 3) {
#pragma unroll
                for (int j = 0; j < 8; j++) { const v2f t = s0[j] + cur[j]; s2[j] = s1[j] + t; }
            } else if (u == 7) {
#pragma unroll
                for (int j = 0; j < 8; j++) { const v2f t = s0[j] + cur[j]; const v2f t2 = s1[j] + t; V[j] = s2[j] + t2; }
            }
        }
        // the upper levels of the stack are touched once per 8 / 16 / 32 leaves: STK_LDS of them (from the top) live in LDS
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));        // (addresses recomputed here, not kept across the walk)
        v2f * const stk_t = stk + tid_o;
#define STK_GET(LV, REG) (M::STK_LDS > (LV) ? stk_t[((LV) * 8 + j) * M::NT] : REG[j])
#define STK_PUT(LV, REG, VAL) do { if constexpr (M::STK_LDS > (LV)) stk_t[((LV) * 8 + j) * M::NT] = (VAL); else REG[j] = (VAL); } while (0)
        const int al = a - a0;                   // (a0 is a multiple of rng: al < rng)
        if ((al & 1) == 0) {
            if (rng > 1) {
#pragma unroll
                for (int j = 0; j < 8; j++) STK_PUT(2, S3, V[j]);
            }
        } else if ((al & 2) == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) { const v2f t = STK_GET(2, S3) + V[j]; if (rng == 2) V[j] = t; else STK_PUT(1, S4, t); }
        } else if ((al & 4) == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const v2f t = STK_GET(2, S3) + V[j]; const v2f t2 = STK_GET(1, S4) + t;
                if (rng == 4) V[j] = t2; else STK_PUT(0, S5, t2);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) { const v2f t = STK_GET(2, S3) + V[j]; const v2f t2 = STK_GET(1, S4) + t; V[j] = STK_GET(0, S5) + t2; }
        }
#undef STK_GET
#undef STK_PUT
    }
#ifdef PF_EXP_STAMP
    st_t3 = wall_clock64();
#endif
    // ---- split walk: V is this part's subtree sum; k_mmq_combine adds the parts in tree order and applies the epilogue ----
    if (A.split > 1) {
        const int64_t n_tiles = (int64_t) gridDim.y * ((RT + M::RGN - 1) / M::RGN) * C;
        const int64_t tile = ((int64_t) bz * ((RT + M::RGN - 1) / M::RGN) + rp) * C + ct;
        v2f * const mine = reinterpret_cast<v2f *>(A.part) + ((blockIdx.z * n_tiles + tile) * M::NT + tid) * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) mine[j] = V[j];
        return;
    }
    // ---- epilogue: row n = column of the tile (lane), tokens along the registers ----
    const int64_t n = (int64_t) (rt0 + rg) * 32 + nn;
    if (A.yq.q != nullptr) {
        // quantised output (MmqQOut): the wave's tile goes through its own patch of LDS (token-major, 33 floats per token) into the
        // quantiser's operand order -- lane = half * 32 + token holds the 16 elements of its half of the block -- and leaves as one
        // unit of the next product's image. Same statements per element as the f32 epilogue followed by k_quant_act_tiles.
        __syncthreads();                     // every wave is through its last step: the chunk buffers are free
        float * const patch = reinterpret_cast<float *>(lds) + wave * (32 * 33);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int tl = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int64_t t = (int64_t) (tt0 + tg) * 32 + tl;
            patch[tl * 33 + nn] = (t < T && n < N) ? apply_epi(epi, V[r >> 1][r & 1], t, n, ldy) : 0.0f;
        }
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = patch[nn * 33 + h * 16 + j];
        if (rt0 + rg < RT) quantize_tile_unit(v, lane, (int64_t) (tt0 + tg) * A.yq.nb + (rt0 + rg), A.yq.dscale, A.yq.off, A.yq.q, A.yq.d, A.yq.s, A.yq.o);
        return;
    }
    if (n < N) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int64_t t = (int64_t) (tt0 + tg) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (t < T) y[t * ldy + n] = apply_epi(epi, V[r >> 1][r & 1], t, n, ldy);
        }
    }
#ifdef PF_EXP_STAMP
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && tid == 0) {
        const unsigned long long st_t4 = wall_clock64();
        y[0] = (float) (st_t1 - st_t0) * 0.01f; y[1] = (float) (st_t2 - st_t0) * 0.01f; y[2] = (float) (st_t3 - st_t0) * 0.01f;
        y[3] = (float) (st_t4 - st_t0) * 0.01f; y[4] = (float) n_steps; y[5] = (float) (st_t0 & 0xFFFFFF) * 0.01f;
    }
#endif
}


// Second half of a split walk: thread (tile, tid) owns the same 16 outputs as in k_mmq_mfma; the parts are added pairwise in tree
// order -- ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)) -- which are exactly the additions of the unsplit walk above level 3.
template <int SPLIT>
__global__ __launch_bounds__(512) void k_mmq_combine(MmqArgs A) {
    constexpr int RGN = 4, NT = 512;
    const int RP = (A.RT + RGN - 1) / RGN, C = A.C;
    const int64_t n_tiles = (int64_t) gridDim.y * RP * C;
    const int bz = blockIdx.y, rp = blockIdx.x / C, ct = blockIdx.x % C;
    const int64_t tile = ((int64_t) bz * RP + rp) * C + ct;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rg = wave % RGN, tg = wave / RGN, nn = lane & 31, h = lane >> 5;
    // (blockIdx.z = register pair j of the thread's 16 outputs: eight times the workgroups, an eighth of the dependent loads each)
    const int j = blockIdx.z;
    v2f P[SPLIT];
#pragma unroll
    for (int z = 0; z < SPLIT; z++) P[z] = (reinterpret_cast<const v2f *>(A.part) + ((z * n_tiles + tile) * NT + tid) * 8)[j];
#pragma unroll
    for (int wd = 1; wd < SPLIT; wd <<= 1) {
#pragma unroll
        for (int z = 0; z < SPLIT; z += 2 * wd) P[z] = P[z] + P[z + wd];
    }
    const Epi epi = A.epi[bz];
    float * __restrict__ const y = A.y[bz];
    const int64_t n = (int64_t) (rp * RGN + rg) * 32 + nn;
    if (n < A.N) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int r = 2 * j + q;
            const int64_t t = (int64_t) (ct * 2 + tg) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (t < A.T) y[t * A.ldy + n] = apply_epi(epi, P[0][q], t, n, A.ldy);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// WKV-5/6 over a sequence (ggml_rwkv_wkv6, rwkv_graph.inc:275,370), pipelined across the lanes of a wave.
//
// out[t][j] = sum_i r_i (u_i k_i v_j + s_ij), accumulated over i = 0 .. 63 IN THAT ORDER (the reference's and the oracle's
// loop), s_ij <- s_ij w_i + k_i v_j. One wave owns ONE value column j of a head; lane i owns the state element s_ij. The
// ordered sum over i would be a 64-step dependent chain per token, so the chain is run as a systolic pipeline instead: at
// step sigma lane i works on token sigma - i, adds its term to the running sum it receives from lane i - 1 (one DPP
// wave_shr:1 add; lane 0 receives 0.0f, as the reference's `o = 0`), and lane 63 emits out[sigma - 63][j]. Every lane is
// busy on a different token each step: ~10 VALU instructions per token and column, 64 x 64 x H lanes wide, instead of the
// 64-step loop of one wave per head of the single-token form (k_wkv6). Per-token operands are read diagonally from an LDS
// ring of 128 tokens (lane i reads row (sigma - i) mod 128, column i: conflict-free), refilled 64 tokens at a time through
// registers, one pair of workgroup barriers per 64 steps. The workgroup = 8 waves = 8 adjacent columns of one head.
// ---------------------------------------------------------------------------------------------------------------
// Layout of the workgroup: 4 waves (one per SIMD), wave = TWO adjacent value columns j, j + 1 as the halves of packed-f32
// registers (v_pk_mul / v_pk_add: one instruction per operation of both columns), 8 columns of one head per workgroup. The rings
// hold RINGP = 128 + 3 rows: rows 128 .. 130 mirror rows 0 .. 2, so the four steps of a group read rows a, a + 1, a + 2, a + 3 of
// one per-lane address with immediate offsets (no wrap inside a group). The reads of group g + 1 are issued before the arithmetic
// of group g. Chunks whose 64 steps are valid for every lane (all but the first and the last ones) skip the validity selects.
template <int WMODE>
__global__ __launch_bounds__(256, 1) void k_wkv6_seq(const float * __restrict__ r, const float * __restrict__ k, const float * __restrict__ v,
                                                  const float * __restrict__ u, int u_per_chan, const float * __restrict__ w,
                                                  const float * __restrict__ state_in, float * __restrict__ state_out, float * __restrict__ out,
                                                  int T, int H) {
    constexpr int S = 64, RING = 128, RINGP = RING + 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float * l_k = reinterpret_cast<float *>(lds_raw);            // [RINGP][k 64 | r 64 | w 64]: one per-lane address, immediate offsets
    v2f * l_v = reinterpret_cast<v2f *>(l_k + 3 * RINGP * 64);   // [4 column pairs][RINGP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t h = blockIdx.x >> 3;
    const int cb = (int) (blockIdx.x & 7) * 8, j = cb + wave * 2;
    const int64_t D = (int64_t) H * S;
    v2f s = *reinterpret_cast<const v2f *>(state_in + h * S * S + (int64_t) lane * S + j);
    const float ui = u_per_chan ? u[h * S + lane] : u[h];
    const float wconst = WMODE == 2 ? 0.0f : (WMODE == 1 ? w[h * S + lane] : w[h]);
    const int n_chunks = (T + 63 + 63) / 64;   // steps 0 .. T + 62

    // staging registers: chunk c = tokens [64 c, 64 c + 64): 64 x 16 float4 per array, four per thread; 64 x 4 column pairs of v
    float4 gk0, gk1, gk2, gk3, gr0, gr1, gr2, gr3, gw0, gw1, gw2, gw3;
    gw0 = gw1 = gw2 = gw3 = make_float4(0.f, 0.f, 0.f, 0.f);
    v2f gv;
    auto issue1 = [&](int c, int i, float4 & ak, float4 & ar, float4 & aw) __attribute__((always_inline)) {
        int t = 64 * c + 16 * i + (tid >> 4);
        t = t < T ? t : T - 1;                         // (clamped rows are never used: their steps are masked)
        const int64_t o = (int64_t) t * D + h * S + (tid & 15) * 4;
        ak = *reinterpret_cast<const float4 *>(k + o);
        ar = *reinterpret_cast<const float4 *>(r + o);
        if (WMODE == 2) aw = *reinterpret_cast<const float4 *>(w + o);
    };
    auto issue = [&](int c) __attribute__((always_inline)) {
        issue1(c, 0, gk0, gr0, gw0); issue1(c, 1, gk1, gr1, gw1); issue1(c, 2, gk2, gr2, gw2); issue1(c, 3, gk3, gr3, gw3);
        int t = 64 * c + (tid >> 2);
        t = t < T ? t : T - 1;
        gv = *reinterpret_cast<const v2f *>(v + (int64_t) t * D + h * S + cb + (tid & 3) * 2);
    };
    auto commit1 = [&](int c, int i, const float4 & ak, const float4 & ar, const float4 & aw) __attribute__((always_inline)) {
        const int row = (c & 1) * 64 + 16 * i + (tid >> 4), col = (tid & 15) * 4;
        *reinterpret_cast<float4 *>(l_k + row * 192 + col) = ak;
        *reinterpret_cast<float4 *>(l_k + row * 192 + 64 + col) = ar;
        if (WMODE == 2) *reinterpret_cast<float4 *>(l_k + row * 192 + 128 + col) = aw;
        if (row < 3) {
            *reinterpret_cast<float4 *>(l_k + (row + RING) * 192 + col) = ak;
            *reinterpret_cast<float4 *>(l_k + (row + RING) * 192 + 64 + col) = ar;
            if (WMODE == 2) *reinterpret_cast<float4 *>(l_k + (row + RING) * 192 + 128 + col) = aw;
        }
    };
    auto commit = [&](int c) __attribute__((always_inline)) {
        commit1(c, 0, gk0, gr0, gw0); commit1(c, 1, gk1, gr1, gw1); commit1(c, 2, gk2, gr2, gw2); commit1(c, 3, gk3, gr3, gw3);
        const int row = (c & 1) * 64 + (tid >> 2);
        l_v[(tid & 3) * RINGP + row] = gv;
        if (row < 3) l_v[(tid & 3) * RINGP + row + RING] = gv;
    };

    float kk0[4], rr0[4], ww0[4], kk1[4], rr1[4], ww1[4];     // two register sets: the group being read and the group being computed
    v2f vv0[4], vv1[4];
    auto read_group = [&](int c, int g, float (&kk)[4], float (&rr)[4], float (&ww)[4], v2f (&vv)[4]) __attribute__((always_inline)) {
        const int a = (64 * c + 4 * g - lane) & (RING - 1);
        const float * pk = l_k + a * 192 + lane;
        const v2f * pv = l_v + wave * RINGP + a;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            kk[e] = pk[e * 192];
            rr[e] = pk[e * 192 + 64];
            ww[e] = WMODE == 2 ? pk[e * 192 + 128] : wconst;
            vv[e] = pv[e];
        }
    };
    float o_a = 0.0f, o_b = 0.0f, cap_a = 0.0f, cap_b = 0.0f;
    // One step (macros: the capture lane must be an assembly-time constant). Running sums come from lane i - 1 (same token, previous
    // step); lane 0 starts from 0.0f like the reference's `o = 0`. Lane q = 4 G + E keeps what lane 63 emits at step 64 c + q:
    // out[64 c + q - 63][j], [j + 1] (v_readlane -> SGPR -> v_writelane; this clang has no writelane builtin).
#define WKV_STEP(G, E, CHECKED, KK, RR, WW, VV)                                                                                  \
    {                                                                                                                            \
        const v2f kv = VV[E] * (v2f){KK[E], KK[E]};                                                                            \
        const v2f ku = kv * (v2f){ui, ui};                                                                                       \
        const v2f temp = ku + s;                                                                                                 \
        const v2f p = temp * (v2f){RR[E], RR[E]};                                                                                  \
        const v2f sw = s * (v2f){WW[E], WW[E]};                                                                                    \
        const v2f sn = sw + kv;                                                                                                  \
        const float in_a = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o_a), 0x138 /* wave_shr:1 */, 0xF, 0xF, false)); \
        const float in_b = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o_b), 0x138, 0xF, 0xF, false));          \
        o_a = in_a + p[0];                                                                                                       \
        o_b = in_b + p[1];                                                                                                       \
        if (CHECKED) {                                                                                                           \
            const int t = 64 * c + 4 * (G) + (E) - lane;                                                                         \
            s = (t >= 0 && t < T) ? sn : s;                                                                                      \
        } else {                                                                                                                 \
            s = sn;                                                                                                              \
        }                                                                                                                        \
        int sa, sb;                                                                                                              \
        asm("v_readlane_b32 %0, %1, 63" : "=s"(sa) : "v"(o_a));                                                                  \
        asm("v_readlane_b32 %0, %1, 63" : "=s"(sb) : "v"(o_b));                                                                  \
        asm("v_writelane_b32 %0, %1, %2" : "+v"(cap_a) : "s"(sa), "n"(4 * (G) + (E)));                                           \
        asm("v_writelane_b32 %0, %1, %2" : "+v"(cap_b) : "s"(sb), "n"(4 * (G) + (E)));                                           \
    }
#define WKV_GROUP_E(G, CHECKED)   /* even group: computes set 0 while set 1 is read */                                            \
    read_group(c, (G) + 1, kk1, rr1, ww1, vv1);                                                                                  \
    WKV_STEP(G, 0, CHECKED, kk0, rr0, ww0, vv0) WKV_STEP(G, 1, CHECKED, kk0, rr0, ww0, vv0)                                       \
    WKV_STEP(G, 2, CHECKED, kk0, rr0, ww0, vv0) WKV_STEP(G, 3, CHECKED, kk0, rr0, ww0, vv0)
#define WKV_GROUP_O(G, CHECKED)   /* odd group */                                                                                \
    if ((G) + 1 < 16) read_group(c, (G) + 1, kk0, rr0, ww0, vv0);                                                                \
    WKV_STEP(G, 0, CHECKED, kk1, rr1, ww1, vv1) WKV_STEP(G, 1, CHECKED, kk1, rr1, ww1, vv1)                                       \
    WKV_STEP(G, 2, CHECKED, kk1, rr1, ww1, vv1) WKV_STEP(G, 3, CHECKED, kk1, rr1, ww1, vv1)
#define WKV_CHUNK(CHECKED)                                                                                                       \
    WKV_GROUP_E(0, CHECKED) WKV_GROUP_O(1, CHECKED) WKV_GROUP_E(2, CHECKED) WKV_GROUP_O(3, CHECKED) WKV_GROUP_E(4, CHECKED)       \
    WKV_GROUP_O(5, CHECKED) WKV_GROUP_E(6, CHECKED) WKV_GROUP_O(7, CHECKED) WKV_GROUP_E(8, CHECKED) WKV_GROUP_O(9, CHECKED)       \
    WKV_GROUP_E(10, CHECKED) WKV_GROUP_O(11, CHECKED) WKV_GROUP_E(12, CHECKED) WKV_GROUP_O(13, CHECKED) WKV_GROUP_E(14, CHECKED)  \
    WKV_GROUP_O(15, CHECKED)

    issue(0);
    for (int c = 0; c < n_chunks; c++) {
        __syncthreads();                 // every wave is done with the steps of chunk c - 1 (which still read chunk c - 2's half)
        commit(c);
        __syncthreads();
        if (c + 1 < n_chunks) issue(c + 1);
        const bool interior = c >= 1 && 64 * c + 64 <= T;      // every (step, lane) of the chunk is a real token
        read_group(c, 0, kk0, rr0, ww0, vv0);
        if (interior) { WKV_CHUNK(false) } else { WKV_CHUNK(true) }
        const int t_out = 64 * c + lane - 63;
        if (t_out >= 0 && t_out < T) *reinterpret_cast<v2f *>(out + (int64_t) t_out * D + h * S + j) = (v2f){cap_a, cap_b};
    }
    *reinterpret_cast<v2f *>(state_out + h * S * S + (int64_t) lane * S + j) = s;
#undef WKV_STEP
#undef WKV_GROUP_E
#undef WKV_GROUP_O
#undef WKV_CHUNK
}

// RWKV-6 data-dependent mixes over a sequence (rwkv_graph.inc:323-346; same statement order as k_v6_mix2): thread = (f, d) keeps
// its R columns of W2 in registers across a tile of tokens; the tokens' tanh'ed low-rank vectors sit in LDS (broadcast reads).
template <int R, int TT>
__global__ __launch_bounds__(256) void k_v6_mix2_seq(V6Mix2Args a, TileOut qo, int T, int D) {
    static_assert(TT == 32, "one token tile of the quantised image per workgroup");
    __shared__ __attribute__((aligned(16))) float l_tl[TT * R];
    __shared__ __attribute__((aligned(16))) float l_out[TT * QT_LD];
    const int dblocks = D / 256;
    const int f = blockIdx.x / dblocks, d = (blockIdx.x % dblocks) * 256 + threadIdx.x;
    const int t0 = blockIdx.y * TT;
    float wc[R];
    const float * col = a.w2 + (int64_t) f * R * D + d;   // W2 transposed at load: [5][R][D]
#pragma unroll
    for (int m = 0; m < R; m++) wc[m] = col[(int64_t) m * D];
    const float maa = a.maa[f][d];
    for (int i = threadIdx.x; i < TT * R; i += 256) {
        const int tt = i / R, m = i - tt * R;
        l_tl[i] = t0 + tt < T ? a.tl[(int64_t) (t0 + tt) * 5 * R + f * R + m] : 0.0f;
    }
    __syncthreads();
    float * const of = a.out[f];
    for (int tt = 0; tt < TT; tt++) {
        float val = 0.0f;                        // (tokens beyond T: zeros in the quantised image)
        if (t0 + tt < T) {
            const float4 * tl4 = reinterpret_cast<const float4 *>(l_tl + tt * R);
            float acc = 0.0f;
#pragma unroll
            for (int m4 = 0; m4 < R / 4; m4++) {
                const float4 t4 = tl4[m4];
                float pr;
                pr = wc[4 * m4] * t4.x; acc += pr;
                pr = wc[4 * m4 + 1] * t4.y; acc += pr;
                pr = wc[4 * m4 + 2] * t4.z; acc += pr;
                pr = wc[4 * m4 + 3] * t4.w; acc += pr;
            }
            const int64_t o = (int64_t) (t0 + tt) * D + d;
            const float mm = (acc + maa) * a.sx[o];
            val = mm + a.xn[o];
            if (of) of[o] = val;
        }
        if (qo.on) l_out[tt * QT_LD + threadIdx.x] = val;
    }
    if (qo.on) {
        __syncthreads();
        quantize_lds_tile(l_out, threadIdx.x, blockIdx.y, (blockIdx.x % dblocks) * 8, qo.nb, qo.dscale, qo.off, qo.q[f], qo.d[f], qo.s[f], qo.o[f]);
    }
}

// Token-shift mixes of a sequence (k_mix) with quantised outputs. One THREAD per (token, block of 32 channels): the block of the token and of
// the token in front sit in its registers, every output is mixed and quantised in place (quantize_block_thread) -- no LDS, no barrier.
// Same statements as k_mix (rwkv_graph.inc:204-215). Workgroup = 32 tokens x 8 blocks (the 256 channels of blockIdx.x).
// (Until round 6: thread = channel with the tile's 32 tokens in registers, each output staged through LDS into the wave-wide quantiser,
//  two barriers per output: 11.7 us per launch, the latency of one workgroup's chain.)
__global__ __launch_bounds__(256) void k_mix_seq_q(MixArgs a, TileOut qo, int T, int D) {
    const int i = threadIdx.x & 31, bl = threadIdx.x >> 5;
    const int b = blockIdx.x * 8 + bl;                    // block of 32 channels
    const int64_t tt = blockIdx.y, t = tt * 32 + i;
    const bool live = t < T;
    const int d0 = b * 32;
    float x[32], xp[32];
    {
        const float4 * xr = reinterpret_cast<const float4 *>(a.xn + (live ? t : 0) * D + d0);
        const float4 * pr = (live && t > 0) ? reinterpret_cast<const float4 *>(a.xn + (t - 1) * D + d0) : reinterpret_cast<const float4 *>(a.carry_in + d0);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float4 f = xr[j], g = pr[j];
            x[4 * j] = f.x; x[4 * j + 1] = f.y; x[4 * j + 2] = f.z; x[4 * j + 3] = f.w;
            xp[4 * j] = g.x; xp[4 * j + 1] = g.y; xp[4 * j + 2] = g.z; xp[4 * j + 3] = g.w;
        }
    }
    if (live && a.mode != 0 && a.sx) {
        float4 * sr = reinterpret_cast<float4 *>(a.sx + t * D + d0);
#pragma unroll
        for (int j = 0; j < 8; j++) sr[j] = make_float4(xp[4 * j] - x[4 * j], xp[4 * j + 1] - x[4 * j + 1], xp[4 * j + 2] - x[4 * j + 2], xp[4 * j + 3] - x[4 * j + 3]);
    }
    if (live && t == T - 1) {
        float4 * cr = reinterpret_cast<float4 *>(a.carry_out + d0);
#pragma unroll
        for (int j = 0; j < 8; j++) cr[j] = make_float4(x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]);
    }
    for (int f = 0; f < a.n_out; f++) {
        const float4 * cr = reinterpret_cast<const float4 *>(a.coef[f] + d0);
        float * const of = a.out[f];
        float val[32];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float4 c4 = cr[j];
            const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float xv = x[4 * j + e], pv = xp[4 * j + e], c = cv[e];
                float v;
                if (a.mode == 0) { const float xc = xv * c, pc = pv * c; v = xc + (pv - pc); }
                else { const float sx = pv - xv; const float sc = sx * c; v = sc + xv; }
                val[4 * j + e] = live ? v : 0.0f;        // (tokens beyond T: zeros in the quantised image)
            }
        }
        if (of && live) {
            float4 * orow = reinterpret_cast<float4 *>(of + t * D + d0);
#pragma unroll
            for (int j = 0; j < 8; j++) orow[j] = make_float4(val[4 * j], val[4 * j + 1], val[4 * j + 2], val[4 * j + 3]);
        }
        quantize_block_thread(val, i, tt * qo.nb + b, qo.dscale, qo.off, qo.q[f], qo.d[f], qo.s[f], qo.o[f]);
    }
}

// Group norm of the WKV output (+ gate) of a sequence with a quantised output (rwkv_graph.inc:279-283, 372-377; head size 64). One THREAD
// per (token, head): the 64 values sit in its registers, the two double sums walk the halving tree of the numerics spec by themselves
// (level o adds element l + o onto element l for l < o, o = 32 .. 1: the additions wave_sum_d makes across lanes, so the same bits as
// k_groupnorm), and the head's two blocks are quantised in place (quantize_block_thread). No lane exchange, no LDS, no barrier; the
// normalised f32 values never go to HBM. A wave = 32 tokens x 2 heads: its stores are contiguous 512-byte runs of the image.
// (The first version of this round kept k_groupnorm's wave-per-(token, head) reductions, 32 tokens in a row per wave, and staged the tile
//  through LDS: 14.7 us per launch against 13.7 + 6.3 for k_groupnorm + the quantiser -- the reductions' latency chain, not bytes.)
__global__ __launch_bounds__(64) void k_groupnorm_seq_q(const float * __restrict__ x, const float * __restrict__ lw, const float * __restrict__ lb, float eps,
                                                        const float * __restrict__ gate, TileOut qo, int T, int D) {
    const int i = threadIdx.x & 31;
    const int h = blockIdx.x * 2 + (threadIdx.x >> 5);
    const int64_t tt = blockIdx.y, t = tt * 32 + i;
    const bool live = t < T;
    float xv[64];
    const float4 * xr = reinterpret_cast<const float4 *>(x + (live ? t : 0) * D + h * 64);
#pragma unroll
    for (int j = 0; j < 16; j++) { const float4 f = xr[j]; xv[4 * j] = f.x; xv[4 * j + 1] = f.y; xv[4 * j + 2] = f.z; xv[4 * j + 3] = f.w; }
    double acc[32];
#pragma unroll
    for (int l = 0; l < 32; l++) acc[l] = (double) xv[l] + (double) xv[l + 32];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
#pragma unroll
        for (int l = 0; l < o; l++) acc[l] = acc[l] + acc[l + o];
    }
    const float mean = (float) (acc[0] / 64.0);
#pragma unroll
    for (int l = 0; l < 32; l++) { const float a = xv[l] - mean, c = xv[l + 32] - mean; acc[l] = (double) (a * a) + (double) (c * c); }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
#pragma unroll
        for (int l = 0; l < o; l++) acc[l] = acc[l] + acc[l + o];
    }
    const float var = (float) (acc[0] / 64.0);
    const float scale = 1.0f / sqrtf(var + eps);
    const float4 * wr = reinterpret_cast<const float4 *>(lw + h * 64), * br = reinterpret_cast<const float4 *>(lb + h * 64);
    const float4 * gr = gate ? reinterpret_cast<const float4 *>(gate + (live ? t : 0) * D + h * 64) : nullptr;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        float y[32];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float4 w4 = wr[half * 8 + j], b4 = br[half * 8 + j];
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
            float gv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if (gr) { const float4 g4 = gr[half * 8 + j]; gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w; }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v = (xv[half * 32 + 4 * j + e] - mean) * scale;
                v = v * wv[e];
                v = v + bv[e];
                if (gr) v *= gv[e];
                y[4 * j + e] = live ? v : 0.0f;                // (tokens beyond T: zeros in the quantised image)
            }
        }
        quantize_block_thread(y, i, tt * qo.nb + 2 * h + half, qo.dscale, qo.off, qo.q[0], qo.d[0], qo.s[0], qo.o[0]);
    }
}

static TileOut tile_out_of(int n, const TileAct * outs, int wtype, int64_t K) {
    TileOut qo;
    const bool hm = wtype == T_Q4_1 || wtype == T_Q5_1, xo = wtype == T_Q5_0;
    for (int i = 0; i < 6; i++) {
        const int j = i < n ? i : 0;
        qo.q[i] = n ? outs[j].q : nullptr; qo.d[i] = n ? outs[j].d : nullptr;
        qo.s[i] = n && hm ? outs[j].s : nullptr; qo.o[i] = n && xo ? outs[j].o : nullptr;
    }
    qo.dscale = wtype == T_Q4_0 ? 0.0625f : 1.0f; qo.off = 16.0f; qo.nb = (int) (K / 32); qo.on = n > 0;
    return qo;
}

// `outs` (5 tile images, or nullptr): the quantised outputs for products with weights of type `wtype`; a.out[f] may then be null
bool launch_v6_mix2_seq(const V6Mix2Args & a, int64_t T, int64_t D, int64_t R, hipStream_t st, const TileAct * outs, int wtype) {
    constexpr int TT = 32;
    if (D % 256 != 0 || !(R == 32 || R == 64)) return false;
    const TileOut qo = tile_out_of(outs ? 5 : 0, outs, wtype, D);
    const int64_t rows = outs ? outs[0].T_pad : T;
    const dim3 grid((unsigned) (5 * D / 256), (unsigned) ((rows + TT - 1) / TT));
    if (R == 32) hipLaunchKernelGGL((k_v6_mix2_seq<32, TT>), grid, dim3(256), 0, st, a, qo, (int) T, (int) D);
    else hipLaunchKernelGGL((k_v6_mix2_seq<64, TT>), grid, dim3(256), 0, st, a, qo, (int) T, (int) D);
    return true;
}

// group norm (+ gate) with the result only as the quantised image `out` (head size 64, D % 256 == 0; no RWKV-7 bonus term)
bool launch_groupnorm_seq_q(const float * x, const float * lw, const float * lb, float eps, const float * gate, int64_t T, int64_t H, int64_t S,
                            const TileAct & out, int wtype, hipStream_t st) {
    const int64_t D = H * S;
    if (S != 64 || D % 256 != 0 || H % 2 != 0 || getenv("RWKV_MI_NO_GN_QUANT") != nullptr) return false;
    const TileOut qo = tile_out_of(1, &out, wtype, D);
    const dim3 grid((unsigned) (H / 2), (unsigned) (out.T_pad / 32));
    hipLaunchKernelGGL(k_groupnorm_seq_q, grid, dim3(64), 0, st, x, lw, lb, eps, gate, qo, (int) T, (int) D);
    return true;
}

// token-shift mix with n_out <= 6 quantised outputs (a.out[f] may be null)
bool launch_mix_seq_q(const MixArgs & a, int64_t T, int64_t D, hipStream_t st, const TileAct * outs, int wtype) {
    if (D % 256 != 0 || a.n_out < 1 || a.n_out > 6) return false;
    const TileOut qo = tile_out_of(a.n_out, outs, wtype, D);
    const dim3 grid((unsigned) (D / 256), (unsigned) (outs[0].T_pad / 32));
    hipLaunchKernelGGL(k_mix_seq_q, grid, dim3(256), 0, st, a, qo, (int) T, (int) D);
    return true;
}

bool launch_wkv6_seq(const float * r, const float * k, const float * v, const float * u, int u_per_chan, const float * w, int w_mode,
                     const float * state_in, float * state_out, float * out, int64_t T, int64_t H, hipStream_t st) {
    const size_t lds = (size_t) (3 * 131 * 64 + 2 * 4 * 131) * sizeof(float);
    prefill_prepare_current_device();   // (dynamic-LDS limit: per device, set once per device)
    const dim3 grid((unsigned) (H * 8));
    switch (w_mode) {
        case 0: hipLaunchKernelGGL((k_wkv6_seq<0>), grid, dim3(256), lds, st, r, k, v, u, u_per_chan, w, state_in, state_out, out, (int) T, (int) H); break;
        case 1: hipLaunchKernelGGL((k_wkv6_seq<1>), grid, dim3(256), lds, st, r, k, v, u, u_per_chan, w, state_in, state_out, out, (int) T, (int) H); break;
        default: hipLaunchKernelGGL((k_wkv6_seq<2>), grid, dim3(256), lds, st, r, k, v, u, u_per_chan, w, state_in, state_out, out, (int) T, (int) H); break;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// WKV-7 over a sequence (rwkv_wkv_v7_impl, rwkv_operators_wkv_v7.inc:37-107).
//
// Per token, head and value row i:  sa = sum_j a_j s_ij  (j = 0 .. 63 IN THAT ORDER, from 0),  s_ij <- (s_ij w_j + v_i k_j) + sa b_j,
// out_i = sum_j s_ij r_j (same order, from 0). The first sum needs the whole row before any element of it may change, so -- unlike
// WKV-6 -- tokens cannot be skewed across the lanes of a row: every token costs one 64-step dependent chain per row, whatever the
// layout. The single-token form (k_wkv7: lane = row, the row's 64 elements in registers) pays that chain once per 64 rows but also
// runs the 64-element update and the second chain serially in every lane: ~580 instructions per token on H waves (40 of the chip's
// 1024 SIMDs at 2.9B). Here a wave owns FOUR rows, lane = 16 * row + q with q = elements 4 q .. 4 q + 3 of the row:
//   * the update is 4 elements per lane (28 instructions instead of 448);
//   * the sa chain: every lane of the row adds the row's 64 products itself, in order, reading each one through a DPP row_newbcast of
//     its owner (w7_row_sum above): 64 dependent full-rate adds per token -- the floor -- and the sum is in all 16 lanes when it ends;
//   * the out chain IS skewed: it only reads. Lane q parks its four products s_ij r_j of token t in an LDS ring (own column, used as an
//     indexed register file: no barrier) and at step sigma adds those of token sigma - 1 - q onto the sum it receives from lane q - 1;
//     lane 15 emits out[sigma - 16] (one step of lag: every lane's entry is from an earlier step and is read a step ahead). Four adds per
//     token instead of 64.
// ~100 instructions per token and wave, 16 H workgroups of 4 waves. Per-token operands are staged through LDS in chunks of 32 tokens
// (global loads of chunk c + 1 in flight during chunk c), read as one 16-byte vector per array and token, one token ahead.
// ---------------------------------------------------------------------------------------------------------------
constexpr int W7_CH = 32;                                                     // tokens per staged chunk (the staging loops assume 32)
constexpr int W7_LDS = (2 * W7_CH * (5 * 64 + 16) + 4 * 16 * 64 * 4) * 4;     // two chunk buffers (r w k a b: 64 each, v: 16 rows) + the out-chain rings

// sa of a row = ((((0 + p_0) + p_1) + ...) + p_63): the row's 16 lanes hold p_{4q .. 4q+3}; every lane of the row adds all 64 products
// itself, in order, each read through a DPP row_newbcast of the lane that owns it (v_add_f32_dpp: one full-rate instruction per
// term). All 16 lanes end with the same, complete sum: no hand-over of partial sums from lane to lane, no LDS permute at the end.
// The instructions are written out by hand: the compiler's hazard recogniser puts two wait states in front of every DPP instruction
// whose ANY source was written by the previous VALU instruction; the hardware needs them for the DPP-permuted source only
// (the products, written long before), not for the accumulator, which is read like any other source 1.
template <int Q> __device__ __forceinline__ float w7_bc(float p) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0x150 + Q /* row_newbcast:Q */, 0xF, 0xF, true));
}
template <int Q> __device__ __forceinline__ void w7_chain(float & x, float p0, float p1, float p2, float p3) {
    if constexpr (Q < 16) {
        x = w7_bc<Q>(p0) + x; x = w7_bc<Q>(p1) + x; x = w7_bc<Q>(p2) + x; x = w7_bc<Q>(p3) + x;
        w7_chain<Q + 1>(x, p0, p1, p2, p3);
    }
}
__device__ __forceinline__ float w7_row_sum(float p0, float p1, float p2, float p3) {
    float x = 0.0f;
#ifdef W7_CHAIN_BUILTIN
    w7_chain<0>(x, p0, p1, p2, p3);
#else
#define W7_A(Q) "v_add_f32_dpp %0, %1, %0 row_newbcast:" #Q " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                "v_add_f32_dpp %0, %2, %0 row_newbcast:" #Q " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                "v_add_f32_dpp %0, %3, %0 row_newbcast:" #Q " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                "v_add_f32_dpp %0, %4, %0 row_newbcast:" #Q " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm("s_nop 1\n\t"                          // (the products may have been written by the instruction just before)
                 W7_A(0) W7_A(1) W7_A(2) W7_A(3) W7_A(4) W7_A(5) W7_A(6) W7_A(7)
                 W7_A(8) W7_A(9) W7_A(10) W7_A(11) W7_A(12) W7_A(13) W7_A(14) W7_A(15)
                 : "+v"(x) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));
#undef W7_A
#endif
    return x;
}

__global__ __launch_bounds__(256) void k_wkv7_seq(const float * __restrict__ r, const float * __restrict__ w, const float * __restrict__ k,
                                                  const float * __restrict__ v, const float * __restrict__ a, const float * __restrict__ b,
                                                  const float * __restrict__ state_in, float * __restrict__ state_out, float * __restrict__ out,
                                                  int T, int H) {
    constexpr int S = 64, CH = W7_CH, TOKF = 5 * 64 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float * l_tok = reinterpret_cast<float *>(lds_raw);                       // [2][CH][r 64 | w 64 | k 64 | a 64 | b 64 | v 16]
    float4 * l_ring = reinterpret_cast<float4 *>(l_tok + 2 * CH * TOKF);      // [4 waves][16 tokens][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t h = blockIdx.x >> 2;
    const int rg = (int) (blockIdx.x & 3);                                    // 16 rows of the head per workgroup
    const int q = lane & 15, rr = lane >> 4;
    const int row_wg = 4 * wave + rr, i = 16 * rg + row_wg;
    const int64_t D = (int64_t) H * S;
    float4 * const ring = l_ring + wave * 16 * 64 + lane;

    v2f s01, s23;                                               // elements 4 q .. 4 q + 3 of row i
    {
        const float4 q4 = *reinterpret_cast<const float4 *>(state_in + h * S * S + (int64_t) i * S + 4 * q);
        s01 = (v2f){q4.x, q4.y}; s23 = (v2f){q4.z, q4.w};
    }
    // staging: chunk c = tokens [CH c, CH c + CH): five arrays of CH x 16 float4 = ten per thread; v: CH x 16 floats = two per thread
    static_assert(CH == 32, "the staging below is written for chunks of 32 tokens");
    float4 g0, g1, g2, g3, g4, g5, g6, g7, g8, g9;     // (named registers: an array indexed inside the lambdas ended up in scratch)
    float gva, gvb;
    auto ld4 = [&](const float * __restrict__ src, int c, int half) __attribute__((always_inline)) -> float4 {
        const int rem = half * 256 + tid;                       // (token, float4) of the chunk: 16 float4 per token
        int t = CH * c + (rem >> 4);
        t = t < T ? t : T - 1;                                  // (clamped tokens are never consumed)
        return *reinterpret_cast<const float4 *>(src + (int64_t) t * D + h * S + (rem & 15) * 4);
    };
    auto ldv = [&](int c, int half) __attribute__((always_inline)) -> float {
        const int e = half * 256 + tid;
        int t = CH * c + (e >> 4);
        t = t < T ? t : T - 1;
        return v[(int64_t) t * D + h * S + 16 * rg + (e & 15)];
    };
    auto issue = [&](int c) __attribute__((always_inline)) {
        g0 = ld4(r, c, 0); g1 = ld4(r, c, 1); g2 = ld4(w, c, 0); g3 = ld4(w, c, 1); g4 = ld4(k, c, 0); g5 = ld4(k, c, 1);
        g6 = ld4(a, c, 0); g7 = ld4(a, c, 1); g8 = ld4(b, c, 0); g9 = ld4(b, c, 1);
        gva = ldv(c, 0); gvb = ldv(c, 1);
    };
    auto st4 = [&](float * buf, int arr, int half, const float4 & val) __attribute__((always_inline)) {
        const int rem = half * 256 + tid;
        *reinterpret_cast<float4 *>(buf + (rem >> 4) * TOKF + arr * 64 + (rem & 15) * 4) = val;
    };
    auto commit = [&](int c) __attribute__((always_inline)) {
        float * buf = l_tok + (c & 1) * CH * TOKF;
        st4(buf, 0, 0, g0); st4(buf, 0, 1, g1); st4(buf, 1, 0, g2); st4(buf, 1, 1, g3); st4(buf, 2, 0, g4); st4(buf, 2, 1, g5);
        st4(buf, 3, 0, g6); st4(buf, 3, 1, g7); st4(buf, 4, 0, g8); st4(buf, 4, 1, g9);
        buf[(tid >> 4) * TOKF + 320 + (tid & 15)] = gva;
        buf[((256 + tid) >> 4) * TOKF + 320 + (tid & 15)] = gvb;
    };
    struct Tok { float4 r, w, k, a, b; float v; };
    auto read_tok = [&](int c, int tt, Tok & o) __attribute__((always_inline)) {
        const float * bt = l_tok + ((c & 1) * CH + tt) * TOKF + 4 * q;
        o.r = *reinterpret_cast<const float4 *>(bt);
        o.w = *reinterpret_cast<const float4 *>(bt + 64);
        o.k = *reinterpret_cast<const float4 *>(bt + 128);
        o.a = *reinterpret_cast<const float4 *>(bt + 192);
        o.b = *reinterpret_cast<const float4 *>(bt + 256);
        o.v = l_tok[((c & 1) * CH + tt) * TOKF + 320 + row_wg];
    };
    float o_run = 0.0f;                                          // the out chain's running sum as this lane last produced it
    float4 pq_n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);           // the ring entry this lane adds at the next step (read a step ahead)
    int64_t t_off = -16 * D;                                     // out row of the token the chain emits at this step (sigma - 16)
    const int64_t lane_off = h * S + i;
    // one step: `upd` = a token is consumed (state update + its products into the ring); the out chain always advances. At step sigma
    // lane q adds the products of token sigma - 1 - q: every lane's entry was written in an earlier step (lane 0's in the one before),
    // so it is read a whole step ahead, right behind the ring write, and its LDS round trip is off the token's path.
    auto step = [&](int sigma, bool upd, const Tok & tk) __attribute__((always_inline)) {
        v2f o01 = {0.0f, 0.0f}, o23 = {0.0f, 0.0f};
        if (upd) {
            const v2f p01 = (v2f){tk.a.x, tk.a.y} * s01, p23 = (v2f){tk.a.z, tk.a.w} * s23;
            // independent of sa: v k, s w and their sum
            const v2f vv = {tk.v, tk.v};
            const v2f t01 = s01 * (v2f){tk.w.x, tk.w.y} + vv * (v2f){tk.k.x, tk.k.y};
            const v2f t23 = s23 * (v2f){tk.w.z, tk.w.w} + vv * (v2f){tk.k.z, tk.k.w};
            const float sa = w7_row_sum(p01.x, p01.y, p23.x, p23.y);
            __builtin_amdgcn_sched_barrier(0);                   // (the out chain below stays below: its ring entry needs no wait there)
            const v2f sv = {sa, sa};
            s01 = t01 + sv * (v2f){tk.b.x, tk.b.y};
            s23 = t23 + sv * (v2f){tk.b.z, tk.b.w};
            o01 = s01 * (v2f){tk.r.x, tk.r.y}; o23 = s23 * (v2f){tk.r.z, tk.r.w};
        }
        {   // (behind the sa chain: the entry read at the end of the previous step has long arrived)
            const float4 pq = pq_n;
            float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o_run), 0x111, 0xF, 0xF, false)) + pq.x;
            o += pq.y; o += pq.z; o += pq.w;
            o_run = o;
            if (q == 15 && sigma >= 16) (out + t_off)[lane_off] = o;      // token sigma - 16 (< T: sigma < T + 16)
            t_off += D;
        }
        if (upd) ring[(sigma & 15) * 64] = make_float4(o01.x, o01.y, o23.x, o23.y);
        pq_n = ring[((sigma - q) & 15) * 64];                    // token sigma - q for step sigma + 1 (lane 15: before step sigma + 1 overwrites the slot)
    };

    const int n_chunks = (T + CH - 1) / CH;
    issue(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; c++) {
        const bool more = c + 1 < n_chunks;
        if (more) issue(c + 1);
        const int n = T - CH * c < CH ? T - CH * c : CH;
        Tok ta, tb;
        read_tok(c, 0, ta);
        if (n == CH) {
            // whole chunk: no conditions inside, and the LDS queue has the same shape on entry as on the back edge (the ring entry is
            // read again -- the same one -- behind the first token's operands), so the operand waits are counted and wait for nothing
            // that was issued after what they need
            pq_n = ring[((CH * c - 1 - q) & 15) * 64];
#pragma unroll 1
            for (int tt = 0; tt < CH - 2; tt += 2) {
                read_tok(c, tt + 1, tb);
                step(CH * c + tt, true, ta);
                read_tok(c, tt + 2, ta);
                step(CH * c + tt + 1, true, tb);
            }
            read_tok(c, CH - 1, tb);
            step(CH * c + CH - 2, true, ta);
            step(CH * c + CH - 1, true, tb);
        } else {
            for (int tt = 0; tt < n; tt += 2) {
                if (tt + 1 < CH) read_tok(c, tt + 1, tb);
                step(CH * c + tt, true, ta);
                if (tt + 1 < n) {
                    if (tt + 2 < CH) read_tok(c, tt + 2, ta);
                    step(CH * c + tt + 1, true, tb);
                }
            }
        }
        if (more) commit(c + 1);
        __syncthreads();
    }
    {
        Tok none{};
        for (int sigma = T; sigma < T + 16; sigma++) step(sigma, false, none);   // drain the out chain
    }
    *reinterpret_cast<float4 *>(state_out + h * S * S + (int64_t) i * S + 4 * q) = make_float4(s01.x, s01.y, s23.x, s23.y);
}

bool launch_wkv7_seq(const float * r, const float * w, const float * k, const float * v, const float * a, const float * b,
                     const float * state_in, float * state_out, float * out, int64_t T, int64_t H, hipStream_t st) {
    prefill_prepare_current_device();
    hipLaunchKernelGGL(k_wkv7_seq, dim3((unsigned) (H * 4)), dim3(256), W7_LDS, st, r, w, k, v, a, b, state_in, state_out, out, (int) T, (int) H);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

// visiting order of the blocks of a row of nb blocks (see the header): leaves in bit-reversed order, a leaf's blocks ascending.
// `out`: the blocks alone (what the staging reads); `walk`: one entry per block plus one per leaf without blocks, with the merge
// code of the kernel (0 = leaf continues; m + 1 = leaf ends, m = trailing one bits of the leaf's number in visiting order).
static void walk_order(int nb, std::vector<unsigned short> & out, std::vector<unsigned> & walk) {
    out.clear(); walk.clear();
    for (int c = 0; c < 64; c++) {
        int l = 0, m = 0;
        for (int k = 0; k < 6; k++) if (c & (1 << k)) l |= 1 << (5 - k);
        while (m < 6 && (c & (1 << m))) m++;
        const size_t first = walk.size();
        for (int b = l; b < nb; b += 64) { out.push_back((unsigned short) b); walk.push_back((unsigned) b); }
        if (walk.size() == first) walk.push_back(0xFFFFu);
        walk.back() |= (unsigned) (m + 1) << 16;
    }
}

static std::mutex g_pf_mu;

// device tables per (device, row length): a few hundred bytes each, kept for the life of the process
struct WalkTab { int * order; int a_start[9]; };
static std::unordered_map<long long, WalkTab> g_walk_tables;

static const WalkTab * walk_table(int nb) {
    std::lock_guard<std::mutex> lk(g_pf_mu);
    int dev = 0;
    (void) hipGetDevice(&dev);
    const long long key = ((long long) dev << 32) | (unsigned) nb;
    auto & walk_tables = g_walk_tables;
    auto it = walk_tables.find(key);
    if (it != walk_tables.end()) return &it->second;
    std::vector<unsigned short> o;
    std::vector<unsigned> wk;
    walk_order(nb, o, wk);
    std::vector<int> o32(o.begin(), o.end());
    // steps before outer iteration a (8 leaves each, in visiting order): where a part of a split walk starts
    WalkTab wt{nullptr, {0}};
    {
        int steps = 0;
        for (int a = 0; a <= 8; a++) {
            wt.a_start[a] = steps;
            if (a == 8) break;
            for (int u = 0; u < 8; u++) {
                const int c = 8 * a + u;
                int l = 0;
                for (int k = 0; k < 6; k++) if (c & (1 << k)) l |= 1 << (5 - k);
                for (int b = l; b < nb; b += 64) steps++;
            }
        }
    }
    int * d = nullptr;
    if (hipMalloc((void **) &d, o32.size() * sizeof(int) + 64) != hipSuccess) return nullptr;
    if (hipMemcpy(d, o32.data(), o32.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void) hipFree(d); return nullptr; }
    wt.order = d;
    walk_tables[key] = wt;
    return &walk_tables[key];
}

// builds (once) the tile-major image of a quantised matrix
bool ensure_pf(const DevTensor & W, hipStream_t st) {
    if (W.pf_qs) return true;
    std::lock_guard<std::mutex> lk(g_pf_mu);
    if (W.pf_qs) return true;
    const int64_t N = W.rows();
    const int nb = (int) (W.cols() / 32);
    const int64_t RT = (N + 31) / 32;
    const bool q8 = W.type == T_Q8_0, qh = W.type == T_Q5_0 || W.type == T_Q5_1;
    const size_t n_qs = (size_t) RT * nb * 32 * (q8 ? 32 : 16), n_sc = (size_t) RT * nb * 32 * 4;
    const size_t total = n_qs + n_sc + (qh ? n_sc : 0);
    uint8_t * base = nullptr;
    if (hipMalloc((void **) &base, total) != hipSuccess) return false;
    uint32_t * psc = reinterpret_cast<uint32_t *>(base + n_qs);
    uint32_t * pqh = qh ? reinterpret_cast<uint32_t *>(base + n_qs + n_sc) : nullptr;
    const int64_t n_thr = RT * nb * 32;
    hipLaunchKernelGGL(k_pf_repack, dim3((unsigned) ((n_thr + 255) / 256)), dim3(256), 0, st, W.type, W.qs, W.qh, W.sc, N, nb, base, psc, pqh);
    if (hipStreamSynchronize(st) != hipSuccess) { (void) hipFree(base); return false; }
    W.pf_sc = psc; W.pf_qh = pqh; W.pf_rows = RT * 32;
    W.pf_qs = base;   // published last
    return true;
}

void free_pf(const DevTensor & W) {
    if (W.pf_qs) (void) hipFree(W.pf_qs);
    W.pf_qs = nullptr; W.pf_sc = nullptr; W.pf_qh = nullptr;
}

size_t tile_act_bytes(int64_t T, int64_t K) {
    const int64_t T_pad = (T + 63) / 64 * 64;
    return (size_t) T_pad * K + 3 * (size_t) T_pad * (K / 32) * 4 + 1024;
}

TileAct tile_act_at(void * base, int64_t T, int64_t K) {
    const int64_t T_pad = (T + 63) / 64 * 64;
    TileAct a;
    a.q = (int8_t *) base;
    a.d = (float *) ((uint8_t *) base + (((size_t) T_pad * K + 255) / 256) * 256);
    a.s = a.d + T_pad * (K / 32);
    a.o = a.s + T_pad * (K / 32);
    a.T_pad = T_pad;
    return a;
}

void launch_quantize_act_tiles_batched(int n, const float * const * xs, int64_t T, int64_t K, int wtype, const TileAct * outs, hipStream_t st) {
    const int nb = (int) (K / 32);
    const int64_t n_tiles = (outs[0].T_pad / 32) * nb;
    const float dscale = wtype == T_Q4_0 ? 0.0625f : 1.0f;
    const bool hm = wtype == T_Q4_1 || wtype == T_Q5_1, xo = wtype == T_Q5_0;
    QuantBatch qb;
    for (int i = 0; i < QUANT_BATCH; i++) {
        const int j = i < n ? i : 0;
        qb.x[i] = xs[j]; qb.q[i] = outs[j].q; qb.d[i] = outs[j].d; qb.s[i] = hm ? outs[j].s : nullptr; qb.o[i] = xo ? outs[j].o : nullptr;
    }
    hipLaunchKernelGGL(k_quant_act_tiles, dim3((unsigned) ((n_tiles + 3) / 4), (unsigned) n), dim3(256), 0, st, qb, T, outs[0].T_pad, nb, dscale, 16.0f);
}

void launch_quantize_act_tiles(const float * x, int64_t T, int64_t K, int wtype, const TileAct & out, hipStream_t st) {
    launch_quantize_act_tiles_batched(1, &x, T, K, wtype, &out, st);
}

template <int FMT>
static bool launch_mmq_mfma_t(int n, const DevTensor * const * Ws, const TileAct * xs, float * const * ys, const Epi * epis, int64_t T, int64_t ldy,
                              const MmqWs * ws, hipStream_t st, const MmqQOut * yq = nullptr) {
    typedef MF<FMT> M;
    const DevTensor & W0 = *Ws[0];
    const int64_t N = W0.rows();
    const int nb = (int) (W0.cols() / 32);
    const WalkTab * wt = walk_table(nb);
    if (!wt) return false;
    MmqArgs A;
    for (int i = 0; i < n; i++) if (!Ws[i]->pf_qs && !ensure_pf(*Ws[i], st)) return false;
    for (int i = 0; i < MMQ_BATCH; i++) {
        const int j = i < n ? i : 0;
        A.w[i] = PfW{Ws[j]->pf_qs, Ws[j]->pf_sc, Ws[j]->pf_qh};
        A.x[i] = PfX{xs[j].q, xs[j].d, xs[j].s, xs[j].o};
        A.y[i] = ys[j];
        A.epi[i] = epis[j];
    }
    const int RT = (int) ((N + 31) / 32), RP = (RT + M::RGN - 1) / M::RGN, C = (int) ((T + 63) / 64);
    // too few tiles for the chip and every leaf of the walk has a block: cut the walk (see MmqArgs)
    int split = 1;
    const int64_t tiles = (int64_t) n * RP * C;
    if (ws && ws->part && nb >= 64) {
        while (split < 8 && tiles * split * 2 <= 256) split *= 2;
        while (split > 1 && (size_t) split * tiles * M::NT * 64 > ws->part_bytes) split /= 2;
    }
    A.N = N; A.T = T; A.ldy = ldy; A.nb = nb; A.RT = RT; A.C = C; A.split = split;
    A.order = wt->order; for (int a = 0; a < 9; a++) A.a_start[a] = wt->a_start[a];
    A.part = ws ? ws->part : nullptr; A.counters = ws ? ws->counters : nullptr;
    if (yq) { if (split != 1 || n != 1 || N % 32 != 0) return false; A.yq = *yq; }
    static_assert(8 * 32 * 33 * 4 <= M::LDS_BYTES, "the quantising epilogue's patches");
    const size_t lds = (size_t) M::LDS_BYTES;
    prefill_prepare_current_device();   // (dynamic-LDS limit: per device, set once per device)
    const dim3 grid((unsigned) (RP < 8 ? RP * C : ((RP + 7) / 8) * 8 * C), (unsigned) n, (unsigned) split);
    hipLaunchKernelGGL((k_mmq_mfma<FMT>), grid, dim3(M::NT), lds, st, A);
    const dim3 cgrid((unsigned) (RP * C), (unsigned) n, 8);
    if (split == 2) hipLaunchKernelGGL(k_mmq_combine<2>, cgrid, dim3(512), 0, st, A);
    else if (split == 4) hipLaunchKernelGGL(k_mmq_combine<4>, cgrid, dim3(512), 0, st, A);
    else if (split == 8) hipLaunchKernelGGL(k_mmq_combine<8>, cgrid, dim3(512), 0, st, A);
    return true;
}

// hipFuncSetAttribute acts on the CURRENT device: the dynamic-LDS limits of the sequence-mode kernels are raised once per DEVICE (a bitmap
// under g_pf_mu), from create_context (engine.hip) and again, as a cheap check, from the launchers -- a process may hold contexts on
// several devices (RWKV_MI_DEVICES, rwkv_mi_init_stage after hipSetDevice, clones on other threads).
void prefill_prepare_current_device() {
    static uint64_t ready[4] = {0, 0, 0, 0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) return;
    std::lock_guard<std::mutex> lk(g_pf_mu);
    if (ready[dev >> 6] & (1ull << (dev & 63))) return;
    ready[dev >> 6] |= 1ull << (dev & 63);
    (void) hipFuncSetAttribute((const void *) k_mmq_mfma<T_Q4_0>, hipFuncAttributeMaxDynamicSharedMemorySize, MF<T_Q4_0>::LDS_BYTES);
    (void) hipFuncSetAttribute((const void *) k_mmq_mfma<T_Q4_1>, hipFuncAttributeMaxDynamicSharedMemorySize, MF<T_Q4_1>::LDS_BYTES);
    (void) hipFuncSetAttribute((const void *) k_mmq_mfma<T_Q5_0>, hipFuncAttributeMaxDynamicSharedMemorySize, MF<T_Q5_0>::LDS_BYTES);
    (void) hipFuncSetAttribute((const void *) k_mmq_mfma<T_Q5_1>, hipFuncAttributeMaxDynamicSharedMemorySize, MF<T_Q5_1>::LDS_BYTES);
    (void) hipFuncSetAttribute((const void *) k_mmq_mfma<T_Q8_0>, hipFuncAttributeMaxDynamicSharedMemorySize, MF<T_Q8_0>::LDS_BYTES);
    const int wkv_lds = (int) ((3 * 131 * 64 + 2 * 4 * 131) * sizeof(float));
    (void) hipFuncSetAttribute((const void *) k_wkv6_seq<0>, hipFuncAttributeMaxDynamicSharedMemorySize, wkv_lds);
    (void) hipFuncSetAttribute((const void *) k_wkv6_seq<1>, hipFuncAttributeMaxDynamicSharedMemorySize, wkv_lds);
    (void) hipFuncSetAttribute((const void *) k_wkv6_seq<2>, hipFuncAttributeMaxDynamicSharedMemorySize, wkv_lds);
    (void) hipFuncSetAttribute((const void *) k_wkv7_seq, hipFuncAttributeMaxDynamicSharedMemorySize, W7_LDS);
    (void) hipGetLastError();
}

bool launch_mmq_mfma_batched(int n, const DevTensor * const * Ws, const TileAct * xs, float * const * ys, const Epi * epis, int64_t T, int64_t ldy,
                             const MmqWs * ws, hipStream_t st) {
    if (n < 1 || n > MMQ_BATCH) return false;
    for (int i = 1; i < n; i++) if (Ws[i]->type != Ws[0]->type || Ws[i]->rows() != Ws[0]->rows() || Ws[i]->cols() != Ws[0]->cols()) return false;
    // RWKV_MI_SEQ_Q=fast: block sums in plain K order (prefill_fast.hip) where the shape fills the chip; the default is the walk below (bit-identical to the serial path)
    if (launch_mmq_fast(n, Ws, xs, ys, epis, T, ldy, st)) return true;
    switch (Ws[0]->type) {
        case T_Q4_0: return launch_mmq_mfma_t<T_Q4_0>(n, Ws, xs, ys, epis, T, ldy, ws, st);
        case T_Q4_1: return launch_mmq_mfma_t<T_Q4_1>(n, Ws, xs, ys, epis, T, ldy, ws, st);
        case T_Q5_0: return launch_mmq_mfma_t<T_Q5_0>(n, Ws, xs, ys, epis, T, ldy, ws, st);
        case T_Q5_1: return launch_mmq_mfma_t<T_Q5_1>(n, Ws, xs, ys, epis, T, ldy, ws, st);
        case T_Q8_0: return launch_mmq_mfma_t<T_Q8_0>(n, Ws, xs, ys, epis, T, ldy, ws, st);
        default: return false;
    }
}

// y = epi(W . x) written ONLY as the quantised tile image `out` of a following product with weights of type out_wtype (exact arm; false:
// not applicable -- the caller runs the plain product and quantises its output as before). Nothing reads partial sums: split walks excluded.
bool launch_mmq_mfma_q(const DevTensor & W, const TileAct & x, int64_t T, const Epi & epi, const TileAct & out, int out_wtype, hipStream_t st) {
    const bool off = getenv("RWKV_MI_NO_EPI_QUANT") != nullptr;      // (A/B, tests: read per call)
    if (off || seq_q_arm() != 0 || !dtype_quantized(out_wtype) || W.rows() % 32 != 0 || out.T_pad < (T + 63) / 64 * 64) return false;
    MmqQOut yq;
    const bool hm = out_wtype == T_Q4_1 || out_wtype == T_Q5_1, xo = out_wtype == T_Q5_0;
    yq.q = out.q; yq.d = out.d; yq.s = hm ? out.s : nullptr; yq.o = xo ? out.o : nullptr;
    yq.dscale = out_wtype == T_Q4_0 ? 0.0625f : 1.0f; yq.off = 16.0f; yq.nb = (int) (W.rows() / 32);
    const DevTensor * Wp = &W;
    float * y = nullptr;
    switch (W.type) {
        case T_Q4_0: return launch_mmq_mfma_t<T_Q4_0>(1, &Wp, &x, &y, &epi, T, W.rows(), nullptr, st, &yq);
        case T_Q4_1: return launch_mmq_mfma_t<T_Q4_1>(1, &Wp, &x, &y, &epi, T, W.rows(), nullptr, st, &yq);
        case T_Q5_0: return launch_mmq_mfma_t<T_Q5_0>(1, &Wp, &x, &y, &epi, T, W.rows(), nullptr, st, &yq);
        case T_Q5_1: return launch_mmq_mfma_t<T_Q5_1>(1, &Wp, &x, &y, &epi, T, W.rows(), nullptr, st, &yq);
        case T_Q8_0: return launch_mmq_mfma_t<T_Q8_0>(1, &Wp, &x, &y, &epi, T, W.rows(), nullptr, st, &yq);
        default: return false;
    }
}

bool launch_mmq_mfma(const DevTensor & W, const TileAct & x, int64_t T, float * y, int64_t ldy, const Epi & epi, hipStream_t st, const MmqWs * ws) {
    const DevTensor * Wp = &W;
    return launch_mmq_mfma_batched(1, &Wp, &x, &y, &epi, T, ldy, ws, st);
}

}  // namespace rwkvmi
