// prefill_mm.h -- what the two sequence-mode GEMM kernels on v_mfma_i32_32x32x32_i8 share (prefill.hip: the walk that reproduces the
// single-token kernel's additions bit for bit; prefill_fast.hip: block sums accumulated in plain K order): the per-step LDS slot layout
// of the tile-major images, the image pointers and the launch arguments.
#pragma once

#include "kdev.h"
#include "model.h"

#include <atomic>

namespace rwkvmi {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));


template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

template <int FMT> struct MF {
    static constexpr bool Q8 = FMT == T_Q8_0;
    static constexpr bool QH = QF<FMT>::QH, HM = QF<FMT>::HM;
    static constexpr bool XO = FMT == T_Q5_0;                   // offset through the activation sums (float, exact)
    // workgroup = RGN x 2 waves: RGN row groups of 32 rows x 2 token groups of 32 tokens. Measured (2048 x 2048 x 1024, Q4_0): RGN = 4
    // (512 threads, 128 x 64 tile, one workgroup per CU) 36.6 us; RGN = 2 (256 threads, 64 x 64 tile, two workgroups per CU covering
    // each other's barrier / DMA stalls) 44.0 us -- the larger tile's operand reuse is worth more than the second workgroup.
    static constexpr int  RGN = 4, NT = RGN * 2 * 64, ROWS = RGN * 32;
    // One step's operands in LDS = a sequence of 1-KiB "DMA rows" (one global_load_lds_dwordx4 of a wave each):
    //   NW rows of weight codes | 2 rows of token codes (one per 32-token tile) | tail: 16-byte pieces of the small arrays, in the order
    //   weight scales (32) [fifth bits (32)] token scales d (16) [s (16)] [o (16)]
    static constexpr int  WC = ROWS * (Q8 ? 32 : 16);           // LDS bytes of weight codes per step
    static constexpr int  NW = WC / 1024;
    static constexpr int  OFF_WC = 0, OFF_XQ = WC, OFF_WSC = OFF_XQ + 2048, OFF_WQH = OFF_WSC + ROWS * 4, OFF_XD = OFF_WQH + (QH ? ROWS * 4 : 0),
                          OFF_XS = OFF_XD + 256, OFF_XO = OFF_XS + (HM ? 256 : 0), SLOT = OFF_XO + (XO ? 256 : 0);
    static constexpr int  P_WSC = ROWS / 4, P_WQH = QH ? ROWS / 4 : 0, P_XD = 16, P_XS = HM ? 16 : 0, P_XO = XO ? 16 : 0;
    static constexpr int  P_TAIL = P_WSC + P_WQH + P_XD + P_XS + P_XO, NTR = (P_TAIL + 63) / 64, NR = NW + 2 + NTR;   // rows per step
    // steps per chunk; stack levels kept in LDS (64 B per thread each) so that two chunks of 8 steps + the levels fit in 160 KiB
    static constexpr bool OVERLAP = !(QH && HM);                // the MFMA of step sigma + 1 runs under the fold of step sigma (Q5_1: no registers for it)
#ifndef PF_CH
#define PF_CH 8
#endif
#ifndef PF_NBUF
#define PF_NBUF 2
#endif
    // NBUF chunk buffers of CH steps: chunk k + NBUF - 1 is issued when chunk k starts, (NBUF - 1) CH steps before its first read; the
    // boundary waits with a counted vmcnt for chunk k only. Measured in round 4 (1.6B Q4_0, average launch of the 1024-token pass,
    // profiles/r04q_mmq_variants.txt): CH 8 x 2 buffers 40.7 us; CH 4 x 2 / 3 / 4 buffers 42.9 / 43.6 / 43.2 us -- the depth of the
    // prefetch does not matter, the extra barriers cost 6 %. The timing-only builds behind -DPF_EXP_* (results invalid: parts of the
    // step are skipped) say where the time is NOT: without the fold 38.6 us, without the LDS operand reads 41.4, without the scale
    // reads 41.3, without the merges 39.7, without the DMAs after the first chunks 39.2, without the boundary wait + barrier 38.3,
    // without all of these together 31.8 us. What is left in that last build -- the nibble unpack, the MFMA, the magic subtraction,
    // the walk's control flow and the launch's fixed part -- is 78 % of the kernel: no single resource the profiler names is the bound.
    static constexpr int CH = PF_CH, NBUF = PF_NBUF, STK_LDS = Q8 ? 1 : 2;
    static constexpr int LDS_BYTES = NBUF * CH * SLOT + STK_LDS * 64 * NT;
    // DMA instructions one wave issues per chunk (its vmcnt share): rows r = sub, sub + NSUB, ... of its step; a tail row is one DMA per
    // small array that has pieces in it (see `seg` in the kernel)
    static constexpr int n_dma(int sub) {
        constexpr int NSUB_ = 8 / CH;
        const int first[5] = {0, P_WSC, P_WSC + P_WQH, P_WSC + P_WQH + P_XD, P_WSC + P_WQH + P_XD + P_XS};
        const int count[5] = {P_WSC, P_WQH, P_XD, P_XS, P_XO};
        int n = 0;
        for (int r = 0; r < NR; r++) {
            if (r % NSUB_ != sub) continue;
            if (r < NW + 2) { n++; continue; }
            const int lo = 64 * (r - NW - 2), hi = lo + 64;
            for (int a = 0; a < 5; a++) if (count[a] != 0 && first[a] < hi && first[a] + count[a] > lo) n++;
        }
        return n;
    }
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

struct PfW { const uint8_t * qs; const uint32_t * sc; const uint32_t * qh; };
struct PfX { const int8_t * q; const float * d; const float * s; const float * o; };

// One launch = up to MMQ_BATCH products of the same shape (blockIdx.y: e.g. the r / k / v / g projections of a layer), each cut
// along the walk into `split` parts (blockIdx.z) when the output has too few tiles to fill the chip: the outer iterations a of the
// walk are the 8 subtrees below level 3 of the reduction tree, so part z walks a in [8 z / split, 8 (z + 1) / split) and ends with
// the sum of its subtree; the workgroup that arrives last on the tile's counter adds the parts in tree order -- the same additions.
constexpr int MMQ_BATCH = 4;
// Round 6: the output of product 0 as the tile-major quantised image of the NEXT product's input (q != nullptr; split == 1, N % 32 == 0):
// a wave's 32 rows x 32 tokens are exactly one (token tile, block) unit of that image, so the epilogue quantises them in place of storing
// f32 -- the 4 F bytes per token of the channel mixing's key vector no longer travel to HBM and back, and its quantiser launch is gone.
struct MmqQOut { int8_t * q = nullptr; float * d = nullptr; float * s = nullptr; float * o = nullptr; float dscale = 1.0f, off = 16.0f; int nb = 0; };
struct MmqArgs {
    PfW w[MMQ_BATCH]; PfX x[MMQ_BATCH]; float * y[MMQ_BATCH]; Epi epi[MMQ_BATCH];
    int64_t N, T, ldy;
    int nb, RT /* row tiles of 32 */, C /* token tiles of 64 */, split;
    const int * order;       // walk order of the blocks
    int a_start[9];          // steps of the walk before outer iteration a = 0 .. 8
    float * part;            // [split][tile][thread][16] partial sums (split > 1)
    int * counters;          // (not read: the parts are added by k_mmq_combine; reserved for a last-arriver variant in one kernel)
    MmqQOut yq;              // see MmqQOut
};


// prefill_fast.hip: the same product with the block sums accumulated in plain K order (RWKV_MI_SEQ_Q = exact (default) | fast); false = this shape
// or format stays on the exact kernel
bool launch_mmq_fast(int n, const DevTensor * const * Ws, const TileAct * xs, float * const * ys, const Epi * epis, int64_t T, int64_t ldy, hipStream_t st);
void mmq_fast_prepare_current_device();
int  seq_q_arm();          // RWKV_MI_SEQ_Q: 0 exact (default), 1 fast, 2 force
extern std::atomic<unsigned long long> g_mmq_fast_launches;

}  // namespace rwkvmi
