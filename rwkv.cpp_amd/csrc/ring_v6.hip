// ring_v6.hip -- the RWKV-6 single-token (decode) step over ALL layers of a stage as ONE persistent launch, with the weights
// streamed through an LDS ring by a dedicated loader wave (LDS-DMA, no registers).
//
// mega_v6.hip keeps the next phase's weights in REGISTERS of the waves that will use them. That couples the stream to the
// compute waves: a wave that has a prefetch in flight cannot poll (vector-memory results return in order per wave), every
// hand-over poll of the workgroup queues behind the bursts in the CU's memory pipe, and nothing streams while the owner
// is in a prologue or a hand-over. DESIGN.md section 7.2 measured the result: ~30 us of hand-over chain per layer and a
// ~19 us weight stream that is NOT hidden behind it (49 us per 7B layer).
//
// Here the stream is decoupled (MI355X_MICROARCH.md, rows ldsdma-fill / prefetch-credit / engine-vs-launches):
//   wave 0      loader: walks the workgroup's private, contiguous weight stream (ring_geom.h: every matrix row this workgroup
//               ever needs, packed once at context creation in the order of use) with 1-KiB global_load_lds_dwordx4 ... nt
//               instructions into a ring in LDS, as far ahead as the ring allows -- through prologues, hand-overs and the WKV
//               phase. It never waits for anything but ring space; while the workgroup gathers a hand-over it keeps fewer
//               fills in flight (row gather-pass).
//   wave 1      comm: the small serial jobs -- the data-dependent mixes of this workgroup's chunk (phase B), the WKV head (phase D,
//               workgroups 0..H-1), quantising the channel-mixing keys.
//   waves 2..7  consumers: row sums out of ring records (ds_read, lane-linear), results published as tagged units.
// No wave holds weight registers, so EVERY non-loader wave can poll: each hand-over is gathered by all seven waves at once
// (a seventh of the units each) instead of by one wave with 32 loads per lane in flight.
// There is no s_barrier after the start-up: waves meet through monotonic counters in LDS (LDS operations of a wave execute in
// order, so a counter bumped after a wave's stores covers them).
//
// Arithmetic, reduction orders and epilogues are those of fused_v6.hip / mega_v6.hip: bit-identical to
// the CPU oracle. Residency / abort rules as for mega_v6.hip: one workgroup per CU, all resident; every wait is bounded.
#include "persist.h"
#include "ring_geom.h"

namespace rwkvmi {

struct R6Cu { unsigned long long base; unsigned chunks, chunks_head; unsigned layer_bytes, pad; };   // per workgroup: stream offset, 1-KiB fills without / with the head, bytes per layer

struct R6P {
    const M6Layer * layers; int n_layers;
    const unsigned char * arena;
    const float * w2b;                                 // W2 of every layer in the chunk-blocked layout (k_block_w2, mega_v6.hip)
    float * x;                                         // plain residual stream: input of the first layer, output of the last
    float * x_out;                                     // where the last layer of the launch leaves it (= x; a pipeline stage: the next stage's x, through the peer mapping)
    const float * sin; float * sout; long long state_stride;
    void * xch; unsigned xch_bytes;                     // the exchange arena ...
    int tl, act5, rkvg, dl, yq, xatt, kq, xffn;         // ... buffers at these unit (16-byte) offsets
    int act_stride;                                    // units between the five mix images
    unsigned * ctl;                                    // [0] tag generation, [1] abort
    const unsigned char * stream; const R6Cu * cus;     // the per-workgroup weight streams
    int layer0, layers_total;                          // this launch covers layers [layer0, layer0 + n_layers) of the stage's layers_total (the streamed rwkv_eval cuts a token into groups)
    int F, DR, R, H;
    int head_wg0;                                      // first of the H workgroups whose comm wave runs a WKV head
    unsigned ring_bytes, mirror_bytes;                 // LDS ring (multiple of 4 KiB) and how much of its head is repeated behind its end
    int inflight, thin;                                // loader: DMA instructions in flight (normal / while the workgroup gathers)
    int look;                                          // consumers: records taken per look at a hand-over's sentinel (hint_take)
    int look_g;                                        // ... at the kq hand-over (value records; RWKV_MI_RING_LOOKG)
    int hthin;                                         // ... while a wave of the workgroup watches a hand-over's sentinel (= inflight: no third level)
    int nap;                                           // extra 64-cycle sleeps between two looks at a gather's sentinel unit
    int burst;                                         // loader: fills issued per round (between two looks at the consumers' positions)
    int dbg;                                           // timing experiment: 8 = the loader alone (every other wave leaves at once; results are WRONG)
    // the head projection folded in behind the last layer (F16 head.weight of a last stage; logits == nullptr: not this launch)
    float * logits; long long lnout_w, lnout_b; int n_vocab;
    // embedding + ln0 in front of the first layer (tok != nullptr: the launch starts from the token id; rwkv_graph.inc:655-658) and the argmax of the
    // logits behind the head (next_tok != nullptr; every launch publishes its workgroups' candidates at `am`, only launches with logits pick)
    const unsigned * tok; unsigned * next_tok; const void * emb; int emb_f16; long long ln0_w, ln0_b; int am;
    long long * trace; int trace_layer;
};

// monotonic words in LDS
enum { FL_LANDED = 0, FL_SWB = 1, FL_SWE = 2 /* wide sweeps begun / ended (the loader keeps fewer fills in flight while they differ) */, FL_DONE = 4 /* [8], 16-byte aligned */, FL_GX = 12, FL_GACT = 13, FL_GYQ = 14, FL_GKQ = 15,
       FL_RED1 = 16, FL_RED2 = 17, FL_PRO = 18, FL_KEYS = 19,
       FL_HX = 20, FL_HACT = 21, FL_HYQ = 22, FL_HKQ = 23, FL_HTL = 24 /* "some wave saw its sentinel turn" per gather kind */,
       FL_PROX = 25 /* consumer waves 4, 5 are done with their share of a prologue's elementwise part */,
       FL_HWB = 26, FL_HWE = 27 /* sentinel watches begun / ended (p.hthin: the loader's depth while a wave of the workgroup watches a hand-over) */,
       FL_XRQ = 28 /* waves done with the deferred quantisation of the ffn receptance input (quant_xr) */,
       FL_AM = 29 /* consumer waves that have left their argmax candidate in l.bc */, FL_WORDS = 32 };

struct R6Lds { size_t x, q1, q2, u, tl, bc, red, out, dl, misc, fl, ring, fixed; };
__host__ __device__ inline R6Lds r6_lds(int D, int F) {
    R6Lds o; size_t p = 0;
    auto take = [&](size_t n) { const size_t r = p; p += (n + 15) / 16 * 16; return r; };
    o.x = take((size_t) D * 4); o.q1 = take(qvec_bytes(D)); o.q2 = take(qvec_bytes(D));
    const size_t ua = 3 * ((qvec_bytes(D) + 15) / 16 * 16), ub = qvec_bytes(F);   // {act, actw, yq} (C..E) share their space with kq (G)
    o.u = take(ua > ub ? ua : ub);
    o.tl = take(5 * 64 * 4); o.bc = take(64 * 16); o.red = take(2 * 256 * 8); o.out = take(64 * 4); o.dl = take(qvec_bytes(256)); o.misc = take(64);
    o.fl = take(FL_WORDS * 4);
    p = (p + 1023) / 1024 * 1024;
    o.ring = p; o.fixed = p;
    return o;
}

// ---------------------------------------------------------------------------------------------------------------
// LDS words shared by the waves of a workgroup. Relaxed accesses: LDS operations of one wave execute in order, which is all the
// ordering the protocols below need; the compiler is kept from moving other accesses across with empty asm statements.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fl_ld(unsigned * f) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void fl_st(unsigned * f, unsigned v) { asm volatile("" ::: "memory"); if ((threadIdx.x & 63) == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void fl_add(unsigned * f, unsigned v) { asm volatile("" ::: "memory"); if ((threadIdx.x & 63) == 0) (void) __hip_atomic_fetch_add(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ bool lds_backoff(Poll & pl, unsigned spin) {
    if ((spin & 1023u) == 1023u) {
        if (__hip_atomic_load(pl.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) pl.dead = true;
        else if (spin > 60000000u) { __hip_atomic_store(pl.ctl + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pl.dead = true; }
    }
    __builtin_amdgcn_s_sleep(1);
    return pl.dead;
}
// waits until *f >= want (bounded: the abort word ends every wait of the grid)
__device__ __forceinline__ void fl_wait(Poll & pl, unsigned * f, unsigned want) {
    for (unsigned spin = 0;; spin++) {
        if (fl_ld(f) >= want || pl.dead) break;
        if (lds_backoff(pl, spin)) break;
    }
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// ring records -> row sums
// ---------------------------------------------------------------------------------------------------------------
// A record in registers. The buffers are read-modify-write operands of the inline-asm reads below ("+v"): a buffer is ONE virtual
// register from rec_reserve() to its last use, whichever of its (conditional) loads ran. Through plain loads the compiler sees
// phi(undef, record) chains over the hand-over wait loops, splits the live ranges around them and copies whole records (v_mov_b64
// chains, then spills); tied operands leave it nothing to split.
template <int FMT> struct RBlk { wv4i q[QF<FMT>::QS / 16]; unsigned qh, sc; };
template <int FMT, int R, int U> struct RawRec { RBlk<FMT> raw[U][R]; };
template <int FMT>
__device__ __forceinline__ void to_raw(RawBlk<FMT> & o, const RBlk<FMT> & r) {
    o.q[0] = make_int4(r.q[0].x, r.q[0].y, r.q[0].z, r.q[0].w);
    if constexpr (QF<FMT>::QS == 32) o.q[1] = make_int4(r.q[1].x, r.q[1].y, r.q[1].z, r.q[1].w);
    o.qh = r.qh; o.sc = r.sc;
}
template <int FMT>
__device__ __forceinline__ void from_raw(RBlk<FMT> & o, const RawBlk<FMT> & r) {
    o.q[0] = wv4i{r.q[0].x, r.q[0].y, r.q[0].z, r.q[0].w};
    if constexpr (QF<FMT>::QS == 32) o.q[1] = wv4i{r.q[1].x, r.q[1].y, r.q[1].z, r.q[1].w};
    o.qh = r.qh; o.sc = r.sc;
}
// start of a buffer's life (no instruction)
template <int FMT, int R, int U>
__device__ __forceinline__ void rec_reserve(RawRec<FMT, R, U> & w) {
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int r = 0; r < R; r++) {
            RBlk<FMT> & o = w.raw[u][r];
            asm volatile("" : "=v"(o.q[0]));
            if constexpr (QF<FMT>::QS == 32) asm volatile("" : "=v"(o.q[1]));
            asm volatile("" : "=v"(o.sc));
            if constexpr (QF<FMT>::QH) asm volatile("" : "=v"(o.qh)); else o.qh = 0u;
        }
}

// A record is read linearly from its ring offset: the first `mirror` bytes of the ring exist a second time right behind its end (the
// loader fills both), so a record that starts near the end continues there -- no wrap arithmetic, every read an immediate offset from
// one base register per access width. ring_lds = LDS address of the ring; the reads are in flight on return (rec_wait).
template <int FMT, int R, int U>
__device__ __forceinline__ void rec_load(RawRec<FMT, R, U> & w, unsigned ring_lds, unsigned off, int lane) {
    constexpr unsigned QS = QF<FMT>::QS, SCB = QF<FMT>::HM ? 4 : 2;
    constexpr unsigned SC0 = U * R * 64 * QS, QH0 = SC0 + U * R * 64 * SCB;
    const unsigned a16 = ring_lds + off + (unsigned) lane * 16u, asc = ring_lds + off + (unsigned) lane * SCB, aqh = ring_lds + off + (unsigned) lane * 4u;
#pragma unroll
    for (int u = 0; u < U; u++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            RBlk<FMT> & o = w.raw[u][r];
            const unsigned k = (unsigned) (u * R + r);
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(o.q[0]) : "v"(a16), "n"(k * (QS / 16) * 1024u));
            if constexpr (QS == 32) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(o.q[1]) : "v"(a16), "n"(k * 2048u + 1024u));
            if constexpr (QF<FMT>::HM) asm volatile("ds_read_b32 %0, %1 offset:%2" : "+v"(o.sc) : "v"(asc), "n"(SC0 + k * 256u));
            else asm volatile("ds_read_u16 %0, %1 offset:%2" : "+v"(o.sc) : "v"(asc), "n"(SC0 + k * 128u));
            if constexpr (QF<FMT>::QH) asm volatile("ds_read_b32 %0, %1 offset:%2" : "+v"(o.qh) : "v"(aqh), "n"(QH0 + k * 256u));
        }
    }
}
// the reads of rec_load have returned (the compiler does not count inline-asm LDS operations: waited for by hand; every register of the
// record passes through the statement, so no use can move above it)
// YOUNGER = LDS operations issued behind this record's reads that may stay in flight (the reads of the record taken right after it + the
// release word: LDS operations of a wave complete in order, so "at most YOUNGER outstanding" means this record's have returned; 0: everything)
template <int N> __device__ __forceinline__ void wait_lgkm() {
    if constexpr (N <= 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    else if constexpr (N == 13) asm volatile("s_waitcnt lgkmcnt(13)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
}
// LDS instructions of one record's reads (rec_load) + the release word
template <int FMT, int R, int U> constexpr int rec_lds_ops() { return U * R * ((QF<FMT>::QS / 16) + 1 + (QF<FMT>::QH ? 1 : 0)) + 1; }
template <int FMT, int R, int U, int YOUNGER = 0>
__device__ __forceinline__ void rec_wait(RawRec<FMT, R, U> & w) {
    wait_lgkm<(YOUNGER > 15 ? 0 : YOUNGER)>();      // (more than the counter holds: wait for everything)
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int r = 0; r < R; r++) {
            RBlk<FMT> & o = w.raw[u][r];
            asm volatile("" : "+v"(o.q[0]));
            if constexpr (QF<FMT>::QS == 32) asm volatile("" : "+v"(o.q[1]));
            asm volatile("" : "+v"(o.sc));
            if constexpr (QF<FMT>::QH) asm volatile("" : "+v"(o.qh));
        }
}

// Round 6: 4-bit codes against NIBBLE PLANES of the activations (R6_DOT8; Q4_0 / Q4_1). The row phases are bound by VALU issue,
// and 12 of a Q4 block's ~26 VALU instructions only unpack its nibbles into bytes for v_dot4. v_dot8_u32_u4 multiplies the packed nibbles as they
// lie: with b = a + 128 (one xor per activation dword, 1 <= b <= 255) = 16 bh + bl,
//     sum w a = 16 sum w bh + sum w bl - 128 sum w,
// three dot8 per code dword (the last against 0x11111111) and no unpacking: 12 + 3 instead of 12 + 8 + 1 instructions per block. The nibble
// planes are built once per phase and lane, in the order the weight nibbles lie in a code dword (byte t of dword j: low nibble = element 4 j + t,
// high nibble = element 16 + 4 j + t). Integers are exact in any order: the block sum -- and everything behind it -- is bit for bit the same.
// Built, bit-identical (parity tests + bench parity green), measured, OFF: 662.0 against 672.5 tokens/s on the 7B (three alternations on one box),
// 1420.8 against 1432.3 on the 1.6B -- a third fewer VALU instructions per block and 1.5 % SLOWER: v_dot8_u32_u4 does not issue at v_dot4's rate on
// this part, or the phases are less issue-bound than the per-SIMD record counts suggest; profiles/r06_dot8_ab.txt.
#ifndef R6_DOT8
#define R6_DOT8 0
#endif
template <int FMT> constexpr bool r6_nib() { return R6_DOT8 != 0 && (FMT == T_Q4_0 || FMT == T_Q4_1); }
// The activation blocks a lane needs are the same for every record of a phase (block 64 u + lane of the image): read once per phase.
// (nibble planes: alo[u] = the low nibbles of b, ahi[u] = the high nibbles, one dword per code dword)
template <int U> struct ActRegs { int4 alo[U], ahi[U]; float dx[U], sx[U]; int asum[U]; };
template <int FMT, int U>
__device__ __forceinline__ void act_load(ActRegs<U> & ar, const QVec & a, int nbk, int lane) {
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int bb = u * WAVE + lane;
        const int b = bb < nbk ? bb : nbk - 1;
        const int4 lo = *reinterpret_cast<const int4 *>(a.q + b * 16);
        const int4 hi = *reinterpret_cast<const int4 *>(a.q + nbk * 16 + b * 16);
        if constexpr (r6_nib<FMT>()) {
            const unsigned l4[4] = {(unsigned) lo.x ^ 0x80808080u, (unsigned) lo.y ^ 0x80808080u, (unsigned) lo.z ^ 0x80808080u, (unsigned) lo.w ^ 0x80808080u};
            const unsigned h4[4] = {(unsigned) hi.x ^ 0x80808080u, (unsigned) hi.y ^ 0x80808080u, (unsigned) hi.z ^ 0x80808080u, (unsigned) hi.w ^ 0x80808080u};
            unsigned bl[4], bh[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bl[j] = (l4[j] & 0x0F0F0F0Fu) | ((h4[j] << 4) & 0xF0F0F0F0u);
                bh[j] = ((l4[j] >> 4) & 0x0F0F0F0Fu) | (h4[j] & 0xF0F0F0F0u);
            }
            ar.alo[u] = make_int4((int) bl[0], (int) bl[1], (int) bl[2], (int) bl[3]);
            ar.ahi[u] = make_int4((int) bh[0], (int) bh[1], (int) bh[2], (int) bh[3]);
        } else { ar.alo[u] = lo; ar.ahi[u] = hi; }
        ar.dx[u] = a.d[b]; ar.sx[u] = a.s[b]; ar.asum[u] = a.isum[b];
    }
}

// The record's rows against the activation registers: lane l accumulates blocks l, l + 64, ... in increasing order (per-lane partials,
// no butterfly). Written breadth first -- every block's codes unpacked, then the eight dot4 steps round-robin over the U x R blocks with
// two integer accumulators each (integer sums are exact in any order), then the scales -- so that the U x R dependent chains interleave
// instead of running back to back.
template <int FMT, int R, int U>
__device__ __forceinline__ void rec_acc(const RawRec<FMT, R, U> & w, const ActRegs<U> & ar, int nbk, int lane, float * acc) {
    constexpr int GB = QF<FMT>::QS == 32 ? 2 : 4;     // blocks unpacked at a time (registers; Q8_0 blocks are twice the size)
    constexpr int GU = (GB / R) < U ? ((GB / R) > 0 ? (GB / R) : 1) : U;
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0f;
    if constexpr (r6_nib<FMT>()) {
        // nibble planes (see act_load): three v_dot8_u32_u4 chains per block, breadth first over the GU x R blocks of a group
#pragma unroll
        for (int u0 = 0; u0 < U; u0 += GU) {
            unsigned sl[GU][R], sh[GU][R], sw[GU][R];
#pragma unroll
            for (int g = 0; g < GU; g++)
#pragma unroll
                for (int r = 0; r < R; r++) { sl[g][r] = 0u; sh[g][r] = 0u; sw[g][r] = 0u; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int g = 0; g < GU; g++) {
                    if (u0 + g < U) {
                        const int u = u0 + g;
                        const unsigned bl = (unsigned) (k == 0 ? ar.alo[u].x : (k == 1 ? ar.alo[u].y : (k == 2 ? ar.alo[u].z : ar.alo[u].w)));
                        const unsigned bh = (unsigned) (k == 0 ? ar.ahi[u].x : (k == 1 ? ar.ahi[u].y : (k == 2 ? ar.ahi[u].z : ar.ahi[u].w)));
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const wv4i & q = w.raw[u][r].q[0];
                            const unsigned c = (unsigned) (k == 0 ? q.x : (k == 1 ? q.y : (k == 2 ? q.z : q.w)));
                            sl[g][r] = __builtin_amdgcn_udot8(c, bl, sl[g][r], false);
                            sh[g][r] = __builtin_amdgcn_udot8(c, bh, sh[g][r], false);
                            sw[g][r] = __builtin_amdgcn_udot8(c, 0x11111111u, sw[g][r], false);
                        }
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < GU; g++) {
                if (u0 + g < U) {
                    const int u = u0 + g;
                    const bool valid = u + 1 < U || u * WAVE + lane < nbk;   // (only the last step of a row can be short)
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        int sv = (int) ((sh[g][r] << 4) + sl[g][r]) - (int) (sw[g][r] << 7);      // sum w a = 16 sum w bh + sum w bl - 128 sum w
                        if constexpr (QF<FMT>::OFF != 0) sv -= QF<FMT>::OFF * ar.asum[u];
                        const float wd = h2f_bits((uint16_t) (w.raw[u][r].sc & 0xFFFFu));
                        const float dd = wd * ar.dx[u];
                        acc[r] = fmaf(dd, valid ? (float) sv : 0.0f, acc[r]);
                        if constexpr (QF<FMT>::HM) acc[r] = fmaf(h2f_bits((uint16_t) (w.raw[u][r].sc >> 16)), valid ? ar.sx[u] : 0.0f, acc[r]);
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int u0 = 0; u0 < U; u0 += GU) {
        WBlk<FMT> wb[GU][R];
        int s0[GU][R], s1[GU][R];
#pragma unroll
        for (int g = 0; g < GU; g++)
#pragma unroll
            for (int r = 0; r < R; r++) { if (u0 + g < U) { RawBlk<FMT> rb; to_raw<FMT>(rb, w.raw[u0 + g][r]); unpack_raw<FMT>(wb[g][r], rb); } s0[g][r] = 0; s1[g][r] = 0; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int g = 0; g < GU; g++) {
                if (u0 + g < U) {
                    const int u = u0 + g;
                    const int alo = k == 0 ? ar.alo[u].x : (k == 1 ? ar.alo[u].y : (k == 2 ? ar.alo[u].z : ar.alo[u].w));
                    const int ahi = k == 0 ? ar.ahi[u].x : (k == 1 ? ar.ahi[u].y : (k == 2 ? ar.ahi[u].z : ar.ahi[u].w));
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        s0[g][r] = __builtin_amdgcn_sdot4(wb[g][r].c[k], alo, s0[g][r], false);
                        s1[g][r] = __builtin_amdgcn_sdot4(wb[g][r].c[4 + k], ahi, s1[g][r], false);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < GU; g++) {
            if (u0 + g < U) {
                const int u = u0 + g;
                const bool valid = u + 1 < U || u * WAVE + lane < nbk;   // (only the last step of a row can be short)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    int sv = s0[g][r] + s1[g][r];
                    if constexpr (QF<FMT>::OFF != 0) sv -= QF<FMT>::OFF * ar.asum[u];
                    const float dd = wb[g][r].d * ar.dx[u];
                    acc[r] = fmaf(dd, valid ? (float) sv : 0.0f, acc[r]);
                    if constexpr (QF<FMT>::HM) acc[r] = fmaf(wb[g][r].m, valid ? ar.sx[u] : 0.0f, acc[r]);
                }
            }
        }
    }
}
// N xor-butterflies (the halving tree of wave_sum_f) written level by level, so that the N dependent chains interleave
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false); v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
#pragma unroll
    for (int i = 0; i < N; i++) { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false); v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + __int_as_float(lane_xor8_i(__float_as_int(v[i])));
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + __int_as_float(lane_xor4_i(__float_as_int(v[i])));
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + __int_as_float(lane_xor2_i(__float_as_int(v[i])));
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + __int_as_float(lane_xor1_i(__float_as_int(v[i])));
}

// ---------------------------------------------------------------------------------------------------------------
// the kernel. EPT = D / 512, NBD = decay rank / 32, UF = 64-block steps of an F-long row, KSL = gather slots per lane for the
// quantised F-vector. Exactly RG_NBLK workgroups of 512 threads.
// ---------------------------------------------------------------------------------------------------------------
#ifndef R6_ROWS_WAIT   /* 1 (debugging aid): every take inside a row phase waits for its LDS reads at once */
#define R6_ROWS_WAIT 0
#endif
#ifndef R6_REPOLL_MISSING
#define R6_REPOLL_MISSING 0
#endif
#ifndef R6_RT_STAMPS   /* 1: every phase stamp from the 100 MHz real-time counter (one clock for all workgroups: tools/trace_ring_cp.py) instead of the shader clock */
#define R6_RT_STAMPS 0
#endif
#if R6_RT_STAMPS
#define R6STAMP(K) do { if (p.trace && li == p.trace_layer && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + (K)] = (long long) __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define R6STAMP(K) do { if (p.trace && li == p.trace_layer && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + (K)] = (long long) __builtin_readcyclecounter(); } while (0)
#endif
#define R6RSTAMP(K) do { if (p.trace && li == p.trace_layer && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + (K)] = (long long) __builtin_amdgcn_s_memrealtime(); } while (0)

template <int FMT, int EPT, int NBD, int UF, int KSL>
struct R6 {
    static constexpr int S = 64, NBLK = RG_NBLK, NC = RG_NC, NG = RG_NC + 1;   // NG: waves that gather (consumers + comm)
    static constexpr int D = EPT * 512;
    static constexpr int nb = D / 32;
    static constexpr int UD = (nb + 63) / 64;                // steps of a D-long row (the last one may be short: zero blocks in the stream)
    static constexpr int RE = D / NBLK;                      // output / receptance / value rows per workgroup
    static constexpr int XT = (RE + NC - 1) / NC;            // rows per x unit (<= 3)
    static constexpr int XSL = (NBLK * NC + NG * 64 - 1) / (NG * 64);   // gather slots per lane for an x-like vector
    static constexpr int DSL = (3 * nb + NG * 64 - 1) / (NG * 64);      // ... for a quantised D-vector
    static_assert(UD >= 1 && XT <= 3 && D % 512 == 0 && nb % RG_HSTEPS == 0 && RE >= NC, "geometry");

    struct Lds {
        float *x, *tl, *bc, *out, *misc;
        unsigned char *q1, *q2, *act, *actw, *yq, *kq, *dl, *ring;
        double * red;
        unsigned * fl;
    };
    static __device__ __forceinline__ RingShape shape(const R6P & p) {
        RingShape s; s.D = D; s.F = p.F; s.R5 = 5 * p.R; s.DR = p.DR;
        s.qs = QF<FMT>::QS; s.scb = QF<FMT>::HM ? 4 : 2; s.qhb = QF<FMT>::QH ? 4 : 0;
        s.bal = BAL ? 1 : 0;
        return s;
    }
    // two-row sets of C and FK dealt by SIMD share (ring_geom.h, RG_BAL_*): the D = 4096 geometry. Built, bit-identical, measured, OFF:
    // 649.4 tokens/s against 662.4 for the even deal on one box (profiles/r06_balance_ab.txt). The deal moved nothing where it was aimed --
    // c4 / c5 finish the r/k/v/g rows at 8100 cycles with four records as with five -- because a SIMD issues one VALU instruction per four
    // cycles whether one wave or two feed it: the row phases are bound by the CU's 4 x 1 issue slots (32 sets in ~8000 cycles), a wave that
    // shares its SIMD with an older one simply gets the slots the older one leaves, and the seven-record waves became the new last ones.
#ifndef R6_BALANCE
#define R6_BALANCE 0
#endif
    static constexpr bool BAL = R6_BALANCE != 0 && EPT == 8;
    template <int PH> static constexpr bool balph() { return BAL && (PH == RG_C || PH == RG_FK); }

    // -----------------------------------------------------------------------------------------------------------
    // cooperative gathers: the NG gathering waves each poll a share of the units and stage it into LDS, then meet on a counter
    // -----------------------------------------------------------------------------------------------------------
    // x-like vector: unit (workgroup b, consumer c) at b * NC + c carries rows b * RE + c + NC * t, t < XT
    // plain != nullptr (the first layer of a launch): the vector lies in plain memory (written before the launch) and is read in the same
    // unit-shaped pieces -- one code path, so that the layer loop has no separate "first layer" branch (around which the register allocator
    // spilled the record buffers that are reserved across it)
    // keep = true (the first layer behind an in-launch embedding): l.x already holds the vector, nothing is stored
    static __device__ __forceinline__ void gather_x(Poll & pl, xrsrc xr, int buf, unsigned tag, int g, int lane, float * lx, const float * plain = nullptr, bool keep = false) {
        constexpr int N = NBLK * NC;
        const int i0 = g * 64 + lane;
        auto stage = [&](int k, const v4u & u) {
            const unsigned i = (unsigned) (i0 + k * NG * 64);
            const unsigned b = (i * 43691u) >> 18;         // i / 6 for i < 2^16
            const unsigned c = i - 6u * b;
            float * dst = lx + b * RE + c;
            dst[0] = __uint_as_float(u.x);
            if (XT > 1 && c + NC < (unsigned) RE) dst[NC] = __uint_as_float(u.y);
            if (XT > 2 && c + 2 * NC < (unsigned) RE) dst[2 * NC] = __uint_as_float(u.z);
        };
        if (plain) {
            v4u v[XSL];
#pragma unroll
            for (int k = 0; k < XSL; k++) {
                const unsigned i = (unsigned) (i0 + k * NG * 64);
                const unsigned ic = i < (unsigned) N ? i : 0u;
                const unsigned b = (ic * 43691u) >> 18, c = ic - 6u * b;
                const float * src = plain + b * RE + c;
                v[k].x = __float_as_uint(src[0]);
                v[k].y = __float_as_uint(src[(XT > 1 && c + NC < (unsigned) RE) ? NC : 0]);
                v[k].z = __float_as_uint(src[(XT > 2 && c + 2 * NC < (unsigned) RE) ? 2 * NC : 0]);
                v[k].w = 0u;
            }
#pragma unroll
            for (int k = 0; k < XSL; k++) if (i0 + k * NG * 64 < N && !keep) stage(k, v[k]);
            return;
        }
        // A sweep that comes back incomplete re-reads only what was missing (R6_REPOLL_MISSING): a unit is staged by the sweep that finds it,
        // and a lane that has its unit of a slot reads the buffer's first unit instead (one address for all of them: one request) -- the
        // retries of a hand-over's last microsecond cost a few requests, not the vector again. (Unconditional loads with a selected
        // address: loads under `if (missing)` are conditional redefinitions, around which the allocator copies and spills.)
        bool have[XSL];
#pragma unroll
        for (int k = 0; k < XSL; k++) have[k] = i0 + k * NG * 64 >= N;
        for (unsigned spin = 0;; spin++) {
            asm volatile("" ::: "memory");
            v4u v[XSL];
#pragma unroll
            for (int k = 0; k < XSL; k++) v[k] = tg_load(xr, (R6_REPOLL_MISSING && have[k]) ? buf : buf + i0 + k * NG * 64);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < XSL; k++) {
                const bool now = !have[k] && tg_ok(v[k], tag);
                if (now) stage(k, v[k]);
                have[k] = have[k] || now; ok = ok && have[k];
            }
            if (__all(ok) || pl.dead) break;
            if (poll_backoff(pl, spin)) break;
        }
    }
    // quantised vector of K elements (3 units per 32-element block, see tq_store_block) into its lohi image
    template <int SL>
    static __device__ __forceinline__ void gather_qvec(Poll & pl, xrsrc xr, int src, int K, unsigned tag, int g, int lane, unsigned char * l) {
        const int nbk = K / 32, n = 3 * nbk;
        const QVec q = qvec_at(l, K);
        unsigned * img = reinterpret_cast<unsigned *>(l);
        const int i0 = g * 64 + lane;
        auto stage = [&](int k, const v4u & u) {
            const unsigned i = (unsigned) (i0 + k * NG * 64);
            const unsigned b = (i * 43691u) >> 17;         // i / 3 for i < 2^16
            const unsigned kk = i - 3u * b;
            const unsigned j0 = 3u * kk, j1 = j0 + 1u;     // dword index of .x / .y within the block's eight code dwords
            img[(j0 < 4u ? 0u : 4u * nbk) + 4u * b + (j0 & 3u)] = u.x;
            img[(j1 < 4u ? 0u : 4u * nbk) + 4u * b + (j1 & 3u)] = u.y;
            if (kk < 2u) {
                const unsigned j2 = j0 + 2u;
                img[(j2 < 4u ? 0u : 4u * nbk) + 4u * b + (j2 & 3u)] = u.z;
            } else {
                q.d[b] = h2f_bits((uint16_t) (u.z & 0xFFFFu)); q.s[b] = h2f_bits((uint16_t) (u.z >> 16));
                q.isum[b] = (int) (short) (u.w >> 16);
            }
        };
        bool have[SL];
#pragma unroll
        for (int k = 0; k < SL; k++) have[k] = i0 + k * NG * 64 >= n;
        for (unsigned spin = 0;; spin++) {   // (a retry re-reads only what was missing: see gather_x)
            asm volatile("" ::: "memory");
            v4u v[SL];
#pragma unroll
            for (int k = 0; k < SL; k++) v[k] = tg_load(xr, (R6_REPOLL_MISSING && have[k]) ? src : src + i0 + k * NG * 64);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < SL; k++) {
                const bool now = !have[k] && tg_ok(v[k], tag);
                if (now) stage(k, v[k]);
                have[k] = have[k] || now; ok = ok && have[k];
            }
            if (__all(ok) || pl.dead) break;
            if (poll_backoff(pl, spin)) break;
        }
    }
    // Stage 1 of a gather. Sweeping all units while the producers are still microseconds away is what the first version did: 256
    // workgroups x 7 waves re-reading up to 24 KB each per round trip, through the fabric (tagged units are read past the L2s) -- as
    // many bytes per layer as the weight stream itself, on a handful of memory channels, and the stream crawled during every gather.
    // Here every wave first watches ONE unit (a different one per wave and workgroup: 64 bytes per attempt) until its tag turns or until
    // another wave of the workgroup has seen its own turn (LDS word); only then the wide sweeps start.
    // idle(): called once per watch round (the consumers use the wait to take their first record of the coming phase out of the ring
    // as soon as it has landed: rows_pre)
    // The look at a sentinel: the tag word of one unit, the same address on every lane. R6_SWATCH = 1 reads it through the SCALAR memory path
    // (s_load_dword glc: past the scalar cache, served by the L2 like an sc1 vector load) -- it does not queue in the CU's vector memory pipe
    // behind the LDS-DMA fills and the sweeps (tools/watch_bench.hip: a chain of idle hand-overs 1.16 us each against 1.62 us).
#ifndef R6_SWATCH
#define R6_SWATCH 0
#endif
    static __device__ __forceinline__ bool look_turned(const Poll & pl, xrsrc xr, int unit, unsigned tag) {
#if R6_SWATCH
        if (!pl.sw) { const v4u v = tg_load(xr, unit); return __builtin_amdgcn_readfirstlane((int) tg_ok(v, tag)) != 0; }   // (R6_SWATCH = 2: the comm wave only)
        const unsigned off = (unsigned) __builtin_amdgcn_readfirstlane(unit) * 16u + 12u;
        const unsigned long long a = (unsigned long long) pl.xch;
        const unsigned long long base = (unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) a) | ((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (a >> 32)) << 32);
        unsigned t;
        asm volatile("s_load_dword %0, %1, %2 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(base), "s"(off) : "memory");
        return (t & 0xFFFFu) == (tag & 0xFFFFu);
#else
        const v4u v = tg_load(xr, unit);
        return __builtin_amdgcn_readfirstlane((int) tg_ok(v, tag)) != 0;
#endif
    }
    template <typename Idle>
    static __device__ __forceinline__ void gather_hint(Poll & pl, xrsrc xr, int unit, unsigned tag, unsigned * go, unsigned gen, int nap, Idle && idle) {
        for (unsigned spin = 0;; spin++) {
            if (fl_ld(go) >= gen || pl.dead) break;
            asm volatile("" ::: "memory");
            if (look_turned(pl, xr, unit, tag)) { fl_st(go, gen); break; }
            idle();
            if (poll_backoff(pl, spin)) break;
            for (int i = 0; i < nap; i++) __builtin_amdgcn_s_sleep(1);
        }
    }
    static __device__ __forceinline__ void gather_hint(Poll & pl, xrsrc xr, int unit, unsigned tag, unsigned * go, unsigned gen, int nap) {
        gather_hint(pl, xr, unit, tag, go, gen, nap, [] {});
    }
    // every gathering wave, after staging its share: arrive, then wait for the others' shares (gen = gathers of this kind so far)
    static __device__ __forceinline__ void gather_meet(Poll & pl, unsigned * f, unsigned gen) {
        fl_add(f, 1u);
        fl_wait(pl, f, (unsigned) NG * gen);
    }
    // A wave's wide sweep of a hand-over (after its sentinel turned) until the workgroup's gather is complete: the loader is thinned
    // for exactly that span -- its fills sit in the same memory pipe as the polls (row gather-pass). NOT while a wave merely watches a
    // sentinel: the comm wave reaches most gathers a whole row phase early, and a loader thinned through the row phases streams at half rate.
    // ... and, as a third level (p.hthin, by default the normal depth), while a wave watches a sentinel: with the records of a phase taken
    // into registers during these waits the loader streams through them, and a poll queues behind whatever the loader has in flight
    static __device__ __forceinline__ void watch_begin(const Lds & l) { fl_add(l.fl + FL_HWB, 1u); }
    static __device__ __forceinline__ void watch_end(const Lds & l) { fl_add(l.fl + FL_HWE, 1u); }
    static __device__ __forceinline__ void sweep_begin(const Lds & l) { fl_add(l.fl + FL_SWB, 1u); }
    static __device__ __forceinline__ void sweep_end(const Lds & l) { fl_add(l.fl + FL_SWE, 1u); }

    // -----------------------------------------------------------------------------------------------------------
    // prologues (consumer waves 0..3 = 256 threads)
    // -----------------------------------------------------------------------------------------------------------
    // The statistics run on consumer waves 0..3 (the reduction tree is defined over 256 partials). The elementwise
    // part behind them -- LayerNorm affine, token-shift mixes, quantisation: most of a prologue -- is spread over all six consumer waves
    // where the row divides that way (D = 4096: 1024 groups of four elements = three per thread of waves 0..3 + two per thread of
    // waves 4, 5, which used to idle through the prologue); every 32-block stays inside eight consecutive lanes.
    // D / 4 groups of four elements = 256 SLO + 128 SHI: D = 4096 -> 3 + 2, 2560 -> 2 + 1, 2048 -> 2 + 0
    static constexpr int Q128 = D / 512;                // groups of four per 128 threads
    static constexpr int SLO = (Q128 + 2) / 3;          // groups of four elements per thread of consumer waves 0..3
    static constexpr int SHI = Q128 - 2 * SLO;          // ... of consumer waves 4, 5
    static_assert(SHI >= 0 && SHI <= SLO, "prologue slots");
    static constexpr int SMAX = SLO > SHI ? SLO : SHI;
    static __device__ __forceinline__ int pslots(int c) { return c < 4 ? SLO : SHI; }
    // element index of slot k of consumer wave c (a missing slot repeats slot 0: loads stay unconditional)
    static __device__ __forceinline__ int pelem(int c, int lane, int k) {
        const int kk = k < pslots(c) ? k : 0;
        if (SHI == 0 && c >= 4) return 4 * lane;        // (waves 4, 5 take no part: any valid address)
        return 4 * (c < 4 ? c * 64 + lane + 256 * kk : 256 * SLO + (c - 4) * 64 * SHI + 64 * kk + lane);
    }
    struct PA { float4 lw[SMAX], lb[SMAX], pv[SMAX], mx[SMAX]; };
    struct PF { float4 lw[SMAX], lb[SMAX], pv[SMAX], mk[SMAX], mr[SMAX]; };
    static __device__ __forceinline__ void issue_pa(PA & pa, const M6Arena & ar, const M6Layer & L, const float * sin_l, int c, int lane) {
        const float * ln1_w = ar.f(L.ln1_w), * ln1_b = ar.f(L.ln1_b), * maa_x = ar.f(L.maa_x);
#pragma unroll
        for (int u = 0; u < SMAX; u++) {
            const int i = pelem(c, lane, u);
            pa.lw[u] = *reinterpret_cast<const float4 *>(ln1_w + i); pa.lb[u] = *reinterpret_cast<const float4 *>(ln1_b + i);
            pa.pv[u] = *reinterpret_cast<const float4 *>(sin_l + D + i); pa.mx[u] = *reinterpret_cast<const float4 *>(maa_x + i);
        }
    }
    static __device__ __forceinline__ void issue_pf(PF & pf, const M6Arena & ar, const M6Layer & L, const float * sin_l, int c, int lane) {
        const float * ln2_w = ar.f(L.ln2_w), * ln2_b = ar.f(L.ln2_b), * fmaa_k = ar.f(L.fmaa_k), * fmaa_r = ar.f(L.fmaa_r);
#pragma unroll
        for (int u = 0; u < SMAX; u++) {
            const int i = pelem(c, lane, u);
            pf.lw[u] = *reinterpret_cast<const float4 *>(ln2_w + i); pf.lb[u] = *reinterpret_cast<const float4 *>(ln2_b + i);
            pf.pv[u] = *reinterpret_cast<const float4 *>(sin_l + i);
            pf.mk[u] = *reinterpret_cast<const float4 *>(fmaa_k + i); pf.mr[u] = *reinterpret_cast<const float4 *>(fmaa_r + i);
        }
    }
    // LayerNorm statistics of the row in l.x: thread pt < 256 owns the partial over elements pt, pt + 256, ...;
    // every wave folds the 256 partials itself (the additions of block_sum_d). Leaves x - mean in l.x. gen = prologues so far.
    static __device__ __forceinline__ float ln_stats(Poll & pl, const Lds & l, int pt, int lane, unsigned gen) {
        constexpr int NP = D / 256;
        float xv[NP];
#pragma unroll
        for (int j = 0; j < NP; j++) xv[j] = l.x[pt + 256 * j];
        double sacc = 0.0;
#pragma unroll
        for (int j = 0; j < NP; j++) sacc += (double) xv[j];
        l.red[pt] = sacc;
        fl_add(l.fl + FL_RED1, 1u);
        fl_wait(pl, l.fl + FL_RED1, 4u * gen);
        const double t1 = (l.red[lane] + l.red[lane + 128]) + (l.red[lane + 64] + l.red[lane + 192]);
        const float mean = (float) (wave_sum_d(t1) / (double) D);
        double s2 = 0.0;
#pragma unroll
        for (int j = 0; j < NP; j++) { const float v = xv[j] - mean; l.x[pt + 256 * j] = v; s2 += (double) (v * v); }
        l.red[256 + pt] = s2;
        fl_add(l.fl + FL_RED2, 1u);
        fl_wait(pl, l.fl + FL_RED2, 4u * gen);
        const double t2 = (l.red[256 + lane] + l.red[256 + lane + 128]) + (l.red[256 + lane + 64] + l.red[256 + lane + 192]);
        const float var = (float) (wave_sum_d(t2) / (double) D);
        return 1.0f / sqrtf(var + 1e-5f);
    }
    // the same scale on a wave that owns no partial (consumer waves 4, 5): behind the second reduction round x - mean is in l.x
    static __device__ __forceinline__ float ln_scale(Poll & pl, const Lds & l, int lane, unsigned gen) {
        fl_wait(pl, l.fl + FL_RED2, 4u * gen);
        const double t2 = (l.red[256 + lane] + l.red[256 + lane + 128]) + (l.red[256 + lane + 64] + l.red[256 + lane + 192]);
        const float var = (float) (wave_sum_d(t2) / (double) D);
        return 1.0f / sqrtf(var + 1e-5f);
    }
    // A: LN1 + token shift + maa_x mix + quantise -> l.q1
    // (gen = prologues so far; eg = 1 when an embedding LayerNorm ran in front of them: the reduction counters are one round further)
    static __device__ __forceinline__ void prologue_A(Poll & pl, const Lds & l, const PA & pa, float * sout_l, bool write_state, int c, int lane, unsigned gen, unsigned eg) {
        const float scale = c < 4 ? ln_stats(pl, l, c * 64 + lane, lane, gen + eg) : ln_scale(pl, l, lane, gen + eg);
        if (c == 0 && lane == 0) l.misc[0] = scale;
        const QVec lq = qvec_at(l.q1, D);
#pragma unroll
        for (int u = 0; u < SMAX; u++) {
            if (u >= pslots(c)) break;
            const int i = pelem(c, lane, u);
            const float4 xc = *reinterpret_cast<const float4 *>(l.x + i);
            const float xs[4] = {xc.x, xc.y, xc.z, xc.w};
            const float lw[4] = {pa.lw[u].x, pa.lw[u].y, pa.lw[u].z, pa.lw[u].w}, lb[4] = {pa.lb[u].x, pa.lb[u].y, pa.lb[u].z, pa.lb[u].w};
            const float pv[4] = {pa.pv[u].x, pa.pv[u].y, pa.pv[u].z, pa.pv[u].w}, mx[4] = {pa.mx[u].x, pa.mx[u].y, pa.mx[u].z, pa.mx[u].w};
            float xn[4], xxx[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float y = xs[j] * scale;
                const float yw = y * lw[j];
                xn[j] = yw + lb[j];
                const float sx = pv[j] - xn[j];
                const float sm = sx * mx[j];
                xxx[j] = sm + xn[j];
            }
            if (write_state) *reinterpret_cast<float4 *>(sout_l + D + i) = make_float4(xn[0], xn[1], xn[2], xn[3]);
            unsigned packed; float d16, s16; int isum;
            quant_vec4(xxx, packed, d16, s16, isum);
            qvec_store4(lq, nb, i, packed, d16, s16, isum);
        }
        fl_add(l.fl + (c < 4 ? FL_PRO : FL_PROX), 1u);
    }
    // F: LN2 + token shift + the two mixes + quantise -> l.q1 (key input), l.q2 (receptance input)
    static __device__ __forceinline__ void prologue_F(Poll & pl, const Lds & l, const PF & pf, float * sout_l, bool write_state, int c, int lane, unsigned gen, unsigned eg) {
        const float scale = c < 4 ? ln_stats(pl, l, c * 64 + lane, lane, gen + eg) : ln_scale(pl, l, lane, gen + eg);
        const QVec qk = qvec_at(l.q1, D), qr = qvec_at(l.q2, D);
#pragma unroll
        for (int u = 0; u < SMAX; u++) {
            if (u >= pslots(c)) break;
            const int i = pelem(c, lane, u);
            const float4 xc = *reinterpret_cast<const float4 *>(l.x + i);
            const float xs[4] = {xc.x, xc.y, xc.z, xc.w};
            const float lw[4] = {pf.lw[u].x, pf.lw[u].y, pf.lw[u].z, pf.lw[u].w}, lb[4] = {pf.lb[u].x, pf.lb[u].y, pf.lb[u].z, pf.lb[u].w};
            const float pv[4] = {pf.pv[u].x, pf.pv[u].y, pf.pv[u].z, pf.pv[u].w};
            const float mk[4] = {pf.mk[u].x, pf.mk[u].y, pf.mk[u].z, pf.mk[u].w}, mr[4] = {pf.mr[u].x, pf.mr[u].y, pf.mr[u].z, pf.mr[u].w};
            float xn[4], xk[4], xr[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float y = xs[j] * scale;
                const float yw = y * lw[j];
                xn[j] = yw + lb[j];
                const float sx = pv[j] - xn[j];
                const float sk = sx * mk[j];
                xk[j] = sk + xn[j];
                const float sr = sx * mr[j];
                xr[j] = sr + xn[j];
            }
            if (write_state) *reinterpret_cast<float4 *>(sout_l + i) = make_float4(xn[0], xn[1], xn[2], xn[3]);
            unsigned packed; float d16, s16; int isum;
            quant_vec4(xk, packed, d16, s16, isum);
            qvec_store4(qk, nb, i, packed, d16, s16, isum);
            if constexpr (DEFER_XR) *reinterpret_cast<float4 *>(l.x + i) = make_float4(xr[0], xr[1], xr[2], xr[3]);   // (this thread is the only reader of these four elements from here on)
            else { quant_vec4(xr, packed, d16, s16, isum); qvec_store4(qr, nb, i, packed, d16, s16, isum); }
        }
        fl_add(l.fl + (c < 4 ? FL_PRO : FL_PROX), 1u);
    }
    // The key rows need the key input only; the kq hand-over behind them is on the layer's critical path, the receptance rows are not
    // (their results are used in the value rows' epilogue, a hand-over later). So the prologue leaves the receptance input as floats in
    // l.x (in place of x - mean) and its quantisation -- a third of the prologue's arithmetic -- runs here, behind the wave's key rows,
    // under the kq hand-over. Same statements on the same values: bit-identical.
#ifndef R6_DEFER_XR
#define R6_DEFER_XR 1
#endif
    static constexpr bool DEFER_XR = R6_DEFER_XR != 0;
    static __device__ __forceinline__ void quant_xr(const Lds & l, int c, int lane) {
        const QVec qr = qvec_at(l.q2, D);
#pragma unroll
        for (int u = 0; u < SMAX; u++) {
            if (u >= pslots(c)) break;
            const int i = pelem(c, lane, u);
            const float4 xc = *reinterpret_cast<const float4 *>(l.x + i);
            const float xr[4] = {xc.x, xc.y, xc.z, xc.w};
            unsigned packed; float d16, s16; int isum;
            quant_vec4(xr, packed, d16, s16, isum);
            qvec_store4(qr, nb, i, packed, d16, s16, isum);
        }
        fl_add(l.fl + FL_XRQ, 1u);
    }
    // everything a prologue leaves in LDS is there (gen = prologues so far)
    static __device__ __forceinline__ void prologue_wait(Poll & pl, const Lds & l, unsigned gen) {
        fl_wait(pl, l.fl + FL_PRO, 4u * gen);
        if constexpr (SHI > 0) fl_wait(pl, l.fl + FL_PROX, 2u * gen);
    }

    // embedding row of the token + ln0 -> l.x (rwkv_graph.inc:655-658; the statements of k_embed_ln0 / block_layernorm in kernels.hip: the same
    // 256 partials, the same tree). Consumer waves 0..3, thread pt owns elements pt, pt + 256, ... through all of it: no meeting but the two
    // reduction rounds. The other gathering waves find l.x complete behind the first layer's gather_meet.
    static __device__ __forceinline__ void embed_ln0(const R6P & p, Poll & pl, const Lds & l, const M6Arena & ar, int pt, int lane) {
        constexpr int NP = D / 256;
        const unsigned tk = p.tok[0];
        const long long row = tk < (unsigned) p.n_vocab ? (long long) tk : 0ll;   // (host-side token ids are range-checked by the API; this guards device-side ones)
        const float * w0 = ar.f(p.ln0_w), * b0 = ar.f(p.ln0_b);
        float wv[NP], bv[NP];
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const int e = pt + 256 * j;
            l.x[e] = p.emb_f16 ? h2f_bits(reinterpret_cast<const uint16_t *>(p.emb)[row * D + e]) : reinterpret_cast<const float *>(p.emb)[row * D + e];
            wv[j] = w0[e]; bv[j] = b0[e];
        }
        const float scale = ln_stats(pl, l, pt, lane, 1u);
#pragma unroll
        for (int j = 0; j < NP; j++) { const int e = pt + 256 * j; const float y = l.x[e] * scale; const float yw = y * wv[j]; l.x[e] = yw + bv[j]; }
    }

    // argmax candidates: greater value, then smaller index (k_argmax's rule in kernels.hip; NaN never wins)
    static __device__ __forceinline__ void am_merge(float & best, int & bi, float ov, int oi) { if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; } }
    static __device__ __forceinline__ void am_wave(float & best, int & bi) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(best, o, WAVE); const int oi = __shfl_xor(bi, o, WAVE); am_merge(best, bi, ov, oi); }
    }

    // -----------------------------------------------------------------------------------------------------------
    // loader wave
    // -----------------------------------------------------------------------------------------------------------
    // Four consecutive 1-KiB fills as ONE statement: M0 (the LDS destination) is set once, the instruction offset advances the global
    // AND the LDS address. (Inline asm: M0 is not preserved around a statement, and through the builtin the compiler would wait for every
    // DMA in flight at the next LDS access; completion is counted by hand.) A wave issues at most one instruction every four cycles, so
    // the loader's instruction count per fill IS its ceiling: 16 instructions per fill (scalar address arithmetic, M0 save / restore per
    // fill) gave 23 GB/s alone in this kernel; this is 2 per fill.
#ifndef R6_DMA_OFFSET_BOTH
#define R6_DMA_OFFSET_BOTH 1
#endif
    template <bool NT>
    static __device__ __forceinline__ void dma_quad(unsigned long long sbase, unsigned voff, unsigned m0dst) {
        unsigned keep;
#if R6_DMA_OFFSET_BOTH
#define R6_QUAD(POL) "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" \
                     "global_load_lds_dwordx4 %1, %2" POL "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" POL "\n\t" \
                     "global_load_lds_dwordx4 %1, %2 offset:2048" POL "\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" POL "\n\ts_mov_b32 m0, %0"
#else
#define R6_QUAD(POL) "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" \
                     "global_load_lds_dwordx4 %1, %2" POL "\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" POL "\n\t" \
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048" POL "\n\t" \
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" POL "\n\ts_mov_b32 m0, %0"
#endif
        if constexpr (NT) asm volatile(R6_QUAD(" nt") : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(m0dst) : "memory");
        else asm volatile(R6_QUAD("") : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(m0dst) : "memory");
#undef R6_QUAD
    }
    static __device__ __forceinline__ void wait_vm(int w) {
        switch (w >> 2) {   // (immediate operand; multiples of four)
            case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1:  asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 2:  asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 3:  asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 4:  asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 5:  asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 6:  asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            case 7:  asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
            case 8:  asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
            case 9:  asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
        }
    }
    // The loader never waits for more than it must: `level` is a known upper bound of the DMA instructions still in flight (fills land in
    // order, so everything before issued - level is in the ring). With room in the ring it issues up to `burst` groups of four fills per
    // round and only waits when more than `w` would be in flight; when the ring is full it retires the fills in flight a few at a time --
    // publishing them to the consumers and looking at their positions in between -- instead of draining the queue.
    // The first `mirror` bytes of the ring are filled twice: once in place, once behind the ring's end (the copy first, so that "the
    // fill has landed" covers it; its source lines stay in L2 for the second read: default policy, everything else is non-temporal).
    static __device__ __forceinline__ void loader_main(const R6P & p, const Lds & l, int lane) {
        const int wave = 0;
        const R6Cu cu = p.cus[blockIdx.x];
        // fills of this launch (multiple of four): the whole stage's as precomputed, or a layer range's (+ the head's rows behind the last layer)
        unsigned total_ = p.logits ? cu.chunks_head : cu.chunks;
        if (p.n_layers != p.layers_total) {
            const unsigned long long bytes = (unsigned long long) p.n_layers * cu.layer_bytes + (p.logits ? (unsigned long long) rg_head(p.n_vocab, D).bytes : 0ull);
            total_ = (unsigned) ((bytes + 4 * RG_CHUNK - 1) / (4 * RG_CHUNK)) * 4u;
        }
        const unsigned total = __builtin_amdgcn_readfirstlane(total_);
        const unsigned long long src0 = (unsigned long long) p.stream + cu.base + (unsigned long long) p.layer0 * cu.layer_bytes;
        const unsigned RB = __builtin_amdgcn_readfirstlane(p.ring_bytes), MIR = __builtin_amdgcn_readfirstlane(p.mirror_bytes);
        const unsigned ring_m0 = __builtin_amdgcn_readfirstlane((unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) l.ring);
        const unsigned voff = (unsigned) lane * 16u;
        const int w_norm = __builtin_amdgcn_readfirstlane(p.inflight) & ~3, w_thin = __builtin_amdgcn_readfirstlane(p.thin) & ~3, w_hint = __builtin_amdgcn_readfirstlane(p.hthin) & ~3;
        Poll pl{p.ctl, false};
        unsigned issued = 0, roff = 0, landed = 0;
        unsigned min_done = 0, thin = 0;
        int level = 0;
        const unsigned burst = (unsigned) __builtin_amdgcn_readfirstlane(p.burst);
        const unsigned s_lo = __builtin_amdgcn_readfirstlane((unsigned) src0), s_hi = __builtin_amdgcn_readfirstlane((unsigned) (src0 >> 32));
        const int li = p.trace_layer;   // (stamps: once per launch)
        R6STAMP(0);
        unsigned stalls = 0, rounds = 0;
        unsigned long long sb = ((unsigned long long) s_hi << 32) | s_lo;
        unsigned * const fland = l.fl + FL_LANDED;
        // (tracing) workgroups 0 and 131 sample their rounds behind the phase stamps: [2][512][4] after the n_blocks * 8 * 32 stamp slots
        long long * const lsamp = (p.trace && (blockIdx.x == 0 || blockIdx.x == 131)) ? p.trace + (size_t) gridDim.x * 8 * 32 + (blockIdx.x == 0 ? 0 : 2048) : nullptr;
        const unsigned samp_lo = (unsigned) p.trace_layer * cu.layer_bytes, samp_hi = samp_lo + cu.layer_bytes;
        unsigned nsamp = 0;
        for (unsigned spin = 0; issued < total;) {
            const unsigned lim0 = min_done >= total * 1024u ? total : ((min_done + RB) >> 12) << 2;
            const unsigned lim = lim0 < total ? lim0 : total;
            // the consumers' positions and the sweep counters for the NEXT round: the reads travel while this round's fills are issued
            asm volatile("" ::: "memory");
            const v4u da = *reinterpret_cast<const v4u *>(l.fl + FL_DONE), db = *reinterpret_cast<const v4u *>(l.fl + FL_DONE + 4);
            const v4u dh = *reinterpret_cast<const v4u *>(l.fl);   // {landed, sweeps begun, sweeps ended, -}
            const uint2 dw = *reinterpret_cast<const uint2 *>(l.fl + FL_HWB);   // {watches begun, ended}
            const int w = thin == 1u ? w_thin : (thin == 2u ? w_hint : w_norm);
            rounds++;
            if (lsamp && nsamp < 512u && issued * 1024u + 4096u > samp_lo && landed * 1024u < samp_hi) {   // (tracing: this workgroup's rounds around the traced layer)
                if (lane == 0) {
                    long long * o = lsamp + (size_t) nsamp * 4;
                    o[0] = (long long) __builtin_amdgcn_s_memrealtime(); o[1] = (long long) issued * 1024 - samp_lo; o[2] = (long long) landed * 1024 - samp_lo;
                    o[3] = (long long) min_done - samp_lo;
                }
                nsamp++;
            }
            if (issued < lim) {
                spin = 0;
                unsigned n = (lim - issued) >> 2;
                n = n < burst ? n : burst;
                for (unsigned k = 0; k < n; k++) {
                    if (roff < MIR) { dma_quad<false>(sb, voff, ring_m0 + RB + roff); dma_quad<true>(sb, voff, ring_m0 + roff); level += 8; }
                    else { dma_quad<true>(sb, voff, ring_m0 + roff); level += 4; }
                    sb += 4096ull;
                    roff += 4096u; roff = roff >= RB ? 0u : roff;
                }
                issued += 4u * n;
                if (level > w) { wait_vm(w); level = w; }
            } else {
                stalls++;
                if (level > 0) { const int x = (level - 1) & ~3; wait_vm(x); level = x; }
                else if (lds_backoff(pl, spin++)) break;
            }
            {
                // (level counts instructions, a mirrored fill is two of them: issued - level never over-states what has landed)
                const unsigned ld = issued > (unsigned) level ? issued - (unsigned) level : 0u;
                if (ld > landed) { landed = ld; asm volatile("" ::: "memory"); __hip_atomic_store(fland, landed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            }
            {
                unsigned m = da.x < da.y ? da.x : da.y; m = m < da.z ? m : da.z; m = m < da.w ? m : da.w;
                m = m < db.x ? m : db.x; m = m < db.y ? m : db.y; m = m < db.z ? m : db.z; m = m < db.w ? m : db.w;
                min_done = __builtin_amdgcn_readfirstlane(m);
                thin = __builtin_amdgcn_readfirstlane(dh.y != dh.z ? 1u : (dw.x != dw.y ? 2u : 0u));
            }
        }
        wait_vm(0);
        fl_st(l.fl + FL_LANDED, total);
        R6STAMP(1);
        if (p.trace && lane == 0) {
            p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 2] = (long long) stalls;
            p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 3] = (long long) rounds;
        }
    }

    // -----------------------------------------------------------------------------------------------------------
    // consumer waves
    // -----------------------------------------------------------------------------------------------------------
    struct Cons {
        RingCu cu;
        unsigned lbase;        // stream position of the current layer's block
        unsigned next_block;   // stream position of this wave's first record behind the current layer (next layer, head, or none)
        unsigned RB;
        unsigned ring_lds;     // LDS address of the ring
        unsigned landed;       // chunks known to have landed
        int c, lane;
        int dbg;               // RWKV_MI_RING_DBG (bit 5: timing experiment, the records' arithmetic replaced by an xor of what was read)
        unsigned look;         // takes per look at a hand-over's sentinel (hint_take)
        long long waited;      // (tracing) cycles spent waiting for the loader
    };
    template <int T, int TE> struct Unroll {
        template <typename F> static __device__ __forceinline__ void run(F && f) { f(std::integral_constant<int, T>{}); Unroll<T + 1, TE>::run(f); }
    };
    template <int TE> struct Unroll<TE, TE> { template <typename F> static __device__ __forceinline__ void run(F &&) {} };

    // Records leave the ring for REGISTERS as soon as they have landed -- while the wave watches the hand-over its phase waits for.
    // Round 3 measured why: at the end of every hand-over the ring is full and nothing is in flight; the consumers drain
    // it in a microsecond or two and then wait a memory latency (~2 us with all 256 loaders restarting at once) for every further byte:
    // about half of each row phase was "waiting for the loader", and the hand-over behind a phase waits for the SLOWEST workgroup. One
    // record per wave in registers (round 3) gave the loader six records of room per hand-over; the register file (512 KB per CU) is
    // three times the LDS, and a consumer wave holds next to nothing while it waits. So a wave now takes up to NP of its records of
    // the coming phase -- all of them where the registers allow -- and the loader streams the phase after that into the room they
    // leave. Record t of the wave lives in buffer t % NP; rows() takes what is still missing (blocking), one record ahead of the
    // arithmetic, so NP = 2 is the old double buffer and NP = 1 the single one (Q8_0's long value rows).
    template <int PH, int R, int U, int NP> struct Pre {
        static constexpr int PH_ = PH, R_ = R, U_ = U, NP_ = NP;
        static constexpr unsigned RECB = (unsigned) (U * R * 64) * (QF<FMT>::QS + (QF<FMT>::HM ? 4 : 2) + (QF<FMT>::QH ? 4 : 0));
        RawRec<FMT, R, U> w[NP];
        unsigned have;         // records of this wave taken out of the ring so far in this phase
        unsigned cnt;          // records this wave owns in this phase
        unsigned pos, ro;      // stream position / ring offset of the next record to take
        unsigned after;        // where this wave's stream continues behind the phase
        unsigned base;         // stream position of the phase's record 0
    };
    // record number of this wave's T-th record of phase PH: six apart, or by the balanced deal (scalar selects on the wave's pair, no table in memory)
    template <int PH, int T>
    static __device__ __forceinline__ unsigned ownj(const Cons & cs) {
        if constexpr (balph<PH>()) {
            constexpr int TT = T < RG_BAL_TMAX ? T : RG_BAL_TMAX - 1;
            constexpr unsigned a = (unsigned) (PH == RG_C ? RG_BAL_C_J[0][TT] : RG_BAL_K_J[0][TT]), b = (unsigned) (PH == RG_C ? RG_BAL_C_J[1][TT] : RG_BAL_K_J[1][TT]),
                               d = (unsigned) (PH == RG_C ? RG_BAL_C_J[2][TT] : RG_BAL_K_J[2][TT]);
            const unsigned pr = (unsigned) cs.c >> 1;
            return (pr == 0u ? a : (pr == 1u ? b : d)) + ((unsigned) cs.c & 1u);
        } else return rg_first_j(cs.cu, PH, cs.c) + (unsigned) (NC * T);
    }
    template <int R, int U> static constexpr unsigned recb() { return (unsigned) (U * R * 64) * (QF<FMT>::QS + (QF<FMT>::HM ? 4 : 2) + (QF<FMT>::QH ? 4 : 0)); }
    template <int PH, int R, int U, int NP>
    static __device__ __forceinline__ void pre_begin(const Cons & cs, Pre<PH, R, U, NP> & pre) {
        constexpr unsigned RECB = recb<R, U>();
        const unsigned j0 = rg_first_j(cs.cu, PH, cs.c);
        pre.have = 0;
        pre.cnt = __builtin_amdgcn_readfirstlane(rg_own_count(cs.cu, PH, cs.c));
        pre.base = __builtin_amdgcn_readfirstlane(cs.lbase + cs.cu.off[PH]);
        pre.pos = __builtin_amdgcn_readfirstlane(cs.lbase + cs.cu.off[PH] + j0 * RECB);
        pre.ro = __builtin_amdgcn_readfirstlane(pre.pos - (pre.pos / cs.RB) * cs.RB);   // (once per phase)
        const unsigned nxt = rg_next_own_in_layer(cs.cu, cs.c, PH + 1);
        pre.after = __builtin_amdgcn_readfirstlane(nxt != RG_NONE ? cs.lbase + nxt : cs.next_block);
#pragma unroll
        for (int i = 0; i < NP; i++) rec_reserve<FMT, R, U>(pre.w[i]);
    }
    // takes record number pre.have (== T, statically) into buffer T % NP and releases its ring space; block = false: only if it has landed
    // WAIT = false: the reads are still in flight on return -- the caller calls rec_wait on the buffer before its first use, with only
    // straight arithmetic on OTHER buffers in between (no hardware interlock covers a register with an LDS read pending: a copy the
    // compiler made in between would copy the old contents)
    template <int T, int PH, int R, int U, int NP, bool WAIT = true>
    static __device__ __forceinline__ bool rec_take(Cons & cs, Poll & pl, const Lds & l, Pre<PH, R, U, NP> & pre, bool block) {
        constexpr unsigned RECB = recb<R, U>(), STRIDE = NC * RECB;
        const unsigned need = (pre.pos + RECB + 1023u) >> 10;
        if (cs.landed < need) {
            cs.landed = fl_ld(l.fl + FL_LANDED);
            if (cs.landed < need) {
                if (!block) return false;
                const long long t0 = (long long) __builtin_readcyclecounter();
                for (unsigned spin = 0;; spin++) {
                    cs.landed = fl_ld(l.fl + FL_LANDED);
                    if (cs.landed >= need || pl.dead) break;
                    if (lds_backoff(pl, spin)) break;
                }
                cs.waited += (long long) __builtin_readcyclecounter() - t0;
            }
        }
        asm volatile("" ::: "memory");
        rec_load<FMT, R, U>(pre.w[T % NP], cs.ring_lds, pre.ro, opq(cs.lane));
        // the reads above are in the LDS queue: the ring may be refilled up to this wave's next record (every lane writes the same word)
        asm volatile("" ::: "memory");
        const bool last = pre.have + 1u >= pre.cnt;
        // this wave's next record: NC records on, or where the balanced deal puts it (behind the last one: not used)
        unsigned nxt;
        if constexpr (balph<PH>()) nxt = pre.base + ownj<PH, T + 1>(cs) * RECB; else nxt = pre.pos + STRIDE;
        __hip_atomic_store(l.fl + FL_DONE + 2 + cs.c, last ? pre.after : nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (WAIT) rec_wait<FMT, R, U>(pre.w[T % NP]);
        pre.have += 1u;
        // (a wave's records are NC records apart: more than one lap of the ring for the long Q8_0 value rows)
        pre.ro += nxt - pre.pos; pre.pos = nxt;
        pre.ro = pre.ro >= cs.RB ? pre.ro - cs.RB : pre.ro;
        if constexpr (!balph<PH>() && STRIDE > 32768u) { pre.ro = pre.ro >= cs.RB ? pre.ro - cs.RB : pre.ro; pre.ro = pre.ro >= cs.RB ? pre.ro - cs.RB : pre.ro; }
        return true;
    }
    // which phases take records ahead (1: r/k/v/g, 2: output, 4: ffn key, 8: ffn value), and how many per wave. D = 2048: none -- a layer
    // block (130 KB per workgroup) nearly fits the ring as it is, and the extra ring checks in the watch loops cost 2.5 %.
    // + 16: W1 + r/k/v/g records during the x hand-over at the top of the layer, + 32: output records during the act hand-over, + 64: ffn
    // key records during the yq hand-over (the phase behind the one the hand-over feeds), + 128: ffn receptance records behind those.
    // (+ 16 only compiles without scratch since the layer loop has no separate first-layer branch: see gather_x)
#ifndef R6_PRE_MASK
#define R6_PRE_MASK 255
#endif
#ifndef R6_PRE_MASK_SMALL   /* the same for D = 2048: rounds 3 / 4 measured -2.5 % and left it off; on round 6's kernel +0.6 % (1427.6 against 1419.5 tokens/s on the 1.6B, two alternations) */
#define R6_PRE_MASK_SMALL 255
#endif
    static constexpr int PRE_MASK = EPT > 4 ? R6_PRE_MASK : R6_PRE_MASK_SMALL;
    // Stage 1 of a gather (gather_hint) with the wait put to use: until the hand-over's sentinel turns, the wave takes its records of
    // the coming phase as they land -- record 0, then 1, ... up to NP -- and never waits for a record once the hand-over is there.
    // Written as a straight sequence of NP steps (each: spin until "record t has landed" or "the hand-over turned"), not as one loop that
    // takes whatever has landed: with every buffer redefined conditionally inside a loop the register allocator splits the buffers'
    // live ranges around it and copies them (v_mov_b64 chains, then spills: tools/vgpr_live.py showed 253 live registers at a wait
    // that holds 90).
    // One step per record: a look at the sentinel (a memory round trip), then -- if the hand-over has not turned -- the record, if it
    // has landed. Six takes back to back kept all seven waves of a workgroup away from their sentinels for a microsecond, and the hand-
    // over behind them waited that long (measured: x hand-over + 0.95 us, kq + 1.5 us); a look before EVERY take (1 - 2 us each under
    // the stream) left the ffn key records in the ring through the yq wait -- three output records and one or two key records was all a
    // wave got to -- and the loader idle for 4 us behind them. `look` takes per look (p.look, RWKV_MI_RING_LOOK); the workgroup's
    // "turned" word in LDS is read before every take. Measured on the 7B Q4_0, same box, tokens/s: round-3 kernel 638 - 654, look 1: 639 - 644,
    // 2: 630, 3: 626, 6: 608 -- every record taken sooner costs the hand-overs more than the row phase behind them gains (the stream
    // and the polls share the memory system: DESIGN.md 7.2c). Default 1.
    // Two phases per wait: `a` is the phase this hand-over feeds, `b` the one behind it in the stream (taken only once every record of
    // `a` is in registers: a wave releases ring space in stream order). With one phase per wait the loader found the ring full again
    // 2 - 3 us into every wait -- the records of the NEXT phase sat in it until the next wait; now they leave during this one, and
    // what streams in behind them is two phases ahead.
    template <int PA_, int RA, int UA, int NA, int PB_, int RB_, int UB_, int NB_, int PC_, int RC, int UC, int NC_>
    static __device__ __forceinline__ void hint_take(Cons & cs, Poll & pl, const Lds & l, xrsrc xr, int unit, unsigned tag, unsigned * go, unsigned gen, int nap,
                                                     bool on_a, Pre<PA_, RA, UA, NA> & a, bool on_b, Pre<PB_, RB_, UB_, NB_> & b, bool on_c, Pre<PC_, RC, UC, NC_> & c3, bool no_handover = false) {
        bool turned = no_handover;            // (the first layer of a launch: nothing to wait for, nothing taken ahead)
        const unsigned look = cs.look;
        auto step = [&](auto & pre, auto tc, bool ok) {
            constexpr int t = decltype(tc)::value;
            using P = typename std::remove_reference<decltype(pre)>::type;
            if (!turned && ok && pre.have == (unsigned) t && (unsigned) t < pre.cnt) {
                const unsigned need = (pre.pos + P::RECB + 1023u) >> 10;
                bool here = false;
                for (unsigned spin = 0;; spin++) {
                    if (fl_ld(go) >= gen || pl.dead) { turned = true; break; }
                    if (cs.landed < need) cs.landed = fl_ld(l.fl + FL_LANDED);
                    const bool landed = cs.landed >= need;
                    // a landed record is taken at once -- except that every `look`-th take in a row is preceded by a look at the sentinel
                    if (landed && (spin > 0u || (unsigned) t % look != 0u)) { here = true; break; }
                    asm volatile("" ::: "memory");
                    if (look_turned(pl, xr, unit, tag)) { fl_st(go, gen); turned = true; break; }
                    if (landed) { here = true; break; }
                    if (poll_backoff(pl, spin)) { turned = true; break; }
                    for (int i = 0; i < nap; i++) __builtin_amdgcn_s_sleep(1);
                }
                if (here) (void) rec_take<t, P::PH_, P::R_, P::U_, P::NP_>(cs, pl, l, pre, false);
            }
        };
        if (!(cs.dbg & 64)) {
            Unroll<0, NA>::run([&](auto tc) { step(a, tc, on_a); });
            Unroll<0, NB_>::run([&](auto tc) { step(b, tc, on_b && a.have >= a.cnt); });
            Unroll<0, NC_>::run([&](auto tc) { step(c3, tc, on_c && a.have >= a.cnt && b.have >= b.cnt); });
        }
        if (!turned) gather_hint(pl, xr, unit, tag, go, gen, nap);
    }
    template <int PA_, int RA, int UA, int NA, int PB_, int RB_, int UB_, int NB_>
    static __device__ __forceinline__ void hint_take(Cons & cs, Poll & pl, const Lds & l, xrsrc xr, int unit, unsigned tag, unsigned * go, unsigned gen, int nap,
                                                     bool on_a, Pre<PA_, RA, UA, NA> & a, bool on_b, Pre<PB_, RB_, UB_, NB_> & b, bool no_handover = false) {
        Pre<RG_W1, 1, 1, 1> none;
        none.have = 0u; none.cnt = 0u; none.pos = 0u; none.ro = 0u; none.after = 0u; none.base = 0u;
        hint_take(cs, pl, l, xr, unit, tag, go, gen, nap, on_a, a, on_b, b, false, none, no_handover);
    }
    template <int PH, int R, int U, int NP>
    static __device__ __forceinline__ void hint_take(Cons & cs, Poll & pl, const Lds & l, xrsrc xr, int unit, unsigned tag, unsigned * go, unsigned gen, int nap, bool on, Pre<PH, R, U, NP> & pre) {
        Pre<RG_W1, 1, 1, 1> none;
        none.have = 0u; none.cnt = 0u; none.pos = 0u; none.ro = 0u; none.after = 0u; none.base = 0u;
        hint_take(cs, pl, l, xr, unit, tag, go, gen, nap, on, pre, false, none, false, none);
    }

    // The records of one phase that belong to this wave: j0, j0 + NC, ...; the first TF exist for every wave that owns any (compile-time),
    // one more ("tail") for some. Statically unrolled: the cursor is scalar arithmetic (stream position and ring offset advance by a
    // constant), the loader's progress is compared against a cached scalar, a record's ring reads are immediate offsets from one base
    // register. epi(integral_constant<t>, j, res) receives the row sums of the wave's t-th record (record j of the phase).
    // (The first version walked the records in a loop with the cursor in a struct: ~300 overhead instructions per record -- half of
    //  them scalar, forty branches -- around ~140 useful ones; a C phase took 13.7 k cycles for 3.6 k cycles of arithmetic.)
    // TMIN: every wave that owns any record of the phase owns at least TMIN (= TF when the records are dealt six apart; the smallest share of
    // a balanced deal). Steps t < TMIN are unconditional code; the others ask `t < cnt` (wave-uniform) -- and where there are several of those
    // (balanced deal) their takes wait for their LDS reads at once: a buffer with reads in flight must not cross a join of control flow,
    // where the register allocator may copy it (the copy would carry the old contents). With every step conditional Q8_0 computed garbage.
    template <int PH, int R, int U, int TF, int NP, int TMIN = TF, typename EpiF>
    static __device__ __forceinline__ void rows(Cons & cs, Poll & pl, const Lds & l, const QVec & act, int nbk, Pre<PH, R, U, NP> & pre, EpiF && epi) {
        static_assert(NP >= 1 && NP <= TF + 1 && TMIN <= TF, "rows: buffers");
        if (pre.cnt == 0u) return;
        constexpr bool SAFE = TMIN < TF || R6_ROWS_WAIT != 0;       // conditional takes wait at once
        auto has = [&](int t) { return t < TMIN || (unsigned) t < pre.cnt; };
        const int ln = opq(cs.lane);
        ActRegs<U> ar;
        act_load<FMT, U>(ar, act, nbk, ln);
        // per-lane partial sums of every record of the phase; ONE interleaved butterfly over all of them at the end (a butterfly is six
        // dependent cross-lane steps: per record they ran back to back, two chains at a time)
        float part[(TF + 1) * R];
#pragma unroll
        for (int i = 0; i < (TF + 1) * R; i++) part[i] = 0.0f;
        auto finish = [&](auto tc, const RawRec<FMT, R, U> & wr) {
            constexpr int t = decltype(tc)::value;
            if (cs.dbg & 32) {
                unsigned x = 0;
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int r = 0; r < R; r++) { x ^= (unsigned) wr.raw[u][r].q[0].x ^ (unsigned) wr.raw[u][r].q[0].y ^ (unsigned) wr.raw[u][r].q[0].z ^ (unsigned) wr.raw[u][r].q[0].w ^ wr.raw[u][r].sc; }
                part[t * R] = __uint_as_float(x & 0x3FFFFFu);
            } else
            rec_acc<FMT, R, U>(wr, ar, nbk, ln, part + t * R);
            // (anchor: the arithmetic of record t stays in front of the ring reads of record t + 2 -- volatile statements keep their
            //  order; without it the compiler issues every record's reads first and all the arithmetic behind the last landed check)
#pragma unroll
            for (int r = 0; r < R; r++) asm volatile("" :: "v"(part[t * R + r]));
        };
        // what is not in registers yet (blocking), up to NP records; then the arithmetic, and behind record t its buffer takes record t + NP
        Unroll<0, NP>::run([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if (has(t) && pre.have <= (unsigned) t) (void) rec_take<t, PH, R, U, NP, (SAFE && t >= TMIN)>(cs, pl, l, pre, true);
        });
        // (record t's reads were issued NP - 1 takes ago; the take issued right before this wait -- of record t - 1 + NP, behind the
        //  arithmetic of record t - 1 -- may stay in flight: its LDS latency overlaps the arithmetic instead of preceding it)
        bool took = false;                                         // the previous iteration issued a take (wave-uniform)
        Unroll<0, TF + 1>::run([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if (has(t)) {
                if (NP >= 2 && took) rec_wait<FMT, R, U, rec_lds_ops<FMT, R, U>()>(pre.w[t % NP]);
                else rec_wait<FMT, R, U>(pre.w[t % NP]);
                finish(tc, pre.w[t % NP]);
                took = false;
                if constexpr (t + NP <= TF) {
                    constexpr bool W = SAFE && t + NP >= TMIN;
                    if (has(t + NP)) { (void) rec_take<t + NP, PH, R, U, NP, W>(cs, pl, l, pre, true); took = !W; }
                }
            }
        });
        wave_sum_n<(TF + 1) * R>(part);
        Unroll<0, TF + 1>::run([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if (has(t)) {
                float res[R];
#pragma unroll
                for (int r = 0; r < R; r++) res[r] = part[t * R + r];
                epi(tc, (int) ownj<PH, t>(cs), res);
            }
        });
    }
    // a phase without records taken ahead
    template <int PH, int R, int U, int TF, int NP, typename EpiF>
    static __device__ __forceinline__ void rows(Cons & cs, Poll & pl, const Lds & l, const QVec & act, int nbk, EpiF && epi) {
        Pre<PH, R, U, NP> pre;
        pre_begin<PH>(cs, pre);
        rows<PH, R, U, TF, NP>(cs, pl, l, act, nbk, pre, epi);
    }

    // buffers per wave and phase = records held in registers at once. r/k/v/g sets (C), output rows (E), ffn key sets (K), ffn receptance
    // rows (R), ffn value rows (G). With records taken ahead (PRE_MASK): as many as the register file holds without spilling
    // (tools/check_ring_regs.sh prints the budget of every instantiation); without: the double buffer (one buffer for Q8_0's 36-register
    // blocks and for the long value rows).
    // (TFx: a wave owns TFx or TFx + 1 records of the phase -- dealt six apart; with the balanced deal of C and FK: up to TFx + 1)
    static constexpr int KSETS = UF * 64 > NBLK ? 32 : 16 /* two-row key sets of a workgroup that owns any */,
                         KCOMM = (KSETS % NC <= 2) ? KSETS % NC : 0 /* ... of which the comm wave takes the last ones (ring_geom.h, rg_key_comm) */;
    static constexpr int TFC = BAL ? RG_BAL_C_N[1] - 1 : (D * 4 / NBLK / 2) / NC, TFE = RE / NC, TFK = BAL ? RG_BAL_K_N[1] - 1 : (KSETS - KCOMM) / NC;
    static constexpr int TMC = BAL ? RG_BAL_C_N[2] : TFC, TMK = BAL ? RG_BAL_K_N[2] : TFK;   // the smallest share of a wave that owns any
    static_assert(!BAL || (D * 4 / NBLK / 2 == RG_BAL_C_SETS && KSETS - KCOMM == RG_BAL_K_SETS), "balanced deal: geometry");
    static constexpr bool Q8 = QF<FMT>::QS == 32, Q5 = QF<FMT>::QH;
    static constexpr int npcap(int want, int tf) { return want < 1 ? 1 : (want > tf + 1 ? tf + 1 : want); }
#ifndef R6_NPC
#define R6_NPC (Q8 ? 3 : (Q5 ? 5 : 6))
#endif
#ifndef R6_NPE
#define R6_NPE 3
#endif
#ifndef R6_NPK
#define R6_NPK ((PRE_MASK & 128) ? (Q8 ? 2 : (Q5 ? 4 : 5)) : (Q8 ? 3 : (Q5 ? 5 : 6)))   /* (with the receptance records held as well: one buffer less) */
#endif
#ifndef R6_NPG
#define R6_NPG (Q8 ? 1 : 3)
#endif
    static constexpr int NPC = npcap((PRE_MASK & 1) ? R6_NPC : (Q8 ? 1 : 2), TFC);
    static constexpr int NPE = npcap((PRE_MASK & 2) ? R6_NPE : (Q8 ? 1 : 2), TFE);
    static constexpr int NPK = npcap((PRE_MASK & 4) ? R6_NPK : (Q8 ? 1 : 2), TFK);
    static constexpr int NPR = npcap((PRE_MASK & 128) ? 3 : (Q8 ? 1 : 2), TFE);
    static constexpr int NPG = npcap((PRE_MASK & 8) ? R6_NPG : 1, TFE);
    // value records are taken ahead where one fits 48 registers (not Q8_0's 63 at F = 14336). Rounds 4 and 5 tested sizeof(RawRec) <= 192 here:
    // the 16-byte aligned code vectors pad every block to 32 bytes, 7 blocks = 224, and the take was off for EVERY format at the 7B geometry
    // (tools/trace_ring.py, "records already in registers": G 0.00) -- R6_G_PRE_OLD=1 rebuilds that.
#ifndef R6_G_PRE_OLD
#define R6_G_PRE_OLD 0
#endif
    static constexpr int rec_regs(int R, int U) { return U * R * (QF<FMT>::QS / 4 + 1 + (QF<FMT>::QH ? 1 : 0)); }
    static constexpr bool G_PRE = R6_G_PRE_OLD ? sizeof(RawRec<FMT, 1, UF>) <= 48 * 4 : rec_regs(1, UF) <= 48;

    static __device__ __forceinline__ void consumer_main(const R6P & p, const Lds & l, int lane, int wave, unsigned base) {
        const int blk = blockIdx.x;
        const int c = wave - 2;                       // consumer index = gather share
        const bool pro = c < 4 || SHI > 0;            // takes part in the prologues' elementwise part
        // (The prologue parameters are loaded by EVERY consumer wave, also where waves 4, 5 take no part: a load under `if (pro)` is a
        //  conditional definition, and the compiler then waits for it and copies it right where it is issued.)
        const int F = p.F, nbF = F / 32;
        Poll pl{p.ctl, false, p.xch, R6_SWATCH == 1};
        const M6Arena ar{p.arena};
        const xrsrc xr = make_xrsrc(p.xch, p.xch_bytes);
        const RingShape sh = shape(p);
        Cons cs;
        cs.cu = rg_cu(sh, blk); cs.lbase = 0; cs.landed = 0; cs.RB = __builtin_amdgcn_readfirstlane(p.ring_bytes); cs.ring_lds = __builtin_amdgcn_readfirstlane((unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) l.ring); cs.c = c; cs.lane = lane; cs.dbg = __builtin_amdgcn_readfirstlane(p.dbg); cs.look = (unsigned) __builtin_amdgcn_readfirstlane(p.look); cs.waited = 0;
        const int mat = (blk * (4 * D / NBLK)) / D;   // which of r, k, v, g this workgroup's sets belong to
        const int cbase = (blk * (4 * D / NBLK)) % D;
        const bool has_dw1 = blk < p.DR;

        // where this wave's stream continues behind a layer: its first record of the next layer, or of the head, or nowhere
        const unsigned first_own = rg_next_own_in_layer(cs.cu, c, 0);
        const RingHead hd = rg_head(p.n_vocab, D);
        const unsigned hbase = (unsigned) p.n_layers * cs.cu.layer_bytes;
        const unsigned head_first = (p.logits && c < hd.hg) ? hbase + rg_head_off(hd, 0, 0, c) : 0xFFFFFFFFu;
        float xown[XT], rrow[XT];
#pragma unroll
        for (int t = 0; t < XT; t++) { const int row = c + NC * t; xown[t] = row < RE ? p.x[blk * RE + row] : 0.0f; rrow[t] = 0.0f; }
        const bool emb_in = p.tok != nullptr;          // the launch starts from the token id (wave-uniform)
        const unsigned eg = emb_in ? 1u : 0u;
        if (emb_in && c < 4) embed_ln0(p, pl, l, ar, c * 64 + opq(lane), opq(lane));
        // Where the prologue parameters are loaded (LayerNorm affine, token-shift source, mix weights: 4 / 5 float4 per slot and thread).
        // R6_LATE_PARAMS = 0 (round 3): a whole phase ahead -- 48 / 60 registers live across a hand-over wait. 1: at the start of the
        // prologue, in flight under the LayerNorm statistics (two reduction rounds, ~1 us; the lines are the same for every workgroup:
        // L2 hits) -- the hand-over waits then hold next to nothing but records taken ahead, which is what the registers are for now.
        // 2 (default): both when the x hand-over's sentinel has turned -- in flight under the sweep and the statistics, behind the records taken
        // during the wait (LN1's a phase ahead as in round 3 would be 48 registers live across the loop's back edge and the wait in which the
        // r/k/v/g records are taken). Measured on the 7B: placement 1 costs 1.5 + 0.6 us per layer in the two prologues (the loads are NOT
        // covered by the statistics).
#ifndef R6_LATE_PARAMS
#define R6_LATE_PARAMS 2
#endif
        // (D = 2048 takes no records ahead and holds nothing across the x hand-over: there LN1's parameters go out a phase ahead as in round 3 --
        //  issued behind the sentinel they cost the 1.6B 1.8 %)
        constexpr bool PA_EARLY = R6_LATE_PARAMS == 0 || (R6_LATE_PARAMS == 2 && (PRE_MASK & 16) == 0);
        constexpr bool PF_EARLY = R6_LATE_PARAMS == 0 || (R6_LATE_PARAMS == 2 && (PRE_MASK & 4) == 0);
        PA pa; PF pf;
        if (PA_EARLY) issue_pa(pa, ar, p.layers[0], p.sin, c, opq(lane));

        for (int li = 0; li < p.n_layers; li++) {
            const M6Layer & L = p.layers[li];
            const float * sin_l = p.sin + (long long) li * p.state_stride;
            float * sout_l = p.sout + (long long) li * p.state_stride;
            const unsigned tagL = base + (unsigned) li * 8u;
            const unsigned g1 = (unsigned) li + 1u;   // generation of this layer's once-per-layer counters
            cs.lbase = (unsigned) li * cs.cu.layer_bytes;
            cs.next_block = li + 1 < p.n_layers ? cs.lbase + cs.cu.layer_bytes + first_own : head_first;
            R6STAMP(0);
            if (p.trace && li == p.trace_layer && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 23] = cs.waited;
            unsigned havepk = 0u;   // (tracing) records of each row phase already in registers when the phase starts, 4 bits per phase
            Pre<RG_W1, 1, UD, 1> pw;
            pre_begin<RG_W1>(cs, pw);
            Pre<RG_C, 2, UD, NPC> pc;
            pre_begin<RG_C>(cs, pc);
            // (per-lane offsets are derived from an opaque copy of the lane index in every phase: left alone, the compiler hoists a
            //  hundred loop-invariant address registers of the gathers out of the layer loop and spills them)
            // ---- A: x, LN1 + mix + quantise, W1 rows ----
            {
                // (the first layer of a launch has no hand-over in front of it: its x lies in plain memory; same statements, see gather_x)
                const bool first = li == 0;
                watch_begin(l);
                hint_take(cs, pl, l, xr, p.xffn + ((blk * 37 + c * 211) & 1023), tagL - 8u + SLOT_XFFN, l.fl + FL_HX, 2u * li + 1u, p.nap, (PRE_MASK & 16) != 0 && !first, pw,
                          (PRE_MASK & 16) != 0 && !first, pc, first);
                watch_end(l);
                if (R6_LATE_PARAMS == 2 && !PA_EARLY) { issue_pa(pa, ar, L, sin_l, c, opq(lane)); __builtin_amdgcn_sched_barrier(0); }
                sweep_begin(l);
                gather_x(pl, xr, p.xffn, tagL - 8u + SLOT_XFFN, c, opq(lane), l.x, first ? p.x : nullptr, first && emb_in);
            }
            gather_meet(pl, l.fl + FL_GX, 2u * li + 1u);
            sweep_end(l);
            R6STAMP(1);
            {   // behind an in-launch embedding the residual rows of this wave start from l.x (a select, not a branch: see gather_x)
                const bool take = li == 0 && emb_in;
#pragma unroll
                for (int t = 0; t < XT; t++) { const int row = c + NC * t; const float v = l.x[blk * RE + (row < RE ? row : 0)]; xown[t] = take ? v : xown[t]; }
            }
            if (R6_LATE_PARAMS == 1) { issue_pa(pa, ar, L, sin_l, c, opq(lane)); __builtin_amdgcn_sched_barrier(0); }
            if (pro) prologue_A(pl, l, pa, sout_l, blk == 0, c, opq(lane), 2u * li + 1u, eg);
            prologue_wait(pl, l, 2u * li + 1u);
            R6STAMP(2);
            havepk |= pw.have;
            rows<RG_W1, 1, UD, 0, 1>(cs, pl, l, qvec_at(l.q1, D), nb, pw, [&](auto, int j, const float (&res)[1]) {
                if (lane == 0) tg_store(xr, p.tl + blk + NBLK * j, __float_as_uint(det_tanhf(res[0])), 0u, 0u, 0u, tagL + SLOT_TL);
            });
            R6STAMP(3);
            // ---- C: the mixed inputs (this workgroup's matrix reads ONE of the five), decay row, r/k/v/g sets ----
            Pre<RG_E, 1, UD, NPE> pe;
            pre_begin<RG_E>(cs, pe);
            {
                const int img = (0x4213 >> (4 * mat)) & 0xF;   // r, k, v, g -> mix image (w, k, v, r, g order)
                watch_begin(l);
                hint_take(cs, pl, l, xr, p.act5 + img * p.act_stride + ((blk * 7 + c * 19) & 127), tagL + SLOT_ACT, l.fl + FL_HACT, g1, p.nap, (PRE_MASK & 1) != 0, pc, (PRE_MASK & 32) != 0, pe);
                watch_end(l);
                sweep_begin(l);
                gather_qvec<DSL>(pl, xr, p.act5 + img * p.act_stride, D, tagL + SLOT_ACT, c, opq(lane), l.act);
                if (has_dw1) gather_qvec<DSL>(pl, xr, p.act5, D, tagL + SLOT_ACT, c, opq(lane), l.actw);
                gather_meet(pl, l.fl + FL_GACT, g1);
                sweep_end(l);
            }
            R6STAMP(4);
            {
                // all row sums first, then ONE epilogue: lane 2 t + r finishes row r of this wave's t-th set (the gate's silu is a double-
                // precision exp: once per phase, not once per record) and lanes 0, 2, 4, ... store their set's unit with one instruction
                constexpr int MAXT = TFC + 1;
                float all[2 * MAXT];
#pragma unroll
                for (int t = 0; t < 2 * MAXT; t++) all[t] = 0.0f;
                if (p.trace && li == p.trace_layer && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 20] = (long long) fl_ld(l.fl + FL_LANDED) * 1024ll - (long long) (cs.lbase + cs.cu.off[RG_C]);
                R6RSTAMP(26);
                havepk |= pc.have << 4;
                rows<RG_C, 2, UD, TFC, NPC, TMC>(cs, pl, l, qvec_at(l.act, D), nb, pc, [&](auto tc, int, const float (&res)[2]) {
                    constexpr int t = decltype(tc)::value;
                    all[2 * t] = res[0]; all[2 * t + 1] = res[1];
                });
                const int ln = opq(lane);
                float v = pick_lane<2 * MAXT>(all, ln);
                if (mat == 3) v = v / (1.0f + det_expf(-v));     // gate: silu
                const int v1 = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xF, 0xF, true);   // lane 2 t collects row 1 (row_shl:1)
                int j = 0x7fff;   // record number of the set lane ln >> 1 finishes (a select chain over this wave's records)
                Unroll<0, MAXT>::run([&](auto tc) { constexpr int t = decltype(tc)::value; const int jt = (t < TMC || (unsigned) t < pc.cnt) ? (int) ownj<RG_C, t>(cs) : 0x7fff; j = (ln >> 1) == t ? jt : j; });
                if (ln < 2 * MAXT && (ln & 1) == 0 && j < (int) cs.cu.n[RG_C])
                    tg_store(xr, p.rkvg + ((mat * D + cbase + 2 * j) >> 1), __float_as_uint(v), (unsigned) v1, 0u, 0u, tagL + SLOT_RKVG);
            }
            R6STAMP(5); R6RSTAMP(27);
            if (p.trace && li == p.trace_layer && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 24] = cs.waited;
            if (PF_EARLY) issue_pf(pf, ar, L, sin_l, c, opq(lane));
            __builtin_amdgcn_sched_barrier(0);
            // ---- E: output projection + residual ----
            Pre<RG_FK, 2, UD, NPK> pk;
            pre_begin<RG_FK>(cs, pk);
            Pre<RG_FR, 1, UD, NPR> pr;
            pre_begin<RG_FR>(cs, pr);
            watch_begin(l);
            hint_take(cs, pl, l, xr, p.yq + ((blk * 7 + c * 19) & 127), tagL + SLOT_YQ, l.fl + FL_HYQ, g1, p.nap, (PRE_MASK & 2) != 0, pe, (PRE_MASK & 64) != 0, pk, (PRE_MASK & 128) != 0, pr);
            watch_end(l);
            sweep_begin(l);
            gather_qvec<DSL>(pl, xr, p.yq, D, tagL + SLOT_YQ, c, opq(lane), l.yq);
            gather_meet(pl, l.fl + FL_GYQ, g1);
            sweep_end(l);
            R6STAMP(6);
            havepk |= pe.have << 8;
            rows<RG_E, 1, UD, RE / NC, NPE>(cs, pl, l, qvec_at(l.yq, D), nb, pe, [&](auto tc, int, const float (&res)[1]) {
                constexpr int t = decltype(tc)::value;
                xown[t] = xown[t] + res[0];
            });
            if (lane == 0) tg_store(xr, p.xatt + blk * NC + c, __float_as_uint(xown[0]), __float_as_uint(xown[XT > 1 ? 1 : 0]), __float_as_uint(xown[XT > 2 ? 2 : 0]), 0u, tagL + SLOT_XATT);
            R6STAMP(7);
            // ---- F: x, LN2 + mixes + quantise, key sets (-> comm quantises them), receptance rows ----
            watch_begin(l);
            hint_take(cs, pl, l, xr, p.xatt + ((blk * 37 + c * 211) & 1023), tagL + SLOT_XATT, l.fl + FL_HX, 2u * li + 2u, p.nap, (PRE_MASK & 4) != 0, pk, (PRE_MASK & 128) != 0, pr);
            watch_end(l);
            if (R6_LATE_PARAMS == 2 && !PF_EARLY) { issue_pf(pf, ar, L, sin_l, c, opq(lane)); __builtin_amdgcn_sched_barrier(0); }
            sweep_begin(l);
            gather_x(pl, xr, p.xatt, tagL + SLOT_XATT, c, opq(lane), l.x);
            gather_meet(pl, l.fl + FL_GX, 2u * li + 2u);
            sweep_end(l);
            R6STAMP(8);
            if (R6_LATE_PARAMS == 1) { issue_pf(pf, ar, L, sin_l, c, opq(lane)); __builtin_amdgcn_sched_barrier(0); }
            if (pro) prologue_F(pl, l, pf, sout_l, blk == 0, c, opq(lane), 2u * li + 2u, eg);
            prologue_wait(pl, l, 2u * li + 2u);
            R6STAMP(9); R6RSTAMP(28);
            if (p.trace && li == p.trace_layer && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 21] = (long long) fl_ld(l.fl + FL_LANDED) * 1024ll - (long long) (cs.lbase + cs.cu.off[RG_FK]);
            havepk |= (pk.have << 12) | (pr.have << 16);
            rows<RG_FK, 2, UD, TFK, NPK, TMK>(cs, pl, l, qvec_at(l.q1, D), nb, pk, [&](auto, int j, const float (&res)[2]) {
                const float v = lane == 1 ? res[1] : res[0];
                const float t = v > 0.0f ? v : 0.0f;
                if (lane < 2) l.out[2 * j + lane] = t * t;
            });
            fl_add(l.fl + FL_KEYS, 1u);
            if constexpr (DEFER_XR) {
                if (pro) quant_xr(l, c, opq(lane));
                fl_wait(pl, l.fl + FL_XRQ, (unsigned) (SHI > 0 ? NC : 4) * g1);
            }
            R6STAMP(10);
            rows<RG_FR, 1, UD, RE / NC, NPR>(cs, pl, l, qvec_at(l.q2, D), nb, pr, [&](auto tc, int, const float (&res)[1]) {
                constexpr int t = decltype(tc)::value;
                rrow[t] = res[0];
            });
            R6STAMP(11); R6RSTAMP(29);
            // ---- G: value projection, x += sigmoid(r) * (Wv k) ----
            Pre<RG_G, 1, UF, NPG> pg;
            pre_begin<RG_G>(cs, pg);
            // (registers: not the long Q8_0 rows of the 7B geometry)
            watch_begin(l);
            { const unsigned lk = cs.look; cs.look = (unsigned) __builtin_amdgcn_readfirstlane(p.look_g);
              hint_take(cs, pl, l, xr, p.kq + ((blk * 5 + c * 173) & 511), tagL + SLOT_KQ, l.fl + FL_HKQ, g1, p.nap, (PRE_MASK & 8) != 0 && G_PRE, pg);
              cs.look = lk; }
            watch_end(l);
            sweep_begin(l);
            gather_qvec<KSL>(pl, xr, p.kq, F, tagL + SLOT_KQ, c, opq(lane), l.kq);
            gather_meet(pl, l.fl + FL_GKQ, g1);
            sweep_end(l);
            R6STAMP(12); R6RSTAMP(30);
            if (p.trace && li == p.trace_layer && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 22] = (long long) fl_ld(l.fl + FL_LANDED) * 1024ll - (long long) (cs.lbase + cs.cu.off[RG_G]);
            havepk |= pg.have << 20;
            rows<RG_G, 1, UF, RE / NC, NPG>(cs, pl, l, qvec_at(l.kq, F), nbF, pg, [&](auto tc, int j, const float (&res)[1]) {
                constexpr int t = decltype(tc)::value;
                const float gte = sigmoid_f(rrow[t]) * res[0];
                xown[t] = xown[t] + gte;
                if (li == p.n_layers - 1 && lane == 0) p.x_out[blk * RE + j] = xown[t];
            });
            if (lane == 0) tg_store(xr, p.xffn + blk * NC + c, __float_as_uint(xown[0]), __float_as_uint(xown[XT > 1 ? 1 : 0]), __float_as_uint(xown[XT > 2 ? 2 : 0]), 0u, tagL + SLOT_XFFN);
            R6STAMP(13); R6RSTAMP(14);
            if (p.trace && li == p.trace_layer && lane == 0) { p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 25] = cs.waited; p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 15] = (long long) havepk; }
            __builtin_amdgcn_sched_barrier(0);   // (the loads below stay behind the value rows: hoisted into them they cost 48 registers at the kernel's peak)
            if (PA_EARLY) {   // the next layer's prologue parameters: in flight while this wave watches the x hand-over's sentinel (not across G: registers)
                // (unconditionally -- behind the last layer the same layer's again: under `if (li + 1 < n_layers)` the parameters are
                //  conditionally redefined, and the old values stay live through the whole layer on the path the compiler cannot rule out)
                const int nl = li + 1 < p.n_layers ? li + 1 : li;
                issue_pa(pa, ar, p.layers[nl], p.sin + (long long) nl * p.state_stride, c, opq(lane));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.logits) head_phase(p, l, cs, pl, xr, ar, hd, hbase, lane, wave, base, eg);
    }

    // -----------------------------------------------------------------------------------------------------------
    // the head behind the last layer: ln_out on the gathered residual stream, logits = head.weight (F16) . fp16(ln_out(x))
    // (rwkv_graph.inc:704-708; ggml's F16 dot: activations rounded to fp16, 32 partial sums k mod 32 accumulated with fma in increasing k,
    // folded 16 / 8 / 4, (p0 + p1) + (p2 + p3) -- the order of k_mvf in kernels.hip: four lanes per row, lane q holds partials 8 q .. 8 q + 7)
    // -----------------------------------------------------------------------------------------------------------
    static __device__ __forceinline__ void head_phase(const R6P & p, const Lds & l, Cons & cs, Poll & pl, xrsrc xr, const M6Arena & ar, const RingHead & hd,
                                                      unsigned hbase, int lane, int wave, unsigned base, unsigned eg) {
        const int blk = blockIdx.x, c = cs.c, li = p.n_layers;       // (li: the layer index the stamps and generations continue with)
        const unsigned tagL = base + (unsigned) li * 8u;
        const int pt = c * 64 + lane;
        R6STAMP(0); R6RSTAMP(16);
        // x after the last layer
        watch_begin(l);
        gather_hint(pl, xr, p.xffn + ((blk * 37 + c * 211) & 1023), tagL - 8u + SLOT_XFFN, l.fl + FL_HX, 2u * li + 1u, p.nap);
        watch_end(l);
        sweep_begin(l);
        gather_x(pl, xr, p.xffn, tagL - 8u + SLOT_XFFN, c, opq(lane), l.x);
        gather_meet(pl, l.fl + FL_GX, 2u * li + 1u);
        sweep_end(l);
        R6STAMP(1);
        if (c < 4) {
            const float scale = ln_stats(pl, l, pt, lane, 2u * li + 1u + eg);
            const float * lw = ar.f(p.lnout_w), * lb = ar.f(p.lnout_b);
#pragma unroll
            for (int u = 0; u < (D + 1023) / 1024; u++) {
                const int i = pt * 4 + u * 1024;
                if (i >= D) break;
                const float4 xc = *reinterpret_cast<const float4 *>(l.x + i);
                const float4 w4 = *reinterpret_cast<const float4 *>(lw + i), b4 = *reinterpret_cast<const float4 *>(lb + i);
                const float xs[4] = {xc.x, xc.y, xc.z, xc.w}, ws[4] = {w4.x, w4.y, w4.z, w4.w}, bs[4] = {b4.x, b4.y, b4.z, b4.w};
                float y[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { const float a = xs[j] * scale; const float aw = a * ws[j]; y[j] = round_f16(aw + bs[j]); }
                *reinterpret_cast<float4 *>(l.x + i) = make_float4(y[0], y[1], y[2], y[3]);
            }
            fl_add(l.fl + FL_PRO, 1u);
        }
        fl_wait(pl, l.fl + FL_PRO, 4u * (2u * li + 1u));
        R6STAMP(2);
        if (p.trace && li == p.trace_layer && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 20] = (long long) fl_ld(l.fl + FL_LANDED) * 1024ll - (long long) hbase;
        long long waited = 0;
        // row groups of this wave
        unsigned * const dn = l.fl + FL_DONE + 2 + c;
        const int ln = opq(lane), q = ln & 3;
        constexpr int CHK = nb / RG_HSTEPS;      // records per row group
        if (c >= hd.hg) { asm volatile("" ::: "memory"); __hip_atomic_store(dn, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        float am_best = -INFINITY; int am_i = 0x7fffffff;   // this lane's argmax candidate over the rows it finishes (q == 0 lanes)
        for (int ps = 0; ps < hd.passes; ps++) {
            const int g = RG_NC * ps + c;
            if (g >= hd.hg) break;
            const unsigned stride = (unsigned) rg_head_npc(hd, ps) * RG_HREC;
            unsigned pos = __builtin_amdgcn_readfirstlane(hbase + rg_head_off(hd, ps, 0, c));
            unsigned ro = __builtin_amdgcn_readfirstlane(pos - (pos / cs.RB) * cs.RB);
            const bool more = RG_NC * (ps + 1) + c < hd.hg;
            const unsigned after = more ? hbase + rg_head_off(hd, ps + 1, 0, c) : 0xFFFFFFFFu;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] = 0.0f;
            for (int k = 0; k < CHK; k++) {
                const unsigned need = (pos + RG_HREC + 1023u) >> 10;
                if (cs.landed < need) {
                    const long long t0 = p.trace ? (long long) __builtin_readcyclecounter() : 0;
                    for (unsigned spin = 0;; spin++) {
                        cs.landed = fl_ld(l.fl + FL_LANDED);
                        if (cs.landed >= need || pl.dead) break;
                        if (lds_backoff(pl, spin)) break;
                    }
                    if (p.trace) waited += (long long) __builtin_readcyclecounter() - t0;
                }
                asm volatile("" ::: "memory");
                const int4 * wp = reinterpret_cast<const int4 *>(l.ring + ro) + ln;
                int4 raw[RG_HSTEPS];
#pragma unroll
                for (int st = 0; st < RG_HSTEPS; st++) raw[st] = wp[st * 64];
                asm volatile("" ::: "memory");
                __hip_atomic_store(dn, k + 1 < CHK ? pos + stride : after, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const float * xk = l.x + k * (RG_HSTEPS * 32) + 8 * q;
#pragma unroll
                for (int st = 0; st < RG_HSTEPS; st++) {
                    const float4 xa = *reinterpret_cast<const float4 *>(xk + st * 32), xb = *reinterpret_cast<const float4 *>(xk + st * 32 + 4);
                    const unsigned u[4] = {(unsigned) raw[st].x, (unsigned) raw[st].y, (unsigned) raw[st].z, (unsigned) raw[st].w};
                    float w[8];
#pragma unroll
                    for (int i = 0; i < 4; i++) { w[2 * i] = h2f_bits((uint16_t) (u[i] & 0xFFFFu)); w[2 * i + 1] = h2f_bits((uint16_t) (u[i] >> 16)); }
                    acc[0] = fmaf(w[0], xa.x, acc[0]); acc[1] = fmaf(w[1], xa.y, acc[1]); acc[2] = fmaf(w[2], xa.z, acc[2]); acc[3] = fmaf(w[3], xa.w, acc[3]);
                    acc[4] = fmaf(w[4], xb.x, acc[4]); acc[5] = fmaf(w[5], xb.y, acc[5]); acc[6] = fmaf(w[6], xb.z, acc[6]); acc[7] = fmaf(w[7], xb.w, acc[7]);
                }
                pos += stride; ro += stride;
                while (ro >= cs.RB) ro -= cs.RB;
            }
            float ps8[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float v = acc[e];
                v += __int_as_float(lane_xor2_i(__float_as_int(v)));   // ps[i] += ps[i + 16]
                v += __int_as_float(lane_xor1_i(__float_as_int(v)));   // ps[i] += ps[i + 8]
                ps8[e] = v;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) ps8[e] += ps8[e + 4];
            const float r = (ps8[0] + ps8[1]) + (ps8[2] + ps8[3]);
            const int vrow = blk * hd.hg * 16 + g * 16 + (ln >> 2);
            if (q == 0) { p.logits[vrow] = r; if (r > am_best) { am_best = r; am_i = vrow; } }   // (rows come in increasing order per lane: ties keep the smaller index)
            if (ps < 4) R6STAMP(3 + ps);
        }
        // this wave's candidate -> l.bc[c]; the comm wave merges the six and publishes the workgroup's (tail_argmax)
        am_wave(am_best, am_i);
        if (ln == 0) { l.bc[2 * c] = am_best; l.bc[2 * c + 1] = __int_as_float(am_i); }
        fl_add(l.fl + FL_AM, 1u);
        R6STAMP(7); R6RSTAMP(17);
        if (p.trace && li == p.trace_layer && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 21] = waited;
    }

    // -----------------------------------------------------------------------------------------------------------
    // comm wave
    // -----------------------------------------------------------------------------------------------------------
    static __device__ __forceinline__ void comm_main(const R6P & p, const Lds & l, int lane, unsigned base) {
        const int wave = 1;
        const int blk = blockIdx.x;
        const int g = NC;                              // gather share
        const int F = p.F, DR = p.DR, R = p.R, H = p.H;
        const int nbF = F / 32;
        Poll pl{p.ctl, false, p.xch, R6_SWATCH != 0};
        const M6Arena ar{p.arena};
        const xrsrc xr = make_xrsrc(p.xch, p.xch_bytes);
        const int mat = (blk * (4 * D / NBLK)) / D;
        const bool has_dw1 = blk < DR;
        // the WKV heads run on workgroups hw0 .. hw0 + H - 1
        const int hw0 = __builtin_amdgcn_readfirstlane(p.head_wg0);
        const bool d_has = blk >= hw0 && blk < hw0 + H;
        const int d_head = blk - hw0;
        const int gpb = (nbF + NBLK - 1) / NBLK;
        // B: 64-element chunks of the five mixes; chunk ch < NBLK on workgroup ch, the rest on workgroups NBLK/4.. (the first quarter
        // runs the WKV heads)
        constexpr int NCH = 5 * (D / 64);
        const int b_extra = NCH - NBLK;
        const int b_chunk2 = (b_extra > 0 && blk >= NBLK / 4 && blk < NBLK / 4 + b_extra) ? NBLK + (blk - NBLK / 4) : -1;

        for (int li = 0; li < p.n_layers; li++) {
            const M6Layer & L = p.layers[li];
            const float * sin_l = p.sin + (long long) li * p.state_stride;
            float * sout_l = p.sout + (long long) li * p.state_stride;
            const unsigned tagL = base + (unsigned) li * 8u;
            const unsigned g1 = (unsigned) li + 1u;
            R6STAMP(0);
            // ---- A ----
            {
                const bool first = li == 0;
                watch_begin(l);
                if (!first) gather_hint(pl, xr, p.xffn + ((blk * 37 + g * 211) & 1023), tagL - 8u + SLOT_XFFN, l.fl + FL_HX, 2u * li + 1u, p.nap);
                watch_end(l);
                sweep_begin(l);
                gather_x(pl, xr, p.xffn, tagL - 8u + SLOT_XFFN, g, opq(lane), l.x, first ? p.x : nullptr, first && p.tok != nullptr);
            }
            gather_meet(pl, l.fl + FL_GX, 2u * li + 1u);
            sweep_end(l);
            R6STAMP(1); R6RSTAMP(17);
            // W2 of this workgroup's chunk(s) (chunk-blocked copy: lane d reads float4 {m .. m+3}) and the chunk's LayerNorm / shift
            // parameters: in flight while the prologue waves run
            const int lnA = opq(lane);
            float4 wB4[2][16]; float wBmaa[2], cw[2], cb_[2], cpv[2];
            int bf[2], bd[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int ch = q == 0 ? (blk < NCH ? blk : 0) : (b_chunk2 >= 0 ? b_chunk2 : 0);
                bf[q] = ch / (D / 64);
                bd[q] = (ch % (D / 64)) * 64 + lnA;
                const float4 * cbp = reinterpret_cast<const float4 *>(p.w2b + L.w2b + (long long) ch * R * 64);
#pragma unroll
                for (int m4 = 0; m4 < 16; m4++) { const int4 t = ldw16(cbp + (m4 < R / 4 ? m4 : R / 4 - 1) * 64 + lnA); wB4[q][m4] = make_float4(__int_as_float(t.x), __int_as_float(t.y), __int_as_float(t.z), __int_as_float(t.w)); }
                wBmaa[q] = ar.f(L.maa[bf[q]])[bd[q]];
                cw[q] = ar.f(L.ln1_w)[bd[q]]; cb_[q] = ar.f(L.ln1_b)[bd[q]]; cpv[q] = sin_l[D + bd[q]];
            }
            // the decay row of this workgroup (time_decay_w1 row blk, blk < DR): lane l holds blocks l, l + 64, ... like a ring record
            RawRec<FMT, 1, UD> dwr;
            {
                const WPl dw1 = ar.w(L.dw1);
                const int drow = has_dw1 ? blk : 0;
#pragma unroll
                for (int u = 0; u < UD; u++) { const int bb = u * 64 + lnA; RawBlk<FMT> rb; rb.qh = 0u; load_raw<FMT>(rb, dw1.qs, dw1.qh, dw1.sc, (long long) drow * nb + (bb < nb ? bb : nb - 1)); from_raw<FMT>(dwr.raw[u][0], rb); }
            }
            __builtin_amdgcn_sched_barrier(0);
            fl_wait(pl, l.fl + FL_PRO, 4u * (2u * li + 1u));   // l.x holds x - mean, l.misc[0] the scale
            R6STAMP(2);
            // ---- B: the data-dependent mixes of this workgroup's chunk(s) (rwkv_graph.inc:313-346) ----
            {
                const int ln = opq(lane);
                const float scale = l.misc[0];
                float cxn[2], csx[2];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const float y = l.x[bd[q]] * scale;
                    const float yw = y * cw[q];
                    cxn[q] = yw + cb_[q];
                    csx[q] = cpv[q] - cxn[q];
                }
                watch_begin(l);
                if (!(p.dbg & 128)) gather_hint(pl, xr, p.tl + ((blk * 7) & 127), tagL + SLOT_TL, l.fl + FL_HTL, g1, p.nap);   // (one unit first, then the sweep; RWKV_MI_RING_DBG bit 7: the sweep at once -- one wave per workgroup, 5 KB per attempt)
                poll_units<5, 64>(pl, xr, p.tl, 5 * R, tagL + SLOT_TL, ln, [&](int i, const v4u & v) { l.tl[i] = __uint_as_float(v.x); });
                watch_end(l);
                __builtin_amdgcn_wave_barrier();
                R6STAMP(3); R6RSTAMP(18);
                // both chunks' sums in ONE loop: a sum is a chain of R dependent adds (its order is the reference's), and the second
                // chain runs in the issue slots the first one leaves empty -- a workgroup with two chunks stores its second image
                // about when the others store their only one, and the act hand-over waits for the last store. (Workgroups without a
                // second chunk run the loop on a copy of chunk 0 and store nothing for it.)
                {
                    const float4 * tla = reinterpret_cast<const float4 *>(l.tl + bf[0] * R), * tlb = reinterpret_cast<const float4 *>(l.tl + bf[1] * R);
                    float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
                    for (int h = 0; h < (R > 32 ? 2 : 1); h++) {
                        float4 ta[8], tb[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) { ta[j] = tla[8 * h + j]; tb[j] = tlb[8 * h + j]; }
#pragma unroll
                        for (int m = 0; m < 32; m++) {
                            acc0 += (&wB4[0][8 * h + (m >> 2)].x)[m & 3] * (&ta[m >> 2].x)[m & 3];
                            acc1 += (&wB4[1][8 * h + (m >> 2)].x)[m & 3] * (&tb[m >> 2].x)[m & 3];
                        }
                    }
                    const float accq[2] = {acc0, acc1};
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const bool has = q == 0 ? blk < NCH : b_chunk2 >= 0;
                        const float mm = (accq[q] + wBmaa[q]) * csx[q];
                        const float o = mm + cxn[q];
                        int qi, isum; float d16, s16;
                        quant_block32(o, qi, d16, s16, isum);
                        if (has) tq_store_block(xr, p.act5 + bf[q] * p.act_stride, bd[q] >> 5, ln & 31, qi, d16, s16, isum, tagL + SLOT_ACT);
                    }
                }
            }
            R6RSTAMP(19); R6STAMP(4);
            // ---- C: stage this workgroup's share of the mixed inputs ----
            {
                const int img = (0x4213 >> (4 * mat)) & 0xF;
                watch_begin(l);
                gather_hint(pl, xr, p.act5 + img * p.act_stride + ((blk * 7 + g * 19) & 127), tagL + SLOT_ACT, l.fl + FL_HACT, g1, p.nap);
                watch_end(l);
                sweep_begin(l);
                gather_qvec<DSL>(pl, xr, p.act5 + img * p.act_stride, D, tagL + SLOT_ACT, g, opq(lane), l.act);
                if (has_dw1) gather_qvec<DSL>(pl, xr, p.act5, D, tagL + SLOT_ACT, g, opq(lane), l.actw);
                gather_meet(pl, l.fl + FL_GACT, g1);
                sweep_end(l);
            }
            R6STAMP(5); R6RSTAMP(20);
            // ---- the decay row against the w mix (rows<> of a consumer wave: per-lane block sums in increasing order, one butterfly) ----
            if (has_dw1) {
                const int ln = opq(lane);
                ActRegs<UD> arw;
                act_load<FMT, UD>(arw, qvec_at(l.actw, D), nb, ln);
                float part[1];
                rec_acc<FMT, 1, UD>(dwr, arw, nb, ln, part);
                wave_sum_n<1>(part);
                if (ln == 0) tg_store(xr, p.dl + blk, __float_as_uint(det_tanhf(part[0])), 0u, 0u, 0u, tagL + SLOT_RKVG);
            }
            // ---- D: WKV head of this workgroup ----
            if (d_has) {
                const int ln = opq(lane);
                const int c = d_head * S + ln;
                RawBlk<FMT> w2[NBD];
                const WPl dw2 = ar.w(L.dw2);
#pragma unroll
                for (int b = 0; b < NBD; b++) load_raw<FMT>(w2[b], dw2.qs, dw2.qh, dw2.sc, (long long) c * NBD + b);
                const float td = ar.f(L.time_decay)[c], uu = ar.f(L.faaaa)[c], lnw = ar.f(L.lnx_w)[c], lnb = ar.f(L.lnx_b)[c];
                float s[S];
                const float * st = sin_l + 2 * D + (long long) d_head * S * S;
#pragma unroll
                for (int i = 0; i < S; i++) s[i] = st[i * S + ln];
                __builtin_amdgcn_sched_barrier(0);
                unsigned dq[6];
                watch_begin(l);
                {
                    const int ptr[2] = {p.dl + ln, p.dl + (NBD > 2 ? 64 + ln : ln)};
                    const bool valid[2] = {true, NBD > 2};
                    v4u dv[2];
                    poll_ptrs<2>(pl, xr, ptr, valid, tagL + SLOT_RKVG, dv);
                    dq[4] = dv[0].x; dq[5] = dv[1].x;
                }
                v4u early[4];
                {
                    asm volatile("" ::: "memory");
                    early[0] = tg_load(xr, p.rkvg + (c >> 1)); early[1] = tg_load(xr, p.rkvg + ((D + c) >> 1));
                    early[2] = tg_load(xr, p.rkvg + ((2 * D + c) >> 1)); early[3] = tg_load(xr, p.rkvg + ((3 * D + c) >> 1));
                }
                // 1. quantise dl (DR = 32 NBD elements) into LDS: half-wave = block
                const QVec ldl = qvec_at(l.dl, NBD * 32);
#pragma unroll
                for (int j = 0; j < (NBD * 32 + 63) / 64; j++) {
                    const int e = j * 64 + ln;
                    const float val = e < NBD * 32 ? __uint_as_float(dq[4 + j]) : 0.0f;
                    int qi, isum; float d16, s16;
                    quant_block32(val, qi, d16, s16, isum);
                    if (e < NBD * 32) qvec_store(ldl, NBD, e >> 5, e & 31, qi, d16, s16, isum);
                }
                __builtin_amdgcn_wave_barrier();
                // 2. decay row of channel c
                float P[NBD];
#pragma unroll
                for (int b = 0; b < NBD; b++) {
                    WBlk<FMT> w;
                    unpack_raw<FMT>(w, w2[b]);
                    const int4 alo = *reinterpret_cast<const int4 *>(ldl.q + b * 16);
                    const int4 ahi = *reinterpret_cast<const int4 *>(ldl.q + NBD * 16 + b * 16);
                    P[b] = blk_fma<FMT>(w, alo, ahi, ldl.d[b], ldl.s[b], ldl.isum[b], 0.0f);
                }
#pragma unroll
                for (int o = NBD / 2; o > 0; o >>= 1)
#pragma unroll
                    for (int i = 0; i < o; i++) P[i] += P[i + o];
                const float wdec = det_expf(-det_expf(P[0] + td));
                // r, k, v, g of channel c: one unit per 2-row set (first read issued above, before the decay was computed)
                {
                    const int ptr[4] = {p.rkvg + (c >> 1), p.rkvg + ((D + c) >> 1), p.rkvg + ((2 * D + c) >> 1), p.rkvg + ((3 * D + c) >> 1)};
                    const bool valid[4] = {true, true, true, true};
                    v4u dv[4] = {early[0], early[1], early[2], early[3]};
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q < 4; q++) ok = ok && tg_ok(dv[q], tagL + SLOT_RKVG);
                    if (!__all(ok)) poll_ptrs<4>(pl, xr, ptr, valid, tagL + SLOT_RKVG, dv);
#pragma unroll
                    for (int q = 0; q < 4; q++) dq[q] = (c & 1) ? dv[q].y : dv[q].x;
                }
                watch_end(l);
                // 3. WKV6 (ggml_rwkv_wkv6): ln j owns value column j; {k, u, r, w}_i are broadcast through LDS
                float4 * bc = reinterpret_cast<float4 *>(l.bc);
                bc[ln] = make_float4(__uint_as_float(dq[1]), uu, __uint_as_float(dq[0]), wdec);
                __builtin_amdgcn_wave_barrier();
                const float vj = __uint_as_float(dq[2]);
                float o = 0.0f;
                float * so = sout_l + 2 * D + (long long) d_head * S * S;
#pragma unroll
                for (int i0 = 0; i0 < S; i0 += 8) {
                    float4 b8[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) b8[u] = bc[i0 + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int i = i0 + u;
                        const float kv = vj * b8[u].x;
                        const float prev = s[i];
                        const float temp = kv * b8[u].y + prev;
                        o += temp * b8[u].z;
                        so[i * S + ln] = prev * b8[u].w + kv;
                    }
                }
                // 4. GroupNorm over the head, * ln_x, gate
                const float mean = (float) (wave_sum_d((double) o) / (double) S);
                const float dv = o - mean;
                const float var = (float) (wave_sum_d((double) (dv * dv)) / (double) S);
                const float scale = 1.0f / sqrtf(var + 64e-5f);
                float y = dv * scale;
                y = y * lnw;
                y = y + lnb;
                y *= __uint_as_float(dq[3]);
                int qi, isum; float d16, s16;
                quant_block32(y, qi, d16, s16, isum);
                tq_store_block(xr, p.yq, 2 * d_head + (ln >> 5), ln & 31, qi, d16, s16, isum, tagL + SLOT_YQ);
            }
            R6STAMP(6); R6RSTAMP(21);
            // ---- E ----
            watch_begin(l);
            gather_hint(pl, xr, p.yq + ((blk * 7 + g * 19) & 127), tagL + SLOT_YQ, l.fl + FL_HYQ, g1, p.nap);
            watch_end(l);
            sweep_begin(l);
            gather_qvec<DSL>(pl, xr, p.yq, D, tagL + SLOT_YQ, g, opq(lane), l.yq);
            gather_meet(pl, l.fl + FL_GYQ, g1);
            sweep_end(l);
            R6STAMP(7); R6RSTAMP(22);
            // ---- F ----
            // the key sets of this workgroup that do not travel through the ring (the last KCOMM of its KSETS: ring_geom.h): this wave is idle
            // through the key rows; their blocks go in flight now, from the planes, like the decay row's
            RawRec<FMT, 2, UD> kxr[KCOMM > 0 ? KCOMM : 1];
            const bool k_has = KCOMM > 0 && blk * gpb * 32 < nbF * 32;
            if constexpr (KCOMM > 0) {
                const WPl fk = ar.w(L.fk);
                const int lnK = opq(lane);
#pragma unroll
                for (int q = 0; q < KCOMM; q++)
#pragma unroll
                    for (int u = 0; u < UD; u++)
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            const long long row = k_has ? (long long) blk * gpb * 32 + 2 * (KSETS - KCOMM + q) + r : 0;
                            const int bb = u * 64 + lnK;
                            RawBlk<FMT> rb; rb.qh = 0u;
                            load_raw<FMT>(rb, fk.qs, fk.qh, fk.sc, row * nb + (bb < nb ? bb : nb - 1));
                            from_raw<FMT>(kxr[q].raw[u][r], rb);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
            watch_begin(l);
            gather_hint(pl, xr, p.xatt + ((blk * 37 + g * 211) & 1023), tagL + SLOT_XATT, l.fl + FL_HX, 2u * li + 2u, p.nap);
            watch_end(l);
            sweep_begin(l);
            gather_x(pl, xr, p.xatt, tagL + SLOT_XATT, g, opq(lane), l.x);
            gather_meet(pl, l.fl + FL_GX, 2u * li + 2u);
            sweep_end(l);
            R6STAMP(8); R6RSTAMP(23);
            if constexpr (KCOMM > 0) {
                if (k_has) {
                    prologue_wait(pl, l, 2u * li + 2u);              // l.q1 holds the quantised key input
                    const int lnK = opq(lane);
                    ActRegs<UD> ark;
                    act_load<FMT, UD>(ark, qvec_at(l.q1, D), nb, lnK);
                    float part[2 * KCOMM];
#pragma unroll
                    for (int q = 0; q < KCOMM; q++) rec_acc<FMT, 2, UD>(kxr[q], ark, nb, lnK, part + 2 * q);
                    wave_sum_n<2 * KCOMM>(part);
                    const float v = pick_lane<2 * KCOMM>(part, lnK);
                    const float t = v > 0.0f ? v : 0.0f;
                    if (lnK < 2 * KCOMM) l.out[2 * (KSETS - KCOMM) + lnK] = t * t;
                    __builtin_amdgcn_wave_barrier();
                }
            }
            fl_wait(pl, l.fl + FL_KEYS, (unsigned) NC * g1);     // every consumer's key sets are in l.out
            R6STAMP(9);
            {
                // quantise this workgroup's key groups (relu^2 outputs): half-wave = group
                const int ln = opq(lane);
                for (int g2 = 0; g2 < (gpb + 1) / 2; g2++) {
                    const int gi = g2 * 2 + (ln >> 5);
                    const int gg = blk * gpb + gi;
                    const bool valid = gi < gpb && gg < nbF;
                    const float v = valid ? l.out[gi * 32 + (ln & 31)] : 0.0f;
                    int qi, isum; float d16, s16;
                    quant_block32(v, qi, d16, s16, isum);
                    tq_store_block(xr, p.kq, valid ? gg : 0, ln & 31, qi, d16, s16, isum, tagL + SLOT_KQ, valid);
                }
            }
            R6STAMP(10); R6RSTAMP(24);
            // ---- G ----
            watch_begin(l);
            gather_hint(pl, xr, p.kq + ((blk * 5 + g * 173) & 511), tagL + SLOT_KQ, l.fl + FL_HKQ, g1, p.nap);
            watch_end(l);
            sweep_begin(l);
            gather_qvec<KSL>(pl, xr, p.kq, F, tagL + SLOT_KQ, g, opq(lane), l.kq);
            gather_meet(pl, l.fl + FL_GKQ, g1);
            sweep_end(l);
            R6STAMP(11); R6RSTAMP(25);
        }
        if (p.logits) {   // the head: this wave's share of the last x hand-over (the consumers do the rest)
            const int li = p.n_layers;
            const unsigned tagL = base + (unsigned) li * 8u;
            watch_begin(l);
            gather_hint(pl, xr, p.xffn + ((blk * 37 + g * 211) & 1023), tagL - 8u + SLOT_XFFN, l.fl + FL_HX, 2u * li + 1u, p.nap);
            watch_end(l);
            sweep_begin(l);
            gather_x(pl, xr, p.xffn, tagL - 8u + SLOT_XFFN, g, opq(lane), l.x);
            gather_meet(pl, l.fl + FL_GX, 2u * li + 1u);
            sweep_end(l);
        }
        tail_argmax(p, l, pl, xr, opq(lane), base);
    }

    // argmax of the logits inside the launch (k_argmax's rule; persist_v47.hip's tail): the six consumer waves leave their candidates in
    // l.bc, this wave merges them and publishes the workgroup's as ONE unit under slot 7 of the last layer's tags; workgroup 0 gathers the
    // 256 units and writes the token where the next launch's embedding reads it (and into the greedy loops' history: ctl[2..6]).
    // EVERY launch publishes its units (without logits: an empty candidate), so the stale content of the buffer is always the previous
    // launch's -- a launch 65536 / (8 layers) tokens back would carry the same 16-bit tag.
    static __device__ __forceinline__ void tail_argmax(const R6P & p, const Lds & l, Poll & pl, xrsrc xr, int lane, unsigned base) {
        const int blk = blockIdx.x;
        const unsigned tagA = base + (unsigned) (p.n_layers - 1) * 8u + 7u;
        float b = -INFINITY; int bi = 0x7fffffff;
        if (p.logits) {
            fl_wait(pl, l.fl + FL_AM, (unsigned) NC);
            if (lane < NC) { b = l.bc[2 * lane]; bi = __float_as_int(l.bc[2 * lane + 1]); }
            am_wave(b, bi);
        }
        if (lane == 0) tg_store(xr, p.am + blk, __float_as_uint(b), (unsigned) bi, 0u, 0u, tagA);
        if (!p.logits || !p.next_tok || blk != 0) return;
        int ptr[NBLK / 64]; bool valid[NBLK / 64]; v4u dv[NBLK / 64];
#pragma unroll
        for (int k = 0; k < NBLK / 64; k++) { ptr[k] = p.am + lane + 64 * k; valid[k] = true; }
        poll_ptrs<NBLK / 64>(pl, xr, ptr, valid, tagA, dv);
        float b3 = -INFINITY; int i3 = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < NBLK / 64; k++) am_merge(b3, i3, __uint_as_float(dv[k].x), (int) dv[k].y);
        am_wave(b3, i3);
        if (lane == 0) {
            // (no element compared greater than -inf: every logit is NaN or -inf -- the token feeds the next embedding lookup and must stay a row of the table)
            const unsigned tokn = i3 == 0x7fffffff ? 0u : (unsigned) i3;
            p.next_tok[0] = tokn;
            if (p.ctl[2] != 0u) {   // greedy loops: history pointer ctl[4..5], position ctl[3], capacity ctl[6]
                unsigned * hist = reinterpret_cast<unsigned *>((unsigned long long) p.ctl[4] | ((unsigned long long) p.ctl[5] << 32));
                const unsigned pos = p.ctl[3];
                if (pos < p.ctl[6]) hist[pos] = tokn;
                p.ctl[3] = pos + 1u;
            }
        }
    }
};

template <int FMT, int EPT, int NBD, int UF, int KSL>
__global__ __launch_bounds__(512) void k6_ring(R6P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef R6<FMT, EPT, NBD, UF, KSL> K;
    const int tid0 = threadIdx.x;
    const int lane = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const R6Lds lo = r6_lds(K::D, p.F);
    typename K::Lds l;
    l.x = reinterpret_cast<float *>(smem + lo.x); l.q1 = smem + lo.q1; l.q2 = smem + lo.q2;
    const size_t qd = (qvec_bytes(K::D) + 15) / 16 * 16;
    l.act = smem + lo.u; l.actw = smem + lo.u + qd; l.yq = smem + lo.u + 2 * qd; l.kq = smem + lo.u;
    l.tl = reinterpret_cast<float *>(smem + lo.tl); l.bc = reinterpret_cast<float *>(smem + lo.bc); l.red = reinterpret_cast<double *>(smem + lo.red);
    l.out = reinterpret_cast<float *>(smem + lo.out); l.dl = smem + lo.dl; l.misc = reinterpret_cast<float *>(smem + lo.misc);
    l.fl = reinterpret_cast<unsigned *>(smem + lo.fl); l.ring = smem + lo.ring;
    if (tid0 < FL_WORDS) l.fl[tid0] = (tid0 >= FL_DONE + 2 && tid0 < FL_DONE + 2 + RG_NC) ? 0u : (tid0 >= FL_DONE && tid0 < FL_DONE + 8 ? 0xFFFFFFFFu : 0u);
    __syncthreads();   // the only workgroup barrier of the launch
    const unsigned base = p.ctl[0];
#ifndef R6_ROLES
#define R6_ROLES 7
#endif
    if (p.dbg & 8) {   // timing experiment: the loader alone (every other wave releases the whole ring and leaves)
        if (wave == 0) K::loader_main(p, l, lane);
        else fl_st(l.fl + FL_DONE + wave, 0xFFFFFFFFu);
        return;
    }
    if (p.trace && lane == 0) p.trace[((long long) blockIdx.x * 8 + wave) * 32 + 31] = (long long) __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11));   // HW_ID[15:0]: wave, SIMD, pipe, CU, SH, SE
    if (wave == 0) { if (R6_ROLES & 1) K::loader_main(p, l, lane); }
    else if (wave == 1) { if (R6_ROLES & 2) K::comm_main(p, l, lane, base); }
    else { if (R6_ROLES & 4) K::consumer_main(p, l, lane, wave, base); }
    if (blockIdx.x == 0 && tid0 == 64) p.ctl[0] = base + (unsigned) p.n_layers * 8u;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

// One workgroup of 64 threads per record: copies the record's rows out of the planes (DevTensor layout) into the stream.
struct PackMat { M6Off w1, dw1, rkvg[4], wo, fk, fr, fv; };
__global__ void k_ring_pack(const unsigned char * __restrict__ arena, PackMat pm, RingShape s, const R6Cu * __restrict__ cus, unsigned char * __restrict__ stream, int layer, int max_rec) {
    const int b = blockIdx.y, q = blockIdx.x, lane = threadIdx.x;
    const RingCu cu = rg_cu(s, b);
    int ph = 0, j = q;
    while (ph < RG_NPHASE && j >= (int) cu.n[ph]) { j -= (int) cu.n[ph]; ph++; }
    if (ph >= RG_NPHASE) return;
    const RingRec rr = rg_rec(s, b, ph, j);
    M6Off mo;
    int64_t nrows;
    switch (ph) {
        case RG_W1:  mo = pm.w1; nrows = s.R5; break;
        case RG_DW1: mo = pm.dw1; nrows = s.DR; break;
        case RG_C:   mo = pm.rkvg[rr.mat]; nrows = s.D; break;
        case RG_E:   mo = pm.wo; nrows = s.D; break;
        case RG_FK:  mo = pm.fk; nrows = s.F; break;
        case RG_FR:  mo = pm.fr; nrows = s.D; break;
        default:     mo = pm.fv; nrows = s.D; break;
    }
    const int nbk = rr.K / 32, U = rg_steps(rr.K);
    unsigned char * dst = stream + cus[b].base + (size_t) layer * cu.layer_bytes + cu.off[ph] + (size_t) j * cu.rec[ph];
    const unsigned char * qs = arena + mo.qs; const unsigned char * qh = arena + mo.qh; const unsigned char * sc = arena + mo.sc;
    for (int u = 0; u < U; u++) {
        for (int r = 0; r < rr.R; r++) {
            const int blkk = u * 64 + lane;
            const int64_t row = rr.row0 + r;
            const bool valid = blkk < nbk && row < nrows;
            const int64_t gb = row * nbk + blkk;
            for (int h = 0; h < s.qs / 16; h++) {
                int4 v = make_int4(0, 0, 0, 0);
                if (valid) v = *reinterpret_cast<const int4 *>(qs + gb * s.qs + h * 16);
                *reinterpret_cast<int4 *>(dst + rg_code_off(s, rr.R, u, r, h) + lane * 16) = v;
            }
            if (s.scb == 4) *reinterpret_cast<uint32_t *>(dst + rg_sc_off(s, rr.R, U, u, r) + lane * 4) = valid ? reinterpret_cast<const uint32_t *>(sc)[gb] : 0u;
            else *reinterpret_cast<uint16_t *>(dst + rg_sc_off(s, rr.R, U, u, r) + lane * 2) = valid ? reinterpret_cast<const uint16_t *>(sc)[gb] : (uint16_t) 0;
            if (s.qhb) *reinterpret_cast<uint32_t *>(dst + rg_qh_off(s, rr.R, U, u, r) + lane * 4) = valid ? reinterpret_cast<const uint32_t *>(qh)[gb] : 0u;
        }
    }
}

// the head's rows in the record layout of ring_geom.h (RingHead): one workgroup of 64 threads per record
__global__ void k_ring_pack_head(const unsigned short * __restrict__ head, int n_vocab, int K, const R6Cu * __restrict__ cus, unsigned char * __restrict__ stream, int n_layers) {
    const int b = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
    const RingHead hd = rg_head(n_vocab, K);
    // r -> (pass, chunk, consumer): records of a pass are chunk-major over its consumers
    int ps = 0, rr = r;
    while (ps < hd.passes && rr >= rg_head_npc(hd, ps) * hd.chunks) { rr -= rg_head_npc(hd, ps) * hd.chunks; ps++; }
    if (ps >= hd.passes) return;
    const int npc = rg_head_npc(hd, ps), k = rr / npc, c = rr % npc;
    const int g = RG_NC * ps + c;
    const long long row = (long long) b * hd.hg * 16 + g * 16 + (lane >> 2);
    const int q = lane & 3;
    unsigned char * dst = stream + cus[b].base + (size_t) n_layers * cus[b].layer_bytes + rg_head_off(hd, ps, k, c);
    for (int st = 0; st < RG_HSTEPS; st++) {
        const int col = 32 * (k * RG_HSTEPS + st) + 8 * q;
        const int4 v = *reinterpret_cast<const int4 *>(head + row * K + col);
        *reinterpret_cast<int4 *>(dst + st * 1024 + lane * 16) = v;
    }
}

__global__ void k_block_w2_ring(const float * __restrict__ src, float * __restrict__ dst, int D, int R) {   // (layout: see k_block_w2, mega_v6.hip)
    const long long n = 5ll * R * D;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const int d = (int) (i % D); const long long fm = i / D; const int m = (int) (fm % R), f = (int) (fm / R);
        const long long o = ((((long long) f * (D / 64) + d / 64) * (R / 4) + m / 4) * 64 + d % 64) * 4 + m % 4;
        dst[o] = src[i];
    }
}

// What a context's ring kernel derives from the WEIGHTS alone -- the per-workgroup streams (3.9 GB + 0.5 GB of head for the 7B Q4_0),
// the chunk-blocked W2, the layer table -- is built once per Model and shared by every context of it (rwkv_clone_context, the decode
// streams of a pipeline stage: the reference's clones share their weights too, rwkv.cpp:109-143). Round 3 built it per context:
// N decode streams cost N x the model in HBM and re-ran k_ring_pack for all layers. Refcounted by the handles; the Model keeps the
// pointer (Model::ring_shared, under Model::derived_mu) only while a handle holds it.
struct RingShared {
    int ref = 0;
    float * w2b = nullptr;
    M6Layer * d_layers = nullptr;
    R6Cu * d_cus = nullptr;
    unsigned char * stream = nullptr;
    int variant = -1, n_blocks = 0, n_layers = 0;
    size_t lds = 0, ring = 0, mirror = 0;
    uint64_t bytes = 0;
    bool head = false;            // the head projection is folded into the launch (F16 head.weight, vocabulary a multiple of 4096)
    uint64_t bytes_head = 0;      // its algorithmic bytes: head.weight, ln_out, the logits written
    bool embed = false;           // embedding + ln0 are folded into the launch (F16 / F32 emb.weight of a stage that owns it)
    uint64_t bytes_embed = 0;     // one row of emb.weight + ln0
    long long lnw_off = 0, lnb_off = 0, ln0w_off = 0, ln0b_off = 0;
    int64_t D = 0, F = 0, DR = 0, R = 0;
};

struct RingV6 {
    int kind = 2;                 // (first member: mega_v6.hip's entry points dispatch on it)
    const Model * model = nullptr;
    RingShared * sh = nullptr;
    void * xch = nullptr;
    unsigned * ctl = nullptr;
    unsigned * h_ctl = nullptr;
    R6P proto{};
    long long * trace = nullptr;
    float * x_out = nullptr;      // ring_v6_set_x_out: the last layer's x goes there instead of back into the launch's x (pipeline stages, runner.cpp)
};

typedef void (*RingKernel)(R6P);
struct RingVariant { int fmt, ept, nbd, uf, ksl; RingKernel fn; };
static const RingVariant g_ring_variants[] = {
#define RING_VARIANTS(FMT) \
    {FMT, 8, 4, 7, 3, k6_ring<FMT, 8, 4, 7, 3>},   /* D 4096, F 14336 (448 blocks: 7 steps, 1344 units), decay rank 128: RWKV-6 7B */ \
    {FMT, 5, 2, 5, 2, k6_ring<FMT, 5, 2, 5, 2>},   /* D 2560, F 8960 (280 blocks: 5 steps, 840 units), decay rank 64: RWKV-6 3B */ \
    {FMT, 4, 2, 4, 2, k6_ring<FMT, 4, 2, 4, 2>}    /* D 2048, F 7168 (224 blocks: 4 steps, 672 units), decay rank 64: RWKV-6 1.6B */
#ifdef R6_ONLY_FMT   /* (register-budget experiments: one format, the 7B geometry) */
    {R6_ONLY_FMT, 8, 4, 7, 3, k6_ring<R6_ONLY_FMT, 8, 4, 7, 3>},
#else
    RING_VARIANTS(T_Q4_0), RING_VARIANTS(T_Q4_1), RING_VARIANTS(T_Q5_0), RING_VARIANTS(T_Q5_1), RING_VARIANTS(T_Q8_0),
#endif
};

static int ring_variant(const Model & m, int n_cu) {
    if (m.arch_major != 6 || m.head_size != 64 || m.layer_end <= m.layer_begin) return -1;
    const int64_t D = m.n_embed(), H = m.head_count;
    const int fmt = (int) m.header.data_type;
    const LayerW & L0 = m.layers[m.layer_begin];
    if (!L0.ffn_key || !L0.att_time_decay_w1 || !L0.att_time_maa_w1) return -1;
    const int64_t F = L0.ffn_key->ne[1], DR = L0.att_time_decay_w1->ne[1], R5 = L0.att_time_maa_w1->ne[1], R = R5 / 5;
    const int64_t NB = RG_NBLK;
    const int64_t gpb = (F / 32 + NB - 1) / NB;
    if (n_cu != NB || H > NB || F % 32 != 0 || F % (gpb * 32) != 0 || DR > NB || DR % 32 != 0 || 5 * (D / 64) > NB + NB / 2 || !(R == 32 || R == 64) || gpb * 32 > 64) return -1;
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        const DevTensor * mats[] = {L.att_receptance, L.att_key, L.att_value, L.att_gate, L.att_output, L.att_time_maa_w1,
                                    L.att_time_decay_w1, L.att_time_decay_w2, L.ffn_key, L.ffn_value, L.ffn_receptance};
        for (const DevTensor * t : mats) if (!t || t->type != fmt) return -1;
        if (L.ffn_key->ne[1] != F || L.att_time_decay_w1->ne[1] != DR || L.att_time_maa_w1->ne[1] != R5) return -1;
    }
    for (size_t v = 0; v < sizeof(g_ring_variants) / sizeof(g_ring_variants[0]); v++) {
        const RingVariant & rv = g_ring_variants[v];
        const int64_t nbF = F / 32;
        if (rv.fmt == fmt && D == rv.ept * 512 && DR == rv.nbd * 32 && (nbF + 63) / 64 == rv.uf && (3 * nbF + 7 * 64 - 1) / (7 * 64) == rv.ksl) return (int) v;
    }
    return -1;
}

static void ring_shared_free(RingShared * sh) {
    if (!sh) return;
    if (sh->d_layers) (void) hipFree(sh->d_layers);
    if (sh->d_cus) (void) hipFree(sh->d_cus);
    if (sh->w2b) (void) hipFree(sh->w2b);
    if (sh->stream) (void) hipFree(sh->stream);
    delete sh;
}

static int env_int(const char * name, int dflt) { const char * e = getenv(name); return e && e[0] ? atoi(e) : dflt; }

// builds the shared images of model m on the current device (nullptr: the model / device does not qualify, or out of memory -- said on stderr)
static RingShared * ring_shared_build(const Model & m) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m.device) != hipSuccess) return nullptr;
    const int NB = prop.multiProcessorCount;
    const int v = ring_variant(m, NB);
    if (v < 0) return nullptr;
    const LayerW & L0 = m.layers[m.layer_begin];
    const int64_t D = m.n_embed(), F = L0.ffn_key->ne[1], DR = L0.att_time_decay_w1->ne[1], R = L0.att_time_maa_w1->ne[1] / 5;
    const int fmt = (int) m.header.data_type;
    RingShape sh; sh.D = (int) D; sh.F = (int) F; sh.R5 = (int) (5 * R); sh.DR = (int) DR;
    sh.qs = fmt == T_Q8_0 ? 32 : 16; sh.scb = (fmt == T_Q4_1 || fmt == T_Q5_1) ? 4 : 2; sh.qhb = (fmt == T_Q5_0 || fmt == T_Q5_1) ? 4 : 0;
    const int n_layers = (int) (m.layer_end - m.layer_begin);
    RingShared * rs = new RingShared();
    rs->variant = v; rs->n_blocks = NB; rs->n_layers = n_layers;
    rs->D = D; rs->F = F; rs->DR = DR; rs->R = R;
    const R6Lds lo = r6_lds((int) D, (int) F);
    const size_t lds_max = 160 * 1024;
    // the ring is filled in 4-KiB groups of four DMA instructions; its head is repeated behind its end for the longest record (rec_load)
    const size_t max_rec_bytes = rg_rec_bytes(sh, 1, (int) F) > rg_rec_bytes(sh, 2, (int) D) ? rg_rec_bytes(sh, 1, (int) F) : rg_rec_bytes(sh, 2, (int) D);
    const size_t mirror = ((max_rec_bytes > RG_HREC ? max_rec_bytes : RG_HREC) + 4095) / 4096 * 4096;
    size_t ring = (size_t) env_int("RWKV_MI_RING_KB", 1024) * 1024;
    if (lo.fixed + mirror + 32 * 1024 > lds_max) { delete rs; return nullptr; }
    if (ring > lds_max - lo.fixed - mirror) ring = lds_max - lo.fixed - mirror;
    ring = ring / 4096 * 4096;
    if (ring < 32 * 1024) ring = 32 * 1024;
    rs->lds = lo.fixed + ring + mirror; rs->ring = ring; rs->mirror = mirror;
    if (hipFuncSetAttribute((const void *) g_ring_variants[v].fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) rs->lds) != hipSuccess) { delete rs; return nullptr; }
    // the head behind the last layer: F16 head.weight of a stage that owns it, rows divisible into 16-row groups per workgroup
    const bool fold_head = m.has_head && m.head && m.head->type == T_F16 && m.head->cols() == D && m.n_vocab() % (RG_NBLK * 16) == 0 && m.ln_out_w && m.ln_out_b
                           && !(getenv("RWKV_MI_RING_NO_HEAD") && getenv("RWKV_MI_RING_NO_HEAD")[0] == '1');
    rs->head = fold_head;
    // per-workgroup streams
    std::vector<R6Cu> hc(RG_NBLK);
    uint64_t total = 0;
    int max_rec = 0;
    for (int b = 0; b < RG_NBLK; b++) {
        const RingCu cu = rg_cu(sh, b);
        int nr = 0; for (int ph = 0; ph < RG_NPHASE; ph++) nr += (int) cu.n[ph];
        max_rec = nr > max_rec ? nr : max_rec;
        const uint64_t bytes = (uint64_t) cu.layer_bytes * n_layers;
        if (bytes + (1u << 20) > 0xFFFFFFFFull) { delete rs; return nullptr; }   // stream positions are 32-bit
        const uint64_t hbytes = fold_head ? rg_head((int) m.n_vocab(), (int) D).bytes : 0;
        if (bytes + hbytes + (1u << 20) > 0xFFFFFFFFull) { delete rs; return nullptr; }
        hc[b].base = total; hc[b].chunks = (unsigned) ((bytes + 4 * RG_CHUNK - 1) / (4 * RG_CHUNK)) * 4u;
        hc[b].chunks_head = (unsigned) ((bytes + hbytes + 4 * RG_CHUNK - 1) / (4 * RG_CHUNK)) * 4u; hc[b].layer_bytes = cu.layer_bytes; hc[b].pad = 0;
        total += (uint64_t) hc[b].chunks_head * RG_CHUNK;
    }
    const size_t w2_layer = (size_t) 5 * R * D;
    bool ok = R % 4 == 0 && hipMalloc((void **) &rs->w2b, w2_layer * n_layers * sizeof(float)) == hipSuccess
           && hipMalloc((void **) &rs->stream, total + 4 * RG_CHUNK) == hipSuccess
           && hipMalloc((void **) &rs->d_cus, hc.size() * sizeof(R6Cu)) == hipSuccess
           && hipMemcpy(rs->d_cus, hc.data(), hc.size() * sizeof(R6Cu), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        (void) hipGetLastError();
        fprintf(stderr, "librwkv: no room for the ring kernel's weight streams (%.2f GB): this model continues on the per-layer launches\n", (double) total / 1e9);
        ring_shared_free(rs);
        return nullptr;
    }
    std::vector<M6Layer> hl;
    const unsigned char * abase = (const unsigned char *) m.arena;
    bool in_arena = true;
    auto off = [&](const void * ptr) -> long long {
        const long long o = (const unsigned char *) ptr - abase;
        if (!ptr || o < 0 || (uint64_t) o >= m.arena_bytes) in_arena = false;
        return o;
    };
    auto f = [&](const DevTensor * t) { return off(t->data); };
    auto pl3 = [&](const DevTensor * t) { M6Off o; o.qs = off(t->qs); o.qh = t->qh ? off(t->qh) : 0; o.sc = off(t->sc); return o; };
    uint64_t bytes = 0;
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        M6Layer d{};
        d.ln1_w = f(L.ln1_w); d.ln1_b = f(L.ln1_b); d.maa_x = f(L.att_time_maa_x);
        d.maa[0] = f(L.att_time_maa_w); d.maa[1] = f(L.att_time_maa_k); d.maa[2] = f(L.att_time_maa_v); d.maa[3] = f(L.att_time_maa_r); d.maa[4] = f(L.att_time_maa_g);
        d.w2b = (long long) hl.size() * 5 * R * D; d.time_decay = f(L.att_time_decay); d.faaaa = f(L.att_time_faaaa);
        d.lnx_w = f(L.att_ln_x_w); d.lnx_b = f(L.att_ln_x_b); d.ln2_w = f(L.ln2_w); d.ln2_b = f(L.ln2_b);
        d.fmaa_k = f(L.ffn_time_maa_k); d.fmaa_r = f(L.ffn_time_maa_r);
        d.w1 = pl3(L.att_time_maa_w1);
        d.rkvg[0] = pl3(L.att_receptance); d.rkvg[1] = pl3(L.att_key); d.rkvg[2] = pl3(L.att_value); d.rkvg[3] = pl3(L.att_gate);
        d.dw1 = pl3(L.att_time_decay_w1); d.dw2 = pl3(L.att_time_decay_w2); d.wo = pl3(L.att_output);
        d.fk = pl3(L.ffn_key); d.fr = pl3(L.ffn_receptance); d.fv = pl3(L.ffn_value);
        if (!in_arena) break;
        hipLaunchKernelGGL(k_block_w2_ring, dim3(512), dim3(256), 0, 0, (const float *) L.att_time_maa_w2->data, rs->w2b + hl.size() * w2_layer, (int) D, (int) R);
        PackMat pm; pm.w1 = d.w1; pm.dw1 = d.dw1; for (int q = 0; q < 4; q++) pm.rkvg[q] = d.rkvg[q]; pm.wo = d.wo; pm.fk = d.fk; pm.fr = d.fr; pm.fv = d.fv;
        hipLaunchKernelGGL(k_ring_pack, dim3((unsigned) max_rec, RG_NBLK), dim3(64), 0, 0, abase, pm, sh, rs->d_cus, rs->stream, (int) hl.size(), max_rec);
        hl.push_back(d);
        const DevTensor * all[] = {L.ln1_w, L.ln1_b, L.att_time_maa_x, L.att_time_maa_w, L.att_time_maa_k, L.att_time_maa_v, L.att_time_maa_r, L.att_time_maa_g,
                                   L.att_time_maa_w1, L.att_time_maa_w2, L.att_time_decay, L.att_time_faaaa, L.att_time_decay_w1, L.att_time_decay_w2,
                                   L.att_receptance, L.att_key, L.att_value, L.att_gate, L.att_output, L.att_ln_x_w, L.att_ln_x_b, L.ln2_w, L.ln2_b,
                                   L.ffn_time_maa_k, L.ffn_time_maa_r, L.ffn_key, L.ffn_value, L.ffn_receptance};
        for (const DevTensor * t : all) if (t) bytes += t->nbytes;
        bytes += 2 * (uint64_t) m.state_per_layer() * sizeof(float);
    }
    rs->bytes = bytes;
    {
        const char * nf = getenv("RWKV_MI_RING_NO_EMBED");   // (measurement aid: the embedding and the argmax as their own launches, as up to round 5)
        if (!(nf && nf[0] == '1') && m.has_embed && m.emb && m.ln0_w && m.ln0_b && (m.emb->type == T_F16 || m.emb->type == T_F32) && in_arena) {
            rs->embed = true;
            rs->ln0w_off = off(m.ln0_w->data); rs->ln0b_off = off(m.ln0_b->data);
            rs->bytes_embed = (uint64_t) D * (m.emb->type == T_F16 ? 2 : 4) + m.ln0_w->nbytes + m.ln0_b->nbytes;
            if (!in_arena) rs->embed = false;
            in_arena = true;   // (ln0 outside the arena only switches the fold off)
        }
    }
    if (fold_head && in_arena) {
        rs->lnw_off = off(m.ln_out_w->data); rs->lnb_off = off(m.ln_out_b->data);
        const RingHead hd = rg_head((int) m.n_vocab(), (int) D);
        hipLaunchKernelGGL(k_ring_pack_head, dim3((unsigned) (hd.hg * hd.chunks), RG_NBLK), dim3(64), 0, 0, (const unsigned short *) m.head->data, (int) m.n_vocab(), (int) D,
                           rs->d_cus, rs->stream, n_layers);
        rs->bytes_head = m.head->nbytes + m.ln_out_w->nbytes + m.ln_out_b->nbytes + (uint64_t) m.n_vocab() * 4;
    }
    ok = in_arena && hipMalloc((void **) &rs->d_layers, hl.size() * sizeof(M6Layer)) == hipSuccess
      && hipMemcpy(rs->d_layers, hl.data(), hl.size() * sizeof(M6Layer), hipMemcpyHostToDevice) == hipSuccess
      && hipDeviceSynchronize() == hipSuccess;
    if (!ok) { ring_shared_free(rs); return nullptr; }
    return rs;
}

static RingShared * ring_shared_acquire(const Model & m) {
    std::lock_guard<std::mutex> lk(m.derived_mu);
    RingShared * rs = (RingShared *) m.ring_shared;
    if (!rs) { rs = ring_shared_build(m); m.ring_shared = rs; }
    if (rs) rs->ref++;
    return rs;
}
static void ring_shared_release(const Model & m, RingShared * rs) {
    if (!rs) return;
    std::lock_guard<std::mutex> lk(m.derived_mu);
    if (--rs->ref > 0) return;
    if (m.ring_shared == rs) m.ring_shared = nullptr;
    ring_shared_free(rs);
}

void ring_v6_destroy(void * h) {
    RingV6 * rg = (RingV6 *) h;
    if (!rg) return;
    if (rg->xch) (void) hipFree(rg->xch);
    if (rg->ctl) (void) hipFree(rg->ctl);
    if (rg->h_ctl) (void) hipHostFree(rg->h_ctl);
    if (rg->trace) (void) hipFree(rg->trace);
    if (rg->model) ring_shared_release(*rg->model, rg->sh);
    delete rg;
}

// Returns nullptr when the model / device does not qualify (the caller tries the register-prefetch kernel, then the seven launches).
// Per context: the exchange arena, the control words, the trace buffer. Everything derived from the weights is shared (RingShared).
void * ring_v6_create(const Model & m) {
    RingShared * rs = ring_shared_acquire(m);
    if (!rs) return nullptr;
    RingV6 * rg = new RingV6();
    rg->model = &m; rg->sh = rs;
    const int64_t D = rs->D, F = rs->F;
    const int NB = rs->n_blocks;
    const int64_t nbD = D / 32, nbF = F / 32;
    const int64_t PAD = 2048;   // polls read whole rounds of 7 x 64 lanes: keep every buffer readable past its end
    auto up = [](int64_t v) { return (v + 63) / 64 * 64; };
    const int64_t act_stride = up(3 * nbD), xunits = up(RG_NBLK * RG_NC);
    const int64_t sizes[9] = {up(1280) + PAD, 5 * act_stride + PAD, 2 * D + PAD, 256 + PAD, act_stride + PAD, xunits + PAD, up(3 * nbF) + PAD, xunits + PAD, up(RG_NBLK) + PAD};
    int64_t units = 0;
    for (int64_t z : sizes) units += z;
    bool ok = hipMalloc(&rg->xch, (size_t) units * 16) == hipSuccess && hipMemset(rg->xch, 0, (size_t) units * 16) == hipSuccess
      && hipMalloc((void **) &rg->ctl, 256) == hipSuccess && hipMemset(rg->ctl, 0, 256) == hipSuccess   // (ctl[2..6]: the greedy history words)
      && hipHostMalloc((void **) &rg->h_ctl, 64, hipHostMallocDefault) == hipSuccess;
    if (ok) { rg->h_ctl[0] = 8u; rg->h_ctl[1] = 0u; }
    const unsigned init[2] = {8u, 0u};
    ok = ok && hipMemcpy(rg->ctl, init, sizeof(init), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok || hipDeviceSynchronize() != hipSuccess) { ring_v6_destroy(rg); return nullptr; }
    R6P & q = rg->proto;
    q.layers = rs->d_layers; q.n_layers = rs->n_layers; q.layer0 = 0; q.layers_total = rs->n_layers;
    q.arena = (const unsigned char *) m.arena; q.w2b = rs->w2b;
    q.state_stride = m.state_per_layer();
    q.xch = rg->xch; q.xch_bytes = (unsigned) (units * 16);
    int u = 0;
    int * slots[9] = {&q.tl, &q.act5, &q.rkvg, &q.dl, &q.yq, &q.xatt, &q.kq, &q.xffn, &q.am};
    for (int i = 0; i < 9; i++) { *slots[i] = u; u += (int) sizes[i]; }
    q.act_stride = (int) act_stride;
    q.ctl = rg->ctl;
    q.stream = rs->stream; q.cus = rs->d_cus;
    q.F = (int) F; q.DR = (int) rs->DR; q.R = (int) rs->R; q.H = (int) m.head_count;
    q.ring_bytes = (unsigned) rs->ring; q.mirror_bytes = (unsigned) rs->mirror;
    // on the workgroups of the value matrix: their r/k/v/g phase is the shortest (no decay row, 5.0 us against 5.8 - 6.4 us), and the
    // hand-over behind that phase waits for the slowest workgroup -- which the head's state loads and polls made the receptance ones
    // (same-box A/B: +0.7 %)
    q.head_wg0 = env_int("RWKV_MI_RING_HEAD_WG", NB / 2);
    if (q.head_wg0 < 0 || q.head_wg0 + (int) m.head_count > NB) q.head_wg0 = 0;
    q.logits = nullptr; q.lnout_w = rs->lnw_off; q.lnout_b = rs->lnb_off; q.n_vocab = (int) m.n_vocab();
    q.tok = nullptr; q.next_tok = nullptr; q.emb = rs->embed ? m.emb->data : nullptr; q.emb_f16 = (rs->embed && m.emb->type == T_F16) ? 1 : 0; q.ln0_w = rs->ln0w_off; q.ln0_b = rs->ln0b_off;
    auto snap = [](int w) { w &= ~3; return w < 4 ? 4 : (w > 52 ? 52 : w); };
    q.inflight = snap(env_int("RWKV_MI_RING_INFLIGHT", 48));
    q.thin = snap(env_int("RWKV_MI_RING_THIN", 16));
    q.hthin = snap(env_int("RWKV_MI_RING_HTHIN", q.inflight));
    q.look = env_int("RWKV_MI_RING_LOOK", 1);
    if (q.look < 1) q.look = 1;
    q.look_g = env_int("RWKV_MI_RING_LOOKG", q.look);
    if (q.look_g < 1) q.look_g = 1;
    q.nap = env_int("RWKV_MI_RING_NAP", 2);
    q.dbg = env_int("RWKV_MI_RING_DBG", 0);
    q.burst = env_int("RWKV_MI_RING_BURST", 24) / 4;   // in groups of four fills
    if (q.burst < 1) q.burst = 1;
    if (q.burst > 6) q.burst = 6;
    return rg;
}

bool ring_v6_trace(void * h, int layer, long long * out, bool fetch) {
    RingV6 * rg = (RingV6 *) h;
    const size_t n = (size_t) rg->sh->n_blocks * 8 * 32, extra = 2 * 512 * 4;   // (+ the loader's round samples of two workgroups)
    if (!rg->trace) { if (hipMalloc((void **) &rg->trace, (n + extra) * 8) != hipSuccess) return false; (void) hipMemset(rg->trace, 0, (n + extra) * 8); }
    rg->proto.trace = rg->trace; rg->proto.trace_layer = layer;
    if (fetch) {
        if (const char * path = getenv("RWKV_MI_RING_LTRACE")) {   // measurement aid: the loader samples as raw int64 [2][512][4]
            std::vector<long long> buf(extra);
            if (hipMemcpy(buf.data(), rg->trace + n, extra * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                if (FILE * f = fopen(path, "wb")) { fwrite(buf.data(), 8, extra, f); fclose(f); }
            }
        }
        return hipMemcpy(out, rg->trace, n * 8, hipMemcpyDeviceToHost) == hipSuccess;
    }
    return true;
}

uint64_t ring_v6_bytes(void * h) { return ((RingV6 *) h)->sh->bytes; }

bool ring_v6_folds_head(void * h) { return ((RingV6 *) h)->sh->head; }

// logits != nullptr (only when ring_v6_folds_head): ln_out + head run inside the launch and the logits land there
void ring_v6_forward(void * h, float * x, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf, float * logits, const uint32_t * tok, uint32_t * next_tok) {
    ring_v6_forward_range(h, x, sin, sout, st, pf, logits, 0, ((RingV6 *) h)->sh->n_layers, tok, next_tok);
}
bool ring_v6_folds_embed(void * h) { return ((RingV6 *) h)->sh->embed; }
bool ring_v6_folds_argmax(void * h) { return ((RingV6 *) h)->sh->head; }
// greedy loops: the kernel appends every token it picks to hist (device memory, n entries) from position 0; nullptr switches it off
bool ring_v6_set_history(void * h, uint32_t * hist, size_t n, hipStream_t st) {
    RingV6 * rg = (RingV6 *) h;
    const unsigned long long a = (unsigned long long) hist;
    const unsigned w[5] = {hist ? 1u : 0u, 0u, (unsigned) (a & 0xFFFFFFFFull), (unsigned) (a >> 32), hist ? (unsigned) (n > 0xFFFFFFFFull ? 0xFFFFFFFFull : n) : 0u};
    return hipMemcpyAsync(rg->ctl + 2, w, sizeof(w), hipMemcpyHostToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
}

// Layers [l0, l1) of the stage in one launch (sin / sout: state of the stage's FIRST layer; x: the residual stream in plain memory, read
// by the first and written by the last layer of the launch). logits (only with l1 == the stage's last layer): ln_out + head inside.
// tok (only with l0 == 0 and ring_v6_folds_embed): the launch starts from the token id; next_tok (only with logits): where the argmax of the logits lands.
void ring_v6_forward_range(void * h, float * x, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf, float * logits, int l0, int l1,
                           const uint32_t * tok, uint32_t * next_tok) {
    RingV6 * rg = (RingV6 *) h;
    R6P q = rg->proto;
    q.x = x;
    q.x_out = (rg->x_out && l1 == rg->sh->n_layers) ? rg->x_out : x;
    q.tok = (rg->sh->embed && l0 == 0) ? tok : nullptr;
    q.layers = rg->sh->d_layers + l0; q.n_layers = l1 - l0; q.layer0 = l0; q.layers_total = rg->sh->n_layers;
    q.sin = sin + (long long) l0 * q.state_stride; q.sout = sout + (long long) l0 * q.state_stride;
    q.logits = (rg->sh->head && l1 == rg->sh->n_layers) ? logits : nullptr;
    q.next_tok = q.logits ? next_tok : nullptr;
    const RingKernel fn = g_ring_variants[rg->sh->variant].fn;
    if (pf && pf->on) {
        if (pf->used * 2 + 2 > pf->events.size()) {
            hipEvent_t a = nullptr, c = nullptr;
            (void) hipEventCreate(&a); (void) hipEventCreate(&c);
            pf->events.push_back(a); pf->events.push_back(c); pf->bytes.push_back(0);
        }
        pf->bytes[pf->used] = rg->sh->bytes * (uint64_t) (l1 - l0) / (uint64_t) rg->sh->n_layers + (q.logits ? rg->sh->bytes_head : 0) + (q.tok ? rg->sh->bytes_embed : 0);
        hipExtLaunchKernelGGL(fn, dim3((unsigned) rg->sh->n_blocks), dim3(512), (uint32_t) rg->sh->lds, st, pf->events[pf->used * 2], pf->events[pf->used * 2 + 1], 0, q);
        pf->used++;
    } else {
        hipLaunchKernelGGL(fn, dim3((unsigned) rg->sh->n_blocks), dim3(512), rg->sh->lds, st, q);
    }
}

void ring_v6_set_x_out(void * h, float * x_out) { ((RingV6 *) h)->x_out = x_out; }

bool ring_v6_ctl_fetch(void * h, hipStream_t st) {
    RingV6 * rg = (RingV6 *) h;
    return hipMemcpyAsync(rg->h_ctl, rg->ctl, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st) == hipSuccess;
}
unsigned * ring_v6_ctl(void * h) { return ((RingV6 *) h)->ctl; }
bool ring_v6_aborted_cached(void * h) { return ((RingV6 *) h)->h_ctl[1] != 0; }
unsigned ring_v6_generation_cached(void * h) { return ((RingV6 *) h)->h_ctl[0]; }
bool ring_v6_clear_abort(void * h, hipStream_t st) {
    RingV6 * rg = (RingV6 *) h;
    rg->h_ctl[1] = 0u;
    return hipMemsetAsync(rg->ctl + 1, 0, sizeof(unsigned), st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
}
bool ring_v6_set_tag(void * h, unsigned base, hipStream_t st) {
    RingV6 * rg = (RingV6 *) h;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    return hipMemcpy(rg->ctl, &base, sizeof(unsigned), hipMemcpyHostToDevice) == hipSuccess;
}

}  // namespace rwkvmi
