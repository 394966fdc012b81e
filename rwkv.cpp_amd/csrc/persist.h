// persist.h -- building blocks shared by the persistent RWKV-6 decode kernels (mega_v6.hip: register prefetch; ring_v6.hip: LDS-DMA
// weight ring): the per-layer offset table, the tagged 16-byte exchange units (store / poll / staging of quantised vectors) and the
// four-elements-per-lane activation quantiser. Arithmetic and orders follow DESIGN.md section 4.
#pragma once

#include "fused_blocks.h"

#include <hip/hip_ext.h>

namespace rwkvmi {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef volatile __attribute__((address_space(1))) v4u gv4u;

// Per-layer table in HBM: byte offsets from the parameter arena. A pointer READ FROM MEMORY is generic to the compiler and
// generic (FLAT) loads also count on the LDS counter -- every LDS wait would then wait for the weight prefetch in flight;
// arena (a kernel argument, known global) + offset keeps the weight stream on vmcnt alone. Fields are fetched with
// scalar loads where they are used, not held across the layer.
struct M6Off { long long qs, qh, sc; };
struct M6Layer {
    long long ln1_w, ln1_b, maa_x, maa[5], w2b /* floats into M6P::w2b */, time_decay, faaaa, lnx_w, lnx_b, ln2_w, ln2_b, fmaa_k, fmaa_r;
    M6Off w1, rkvg[4], dw1, dw2, wo, fk, fr, fv;
};
struct M6Arena {
    const unsigned char * base;
    __device__ __forceinline__ const float * f(long long off) const { return reinterpret_cast<const float *>(base + off); }
    __device__ __forceinline__ WPl w(const M6Off & o) const {
        return WPl{base + o.qs, reinterpret_cast<const uint32_t *>(base + o.qh), reinterpret_cast<const void *>(base + o.sc)};
    }
};


// ---------------------------------------------------------------------------------------------------------------
// tagged exchange
// ---------------------------------------------------------------------------------------------------------------

// A unit is 16 bytes {p0, p1, p2, (aux16 << 16) | tag16}, written with ONE 16-byte store and read with ONE 16-byte load
// (sc0 sc1: past the non-coherent caches). 16-byte aligned vector accesses are single-copy on this memory system
// (tools/tear16.hip: 3e8 concurrent reads against 5e7 updates from other XCDs, no torn unit), so a unit whose tag matches
// carries its whole payload. The tag is a 16-bit rolling generation; the stale content of a unit is always the previous
// generation of the same buffer.
// Accesses go through one raw buffer descriptor over the exchange arena with the sc1 (agent scope) cache policy only:
// past the per-XCD L2 for foreign lines, but not the system-scope path a volatile access would take (measured ~2x slower).
typedef __amdgpu_buffer_rsrc_t xrsrc;
__device__ __forceinline__ xrsrc make_xrsrc(void * base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int) bytes, 0x00020000); }
__device__ __forceinline__ void tg_store(xrsrc xr, int unit, unsigned a, unsigned b, unsigned c, unsigned aux16, unsigned tag) {
    const v4u v = {a, b, c, (aux16 << 16) | (tag & 0xFFFFu)};
    __builtin_amdgcn_raw_buffer_store_b128(v, xr, unit * 16, 0, 16);
}
__device__ __forceinline__ v4u tg_load(xrsrc xr, int unit) { return __builtin_amdgcn_raw_buffer_load_b128(xr, unit * 16, 0, 16); }
__device__ __forceinline__ bool tg_ok(const v4u & v, unsigned tag) { return (v.w & 0xFFFFu) == (tag & 0xFFFFu); }

struct Poll { unsigned * ctl; bool dead; const void * xch = nullptr; /* base of the exchange arena (ring_v6.hip: the scalar-path look at a sentinel) */ bool sw = false; /* ... this wave looks through the scalar path */ };

__device__ __forceinline__ bool poll_backoff(Poll & pl, unsigned spin) {
    if ((spin & 63u) == 63u) {
        if (__hip_atomic_load(pl.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) pl.dead = true;
        else if (spin > 3000000u) { __hip_atomic_store(pl.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pl.dead = true; }
    }
    __builtin_amdgcn_s_sleep(1);
    return pl.dead;
}

// A long wait (tens of microseconds: a workgroup that has nothing to do until a far-away phase) on ONE unit, every lane of the wave reading the
// same address (one request per attempt) with SL x 64 clocks of sleep between attempts -- the full poll that follows it finds the data
// there or nearly there. A full-width poll spinning for that long is N x 1 KiB per attempt and wave against the very lines the running
// phases hand over through.
template <int SL>
__device__ __forceinline__ void calm_wait(Poll & pl, xrsrc xr, int unit, unsigned tag) {
    for (unsigned spin = 0;; spin++) {
        asm volatile("" ::: "memory");
        const v4u v = tg_load(xr, unit);   // (the lanes may watch different units: all of them)
        if (__all(tg_ok(v, tag)) || pl.dead) break;
        if ((spin & 63u) == 63u) {
            if (__hip_atomic_load(pl.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) pl.dead = true;
            else if (spin > 3000000u) { __hip_atomic_store(pl.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pl.dead = true; }
        }
        __builtin_amdgcn_s_sleep(SL);
    }
}

// The watch in front of a hand-over's sweep, where the wait is on the layer's critical path: FOUR reads of the unit's tag word in flight,
// issued SL x 64 clocks apart and checked as the oldest lands (reads of a wave return in order) -- the unit is sampled every SL x 64
// clocks instead of once per memory round trip (~0.6 us), at 64 bytes per read. All four land in ONE register: a younger read that
// overtakes the check only shows a fresher tag. The loop is one block of assembly: written with the builtins the compiler waits for ALL
// reads (vmcnt(0)) in front of every check -- the one-deep watch again -- and a tied operand around a loop gets copied while in flight.
// The register must stay reserved until the reads still in flight when the watch ends have landed: the caller keeps the Watch4 alive
// across its sweep (whose own reads are younger and waited for) and hands it to watch_done() behind it; tools/check_watch_regs.py reads
// the build's assembly and fails when anything writes the register between the two markers.
typedef unsigned v4s __attribute__((ext_vector_type(4)));
struct Watch4 { unsigned v; };
template <int SL>
__device__ __forceinline__ void watch4(Watch4 & w, Poll & pl, void * base, unsigned bytes, int unit, unsigned tag) {
    const unsigned long long ba = (unsigned long long) base;
    v4s rs;
    rs.x = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) ba);
    rs.y = (unsigned) __builtin_amdgcn_readfirstlane((int) ((unsigned) (ba >> 32) & 0xFFFFu));
    rs.z = (unsigned) __builtin_amdgcn_readfirstlane((int) bytes);
    rs.w = 0x00020000u;
    const int off = unit * 16 + 12;
    const unsigned want = (unsigned) __builtin_amdgcn_readfirstlane((int) (tag & 0xFFFFu));
    unsigned v = 0u, t, left;
    for (unsigned round = 0;; round++) {
        asm volatile("s_waitcnt vmcnt(0)\n\t"
                     "buffer_load_dword %0, %3, %4, 0 offen sc1\n\ts_sleep %6\n\t"
                     "buffer_load_dword %0, %3, %4, 0 offen sc1\n\ts_sleep %6\n\t"
                     "buffer_load_dword %0, %3, %4, 0 offen sc1\n\ts_sleep %6\n\t"
                     "buffer_load_dword %0, %3, %4, 0 offen sc1\n\ts_sleep %6\n\t"
                     "s_movk_i32 %2, 0x100\n"
                     "1:\n\t"
                     "s_waitcnt vmcnt(3)\n\t"
                     "v_and_b32 %1, 0xffff, %0\n\t"
                     "v_cmp_eq_u32 vcc, %5, %1\n\t"
                     "s_cmp_eq_u64 vcc, exec\n\t"            // (every lane: the lanes may watch different units)
                     "s_cbranch_scc1 2f\n\t"
                     "buffer_load_dword %0, %3, %4, 0 offen sc1\n\t"
                     "s_sleep %6\n\t"
                     "s_sub_u32 %2, %2, 1\n\t"
                     "s_cmp_lg_u32 %2, 0\n\t"
                     "s_cbranch_scc1 1b\n"
                     "2:\n\t"
                     "; WATCH4_BEGIN %0"
                     : "=&v"(v), "=&v"(t), "=&s"(left) : "v"(off), "s"(rs), "s"(want), "n"(SL) : "vcc", "scc", "memory");
        if (left != 0u || pl.dead) break;
        if (__hip_atomic_load(pl.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) pl.dead = true;
        else if (round > 30000u) { __hip_atomic_store(pl.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pl.dead = true; }
        if (pl.dead) break;
    }
    w.v = v;
}
__device__ __forceinline__ void watch_done(Watch4 & w) { asm volatile("; WATCH4_END %0" : : "v"(w.v) : "memory"); }
// ... and where no sweep of this wave follows the watch: wait for the reads still in flight before the register is released
__device__ __forceinline__ void watch_drain(Watch4 & w) { asm volatile("s_waitcnt vmcnt(0)\n\t; WATCH4_END %0" : : "v"(w.v) : "memory"); }

// Core: N units per lane given by address; all loads of an attempt are issued together; an attempt succeeds for the wave
// when every lane saw the expected tag on all of its valid units (invalid slots carry a harmless duplicate address).
template <int N>
__device__ __forceinline__ void poll_ptrs(Poll & pl, xrsrc xr, const int (&ptr)[N], const bool (&valid)[N], unsigned tag, v4u (&out)[N]) {
    for (unsigned spin = 0;; spin++) {
        asm volatile("" ::: "memory");   // the loads below are not volatile (that would make them system scope): keep them in the loop
#pragma unroll
        for (int u = 0; u < N; u++) out[u] = tg_load(xr, ptr[u]);
        bool ok = true;
#pragma unroll
        for (int u = 0; u < N; u++) ok = ok && (!valid[u] || tg_ok(out[u], tag));
        if (__all(ok) || pl.dead) break;
        if (poll_backoff(pl, spin)) break;
    }
}

// Lanes tid, tid + NT, ... own units of a contiguous range; sink(i, unit) runs once per unit.
// The buffer is padded to MAXU * NT units, so every slot is loaded unclamped (base + immediate offset addressing, no
// per-slot address registers); slots past n are simply not checked.
template <int MAXU, int NT, typename Sink>
__device__ __forceinline__ void poll_units(Poll & pl, xrsrc xr, int src, int n, unsigned tag, int tid, Sink && sink) {
    static_assert(MAXU <= 32, "poll_units: range too long for one round");
    const int mine = src + tid;
    v4u v[MAXU];
    for (unsigned spin = 0;; spin++) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < MAXU; u++) v[u] = tg_load(xr, mine + u * NT);
        bool ok = true;
#pragma unroll
        for (int u = 0; u < MAXU; u++) ok = ok && (tid + u * NT >= n || tg_ok(v[u], tag));
        if (__all(ok) || pl.dead) break;
        if (poll_backoff(pl, spin)) break;
    }
#pragma unroll
    for (int u = 0; u < MAXU; u++) if (tid + u * NT < n) sink(tid + u * NT, v[u]);
}

// A quantised vector of K elements travels as 3 units per 32-element block b (dwords q0..q7 of the block, elements 4j..4j+3
// in dword j):  unit 3b = {q0, q1, q2},  3b+1 = {q3, q4, q5},  3b+2 = {q6, q7, fp16 pair {d, s}} with the integer sum in aux16.
__device__ __forceinline__ unsigned f16_bits(float v) { return (unsigned) __half_as_ushort(__float2half_rn(v)); }

// One block from the 32 lanes of a half-wave (lane e holds element e): quads pack dwords, lanes 0..2 of the half-wave gather
// three each and store one unit each.
__device__ __forceinline__ void tq_store_block(xrsrc xr, int base, int blk, int e, int qi, float d16, float s16, int isum, unsigned tag, bool valid = true) {
    int w = (qi & 0xFF) << (8 * (e & 3));
    w |= lane_xor1_i(w);
    w |= lane_xor2_i(w);                       // every lane of quad j holds dword j
    const int half0 = (int) (threadIdx.x & 32);
    const int k = e < 3 ? e : 0;               // unit of this lane
    const int g0 = __shfl(w, half0 + 4 * (3 * k), WAVE);
    const int g1 = __shfl(w, half0 + 4 * (3 * k + 1), WAVE);
    const int g2 = __shfl(w, half0 + 4 * ((3 * k + 2) & 7), WAVE);
    if (!valid || e >= 3) return;
    if (e < 2) tg_store(xr, base + 3 * blk + e, (unsigned) g0, (unsigned) g1, (unsigned) g2, 0u, tag);
    else tg_store(xr, base + 3 * blk + 2, (unsigned) g0, (unsigned) g1, f16_bits(d16) | (f16_bits(s16) << 16), (unsigned) isum & 0xFFFFu, tag);
}

// Staging of a quantised vector into its lohi image in LDS, lanes 0..63 of ONE wave owning units lane, lane + 64, ...
// Unit i = 3 b + k carries dwords j = 3 k + t (t = 0, 1, 2) of block b; j < 4 lands in the lo plane, 4 <= j < 8 in the hi
// plane, j = 8 is the fp16 pair. 64 = 3 * 21 + 1, so (b, k) of slot u follow from (lane / 3, lane % 3) without a division
// per unit. (The first version divided, branched three ways per unit and cost ~1300 instructions for the F-vector -- on
// the one wave every other wave of the workgroup is waiting for.)
template <int MAXU, int NT>
__device__ __forceinline__ void stage_qvec(Poll & pl, xrsrc xr, int src, int K, unsigned tag, unsigned char * l, int tid) {
    static_assert(NT == 64, "stage_qvec: one wave");
    const int nb = K / 32;
    const QVec q = qvec_at(l, K);
    unsigned * img = reinterpret_cast<unsigned *>(l);
    const int b0 = tid / 3, k0 = tid - 3 * b0;
    poll_units<MAXU, NT>(pl, xr, src, 3 * nb, tag, tid, [&](int i, const v4u & v) {
        const int u = (i - tid) >> 6;              // slot (compile-time after unrolling)
        const int kk = k0 + u % 3;                  // 0..4
        const int k = kk >= 3 ? kk - 3 : kk;
        const int b = b0 + 21 * u + u / 3 + (kk >= 3 ? 1 : 0);
        const int j0 = 3 * k;                       // dword index of v.x within the block's eight code dwords
        const int a0 = (j0 < 4 ? 0 : 4 * nb) + 4 * b + (j0 & 3);
        const int j1 = j0 + 1;
        const int a1 = (j1 < 4 ? 0 : 4 * nb) + 4 * b + (j1 & 3);
        img[a0] = v.x;
        img[a1] = v.y;
        if (k < 2) {
            const int j2 = j0 + 2;
            img[(j2 < 4 ? 0 : 4 * nb) + 4 * b + (j2 & 3)] = v.z;
        } else {
            q.d[b] = h2f_bits((uint16_t) (v.z & 0xFFFFu)); q.s[b] = h2f_bits((uint16_t) (v.z >> 16));
            q.isum[b] = (int) (short) (v.w >> 16);
        }
    });
}

// Opaque copy: derived per-lane offsets (poll addresses, row offsets) are recomputed where they are used instead of
// being hoisted out of the layer loop as ~100 loop-invariant registers.
__device__ __forceinline__ int opq(int v) { asm volatile("" : "+v"(v)); return v; }
// same for a wave-uniform value: the per-layer table entries are re-fetched (scalar loads) in the phase that uses them
// instead of occupying ~80 SGPRs across the whole layer
__device__ __forceinline__ int opq_s(int v) { asm volatile("" : "+s"(v)); return v; }

// Branch-free issue for a (wave-uniform) optional job: an absent job loads block 0 of row 0 on every lane -- one 16-byte
// request, no bandwidth -- so that the issue sequence is straight-line code. With control flow around the loads the
// compiler's wait-count bookkeeping degrades to s_waitcnt vmcnt(0), i.e. "wait for the whole prefetch", at every use.
template <int FMT, int R, int U>
__device__ __forceinline__ void batch_issue_opt(bool has, Batch<FMT, R, U> & bt, const WPl & w, int row0, int N, int nb, int bbase, int lane) {
    batch_issue<FMT, R, U>(bt, w.qs, w.qh, w.sc, has ? row0 : 0, has ? N : 1, has ? nb : 1, has ? bbase : 0, has ? lane : 0);
}

// lane r of the wave picks res[r] (res is wave-uniform after the butterfly)
template <int R>
__device__ __forceinline__ float pick_lane(const float (&res)[R], int lane) {
    // (the values pass through an opaque copy: otherwise LLVM turns the select chain into an indexed load from a
    //  scratch array, and a scratch load waits behind every prefetch load in flight)
    float v = res[0];
#pragma unroll
    for (int r = 1; r < R; r++) { float t = res[r]; asm volatile("" : "+v"(t)); v = lane == r ? t : v; }
    return v;
}

enum { SLOT_TL = 0, SLOT_ACT = 1, SLOT_RKVG = 2, SLOT_YQ = 3, SLOT_XATT = 4, SLOT_KQ = 5, SLOT_XFFN = 6 };

__host__ __device__ inline size_t m6_round16(size_t v) { return (v + 15) / 16 * 16; }


// 4 consecutive elements per lane, 8 lanes per 32-element block (ggml quantize_row_q8_0 / q8_1; max and integer sum are
// order-free, so this is the same result as quant_block32 on the lane-per-element layout)
__device__ __forceinline__ void quant_vec4(const float (&v)[4], unsigned & packed, float & d16, float & s16, int & isum) {
    float am = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    am = fmaxf(am, __int_as_float(lane_xor1_i(__float_as_int(am))));
    am = fmaxf(am, __int_as_float(lane_xor2_i(__float_as_int(am))));
    am = fmaxf(am, __int_as_float(lane_xor4_i(__float_as_int(am))));
    const float dd = am / 127.0f;
    const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
    int q[4];
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = (int) roundf(v[j] * id);
    int sm = (q[0] + q[1]) + (q[2] + q[3]);
    sm += lane_xor1_i(sm);
    sm += lane_xor2_i(sm);
    sm += lane_xor4_i(sm);
    packed = (unsigned) (q[0] & 0xFF) | ((unsigned) (q[1] & 0xFF) << 8) | ((unsigned) (q[2] & 0xFF) << 16) | ((unsigned) (q[3] & 0xFF) << 24);
    isum = sm;
    d16 = round_f16(dd);
    s16 = round_f16((float) sm * dd);
}

// store the packed dword of elements [i, i+4) (i % 4 == 0) + the block scalars into a lohi image in LDS
__device__ __forceinline__ void qvec_store4(const QVec & v, int nb, int i, unsigned packed, float d16, float s16, int isum) {
    const int blk = i >> 5, e = i & 31;
    *reinterpret_cast<unsigned *>(v.q + (e < 16 ? 0 : nb * 16) + blk * 16 + (e & 15)) = packed;
    if (e == 0) { v.d[blk] = d16; v.s[blk] = s16; v.isum[blk] = isum; }
}

}  // namespace rwkvmi
