// mega_v6.hip -- the RWKV-6 single-token (decode) step over ALL layers of a stage as ONE persistent launch.
//
// Why: at batch 1 every phase of a layer (fused_v6.hip's seven launches) is an all-to-all dependency, and a launch boundary
// costs ~5 us of ramp + drain during which no weight bytes move; measured, the seven-launch path spends more time in those
// gaps than in streaming (profiles/). Here one workgroup per CU stays resident for the whole token and
//   * every wave owns a FIXED slice of every matrix (rows), so the weights of phase P+1 are loaded into registers while
//     phase P is still waiting for its inputs: weights do not depend on activations, only the order of use does;
//   * phases are chained by DATAFLOW, not by barriers: every value that crosses workgroups is an 8-byte {payload, tag}
//     unit written with one agent-scope atomic store and polled with agent-scope atomic loads until the tag of the
//     expected (token, layer, phase) shows up. No fence, no counter, no s_waitcnt vmcnt(0) on the producer side, so the
//     weight stream of the next phase stays in flight across the hand-over.
// Per wave the vector-memory queue returns in order, which fixes the schedule inside a phase:
//   poll the phase's inputs -> issue the NEXT phase's weight loads -> compute with this phase's weights (landed while the
//   wave was polling) -> tagged stores -> next phase's poll (returns once the weights in front of it have landed).
//
// Arithmetic, reduction orders and epilogues are those of fused_v6.hip / kernels.hip: the results are
// bit-identical to the seven-launch path and to the CPU oracle; only the distribution of rows over waves differs.
//
// Residency: the grid is one workgroup per CU (512 threads, ~100 KB LDS) and all of them must be resident at once, i.e. the
// device must not be shared with another process' persistent kernels. Polls are bounded: on timeout the abort word is set,
// every workgroup drains without waiting, and the host reports the step as failed (it never hangs the device).
#include "fused_blocks.h"

#include <hip/hip_ext.h>

namespace rwkvmi {

typedef unsigned long long u64;

struct M6Layer {
    const float *ln1_w, *ln1_b, *maa_x, *maa[5], *w2t, *time_decay, *faaaa, *lnx_w, *lnx_b, *ln2_w, *ln2_b, *fmaa_k, *fmaa_r;
    WPl w1, rkvg[4], dw1, dw2, wo, fk, fr, fv;
};

struct M6P {
    const M6Layer * layers; int n_layers;
    float * x;                                       // plain residual stream: input of the first layer, output of the last
    const float * sin; float * sout; long long state_stride;
    u64 *tl, *act5, *rkvg, *dl, *yq, *xatt, *kq, *rr, *xffn;   // tagged exchange buffers
    long long act_stride;                            // units between the five mix images
    unsigned * ctl;                                  // [0] tag generation, [1] abort
    int F, DR, R, H, gpb;
};

// ---------------------------------------------------------------------------------------------------------------
// tagged exchange
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ void tg_store(u64 * p, unsigned payload, unsigned tag) {
    __hip_atomic_store(p, ((u64) tag << 32) | (u64) payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 tg_load(const u64 * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Poll { unsigned * ctl; bool dead; };

// Core: N units per lane given by address; all loads of an attempt are issued together; an attempt succeeds for the wave
// when every lane saw the expected tag on all of its valid units (invalid slots carry a harmless duplicate address).
template <int N>
__device__ __forceinline__ void poll_ptrs(Poll & pl, const u64 * const (&ptr)[N], const bool (&valid)[N], unsigned tag, unsigned (&out)[N]) {
    u64 v[N];
    for (unsigned spin = 0;; spin++) {
#pragma unroll
        for (int u = 0; u < N; u++) v[u] = tg_load(ptr[u]);
        bool ok = true;
#pragma unroll
        for (int u = 0; u < N; u++) ok = ok && (!valid[u] || (unsigned) (v[u] >> 32) == tag);
        if (__all(ok) || pl.dead) break;
        if ((spin & 63u) == 63u) {
            if (__hip_atomic_load(pl.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) pl.dead = true;
            else if (spin > 3000000u) { __hip_atomic_store(pl.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pl.dead = true; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int u = 0; u < N; u++) out[u] = (unsigned) v[u];
}

// Threads tid, tid + NT, ... own units of a contiguous range; sink(i, payload) runs once per unit afterwards.
template <int MAXU, int NT, typename Sink>
__device__ __forceinline__ void poll_units(Poll & pl, const u64 * src, int n, unsigned tag, int tid, Sink && sink) {
    const u64 * ptr[MAXU];
    bool valid[MAXU];
    unsigned out[MAXU];
#pragma unroll
    for (int u = 0; u < MAXU; u++) { const int i = tid + u * NT; valid[u] = i < n; ptr[u] = src + (valid[u] ? i : n - 1); }
    poll_ptrs<MAXU>(pl, ptr, valid, tag, out);
#pragma unroll
    for (int u = 0; u < MAXU; u++) if (valid[u]) sink(tid + u * NT, out[u]);
}

// A quantised vector of K elements travels as 10 units per 32-element block: units [0, 8 nb) are the dwords of the lohi
// q image, [8 nb, 9 nb) the fp16 pair {d, s}, [9 nb, 10 nb) the integer sum.
__device__ __forceinline__ unsigned f16_bits(float v) { return (unsigned) __half_as_ushort(__float2half_rn(v)); }

// One block from the 32 lanes of a half-wave (lane e holds element e): 4 lanes pack a dword, the quad leader stores it.
__device__ __forceinline__ void tq_store_block(u64 * base, int nb, int blk, int e, int qi, float d16, float s16, int isum, unsigned tag, bool valid = true) {
    int w = (qi & 0xFF) << (8 * (e & 3));
    w |= lane_xor1_i(w);
    w |= lane_xor2_i(w);
    if (!valid) return;
    if ((e & 3) == 0) tg_store(base + (e < 16 ? 0 : 4 * nb) + blk * 4 + ((e & 15) >> 2), (unsigned) w, tag);
    if (e == 0) {
        tg_store(base + 8 * nb + blk, f16_bits(d16) | (f16_bits(s16) << 16), tag);
        tg_store(base + 9 * nb + blk, (unsigned) isum, tag);
    }
}

template <int MAXU, int NT>
__device__ __forceinline__ void stage_qvec(Poll & pl, const u64 * src, int K, unsigned tag, unsigned char * l, int tid) {
    const int nb = K / 32;
    const QVec q = qvec_at(l, K);
    poll_units<MAXU, NT>(pl, src, 10 * nb, tag, tid, [&](int i, unsigned v) {
        if (i < 8 * nb) reinterpret_cast<unsigned *>(l)[i] = v;
        else if (i < 9 * nb) { q.d[i - 8 * nb] = h2f_bits((uint16_t) (v & 0xFFFFu)); q.s[i - 8 * nb] = h2f_bits((uint16_t) (v >> 16)); }
        else q.isum[i - 9 * nb] = (int) v;
    });
}

// Waves without work in a phase define their staging registers too (zeros): a conditional issue alone would keep the
// previous iteration's values live around the whole layer loop and push the kernel into scratch.
template <int FMT, int R, int U>
__device__ __forceinline__ void batch_zero(Batch<FMT, R, U> & bt) {
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int k = 0; k < QF<FMT>::QS / 16; k++) bt.raw[u][r].q[k] = make_int4(0, 0, 0, 0);
            bt.raw[u][r].qh = 0; bt.raw[u][r].sc = 0;
        }
}

// Opaque copy: derived per-lane offsets (poll addresses, row offsets) are recomputed where they are used instead of
// being hoisted out of the layer loop as ~100 loop-invariant registers.
__device__ __forceinline__ int opq(int v) { asm volatile("" : "+v"(v)); return v; }

// lane r of the wave picks res[r] (res is wave-uniform after the butterfly)
template <int R>
__device__ __forceinline__ float pick_lane(const float (&res)[R], int lane) {
    float v = res[0];
#pragma unroll
    for (int r = 1; r < R; r++) v = lane == r ? res[r] : v;
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// the kernel. EPT = D / 512 (elements of a D-vector per thread), KQU = poll slots per thread for the F-vector,
// NBD = decay rank / 32.
// ---------------------------------------------------------------------------------------------------------------

enum { SLOT_TL = 0, SLOT_ACT = 1, SLOT_RKVG = 2, SLOT_YQ = 3, SLOT_XATT = 4, SLOT_KQ = 5, SLOT_XFFN = 6 };

__host__ __device__ inline size_t m6_round16(size_t v) { return (v + 15) / 16 * 16; }

struct M6Lds { size_t x, xn, sx, q1, q2, act, actw, yq, kq, tl, red, out, dl, total; };
__host__ __device__ inline M6Lds m6_lds(int D, int F) {
    M6Lds o; size_t p = 0;
    auto take = [&](size_t n) { const size_t r = p; p += m6_round16(n); return r; };
    o.x = take((size_t) D * 4); o.xn = take((size_t) D * 4); o.sx = take((size_t) D * 4);
    o.q1 = take(qvec_bytes(D)); o.q2 = take(qvec_bytes(D)); o.act = take(qvec_bytes(D)); o.actw = take(qvec_bytes(D)); o.yq = take(qvec_bytes(D));
    o.kq = take(qvec_bytes(F)); o.tl = take(1280 * 4); o.red = take(258 * 8); o.out = take(2 * 32 * 4); o.dl = take(8 * 192);
    o.total = p;
    return o;
}

template <int FMT, int EPT, int KQU, int NBD>
__global__ __launch_bounds__(512) void k6_mega(M6P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = 512, S = 64;
    constexpr int D = EPT * NT;
    constexpr int nb = D / 32;                     // blocks of a D-vector
    constexpr int UD = nb / 64 > 0 ? nb / 64 : 1;  // 64-block steps covering K = D
    constexpr int DU = (10 * nb + NT - 1) / NT;    // poll slots per thread for a quantised D-vector
    const int tid0 = threadIdx.x;
    const int tid = tid0, lane = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);   // wave-uniform: row bases and work predicates stay in SGPRs
    const int blk = blockIdx.x, NB = gridDim.x;
    const int F = p.F, DR = p.DR, R = p.R, H = p.H;
    const int nbF = F / 32;

    const M6Lds lo = m6_lds(D, F);
    float * l_x = reinterpret_cast<float *>(smem + lo.x);
    float * l_xn = reinterpret_cast<float *>(smem + lo.xn);
    float * l_sx = reinterpret_cast<float *>(smem + lo.sx);
    unsigned char * l_q1 = smem + lo.q1;
    unsigned char * l_q2 = smem + lo.q2;
    unsigned char * l_act = smem + lo.act;
    unsigned char * l_actw = smem + lo.actw;
    unsigned char * l_yq = smem + lo.yq;
    unsigned char * l_kq = smem + lo.kq;
    float * l_tl = reinterpret_cast<float *>(smem + lo.tl);
    double * red = reinterpret_cast<double *>(smem + lo.red);
    float * l_out = reinterpret_cast<float *>(smem + lo.out);
    unsigned char * l_dl = smem + lo.dl + wave * 192;

    Poll pl{p.ctl, false};
    const unsigned base = p.ctl[0];

    // ---- static work assignment ----
    // A: W1 row (5R rows, one per wave, interleaved over workgroups)
    const int a_row = wave * NB + blk;
    const bool a_has = a_row < 5 * R;
    // B: 64-element chunk of the five mixes
    const int b_chunk = wave * NB + blk;
    const bool b_has = b_chunk < 5 * (D / 64);
    const int b_f = b_has ? b_chunk / (D / 64) : 0;
    const int b_d0 = (b_has ? b_chunk % (D / 64) : 0) * 64;
    // C: 8-row set of r/k/v/g (workgroup-major: one activation image per workgroup) + one decay-W1 row on wave 7
    const int c_set = blk * 8 + wave;
    const bool c_has = c_set < 4 * (D / 8);
    const int c_mat = c_has ? c_set / (D / 8) : 0;
    const int c_row0 = (c_has ? c_set % (D / 8) : 0) * 8;
    const int c_act = (0x4213 >> (4 * c_mat)) & 0xF;   // r,k,v,g -> mix image (w,k,v,r,g order)
    const int blk_mat = (blk * 8) / (D / 8);           // matrix of this workgroup's sets (uniform: (D/8) % 8 == 0)
    const bool blk_c_has = blk * 8 < 4 * (D / 8);
    const int blk_act = (0x4213 >> (4 * (blk_mat & 3))) & 0xF;
    const int c_xrow = blk + NB * (7 - wave);
    const bool c_xhas = c_xrow < DR;
    const bool blk_xhas = blk < DR;                    // wave 7 of this workgroup has a decay row
    // D: head
    const int d_head = blk + NB * wave;
    const bool d_has = d_head < H;
    // E / G / F-receptance: 2-row set, interleaved over workgroups (the same wave owns x[n] in E and G)
    const int e_set = wave * NB + blk;
    const bool e_has = e_set < D / 2;
    const int e_row0 = e_has ? e_set * 2 : 0;
    // F: key groups of 32 rows, gpb consecutive groups per workgroup, wave w owns rows 4w..4w+3 of each
    const int GK = nbF;
    const int gpb = p.gpb;

    float xown = 0.0f;  // lane r < 2 of an owner wave: x[e_row0 + r]
    if (e_has && lane < 2) xown = p.x[e_row0 + lane];

    // ---- prefetch registers ----
    struct PA { float lw[EPT], lb[EPT], pv[EPT], mx[EPT]; } pa;
    struct PF { float lw[EPT], lb[EPT], pv[EPT], mk[EPT], mr[EPT]; } pf;
    Batch<FMT, 1, UD> wA;
    float wB[64]; float wBmaa = 0.0f;
    Batch<FMT, 8, UD> wC; Batch<FMT, 1, UD> wCx;
    struct WD { float s[S]; RawBlk<FMT> w2[NBD]; float td, u, lw, lb; } wD;
    Batch<FMT, 2, UD> wE;
    Batch<FMT, 4, UD> wFk[2]; Batch<FMT, 2, UD> wFr;
    Batch<FMT, 2, 4> wG[2];

    auto issue_A = [&](const M6Layer & L, const float * sin_l, int tid, int lane) {
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const int i = tid + u * NT;
            pa.lw[u] = L.ln1_w[i]; pa.lb[u] = L.ln1_b[i]; pa.pv[u] = sin_l[D + i]; pa.mx[u] = L.maa_x[i];
        }
        if (a_has) batch_issue<FMT, 1, UD>(wA, L.w1.qs, L.w1.qh, L.w1.sc, a_row, 5 * R, nb, 0, lane);
        else batch_zero<FMT, 1, UD>(wA);
    };

    issue_A(p.layers[0], p.sin, tid, lane);

    for (int li = 0; li < p.n_layers; li++) {
        const M6Layer & L = p.layers[li];
        const float * sin_l = p.sin + (long long) li * p.state_stride;
        float * sout_l = p.sout + (long long) li * p.state_stride;
        const unsigned tagL = base + (unsigned) li * 8u;
        unsigned dq[6] = {0, 0, 0, 0, 0, 0};
        unsigned rrv[1] = {0};

        // =========================================== A ===========================================
        {
        const int tid = opq(tid0), lane = tid & 63;
        if (li == 0) {
#pragma unroll
            for (int u = 0; u < EPT; u++) l_x[tid + u * NT] = p.x[tid + u * NT];
        } else {
            poll_units<EPT, NT>(pl, p.xffn, D, tagL - 8u + SLOT_XFFN, tid, [&](int i, unsigned v) { l_x[i] = __uint_as_float(v); });
        }
        __syncthreads();
        // next phase's weights: the W2 column of this wave's chunk
        if (b_has) {
            const int b_d = b_d0 + lane;
            const float * col = L.w2t + (long long) b_f * R * D + b_d;
#pragma unroll
            for (int m = 0; m < 64; m++) wB[m] = col[(long long) (m < R ? m : R - 1) * D];
            wBmaa = L.maa[b_f][b_d];
        } else {
#pragma unroll
            for (int m = 0; m < 64; m++) wB[m] = 0.0f;
            wBmaa = 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const bool pro = tid < 256;
            double sacc = 0.0;
            if (pro) for (int i = tid; i < D; i += 256) sacc += (double) l_x[i];
            const float mean = (float) (block_sum_d_8w(sacc, red) / (double) D);
            double s2 = 0.0;
            if (pro) for (int i = tid; i < D; i += 256) { const float v = l_x[i] - mean; l_x[i] = v; s2 += (double) (v * v); }
            const float var = (float) (block_sum_d_8w(s2, red) / (double) D);
            const float scale = 1.0f / sqrtf(var + 1e-5f);
            const QVec lq = qvec_at(l_q1, D);
            auto fin = [&](int u) -> float {
                const int i = tid + u * NT;
                const float y = l_x[i] * scale;
                const float yw = y * pa.lw[u];
                const float xn = yw + pa.lb[u];
                const float sx = pa.pv[u] - xn;
                const float sm = sx * pa.mx[u];
                l_xn[i] = xn; l_sx[i] = sx;
                if (blk == 0) sout_l[D + i] = xn;
                return sm + xn;
            };
            int u0 = 0;
#pragma unroll
            for (; u0 + 3 < EPT; u0 += 4) {
                float xv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) xv[u] = fin(u0 + u);
                int qi[4], isum[4]; float d16[4], s16[4];
                quant_blocks<4>(xv, qi, d16, s16, isum);
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = tid + (u0 + u) * NT; qvec_store(lq, nb, i >> 5, i & 31, qi[u], d16[u], s16[u], isum[u]); }
            }
#pragma unroll
            for (; u0 < EPT; u0++) {
                const float xxx = fin(u0);
                int qi, isum; float d16, s16;
                quant_block32(xxx, qi, d16, s16, isum);
                const int i = tid + u0 * NT;
                qvec_store(lq, nb, i >> 5, i & 31, qi, d16, s16, isum);
            }
            __syncthreads();
            if (a_has) {
                float res[1];
                rows_finish<FMT, 1, UD>(wA, L.w1.qs, L.w1.qh, L.w1.sc, a_row, 5 * R, nb, lq, lane, res);
                if (lane == 0) tg_store(p.tl + a_row, __float_as_uint(det_tanhf(res[0])), tagL + SLOT_TL);
            }
        }
        }

        // =========================================== B ===========================================
        {
        const int tid = opq(tid0), lane = tid & 63;
        poll_units<3, NT>(pl, p.tl, 5 * R, tagL + SLOT_TL, tid, [&](int i, unsigned v) { l_tl[i] = __uint_as_float(v); });
        __syncthreads();
        if (c_has) batch_issue<FMT, 8, UD>(wC, L.rkvg[c_mat].qs, L.rkvg[c_mat].qh, L.rkvg[c_mat].sc, c_row0, D, nb, 0, lane);
        else batch_zero<FMT, 8, UD>(wC);
        if (c_xhas) batch_issue<FMT, 1, UD>(wCx, L.dw1.qs, L.dw1.qh, L.dw1.sc, c_xrow, DR, nb, 0, lane);
        else batch_zero<FMT, 1, UD>(wCx);
        __builtin_amdgcn_sched_barrier(0);
        if (b_has) {
            const int b_d = b_d0 + lane;
            const float * tlf = l_tl + b_f * R;
            float acc = 0.0f;
#pragma unroll
            for (int m = 0; m < 64; m++) if (m < R) acc += wB[m] * tlf[m];
            const float mm = (acc + wBmaa) * l_sx[b_d];
            const float o = mm + l_xn[b_d];
            int qi, isum; float d16, s16;
            quant_block32(o, qi, d16, s16, isum);
            tq_store_block(p.act5 + (long long) b_f * p.act_stride, nb, b_d >> 5, lane & 31, qi, d16, s16, isum, tagL + SLOT_ACT);
        }
        }

        // =========================================== C ===========================================
        {
        const int tid = opq(tid0), lane = tid & 63;
        if (blk_c_has) stage_qvec<DU, NT>(pl, p.act5 + (long long) blk_act * p.act_stride, D, tagL + SLOT_ACT, l_act, tid);
        if (blk_xhas) stage_qvec<DU, NT>(pl, p.act5, D, tagL + SLOT_ACT, l_actw, tid);
        __syncthreads();
        if (d_has) {
            const float * st = sin_l + 2 * D + (long long) d_head * S * S;
#pragma unroll
            for (int i = 0; i < S; i++) wD.s[i] = st[i * S + lane];
            const int c = d_head * S + lane;
#pragma unroll
            for (int b = 0; b < NBD; b++) load_raw<FMT>(wD.w2[b], L.dw2.qs, L.dw2.qh, L.dw2.sc, (long long) c * NBD + b);
            wD.td = L.time_decay[c]; wD.u = L.faaaa[c]; wD.lw = L.lnx_w[c]; wD.lb = L.lnx_b[c];
        } else {
#pragma unroll
            for (int i = 0; i < S; i++) wD.s[i] = 0.0f;
#pragma unroll
            for (int b = 0; b < NBD; b++) { wD.w2[b].q[0] = make_int4(0, 0, 0, 0); if (QF<FMT>::QS == 32) wD.w2[b].q[QF<FMT>::QS / 16 - 1] = make_int4(0, 0, 0, 0); wD.w2[b].qh = 0; wD.w2[b].sc = 0; }
            wD.td = 0.0f; wD.u = 0.0f; wD.lw = 0.0f; wD.lb = 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c_has) {
            const QVec la = qvec_at(l_act, D);
            float res[8];
            rows_finish<FMT, 8, UD>(wC, L.rkvg[c_mat].qs, L.rkvg[c_mat].qh, L.rkvg[c_mat].sc, c_row0, D, nb, la, lane, res);
            float v = pick_lane<8>(res, lane);
            if (c_mat == 3) v = v / (1.0f + det_expf(-v));
            if (lane < 8) tg_store(p.rkvg + (long long) c_mat * D + c_row0 + lane, __float_as_uint(v), tagL + SLOT_RKVG);
        }
        if (c_xhas) {
            const QVec la = qvec_at(l_actw, D);
            float res[1];
            rows_finish<FMT, 1, UD>(wCx, L.dw1.qs, L.dw1.qh, L.dw1.sc, c_xrow, DR, nb, la, lane, res);
            if (lane == 0) tg_store(p.dl + c_xrow, __float_as_uint(det_tanhf(res[0])), tagL + SLOT_RKVG);
        }
        }

        // =========================================== D ===========================================
        {
        const int tid = opq(tid0), lane = tid & 63;
        if (d_has) {
            const int c = d_head * S + lane;
            const u64 * ptr[6] = {p.rkvg + c, p.rkvg + D + c, p.rkvg + 2 * D + c, p.rkvg + 3 * D + c, p.dl + lane, p.dl + (NBD > 2 ? 64 + lane : lane)};
            const bool valid[6] = {true, true, true, true, true, NBD > 2};
            poll_ptrs<6>(pl, ptr, valid, tagL + SLOT_RKVG, dq);
        }
        if (e_has) batch_issue<FMT, 2, UD>(wE, L.wo.qs, L.wo.qh, L.wo.sc, e_row0, D, nb, 0, lane);
        else batch_zero<FMT, 2, UD>(wE);
        __builtin_amdgcn_sched_barrier(0);
        if (d_has) {
            const int c = d_head * S + lane;
            // 1. quantise dl (DR = 32 NBD elements) into this wave's LDS slot: half-wave = block
            const QVec ldl = qvec_at(l_dl, NBD * 32);
#pragma unroll
            for (int j = 0; j < (NBD * 32 + 63) / 64; j++) {
                const int e = j * 64 + lane;
                const float val = e < NBD * 32 ? __uint_as_float(dq[4 + j]) : 0.0f;
                int qi, isum; float d16, s16;
                quant_block32(val, qi, d16, s16, isum);
                if (e < NBD * 32) qvec_store(ldl, NBD, e >> 5, e & 31, qi, d16, s16, isum);
            }
            __builtin_amdgcn_wave_barrier();
            // 2. decay row of channel c (order of the 64-entry halving tree, zeros elsewhere)
            float P[NBD];
#pragma unroll
            for (int b = 0; b < NBD; b++) {
                WBlk<FMT> w;
                unpack_raw<FMT>(w, wD.w2[b]);
                const int4 alo = *reinterpret_cast<const int4 *>(ldl.q + b * 16);
                const int4 ahi = *reinterpret_cast<const int4 *>(ldl.q + NBD * 16 + b * 16);
                P[b] = blk_fma<FMT>(w, alo, ahi, ldl.d[b], ldl.s[b], ldl.isum[b], 0.0f);
            }
#pragma unroll
            for (int o = NBD / 2; o > 0; o >>= 1)
#pragma unroll
                for (int i = 0; i < o; i++) P[i] += P[i + o];
            const float wdec = det_expf(-det_expf(P[0] + wD.td));
            // 3. WKV6: lane j owns value column j; r_i, k_i, u_i, w_i are broadcast from lane i
            const int rr_i = (int) dq[0], kk_i = (int) dq[1], uu_i = __float_as_int(wD.u), ww_i = __float_as_int(wdec);
            const float vj = __uint_as_float(dq[2]);
            float o = 0.0f;
            float * so = sout_l + 2 * D + (long long) d_head * S * S;
#pragma unroll
            for (int i = 0; i < S; i++) {
                const float ki = __int_as_float(__builtin_amdgcn_readlane(kk_i, i));
                const float ui = __int_as_float(__builtin_amdgcn_readlane(uu_i, i));
                const float ri = __int_as_float(__builtin_amdgcn_readlane(rr_i, i));
                const float wi = __int_as_float(__builtin_amdgcn_readlane(ww_i, i));
                const float kv = vj * ki;
                const float prev = wD.s[i];
                const float temp = kv * ui + prev;
                o += temp * ri;
                so[i * S + lane] = prev * wi + kv;
            }
            // 4. GroupNorm over the head, * ln_x, gate
            const float mean = (float) (wave_sum_d((double) o) / (double) S);
            const float dv = o - mean;
            const float var = (float) (wave_sum_d((double) (dv * dv)) / (double) S);
            const float scale = 1.0f / sqrtf(var + 64e-5f);
            float y = dv * scale;
            y = y * wD.lw;
            y = y + wD.lb;
            y *= __uint_as_float(dq[3]);
            int qi, isum; float d16, s16;
            quant_block32(y, qi, d16, s16, isum);
            tq_store_block(p.yq, nb, 2 * d_head + (lane >> 5), lane & 31, qi, d16, s16, isum, tagL + SLOT_YQ);
            (void) c;
        }
        }

        // =========================================== E ===========================================
        {
        const int tid = opq(tid0), lane = tid & 63;
        stage_qvec<DU, NT>(pl, p.yq, D, tagL + SLOT_YQ, l_yq, tid);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const int i = tid + u * NT;
            pf.lw[u] = L.ln2_w[i]; pf.lb[u] = L.ln2_b[i]; pf.pv[u] = sin_l[i]; pf.mk[u] = L.fmaa_k[i]; pf.mr[u] = L.fmaa_r[i];
        }
#pragma unroll
        for (int gi = 0; gi < 2; gi++) {
            const int g = blk * gpb + gi;
            if (gi < gpb && g < GK) batch_issue<FMT, 4, UD>(wFk[gi], L.fk.qs, L.fk.qh, L.fk.sc, g * 32 + wave * 4, F, nb, 0, lane);
            else batch_zero<FMT, 4, UD>(wFk[gi]);
        }
        if (e_has) batch_issue<FMT, 2, UD>(wFr, L.fr.qs, L.fr.qh, L.fr.sc, e_row0, D, nb, 0, lane);
        else batch_zero<FMT, 2, UD>(wFr);
        __builtin_amdgcn_sched_barrier(0);
        if (e_has) {
            const QVec la = qvec_at(l_yq, D);
            float res[2];
            rows_finish<FMT, 2, UD>(wE, L.wo.qs, L.wo.qh, L.wo.sc, e_row0, D, nb, la, lane, res);
            const float v = pick_lane<2>(res, lane);
            if (lane < 2) { xown = xown + v; tg_store(p.xatt + e_row0 + lane, __float_as_uint(xown), tagL + SLOT_XATT); }
        }
        }

        // =========================================== F ===========================================
        {
        const int tid = opq(tid0), lane = tid & 63;
        poll_units<EPT, NT>(pl, p.xatt, D, tagL + SLOT_XATT, tid, [&](int i, unsigned v) { l_x[i] = __uint_as_float(v); });
        __syncthreads();
        {
            const bool pro = tid < 256;
            double sacc = 0.0;
            if (pro) for (int i = tid; i < D; i += 256) sacc += (double) l_x[i];
            const float mean = (float) (block_sum_d_8w(sacc, red) / (double) D);
            double s2 = 0.0;
            if (pro) for (int i = tid; i < D; i += 256) { const float v = l_x[i] - mean; l_x[i] = v; s2 += (double) (v * v); }
            const float var = (float) (block_sum_d_8w(s2, red) / (double) D);
            const float scale = 1.0f / sqrtf(var + 1e-5f);
            const QVec qk = qvec_at(l_q1, D), qr = qvec_at(l_q2, D);
            auto fin = [&](int u, float & xk, float & xr) {
                const int i = tid + u * NT;
                const float y = l_x[i] * scale;
                const float yw = y * pf.lw[u];
                const float xn = yw + pf.lb[u];
                const float sx = pf.pv[u] - xn;
                const float sk = sx * pf.mk[u];
                xk = sk + xn;
                const float sr = sx * pf.mr[u];
                xr = sr + xn;
                if (blk == 0) sout_l[i] = xn;
            };
            int u0 = 0;
#pragma unroll
            for (; u0 + 3 < EPT; u0 += 4) {
                float xk[4], xr[4];
#pragma unroll
                for (int u = 0; u < 4; u++) fin(u0 + u, xk[u], xr[u]);
                int qi[4], isum[4]; float d16[4], s16[4];
                quant_blocks<4>(xk, qi, d16, s16, isum);
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = tid + (u0 + u) * NT; qvec_store(qk, nb, i >> 5, i & 31, qi[u], d16[u], s16[u], isum[u]); }
                quant_blocks<4>(xr, qi, d16, s16, isum);
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = tid + (u0 + u) * NT; qvec_store(qr, nb, i >> 5, i & 31, qi[u], d16[u], s16[u], isum[u]); }
            }
#pragma unroll
            for (; u0 < EPT; u0++) {
                float xk, xr;
                fin(u0, xk, xr);
                const int i = tid + u0 * NT;
                int qi, isum; float d16, s16;
                quant_block32(xk, qi, d16, s16, isum);
                qvec_store(qk, nb, i >> 5, i & 31, qi, d16, s16, isum);
                quant_block32(xr, qi, d16, s16, isum);
                qvec_store(qr, nb, i >> 5, i & 31, qi, d16, s16, isum);
            }
            // value-projection rows of this wave (K = F): up to 8 steps of 64 blocks in two batches
            if (e_has) batch_issue<FMT, 2, 4>(wG[0], L.fv.qs, L.fv.qh, L.fv.sc, e_row0, D, nbF, 0, lane);
            else batch_zero<FMT, 2, 4>(wG[0]);
            if (e_has && nbF > 256) batch_issue<FMT, 2, 4>(wG[1], L.fv.qs, L.fv.qh, L.fv.sc, e_row0, D, nbF, 256, lane);
            else batch_zero<FMT, 2, 4>(wG[1]);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
#pragma unroll
            for (int gi = 0; gi < 2; gi++) {
                const int g = blk * gpb + gi;
                if (gi < gpb && g < GK) {
                    float res[4];
                    rows_finish<FMT, 4, UD>(wFk[gi], L.fk.qs, L.fk.qh, L.fk.sc, g * 32 + wave * 4, F, nb, qk, lane, res);
                    const float v = pick_lane<4>(res, lane);
                    const float t = v > 0.0f ? v : 0.0f;
                    if (lane < 4) l_out[gi * 32 + wave * 4 + lane] = t * t;
                }
            }
            if (e_has) {
                float res[2];
                rows_finish<FMT, 2, UD>(wFr, L.fr.qs, L.fr.qh, L.fr.sc, e_row0, D, nb, qr, lane, res);
                const float v = pick_lane<2>(res, lane);
                if (lane < 2) tg_store(p.rr + e_row0 + lane, __float_as_uint(v), tagL + SLOT_KQ);
            }
            __syncthreads();
            if (wave == 0) {   // quantise this workgroup's key groups (relu^2 outputs): half-wave = group
                const int gi = lane >> 5;
                const int g = blk * gpb + gi;
                const bool valid = gi < gpb && g < GK;
                const float v = valid ? l_out[gi * 32 + (lane & 31)] : 0.0f;
                int qi, isum; float d16, s16;
                quant_block32(v, qi, d16, s16, isum);
                tq_store_block(p.kq, nbF, valid ? g : 0, lane & 31, qi, d16, s16, isum, tagL + SLOT_KQ, valid);
            }
        }
        }

        // =========================================== G ===========================================
        {
        const int tid = opq(tid0), lane = tid & 63;
        stage_qvec<KQU, NT>(pl, p.kq, F, tagL + SLOT_KQ, l_kq, tid);
        {
            const u64 * ptr[1] = {p.rr + e_row0 + (lane < 2 ? lane : 0)};
            const bool valid[1] = {e_has && lane < 2};
            if (e_has) poll_ptrs<1>(pl, ptr, valid, tagL + SLOT_KQ, rrv);
        }
        __syncthreads();
        issue_A(p.layers[li + 1 < p.n_layers ? li + 1 : li], li + 1 < p.n_layers ? sin_l + p.state_stride : sin_l, tid, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (e_has) {
            const QVec lk = qvec_at(l_kq, F);
            float acc[2] = {0.0f, 0.0f};
            batch_consume<FMT, 2, 4>(wG[0], nbF, 0, lane, lk, acc);
            if (nbF > 256) batch_consume<FMT, 2, 4>(wG[1], nbF, 256, lane, lk, acc);
            float res[2];
            res[0] = wave_sum_f(acc[0]); res[1] = wave_sum_f(acc[1]);
            const float v = pick_lane<2>(res, lane);
            if (lane < 2) {
                const float gte = sigmoid_f(__uint_as_float(rrv[0])) * v;
                xown = xown + gte;
                tg_store(p.xffn + e_row0 + lane, __float_as_uint(xown), tagL + SLOT_XFFN);
                p.x[e_row0 + lane] = xown;
            }
        }
        }
    }
    if (blk == 0 && tid == 0) p.ctl[0] = base + (unsigned) p.n_layers * 8u;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

struct MegaV6 {
    M6Layer * d_layers = nullptr;
    void * xch = nullptr;
    unsigned * ctl = nullptr;
    M6P proto{};
    int variant = -1, n_blocks = 0;
    size_t lds = 0;
    uint64_t bytes = 0;   // algorithmic bytes of one launch: every layer tensor once + the recurrent state read and written
};

typedef void (*MegaKernel)(M6P);
struct MegaVariant { int fmt, ept, kqu, nbd; MegaKernel fn; };
static const MegaVariant g_variants[] = {
    {T_Q4_0, 8, 9, 4, k6_mega<T_Q4_0, 8, 9, 4>},
    {T_Q4_0, 4, 5, 2, k6_mega<T_Q4_0, 4, 5, 2>},
};

static int mega_variant(const Model & m, int n_cu) {
    if (m.arch_major != 6 || m.head_size != 64 || m.layer_end <= m.layer_begin) return -1;
    const int64_t D = m.n_embed(), H = m.head_count;
    const int fmt = (int) m.header.data_type;
    const LayerW & L0 = m.layers[m.layer_begin];
    if (!L0.ffn_key || !L0.att_time_decay_w1 || !L0.att_time_maa_w1) return -1;
    const int64_t F = L0.ffn_key->ne[1], DR = L0.att_time_decay_w1->ne[1], R5 = L0.att_time_maa_w1->ne[1], R = R5 / 5;
    const int64_t NB = n_cu, W = NB * 8;
    if (H > NB || DR > NB || F % 32 != 0 || F / 32 > 2 * NB || 5 * (D / 64) > W || D / 2 > W || R > 64 || R5 > 1280 || R5 > W) return -1;
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        const DevTensor * mats[] = {L.att_receptance, L.att_key, L.att_value, L.att_gate, L.att_output, L.att_time_maa_w1,
                                    L.att_time_decay_w1, L.att_time_decay_w2, L.ffn_key, L.ffn_value, L.ffn_receptance};
        for (const DevTensor * t : mats) if (!t || t->type != fmt) return -1;
        if (L.ffn_key->ne[1] != F || L.att_time_decay_w1->ne[1] != DR || L.att_time_maa_w1->ne[1] != R5) return -1;
    }
    for (size_t v = 0; v < sizeof(g_variants) / sizeof(g_variants[0]); v++) {
        const MegaVariant & mv = g_variants[v];
        if (mv.fmt == fmt && D == mv.ept * 512 && 10 * (F / 32) <= (int64_t) mv.kqu * 512 && DR == mv.nbd * 32) return (int) v;
    }
    return -1;
}

void mega_v6_destroy(void * h) {
    MegaV6 * mg = (MegaV6 *) h;
    if (!mg) return;
    if (mg->d_layers) (void) hipFree(mg->d_layers);
    if (mg->xch) (void) hipFree(mg->xch);
    if (mg->ctl) (void) hipFree(mg->ctl);
    delete mg;
}

// Returns nullptr when the model / device does not qualify (the caller keeps the seven-launch path).
void * mega_v6_create(const Model & m) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m.device) != hipSuccess) return nullptr;
    const int NB = prop.multiProcessorCount;
    const int v = mega_variant(m, NB);
    if (v < 0) return nullptr;
    const LayerW & L0 = m.layers[m.layer_begin];
    const int64_t D = m.n_embed(), F = L0.ffn_key->ne[1], DR = L0.att_time_decay_w1->ne[1], R = L0.att_time_maa_w1->ne[1] / 5;
    MegaV6 * mg = new MegaV6();
    mg->variant = v; mg->n_blocks = NB;
    mg->lds = m6_lds((int) D, (int) F).total;
    if (mg->lds > (size_t) prop.sharedMemPerBlock && hipFuncSetAttribute((const void *) g_variants[v].fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) mg->lds) != hipSuccess) {
        delete mg; return nullptr;
    }
    (void) hipFuncSetAttribute((const void *) g_variants[v].fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) mg->lds);
    std::vector<M6Layer> hl;
    auto f = [](const DevTensor * t) { return (const float *) t->data; };
    uint64_t bytes = 0;
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        M6Layer d{};
        d.ln1_w = f(L.ln1_w); d.ln1_b = f(L.ln1_b); d.maa_x = f(L.att_time_maa_x);
        d.maa[0] = f(L.att_time_maa_w); d.maa[1] = f(L.att_time_maa_k); d.maa[2] = f(L.att_time_maa_v); d.maa[3] = f(L.att_time_maa_r); d.maa[4] = f(L.att_time_maa_g);
        d.w2t = f(L.att_time_maa_w2); d.time_decay = f(L.att_time_decay); d.faaaa = f(L.att_time_faaaa);
        d.lnx_w = f(L.att_ln_x_w); d.lnx_b = f(L.att_ln_x_b); d.ln2_w = f(L.ln2_w); d.ln2_b = f(L.ln2_b);
        d.fmaa_k = f(L.ffn_time_maa_k); d.fmaa_r = f(L.ffn_time_maa_r);
        d.w1 = planes(L.att_time_maa_w1);
        d.rkvg[0] = planes(L.att_receptance); d.rkvg[1] = planes(L.att_key); d.rkvg[2] = planes(L.att_value); d.rkvg[3] = planes(L.att_gate);
        d.dw1 = planes(L.att_time_decay_w1); d.dw2 = planes(L.att_time_decay_w2); d.wo = planes(L.att_output);
        d.fk = planes(L.ffn_key); d.fr = planes(L.ffn_receptance); d.fv = planes(L.ffn_value);
        hl.push_back(d);
        const DevTensor * all[] = {L.ln1_w, L.ln1_b, L.att_time_maa_x, L.att_time_maa_w, L.att_time_maa_k, L.att_time_maa_v, L.att_time_maa_r, L.att_time_maa_g,
                                   L.att_time_maa_w1, L.att_time_maa_w2, L.att_time_decay, L.att_time_faaaa, L.att_time_decay_w1, L.att_time_decay_w2,
                                   L.att_receptance, L.att_key, L.att_value, L.att_gate, L.att_output, L.att_ln_x_w, L.att_ln_x_b, L.ln2_w, L.ln2_b,
                                   L.ffn_time_maa_k, L.ffn_time_maa_r, L.ffn_key, L.ffn_value, L.ffn_receptance};
        for (const DevTensor * t : all) if (t) bytes += t->nbytes;
        bytes += 2 * (uint64_t) m.state_per_layer() * sizeof(float);
    }
    mg->bytes = bytes;
    const int64_t nbD = D / 32, nbF = F / 32;
    const int64_t act_stride = (10 * nbD + 31) / 32 * 32;
    const int64_t units = 1280 + 5 * act_stride + 4 * D + 256 + act_stride + D + (10 * nbF + 31) / 32 * 32 + D + D;
    bool ok = hipMalloc((void **) &mg->d_layers, hl.size() * sizeof(M6Layer)) == hipSuccess
           && hipMemcpy(mg->d_layers, hl.data(), hl.size() * sizeof(M6Layer), hipMemcpyHostToDevice) == hipSuccess
           && hipMalloc(&mg->xch, (size_t) units * 8) == hipSuccess && hipMemset(mg->xch, 0, (size_t) units * 8) == hipSuccess
           && hipMalloc((void **) &mg->ctl, 256) == hipSuccess;
    const unsigned init[2] = {8u, 0u};
    ok = ok && hipMemcpy(mg->ctl, init, sizeof(init), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { mega_v6_destroy(mg); return nullptr; }
    M6P & q = mg->proto;
    q.layers = mg->d_layers; q.n_layers = (int) hl.size();
    q.state_stride = m.state_per_layer();
    u64 * u = (u64 *) mg->xch;
    q.tl = u; u += 1280;
    q.act5 = u; u += 5 * act_stride; q.act_stride = act_stride;
    q.rkvg = u; u += 4 * D;
    q.dl = u; u += 256;
    q.yq = u; u += act_stride;
    q.xatt = u; u += D;
    q.kq = u; u += (10 * nbF + 31) / 32 * 32;
    q.rr = u; u += D;
    q.xffn = u; u += D;
    q.ctl = mg->ctl;
    q.F = (int) F; q.DR = (int) DR; q.R = (int) R; q.H = (int) m.head_count;
    q.gpb = (int) ((nbF + NB - 1) / NB);
    return mg;
}

uint64_t mega_v6_bytes(void * h) { return ((MegaV6 *) h)->bytes; }

// sin / sout: state of the stage's FIRST layer. One launch covers every layer of the stage.
void mega_v6_forward(void * h, float * x, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf) {
    MegaV6 * mg = (MegaV6 *) h;
    M6P q = mg->proto;
    q.x = x; q.sin = sin; q.sout = sout;
    const MegaKernel fn = g_variants[mg->variant].fn;
    if (pf && pf->on) {
        if (pf->used * 2 + 2 > pf->events.size()) {
            hipEvent_t a = nullptr, c = nullptr;
            (void) hipEventCreate(&a); (void) hipEventCreate(&c);
            pf->events.push_back(a); pf->events.push_back(c); pf->bytes.push_back(0);
        }
        pf->bytes[pf->used] = mg->bytes;
        hipExtLaunchKernelGGL(fn, dim3((unsigned) mg->n_blocks), dim3(512), (uint32_t) mg->lds, st, pf->events[pf->used * 2], pf->events[pf->used * 2 + 1], 0, q);
        pf->used++;
    } else {
        hipLaunchKernelGGL(fn, dim3((unsigned) mg->n_blocks), dim3(512), mg->lds, st, q);
    }
}

// true when a poll timed out in some launch since creation (co-residency lost or a bug): results are not valid
bool mega_v6_aborted(void * h) {
    unsigned c[2] = {0, 0};
    if (hipMemcpy(c, ((MegaV6 *) h)->ctl, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) return true;
    return c[1] != 0;
}

}  // namespace rwkvmi
