// mega_v6.hip -- the RWKV-6 single-token (decode) step over ALL layers of a stage as ONE persistent launch.
//
// Why: at batch 1 every phase of a layer (fused_v6.hip's seven launches) is an all-to-all dependency, and a launch boundary
// costs ~5 us of ramp + drain during which no weight bytes move (profiles/). Here one workgroup per CU
// stays resident for the whole token:
//   * every wave owns a FIXED slice of every matrix (rows), so the weights of the next phase are put in flight into
//     registers as soon as the current phase's rows are done -- weights do not depend on activations, only the order of
//     use does -- and stream while the activations are handed over;
//   * phases are chained by DATAFLOW, not by barriers: every value that crosses workgroups is a 16-byte {payload, tag}
//     unit written with one store and polled until the tag of the expected (token, layer, phase) shows up. No fence, no
//     counter, no wait for outstanding loads on the producer side.
// Vector-memory results return in order per wave, so a wave that polls cannot have a weight prefetch in flight: inside a
// workgroup wave 0 (the "comm" wave) does all the polling into LDS and the other seven never wait on anything but the
// workgroup barrier and their own loads. A grid-wide barrier costs 3.2-3.9 us on this part, a tagged hand-over of a
// 4096-float vector to all 256 workgroups 3.5 us, of a small vector 1.5-2.2 us (tools/barrier_bench.hip, xchg_bench.hip).
//
// Arithmetic, reduction orders and epilogues are those of fused_v6.hip / kernels.hip: the results are
// bit-identical to the seven-launch path and to the CPU oracle; only the distribution of rows over waves differs.
//
// Residency: the grid is one workgroup per CU (512 threads, ~105 KB LDS) and all of them must be resident at once, i.e. the
// device must not be shared with another process' persistent kernels. Polls are bounded: on timeout the abort word is set,
// every workgroup drains without waiting, and the host reports the step as failed (it never hangs the device).
#include "persist.h"

#include <cstring>

namespace rwkvmi {

struct M6P {
    const M6Layer * layers; int n_layers;
    const unsigned char * arena;
    const float * w2b;                               // W2 of every layer in the chunk-blocked layout (see k_block_w2)
    float * x;                                       // plain residual stream: input of the first layer, output of the last
    const float * sin; float * sout; long long state_stride;
    void * xch; unsigned xch_bytes;                   // the exchange arena: every tagged buffer lives in it ...
    int tl, act5, rkvg, dl, yq, xatt, kq, rr, xffn;   // ... at these unit (16-byte) offsets
    long long act_stride;                            // units between the five mix images
    unsigned * ctl;                                  // [0] tag generation, [1] abort
    int F, DR, R, H, gpb;
    long long * trace; int trace_layer;
};

// ---------------------------------------------------------------------------------------------------------------
// the kernel. EPT = D / 512, KQU = ceil(10 (F/32) / 64) poll slots per lane for the F-vector, NBD = decay rank / 32,
// GPB = key groups (32 rows of ffn.key) per workgroup = ceil((F/32) / 256). The grid is exactly NBLK workgroups.
//
// Roles inside a workgroup (8 waves):
//   wave 0      "comm": polls every tagged input of the workgroup into LDS, runs the WKV head (phase D) of workgroups
//               0..H-1 and quantises the key groups. It never has weight loads in flight, so its polls return at
//               memory latency instead of queueing behind a prefetch (vector-memory results return in order per wave).
//   waves 1..7  "workers": own fixed rows of every matrix, keep the NEXT phase's weights in flight, and only ever wait
//               on the workgroup barrier and on their own loads. They never poll.
// The two roles run different code (two loops over the layers) with the same sequence of workgroup barriers.
// ---------------------------------------------------------------------------------------------------------------

struct M6Lds { size_t x, xn, sx, q1, q2, act, actw, yq, kq, tl, red, out, rr, dl, total; };
__host__ __device__ inline M6Lds m6_lds(int D, int F) {
    M6Lds o; size_t p = 0;
    auto take = [&](size_t n) { const size_t r = p; p += m6_round16(n); return r; };
    o.x = take((size_t) D * 4); o.xn = take((size_t) D * 4); o.sx = take((size_t) D * 4);
    o.q1 = take(qvec_bytes(D)); o.q2 = take(qvec_bytes(D)); o.act = take(qvec_bytes(D)); o.actw = take(qvec_bytes(D)); o.yq = take(qvec_bytes(D));
    o.kq = take(qvec_bytes(F)); o.tl = take(1280 * 4); o.red = take(2 * 256 * 8); o.out = take(2 * 32 * 4); o.rr = take(64 * 4); o.dl = take(192);
    o.total = p;
    return o;
}

// RSTAMP: the 100 MHz real-time counter, consistent across XCDs (hand-over latencies); STAMP: the shader clock (phase lengths)
#define RSTAMP(K) do { if (p.trace && li == p.trace_layer && (tidst & 63) == 0) p.trace[((long long) blockIdx.x * 8 + (tidst >> 6)) * 32 + (K)] = (long long) __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP(K) do { if (p.trace && li == p.trace_layer && (tidst & 63) == 0) p.trace[((long long) blockIdx.x * 8 + (tidst >> 6)) * 32 + (K)] = (long long) __builtin_readcyclecounter(); } while (0)

template <int FMT, int EPT, int KQU, int NBD, int GPB>
struct K6 {
    static constexpr int NT = 512, S = 64, NBLK = 256, NWK = 7;
    static constexpr int D = EPT * NT;
    static constexpr int nb = D / 32;
    static constexpr int UD = nb / 64 > 0 ? nb / 64 : 1;
    static constexpr int V4 = D / (4 * NT);              // float4 groups per thread in the prologues
    static constexpr int NOWN = 8;                       // row owners per workgroup: the 7 workers and the comm wave
    static constexpr int RPB_C = 4 * D / NBLK;           // r/k/v/g rows per workgroup (all of one matrix)
    static constexpr int NSC = RPB_C / 2 / NOWN;         // 2-row sets per owner
    static constexpr int RPB_E = D / NBLK;               // output / receptance / value rows per workgroup
    static constexpr int NSE = RPB_E / NOWN;             // rows per owner (consecutive)
    static constexpr int NSK = GPB * 16 / NOWN;          // key 2-row sets per owner
    static constexpr int XU = NBLK * NOWN / 64;          // poll slots per lane for x (one unit per owner: its rows)
    static constexpr int DU = (3 * nb + 63) / 64;        // ... for a quantised D-vector
    static_assert(RPB_C % (2 * NOWN) == 0 && RPB_E % NOWN == 0 && (GPB * 16) % NOWN == 0 && NSE <= 3, "row ownership must divide evenly");
    // sink of an x-like vector: unit i = (workgroup, owner) -> its rows of dst
    static __device__ __forceinline__ void x_sink(float * dst, int i, const v4u & v) {
        const int st = (i / NOWN) * RPB_E + (i % NOWN) * NSE;
        dst[st] = __uint_as_float(v.x);
        if (NSE > 1) dst[st + 1] = __uint_as_float(v.y);
        if (NSE > 2) dst[st + 2] = __uint_as_float(v.z);
    }

    struct Lds {
        float *x, *xn, *sx, *tl, *out, *rr;
        unsigned char *q1, *q2, *act, *actw, *yq, *kq, *dl;
        double * red;
    };

    struct PA { float4 lw[V4], lb[V4], pv[V4], mx[V4]; };
    struct PF { float4 lw[V4], lb[V4], pv[V4], mk[V4], mr[V4]; };

    // LayerNorm statistics of the row in l.x (256 partials, threads 0..255); leaves x - mean in l.x
    // block_sum_d_1b (kdev.h): ONE workgroup barrier per reduction instead of three. A workgroup barrier costs ~0.5 us here
    // (the eight waves arrive unevenly): 8 fewer per layer were worth 4 %.
    static __device__ __forceinline__ float ln_stats(const Lds & l, int tid) {
        // Thread t < 256 owns the partial over elements t, t + 256, ... (summed in that order). The D / 256 values are read
        // into registers in one batch: as a rolled loop every iteration waited for its own LDS read (~130 cycles x 16 x 2).
        constexpr int NP = D / 256;
        const bool pro = tid < 256;
        float xv[NP];
        if (pro) {
#pragma unroll
            for (int j = 0; j < NP; j++) xv[j] = l.x[tid + 256 * j];
        }
        double sacc = 0.0;
        if (pro) {
#pragma unroll
            for (int j = 0; j < NP; j++) sacc += (double) xv[j];
        }
        const float mean = (float) (block_sum_d_1b(sacc, l.red) / (double) D);
        double s2 = 0.0;
        if (pro) {
#pragma unroll
            for (int j = 0; j < NP; j++) { const float v = xv[j] - mean; l.x[tid + 256 * j] = v; s2 += (double) (v * v); }
        }
        const float var = (float) (block_sum_d_1b(s2, l.red + 256) / (double) D);
        return 1.0f / sqrtf(var + 1e-5f);
    }

    static __device__ __forceinline__ void issue_pa(PA & pa, const M6Arena & ar, const M6Layer & L, const float * sin_l, int tid) {
        const float * ln1_w = ar.f(L.ln1_w), * ln1_b = ar.f(L.ln1_b), * maa_x = ar.f(L.maa_x);
#pragma unroll
        for (int u = 0; u < V4; u++) {
            const int i = tid * 4 + u * 4 * NT;
            pa.lw[u] = *reinterpret_cast<const float4 *>(ln1_w + i); pa.lb[u] = *reinterpret_cast<const float4 *>(ln1_b + i);
            pa.pv[u] = *reinterpret_cast<const float4 *>(sin_l + D + i); pa.mx[u] = *reinterpret_cast<const float4 *>(maa_x + i);
        }
    }
    static __device__ __forceinline__ void issue_pf(PF & pf, const M6Arena & ar, const M6Layer & L, const float * sin_l, int tid) {
        const float * ln2_w = ar.f(L.ln2_w), * ln2_b = ar.f(L.ln2_b), * fmaa_k = ar.f(L.fmaa_k), * fmaa_r = ar.f(L.fmaa_r);
#pragma unroll
        for (int u = 0; u < V4; u++) {
            const int i = tid * 4 + u * 4 * NT;
            pf.lw[u] = *reinterpret_cast<const float4 *>(ln2_w + i); pf.lb[u] = *reinterpret_cast<const float4 *>(ln2_b + i);
            pf.pv[u] = *reinterpret_cast<const float4 *>(sin_l + i);
            pf.mk[u] = *reinterpret_cast<const float4 *>(fmaa_k + i); pf.mr[u] = *reinterpret_cast<const float4 *>(fmaa_r + i);
        }
    }

    // A: LN1 + token shift + maa_x mix + quantise, every workgroup redundantly (all 8 waves). Barriers: 6 + 1.
    static __device__ __forceinline__ void prologue_A(const Lds & l, const PA & pa, float * sout_l, bool write_state, int tid) {
        const float scale = ln_stats(l, tid);
        const QVec lq = qvec_at(l.q1, D);
#pragma unroll
        for (int u = 0; u < V4; u++) {
            const int i = tid * 4 + u * 4 * NT;
            const float4 xc = *reinterpret_cast<const float4 *>(l.x + i);
            const float xs[4] = {xc.x, xc.y, xc.z, xc.w};
            const float lw[4] = {pa.lw[u].x, pa.lw[u].y, pa.lw[u].z, pa.lw[u].w}, lb[4] = {pa.lb[u].x, pa.lb[u].y, pa.lb[u].z, pa.lb[u].w};
            const float pv[4] = {pa.pv[u].x, pa.pv[u].y, pa.pv[u].z, pa.pv[u].w}, mx[4] = {pa.mx[u].x, pa.mx[u].y, pa.mx[u].z, pa.mx[u].w};
            float xn[4], sx[4], xxx[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float y = xs[j] * scale;
                const float yw = y * lw[j];
                xn[j] = yw + lb[j];
                sx[j] = pv[j] - xn[j];
                const float sm = sx[j] * mx[j];
                xxx[j] = sm + xn[j];
            }
            *reinterpret_cast<float4 *>(l.xn + i) = make_float4(xn[0], xn[1], xn[2], xn[3]);
            *reinterpret_cast<float4 *>(l.sx + i) = make_float4(sx[0], sx[1], sx[2], sx[3]);
            if (write_state) *reinterpret_cast<float4 *>(sout_l + D + i) = make_float4(xn[0], xn[1], xn[2], xn[3]);
            unsigned packed; float d16, s16; int isum;
            quant_vec4(xxx, packed, d16, s16, isum);
            qvec_store4(lq, nb, i, packed, d16, s16, isum);
        }
        __syncthreads();
    }

    // F: LN2 + token shift + the two mixes + quantise (all 8 waves). Barriers: 6 + 1.
    static __device__ __forceinline__ void prologue_F(const Lds & l, const PF & pf, float * sout_l, bool write_state, int tid) {
        const float scale = ln_stats(l, tid);
        const QVec qk = qvec_at(l.q1, D), qr = qvec_at(l.q2, D);
#pragma unroll
        for (int u = 0; u < V4; u++) {
            const int i = tid * 4 + u * 4 * NT;
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
            quant_vec4(xr, packed, d16, s16, isum);
            qvec_store4(qr, nb, i, packed, d16, s16, isum);
        }
        __syncthreads();
    }

    // -----------------------------------------------------------------------------------------------------------
    // the big row phases, shared by the 8 row owners of a workgroup (workers own = 0..6, comm wave own = 7)
    // -----------------------------------------------------------------------------------------------------------
    struct Rows {
        Batch<FMT, 2, UD> wC[NSC];
        Batch<FMT, 1, UD> wE[NSE], wFr[NSE];
        Batch<FMT, 2, UD> wFk[NSK];
        Batch<FMT, 1, 4> wG[NSE][2];
        float xown[NSE];
    };
    static __device__ __forceinline__ int c_mat() { return ((int) blockIdx.x * RPB_C) / D; }
    static __device__ __forceinline__ int c_base() { return ((int) blockIdx.x * RPB_C) % D; }

    // part 0: the first half of the owner's row sets, part 1: the second half (issued once the activation image has been staged: the
    // comm wave's polls for it then queue behind half the burst; the second half lands under the first half's arithmetic); part 2: all
    static __device__ __forceinline__ void issue_C(Rows & r, const M6Arena & ar, const M6Layer & L, int own, int lane, int part = 2) {
        const WPl w = ar.w(L.rkvg[c_mat()]);
#pragma unroll
        for (int si = 0; si < NSC; si++) {
            if (part == 2 || (part == 0) == (si < NSC / 2))
                batch_issue<FMT, 2, UD>(r.wC[si], w.qs, w.qh, w.sc, c_base() + 2 * (own + si * NOWN), D, nb, 0, lane);
        }
    }
    static __device__ __forceinline__ void compute_C(Rows & r, const Lds & l, xrsrc xr, const M6P & p, unsigned tagL, int own, int lane) {
        const QVec la = qvec_at(l.act, D);
        const int mat = c_mat();
        // all row sums first, then ONE epilogue with lane 2 si + r finishing row r of set si (the gate's silu is a double-
        // precision exp: four of them back to back put the 64 gate workgroups behind everyone else)
        float all[2 * NSC];
#pragma unroll
        for (int si = 0; si < NSC; si++) {
            float res[2];
            rows_finish<FMT, 2, UD>(r.wC[si], nullptr, nullptr, nullptr, 0, D, nb, la, lane, res);
            all[2 * si] = res[0]; all[2 * si + 1] = res[1];
        }
        float v = pick_lane<2 * NSC>(all, lane);
        if (mat == 3) v = v / (1.0f + det_expf(-v));
        const int v1 = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xF, 0xF, true);   // lane 2 si collects row 1 (row_shl:1)
        if (lane < 2 * NSC && (lane & 1) == 0) {
            const int row0 = c_base() + 2 * (own + (lane >> 1) * NOWN);
            tg_store(xr, p.rkvg + ((mat * D + row0) >> 1), __float_as_uint(v), (unsigned) v1, 0u, 0u, tagL + SLOT_RKVG);
        }
    }
    static __device__ __forceinline__ void issue_E(Rows & r, const M6Arena & ar, const M6Layer & L, int own, int lane) {
        const WPl w = ar.w(L.wo);
#pragma unroll
        for (int si = 0; si < NSE; si++) batch_issue<FMT, 1, UD>(r.wE[si], w.qs, w.qh, w.sc, (int) blockIdx.x * RPB_E + own * NSE + si, D, nb, 0, lane);
    }
    static __device__ __forceinline__ void x_store(xrsrc xr, int buf, const float (&v)[NSE], int own, int lane, unsigned tag) {
        if (lane == 0) tg_store(xr, buf + (int) blockIdx.x * NOWN + own, __float_as_uint(v[0]), __float_as_uint(v[NSE > 1 ? 1 : 0]), __float_as_uint(v[NSE > 2 ? 2 : 0]), 0u, tag);
    }
    static __device__ __forceinline__ void compute_E(Rows & r, const Lds & l, xrsrc xr, const M6P & p, unsigned tagL, int own, int lane) {
        const QVec la = qvec_at(l.yq, D);
#pragma unroll
        for (int si = 0; si < NSE; si++) {
            float res[1];
            rows_finish<FMT, 1, UD>(r.wE[si], nullptr, nullptr, nullptr, 0, D, nb, la, lane, res);
            r.xown[si] = r.xown[si] + res[0];
        }
        x_store(xr, p.xatt, r.xown, own, lane, tagL + SLOT_XATT);
    }
    static __device__ __forceinline__ bool k_valid(const M6P & p, int own, int si) { return (int) blockIdx.x * GPB * 32 + 2 * (own + si * NOWN) < p.F; }
    static __device__ __forceinline__ void issue_Fk(Rows & r, const M6P & p, const M6Arena & ar, const M6Layer & L, int own, int lane) {
        const WPl w = ar.w(L.fk);
#pragma unroll
        for (int si = 0; si < NSK; si++) batch_issue_opt<FMT, 2, UD>(k_valid(p, own, si), r.wFk[si], w, (int) blockIdx.x * GPB * 32 + 2 * (own + si * NOWN), p.F, nb, 0, lane);
    }
    static __device__ __forceinline__ void issue_Fr(Rows & r, const M6Arena & ar, const M6Layer & L, int own, int lane) {
        const WPl w = ar.w(L.fr);
#pragma unroll
        for (int si = 0; si < NSE; si++) batch_issue<FMT, 1, UD>(r.wFr[si], w.qs, w.qh, w.sc, (int) blockIdx.x * RPB_E + own * NSE + si, D, nb, 0, lane);
    }
    static __device__ __forceinline__ void compute_F(Rows & r, const Lds & l, xrsrc xr, const M6P & p, unsigned tagL, int own, int lane) {
        const QVec qk = qvec_at(l.q1, D), qr = qvec_at(l.q2, D);
#pragma unroll
        for (int si = 0; si < NSK; si++) {
            if (k_valid(p, own, si)) {
                float res[2];
                rows_finish<FMT, 2, UD>(r.wFk[si], nullptr, nullptr, nullptr, 0, p.F, nb, qk, lane, res);
                const float v = pick_lane<2>(res, lane);
                const float t = v > 0.0f ? v : 0.0f;
                if (lane < 2) l.out[2 * (own + si * NOWN) + lane] = t * t;
            }
        }
        float rrow[NSE];
#pragma unroll
        for (int si = 0; si < NSE; si++) {
            float res[1];
            rows_finish<FMT, 1, UD>(r.wFr[si], nullptr, nullptr, nullptr, 0, D, nb, qr, lane, res);
            rrow[si] = res[0];
        }
        x_store(xr, p.rr, rrow, own, lane, tagL + SLOT_KQ);
    }
    // part 0: the first four block-steps of every row, part 1: the rest. Part 1 goes out only after the k hand-over has been staged: the
    // comm wave's polls for k then queue behind half the burst in the CU's memory pipe, and the second half lands under the first
    // half's arithmetic.
    static __device__ __forceinline__ void issue_G(Rows & r, const M6P & p, const M6Arena & ar, const M6Layer & L, int own, int lane, int part) {
        const WPl w = ar.w(L.fv);
        const int nbF = p.F / 32;
#pragma unroll
        for (int si = 0; si < NSE; si++) {
            const int row = (int) blockIdx.x * RPB_E + own * NSE + si;
            if (part == 0) batch_issue<FMT, 1, 4>(r.wG[si][0], w.qs, w.qh, w.sc, row, D, nbF, 0, lane);
            else batch_issue_opt<FMT, 1, 4>(nbF > 256, r.wG[si][1], w, row, D, nbF, 256, lane);
        }
    }
    static __device__ __forceinline__ void compute_G(Rows & r, const Lds & l, xrsrc xr, const M6P & p, unsigned tagL, int own, int lane) {
        const int nbF = p.F / 32;
        const QVec lk = qvec_at(l.kq, p.F);
#pragma unroll
        for (int si = 0; si < NSE; si++) {
            float acc[1] = {0.0f};
            batch_consume<FMT, 1, 4>(r.wG[si][0], nbF, 0, lane, lk, acc);
            if (nbF > 256) batch_consume<FMT, 1, 4>(r.wG[si][1], nbF, 256, lane, lk, acc);
            const float v = wave_sum_f(acc[0]);
            const float gte = sigmoid_f(l.rr[own * NSE + si]) * v;
            r.xown[si] = r.xown[si] + gte;
            if (lane == 0) p.x[(int) blockIdx.x * RPB_E + own * NSE + si] = r.xown[si];
        }
        x_store(xr, p.xffn, r.xown, own, lane, tagL + SLOT_XFFN);
    }
    static __device__ __forceinline__ void load_xown(Rows & r, const M6P & p, int own) {
#pragma unroll
        for (int si = 0; si < NSE; si++) r.xown[si] = p.x[(int) blockIdx.x * RPB_E + own * NSE + si];
    }

    // -----------------------------------------------------------------------------------------------------------
    // comm wave
    // -----------------------------------------------------------------------------------------------------------
    static __device__ __forceinline__ void comm_main(const M6P & p, const Lds & l, int lane0, unsigned base) {
        const int blk = blockIdx.x;
        const int F = p.F, DR = p.DR, R = p.R, H = p.H;
        const int nbF = F / 32;
        Poll pl{p.ctl, false};
        const M6Arena ar{p.arena};
        const xrsrc xr = make_xrsrc(p.xch, p.xch_bytes);
        // which activation images this workgroup's workers read in C
        const int blk_mat = (blk * RPB_C) / D;
        const int blk_act = (0x4213 >> (4 * blk_mat)) & 0xF;   // r,k,v,g -> mix image (w,k,v,r,g order)
        bool blk_xhas = false;                                 // some worker here owns a decay-W1 row
        for (int wk = 0; wk < NWK; wk++) { const int g = wk * NBLK + blk - 5 * R; blk_xhas = blk_xhas || (g >= 0 && g < DR); }
        const bool d_has = blk < H;
        const int d_head = blk;
        // B: 64-element chunks of the five mixes; chunk c < NBLK on workgroup c, the rest on workgroups NBLK/4.. (the first
        // quarter runs the WKV heads, the last quarter the gate rows with their silu epilogue)
        constexpr int NCH = 5 * (D / 64);
        const int b_extra = NCH - NBLK;   // chunks beyond one per workgroup (host guarantees <= NBLK)
        const int b_chunk2 = (b_extra > 0 && blk >= NBLK / 4 && blk < NBLK / 4 + b_extra) ? NBLK + (blk - NBLK / 4) : -1;
        PA pa; PF pf;
        constexpr int own = NOWN - 1;   // the comm wave is the eighth row owner: its weights go in flight right before the poll that precedes their use
        Rows r;
        load_xown(r, p, own);
        issue_pa(pa, ar, p.layers[0], p.sin, lane0);

        for (int li = 0; li < p.n_layers; li++) {
            const M6Layer & L = p.layers[li];
            const float * sin_l = p.sin + (long long) li * p.state_stride;
            float * sout_l = p.sout + (long long) li * p.state_stride;
            const unsigned tagL = base + (unsigned) li * 8u;
            const int lane = opq(lane0) & 63;
            const int tidst = lane0;
            STAMP(0);

            // ---- A ----
            if (li == 0) {
#pragma unroll 8
                for (int u = 0; u < D / 64; u++) l.x[lane + u * 64] = p.x[lane + u * 64];
            } else {
                poll_units<XU, 64>(pl, xr, p.xffn, NBLK * NOWN, tagL - 8u + SLOT_XFFN, lane, [&](int i, const v4u & v) { x_sink(l.x, i, v); });
            }
            // W2 of this workgroup's chunk(s): chunk-blocked copy, lane d reads float4 {m .. m+3}; in flight across the prologue
            float4 wB4[2][16]; float wBmaa[2];
            int bf[2], bd[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int ch = q == 0 ? (blk < NCH ? blk : 0) : (b_chunk2 >= 0 ? b_chunk2 : 0);
                bf[q] = ch / (D / 64);
                bd[q] = (ch % (D / 64)) * 64 + lane;
                const float4 * cb = reinterpret_cast<const float4 *>(p.w2b + L.w2b + (long long) ch * R * 64);
#pragma unroll
                for (int m4 = 0; m4 < 16; m4++) { const int4 t = ldw16(cb + (m4 < R / 4 ? m4 : R / 4 - 1) * 64 + lane); wB4[q][m4] = make_float4(__int_as_float(t.x), __int_as_float(t.y), __int_as_float(t.z), __int_as_float(t.w)); }
                wBmaa[q] = ar.f(L.maa[bf[q]])[bd[q]];
            }
            __builtin_amdgcn_sched_barrier(0);
            STAMP(1); RSTAMP(17);
            __syncthreads();
            prologue_A(l, pa, sout_l, blk == 0, lane);
            STAMP(2);
            // ---- B: the data-dependent mixes of this workgroup's chunk(s) (rwkv_graph.inc:313-346) ----
            {
                // (the W2 columns of this workgroup's chunk(s) were put in flight before the prologue)
                __builtin_amdgcn_sched_barrier(0);
                poll_units<5, 64>(pl, xr, p.tl, 5 * R, tagL + SLOT_TL, lane, [&](int i, const v4u & v) { l.tl[i] = __uint_as_float(v.x); });
                __builtin_amdgcn_wave_barrier();
                STAMP(3); RSTAMP(18);
                __syncthreads();   // releases the workers' r/k/v/g stream
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const bool has = q == 0 ? blk < NCH : b_chunk2 >= 0;
                    if (has) {
                        const float * tlf = l.tl + bf[q] * R;
                        // R is 32 or 64: two straight-line halves, each reading its 32 tl values as eight 16-byte LDS loads up
                        // front. (A per-term `if (m < R)` compiled into 64 basic blocks, and a per-term scalar LDS read into 64
                        // separate waits on the LDS counter: 4 us and 1.7 us per chunk instead of 0.5.)
                        const float4 * tl4 = reinterpret_cast<const float4 *>(tlf);
                        float acc = 0.0f;
                        {
                            float4 t4[8];
#pragma unroll
                            for (int j = 0; j < 8; j++) t4[j] = tl4[j];
#pragma unroll
                            for (int m = 0; m < 32; m++) acc += (&wB4[q][m >> 2].x)[m & 3] * (&t4[m >> 2].x)[m & 3];
                        }
                        if (R > 32) {
                            float4 t4[8];
#pragma unroll
                            for (int j = 0; j < 8; j++) t4[j] = tl4[8 + j];
#pragma unroll
                            for (int m = 0; m < 32; m++) acc += (&wB4[q][8 + (m >> 2)].x)[m & 3] * (&t4[m >> 2].x)[m & 3];
                        }
                        const float mm = (acc + wBmaa[q]) * l.sx[bd[q]];
                        const float o = mm + l.xn[bd[q]];
                        int qi, isum; float d16, s16;
                        quant_block32(o, qi, d16, s16, isum);
                        tq_store_block(xr, p.act5 + bf[q] * (int) p.act_stride, bd[q] >> 5, lane & 31, qi, d16, s16, isum, tagL + SLOT_ACT);
                    }
                }
            }
            RSTAMP(19);
            // ---- C ----
            issue_C(r, ar, L, own, lane);
            stage_qvec<DU, 64>(pl, xr, p.act5 + blk_act * (int) p.act_stride, D, tagL + SLOT_ACT, l.act, lane);
            if (blk_xhas) stage_qvec<DU, 64>(pl, xr, p.act5, D, tagL + SLOT_ACT, l.actw, lane);
            STAMP(4); RSTAMP(20);
            __syncthreads();
            compute_C(r, l, xr, p, tagL, own, lane);
            RSTAMP(21);
            // ---- D: WKV head of this workgroup ----
            if (d_has) {
                const int c = d_head * S + lane;
                RawBlk<FMT> w2[NBD];
                const WPl dw2 = ar.w(L.dw2);
#pragma unroll
                for (int b = 0; b < NBD; b++) load_raw<FMT>(w2[b], dw2.qs, dw2.qh, dw2.sc, (long long) c * NBD + b);
                const float td = ar.f(L.time_decay)[c], uu = ar.f(L.faaaa)[c], lnw = ar.f(L.lnx_w)[c], lnb = ar.f(L.lnx_b)[c];
                // the head's state column goes in flight together with the decay-W2 blocks, BEFORE the dl poll: one memory round trip
                // for both instead of two on this chain (the exp routine's constants live in scalar registers now, so the 64 extra live
                // registers no longer push them into scratch)
                float s[S];
                const float * st = sin_l + 2 * D + (long long) d_head * S * S;
#pragma unroll
                for (int i = 0; i < S; i++) s[i] = st[i * S + lane];
                __builtin_amdgcn_sched_barrier(0);
                // dl arrives first (the decay rows are the first thing the workers finish): one unit per value
                unsigned dq[6];
                {
                    const int ptr[2] = {p.dl + lane, p.dl + (NBD > 2 ? 64 + lane : lane)};
                    const bool valid[2] = {true, NBD > 2};
                    v4u dv[2];
                    poll_ptrs<2>(pl, xr, ptr, valid, tagL + SLOT_RKVG, dv);
                    dq[4] = dv[0].x; dq[5] = dv[1].x;
                }
                // speculative first read of this channel's r / k / v / g units: in flight while the decay is computed
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
                // r,k,v,g of channel c: one unit per 2-row set. A first read was put in flight before the decay was computed (see above): when
                // the four producers were done by then -- the usual case, the gate rows' silu aside -- its result is used and the chain saves a
                // memory round trip; otherwise the ordinary poll takes over.
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
                // 3. WKV6 (ggml_rwkv_wkv6): lane j owns value column j; {k, u, r, w}_i are broadcast through LDS (l.tl is free here)
                float4 * bc = reinterpret_cast<float4 *>(l.tl);
                bc[lane] = make_float4(__uint_as_float(dq[1]), uu, __uint_as_float(dq[0]), wdec);
                __builtin_amdgcn_wave_barrier();
                const float vj = __uint_as_float(dq[2]);
                float o = 0.0f;
                float * so = sout_l + 2 * D + (long long) d_head * S * S;
#pragma unroll
                for (int i0 = 0; i0 < S; i0 += 8) {
                    float4 b8[8];   // eight broadcast reads in flight, not one LDS round trip per step
#pragma unroll
                    for (int u = 0; u < 8; u++) b8[u] = bc[i0 + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int i = i0 + u;
                        const float kv = vj * b8[u].x;
                        const float prev = s[i];
                        const float temp = kv * b8[u].y + prev;
                        o += temp * b8[u].z;
                        so[i * S + lane] = prev * b8[u].w + kv;   // (deferring these 64 stores behind the yq hand-over was tried: the comm wave's next polls then queue behind them)
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
                tq_store_block(xr, p.yq, 2 * d_head + (lane >> 5), lane & 31, qi, d16, s16, isum, tagL + SLOT_YQ);
            }
            STAMP(5); RSTAMP(22);
            // ---- E ----
            issue_E(r, ar, L, own, lane);
            stage_qvec<DU, 64>(pl, xr, p.yq, D, tagL + SLOT_YQ, l.yq, lane);
            STAMP(6); RSTAMP(23);
            __syncthreads();
            compute_E(r, l, xr, p, tagL, own, lane);
            RSTAMP(24);
            issue_pf(pf, ar, L, sin_l, lane0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- F ----
            poll_units<XU, 64>(pl, xr, p.xatt, NBLK * NOWN, tagL + SLOT_XATT, lane, [&](int i, const v4u & v) { x_sink(l.x, i, v); });
            issue_Fk(r, p, ar, L, own, lane);
            issue_Fr(r, ar, L, own, lane);
            STAMP(7); RSTAMP(25);
            __syncthreads();
            prologue_F(l, pf, sout_l, blk == 0, lane);
            STAMP(8);
            compute_F(r, l, xr, p, tagL, own, lane);
            __syncthreads();   // every owner's key rows are in l.out
            STAMP(9);
            {
                // quantise this workgroup's key groups (relu^2 outputs): half-wave = group
#pragma unroll
                for (int g2 = 0; g2 < (GPB + 1) / 2; g2++) {
                    const int gi = g2 * 2 + (lane >> 5);
                    const int g = blk * GPB + gi;
                    const bool valid = gi < GPB && g < nbF;
                    const float v = valid ? l.out[gi * 32 + (lane & 31)] : 0.0f;
                    int qi, isum; float d16, s16;
                    quant_block32(v, qi, d16, s16, isum);
                    tq_store_block(xr, p.kq, valid ? g : 0, lane & 31, qi, d16, s16, isum, tagL + SLOT_KQ, valid);
                }
            }
            STAMP(10); RSTAMP(26);
            // ---- G ----
            {
                const int nl = li + 1 < p.n_layers ? li + 1 : li;
                issue_pa(pa, ar, p.layers[nl], p.sin + (long long) nl * p.state_stride, lane0);
            }
            issue_G(r, p, ar, L, own, lane, 0);
            __syncthreads();   // releases the workers' value-projection stream
            stage_qvec<KQU, 64>(pl, xr, p.kq, F, tagL + SLOT_KQ, l.kq, lane);
            poll_units<1, 64>(pl, xr, p.rr + blk * NOWN, NOWN, tagL + SLOT_KQ, lane, [&](int i, const v4u & v) { x_sink(l.rr, i, v); });
            STAMP(11); RSTAMP(27);
            __syncthreads();
            issue_G(r, p, ar, L, own, lane, 1);
            compute_G(r, l, xr, p, tagL, own, lane);
            RSTAMP(28);
            STAMP(12);
        }
    }

    // -----------------------------------------------------------------------------------------------------------
    // worker waves
    // -----------------------------------------------------------------------------------------------------------
    static __device__ __forceinline__ void worker_main(const M6P & p, const Lds & l, int tid0, int wave) {
        const int blk = blockIdx.x;
        const int own = wave - 1;
        const int gwk = own * NBLK + blk;
        const int DR = p.DR, R = p.R;
        const unsigned base = p.ctl[0];
        const M6Arena ar{p.arena};
        const xrsrc xr = make_xrsrc(p.xch, p.xch_bytes);
        // small jobs
        const int a_row = gwk; const bool a_has = a_row < 5 * R;                      // W1 row
        const int x_row = gwk - 5 * R; const bool x_has = x_row >= 0 && x_row < DR;   // decay-W1 row

        Rows r;
        load_xown(r, p, own);
        PA pa; PF pf;
        Batch<FMT, 1, UD> wA, wCx;

        auto issue_A = [&](const M6Layer & L, const float * sin_l, int tid, int lane) {
            issue_pa(pa, ar, L, sin_l, tid);
            batch_issue_opt<FMT, 1, UD>(a_has, wA, ar.w(L.w1), a_row, 5 * R, nb, 0, lane);
        };
        issue_A(p.layers[0], p.sin, tid0, tid0 & 63);

        for (int li = 0; li < p.n_layers; li++) {
            const float * sin_l = p.sin + (long long) li * p.state_stride;
            float * sout_l = p.sout + (long long) li * p.state_stride;
            const unsigned tagL = base + (unsigned) li * 8u;
            const int tidst = tid0;
            STAMP(0);
            // ---- A: prologue, W1 row; then the r/k/v/g rows (+ the decay row) stream while the comm waves run the mixes ----
            {
                const int tid = opq(tid0), lane = tid & 63;
                const M6Layer & L = p.layers[opq_s(li)];
                STAMP(1);
                __syncthreads();                       // x staged
                STAMP(2);
                prologue_A(l, pa, sout_l, blk == 0, tid);
                STAMP(3);
                if (a_has) {
                    float res[1];
                    rows_finish<FMT, 1, UD>(wA, nullptr, nullptr, nullptr, 0, 5 * R, nb, qvec_at(l.q1, D), lane, res);
                    if (lane == 0) tg_store(xr, p.tl + a_row, __float_as_uint(det_tanhf(res[0])), 0u, 0u, 0u, tagL + SLOT_TL);
                }
                STAMP(4); RSTAMP(17);
                __syncthreads();   // the r/k/v/g stream starts once the comm wave has polled tl (+1.7 %: its polls are not behind the stream)
                issue_C(r, ar, L, own, lane, 0);
                batch_issue_opt<FMT, 1, UD>(x_has, wCx, ar.w(L.dw1), x_row, DR, nb, 0, lane);
                STAMP(5);
            }
            // ---- C: decay row first (every head waits for all of dl), r/k/v/g rows; then output-projection and key rows go
            //         in flight: the workers idle through the WKV phase ----
            {
                const int tid = opq(tid0), lane = tid & 63;
                const M6Layer & L = p.layers[opq_s(li)];
                STAMP(6);
                __syncthreads();                       // activation image(s) staged
                STAMP(7);
                issue_C(r, ar, L, own, lane, 1);
                if (x_has) {
                    float res[1];
                    rows_finish<FMT, 1, UD>(wCx, nullptr, nullptr, nullptr, 0, DR, nb, qvec_at(l.actw, D), lane, res);
                    if (lane == 0) tg_store(xr, p.dl + x_row, __float_as_uint(det_tanhf(res[0])), 0u, 0u, 0u, tagL + SLOT_RKVG);
                }
                compute_C(r, l, xr, p, tagL, own, lane);
                RSTAMP(21);
                issue_E(r, ar, L, own, lane);
                issue_Fk(r, p, ar, L, own, lane);
            }
            // ---- E (workers have no part in D); then the channel-mixing prologue's parameters: they land while the comm
            //         wave waits for x_att ----
            {
                const int tid = opq(tid0), lane = tid & 63;
                const M6Layer & L = p.layers[opq_s(li)];
                STAMP(8);
                __syncthreads();                       // yq staged
                STAMP(9);
                compute_E(r, l, xr, p, tagL, own, lane);
                RSTAMP(24);
                issue_pf(pf, ar, L, sin_l, tid);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- F: prologue; receptance rows go in flight under the key rows; after the rows the value-projection rows
            //         (K = F) stream while the comm waves quantise and hand over k ----
            {
                const int tid = opq(tid0), lane = tid & 63;
                const M6Layer & L = p.layers[opq_s(li)];
                STAMP(10);
                __syncthreads();                       // x_att staged
                STAMP(11);
                prologue_F(l, pf, sout_l, blk == 0, tid);
                STAMP(12);
                issue_Fr(r, ar, L, own, lane);
                compute_F(r, l, xr, p, tagL, own, lane);
                STAMP(13); RSTAMP(29);
                __syncthreads();                       // key rows in l.out -> comm quantises them
                // the stream below fills the CU's memory pipe for ~6 us: the comm wave's k stores and own loads go in first
                // (second barrier; +4 % over issuing straight after the first one)
                __syncthreads();
                issue_G(r, p, ar, L, own, lane, 0);
            }
            // ---- G; then the next layer's prologue parameters and W1 row ----
            {
                const int tid = opq(tid0), lane = tid & 63;
                STAMP(14);
                __syncthreads();                       // kq and rr staged
                STAMP(15);
                {
                    const M6Layer & L = p.layers[opq_s(li)];
                    issue_G(r, p, ar, L, own, tid & 63, 1);
                }
                compute_G(r, l, xr, p, tagL, own, lane);
                STAMP(16); RSTAMP(28);
                const int nl = li + 1 < p.n_layers ? li + 1 : li;
                issue_A(p.layers[nl], p.sin + (long long) nl * p.state_stride, tid, lane);
            }
        }
    }
};

template <int FMT, int EPT, int KQU, int NBD, int GPB>
__global__ __launch_bounds__(512) void k6_mega(M6P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef K6<FMT, EPT, KQU, NBD, GPB> K;
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);   // wave-uniform: roles, row bases and predicates stay scalar
    const M6Lds lo = m6_lds(K::D, p.F);
    typename K::Lds l;
    l.x = reinterpret_cast<float *>(smem + lo.x); l.xn = reinterpret_cast<float *>(smem + lo.xn); l.sx = reinterpret_cast<float *>(smem + lo.sx);
    l.q1 = smem + lo.q1; l.q2 = smem + lo.q2; l.act = smem + lo.act; l.actw = smem + lo.actw; l.yq = smem + lo.yq; l.kq = smem + lo.kq;
    l.tl = reinterpret_cast<float *>(smem + lo.tl); l.red = reinterpret_cast<double *>(smem + lo.red);
    l.out = reinterpret_cast<float *>(smem + lo.out); l.rr = reinterpret_cast<float *>(smem + lo.rr); l.dl = smem + lo.dl;
    const unsigned base = p.ctl[0];
    if (wave == 0) K::comm_main(p, l, tid0, base);
    else K::worker_main(p, l, tid0, wave);
    if (blockIdx.x == 0 && tid0 == 0) p.ctl[0] = base + (unsigned) p.n_layers * 8u;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

// dst[(((f * D/64 + chunk) * R/4 + m/4) * 64 + lane) * 4 + m%4] = src[(f * R + m) * D + chunk * 64 + lane]   (src = W2 as [5][R][D])
__global__ void k_block_w2(const float * __restrict__ src, float * __restrict__ dst, int D, int R) {
    const long long n = 5ll * R * D;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const int d = (int) (i % D); const long long fm = i / D; const int m = (int) (fm % R), f = (int) (fm / R);
        const long long o = ((((long long) f * (D / 64) + d / 64) * (R / 4) + m / 4) * 64 + d % 64) * 4 + m % 4;
        dst[o] = src[i];
    }
}

struct MegaV6 {
    int kind = 1;                 // 1: this file's kernel, 2: ring_v6.hip's (its handle starts with the same member)
    float * w2b = nullptr;
    M6Layer * d_layers = nullptr;
    void * xch = nullptr;
    unsigned * ctl = nullptr;
    unsigned * h_ctl = nullptr;   // pinned host mirror of ctl[0..1], refreshed by mega_v6_ctl_fetch on the caller's stream
    M6P proto{};
    long long * trace = nullptr;
    int variant = -1, n_blocks = 0;
    size_t lds = 0;
    uint64_t bytes = 0;   // algorithmic bytes of one launch: every layer tensor once + the recurrent state read and written
};

typedef void (*MegaKernel)(M6P);
struct MegaVariant { int fmt, ept, kqu, nbd, gpb; MegaKernel fn; };
static const MegaVariant g_variants[] = {
#define MEGA_VARIANTS(FMT) \
    {FMT, 8, 21, 4, 2, k6_mega<FMT, 8, 21, 4, 2>},   /* D 4096, F <= 14336, decay rank 128 */ \
    {FMT, 4, 11, 2, 1, k6_mega<FMT, 4, 11, 2, 1>}    /* D 2048, F <= 7168,  decay rank 64  */
    MEGA_VARIANTS(T_Q4_0), MEGA_VARIANTS(T_Q4_1), MEGA_VARIANTS(T_Q5_0), MEGA_VARIANTS(T_Q5_1), MEGA_VARIANTS(T_Q8_0),
};

static int mega_variant(const Model & m, int n_cu) {
    if (m.arch_major != 6 || m.head_size != 64 || m.layer_end <= m.layer_begin) return -1;
    const int64_t D = m.n_embed(), H = m.head_count;
    const int fmt = (int) m.header.data_type;
    const LayerW & L0 = m.layers[m.layer_begin];
    if (!L0.ffn_key || !L0.att_time_decay_w1 || !L0.att_time_maa_w1) return -1;
    const int64_t F = L0.ffn_key->ne[1], DR = L0.att_time_decay_w1->ne[1], R5 = L0.att_time_maa_w1->ne[1], R = R5 / 5;
    const int64_t NB = 256, NW = NB * 7;   // the kernel is laid out for exactly 256 workgroups (one per CU of an MI355X)
    if (n_cu != NB || H > NB || F % 32 != 0 || 5 * (D / 64) + R5 + DR > NW || !(R == 32 || R == 64)) return -1;
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        const DevTensor * mats[] = {L.att_receptance, L.att_key, L.att_value, L.att_gate, L.att_output, L.att_time_maa_w1,
                                    L.att_time_decay_w1, L.att_time_decay_w2, L.ffn_key, L.ffn_value, L.ffn_receptance};
        for (const DevTensor * t : mats) if (!t || t->type != fmt) return -1;
        if (L.ffn_key->ne[1] != F || L.att_time_decay_w1->ne[1] != DR || L.att_time_maa_w1->ne[1] != R5) return -1;
    }
    for (size_t v = 0; v < sizeof(g_variants) / sizeof(g_variants[0]); v++) {
        const MegaVariant & mv = g_variants[v];
        if (mv.fmt == fmt && D == mv.ept * 512 && 3 * (F / 32) <= (int64_t) mv.kqu * 64 && DR == mv.nbd * 32 && (F / 32 + NB - 1) / NB == mv.gpb) return (int) v;
    }
    return -1;
}

static bool is_ring(void * h) { return h && *(const int *) h == 2; }
static bool is_p47(void * h) { return h && *(const int *) h == 3; }

void mega_v6_destroy(void * h) {
    if (is_ring(h)) { ring_v6_destroy(h); return; }
    if (is_p47(h)) { p47_destroy(h); return; }
    MegaV6 * mg = (MegaV6 *) h;
    if (!mg) return;
    if (mg->d_layers) (void) hipFree(mg->d_layers);
    if (mg->w2b) (void) hipFree(mg->w2b);
    if (mg->xch) (void) hipFree(mg->xch);
    if (mg->ctl) (void) hipFree(mg->ctl);
    if (mg->h_ctl) (void) hipHostFree(mg->h_ctl);
    if (mg->trace) (void) hipFree(mg->trace);
    delete mg;
}

// Returns nullptr when the model / device does not qualify (the caller keeps the seven-launch path).
// RWKV_MI_PERSIST = ring | regs picks one of the two persistent kernels; by default the LDS-DMA ring kernel (ring_v6.hip) is tried first.
void * mega_v6_create(const Model & m) {
    const char * pk = getenv("RWKV_MI_PERSIST");
    const bool want_ring = !(pk && strcmp(pk, "regs") == 0), want_regs = !(pk && strcmp(pk, "ring") == 0);
    if (want_ring) { void * r = ring_v6_create(m); if (r || !want_regs) return r; }
    return mega_v6_create_kind(m, 1);
}

void * mega_v6_create_kind(const Model & m, int kind) {
    if (kind == 2) return ring_v6_create(m);
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
    const size_t w2_layer = (size_t) 5 * R * D;
    if (R % 4 != 0 || hipMalloc((void **) &mg->w2b, w2_layer * (m.layer_end - m.layer_begin) * sizeof(float)) != hipSuccess) { delete mg; return nullptr; }
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
        hipLaunchKernelGGL(k_block_w2, dim3(512), dim3(256), 0, 0, (const float *) L.att_time_maa_w2->data, mg->w2b + hl.size() * w2_layer, (int) D, (int) R);
        hl.push_back(d);
        const DevTensor * all[] = {L.ln1_w, L.ln1_b, L.att_time_maa_x, L.att_time_maa_w, L.att_time_maa_k, L.att_time_maa_v, L.att_time_maa_r, L.att_time_maa_g,
                                   L.att_time_maa_w1, L.att_time_maa_w2, L.att_time_decay, L.att_time_faaaa, L.att_time_decay_w1, L.att_time_decay_w2,
                                   L.att_receptance, L.att_key, L.att_value, L.att_gate, L.att_output, L.att_ln_x_w, L.att_ln_x_b, L.ln2_w, L.ln2_b,
                                   L.ffn_time_maa_k, L.ffn_time_maa_r, L.ffn_key, L.ffn_value, L.ffn_receptance};
        for (const DevTensor * t : all) if (t) bytes += t->nbytes;
        bytes += 2 * (uint64_t) m.state_per_layer() * sizeof(float);
    }
    mg->bytes = bytes;
    if (!in_arena) { delete mg; return nullptr; }
    const int64_t nbD = D / 32, nbF = F / 32;
    const int64_t PAD = 2048;   // polls read whole 64-lane rounds: keep every buffer readable past its end
    auto up = [](int64_t v) { return (v + 63) / 64 * 64; };
    const int64_t act_stride = up(3 * nbD), xunits = up(256 * 8);
    const int64_t sizes[9] = {up(1280) + PAD, 5 * act_stride + PAD, 2 * D + PAD, 256 + PAD, act_stride + PAD, xunits + PAD, up(3 * nbF) + PAD, xunits + PAD, xunits + PAD};
    int64_t units = 0;
    for (int64_t z : sizes) units += z;
    bool ok = hipMalloc((void **) &mg->d_layers, hl.size() * sizeof(M6Layer)) == hipSuccess
           && hipMemcpy(mg->d_layers, hl.data(), hl.size() * sizeof(M6Layer), hipMemcpyHostToDevice) == hipSuccess
           && hipMalloc(&mg->xch, (size_t) units * 16) == hipSuccess && hipMemset(mg->xch, 0, (size_t) units * 16) == hipSuccess
           && hipMalloc((void **) &mg->ctl, 256) == hipSuccess
           && hipHostMalloc((void **) &mg->h_ctl, 64, hipHostMallocDefault) == hipSuccess;
    if (ok) { mg->h_ctl[0] = 8u; mg->h_ctl[1] = 0u; }
    const unsigned init[2] = {8u, 0u};
    ok = ok && hipMemcpy(mg->ctl, init, sizeof(init), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { mega_v6_destroy(mg); return nullptr; }
    M6P & q = mg->proto;
    q.layers = mg->d_layers; q.n_layers = (int) hl.size();
    q.arena = abase; q.w2b = mg->w2b;
    if (hipDeviceSynchronize() != hipSuccess) { mega_v6_destroy(mg); return nullptr; }
    q.state_stride = m.state_per_layer();
    q.xch = mg->xch; q.xch_bytes = (unsigned) (units * 16);
    int u = 0;
    int * slots[9] = {&q.tl, &q.act5, &q.rkvg, &q.dl, &q.yq, &q.xatt, &q.kq, &q.rr, &q.xffn};
    for (int i = 0; i < 9; i++) { *slots[i] = u; u += (int) sizes[i]; }
    q.act_stride = act_stride;
    q.ctl = mg->ctl;
    q.F = (int) F; q.DR = (int) DR; q.R = (int) R; q.H = (int) m.head_count;
    q.gpb = (int) ((nbF + NB - 1) / NB);
    return mg;
}

// debug: cycle stamps of one layer (16 per wave) for the next launches; out must hold n_blocks * 8 * 32 values
bool mega_v6_trace(void * h, int layer, long long * out, bool fetch) {
    if (is_ring(h)) return ring_v6_trace(h, layer, out, fetch);
    if (is_p47(h)) return p47_trace(h, layer, out, fetch);
    MegaV6 * mg = (MegaV6 *) h;
    const size_t n = (size_t) mg->n_blocks * 8 * 32;
    if (!mg->trace) { if (hipMalloc((void **) &mg->trace, n * 8) != hipSuccess) return false; (void) hipMemset(mg->trace, 0, n * 8); }
    mg->proto.trace = mg->trace; mg->proto.trace_layer = layer;
    if (fetch) return hipMemcpy(out, mg->trace, n * 8, hipMemcpyDeviceToHost) == hipSuccess;
    return true;
}

int mega_v6_kind(void * h) { return h ? *(const int *) h : 0; }
uint64_t mega_v6_bytes(void * h) { if (is_ring(h)) return ring_v6_bytes(h); if (is_p47(h)) return p47_bytes(h); return ((MegaV6 *) h)->bytes; }

// sin / sout: state of the stage's FIRST layer. One launch covers every layer of the stage.
bool mega_v6_folds_head(void * h) { return (is_ring(h) && ring_v6_folds_head(h)) || (is_p47(h) && p47_folds_head(h)); }

bool mega_v6_has_range(void * h) { return is_ring(h) || is_p47(h); }
bool mega_v6_folds_embed(void * h) { return (is_p47(h) && p47_folds_embed(h)) || (is_ring(h) && ring_v6_folds_embed(h)); }
bool mega_v6_folds_argmax(void * h) { return (is_p47(h) && p47_folds_head(h)) || (is_ring(h) && ring_v6_folds_argmax(h)); }
bool mega_v6_set_history(void * h, uint32_t * hist, size_t n, hipStream_t st) {
    if (is_p47(h)) return p47_set_history(h, hist, n, st);
    return is_ring(h) && ring_v6_set_history(h, hist, n, st);
}
// Pipeline stages: the launch that runs the stage's last layer writes the residual stream to x_out (the NEXT stage's input buffer, on this
// or on a peer device) instead of back into its own x; nullptr restores the in-place form. false: this kernel has no such output.
bool mega_v6_set_x_out(void * h, float * x_out) {
    if (is_ring(h)) { ring_v6_set_x_out(h, x_out); return true; }
    if (is_p47(h)) { p47_set_x_out(h, x_out); return true; }
    return x_out == nullptr;
}
void mega_v6_forward_range(void * h, float * x, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf, float * logits, int l0, int l1, float * v_first,
                           const uint32_t * tok, uint32_t * next_tok) {
    if (is_p47(h)) { p47_forward_range(h, x, v_first, sin, sout, st, pf, l0, l1, logits, tok, next_tok); return; }
    ring_v6_forward_range(h, x, sin, sout, st, pf, logits, l0, l1, tok, next_tok);
}

void mega_v6_forward(void * h, float * x, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf, float * logits, float * v_first,
                     const uint32_t * tok, uint32_t * next_tok) {
    if (is_ring(h)) { ring_v6_forward(h, x, sin, sout, st, pf, logits, tok, next_tok); return; }
    if (is_p47(h)) { p47_forward_range(h, x, v_first, sin, sout, st, pf, 0, p47_layers(h), logits, tok, next_tok); return; }
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

// The abort word (a poll timed out: co-residency lost or a bug; results since then are not valid) is read through a pinned
// host mirror: an asynchronous copy on the caller's stream, checked after the caller's own stream synchronisation. (A
// blocking hipMemcpy would go through the legacy null stream and couple every blocking stream of the process.)
bool mega_v6_ctl_fetch(void * h, hipStream_t st) {
    if (is_ring(h)) return ring_v6_ctl_fetch(h, st);
    if (is_p47(h)) return p47_ctl_fetch(h, st);
    MegaV6 * mg = (MegaV6 *) h;
    return hipMemcpyAsync(mg->h_ctl, mg->ctl, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st) == hipSuccess;
}
bool mega_v6_aborted_cached(void * h) { if (is_ring(h)) return ring_v6_aborted_cached(h); if (is_p47(h)) return p47_aborted_cached(h); return ((MegaV6 *) h)->h_ctl[1] != 0; }
bool mega_v6_aborted(void * h, hipStream_t st) {
    if (!mega_v6_ctl_fetch(h, st) || hipStreamSynchronize(st) != hipSuccess) return true;
    return mega_v6_aborted_cached(h);
}
// clears the abort word (after the caller has drained the stream), so that the handle -- or the context that drops it -- is usable again
bool mega_v6_clear_abort(void * h, hipStream_t st) {
    if (is_ring(h)) return ring_v6_clear_abort(h, st);
    if (is_p47(h)) return p47_clear_abort(h, st);
    MegaV6 * mg = (MegaV6 *) h;
    mg->h_ctl[1] = 0u;
    return hipMemsetAsync(mg->ctl + 1, 0, sizeof(unsigned), st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
}
unsigned * mega_v6_ctl(void * h) { return is_ring(h) ? ring_v6_ctl(h) : (is_p47(h) ? p47_ctl(h) : ((MegaV6 *) h)->ctl); }
// Test hook: the abort word set from the host, as a poll that timed out would set it -- the next launch drains at once, the host finds the
// word behind it and the context falls back to the per-layer launches (engine.hip, recover_from_abort).
bool mega_v6_force_abort(void * h, hipStream_t st) {
    if (!h || hipStreamSynchronize(st) != hipSuccess) return false;
    const unsigned one = 1u;
    return hipMemcpy(mega_v6_ctl(h) + 1, &one, sizeof(one), hipMemcpyHostToDevice) == hipSuccess;
}
// Why no persistent kernel serves this model on this device (nullptr: one does). The kernels give every CU one workgroup and hand vectors
// over between them inside the launch: they need all 256 CUs of an unpartitioned MI355X (a CPX / DPX partition or another part reports
// fewer), quantised matrices of one format, 64-wide heads and a geometry that has an instantiation.
const char * persist_unavailable_reason(const Model & m) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m.device) != hipSuccess) return "the device properties could not be read";
    static thread_local char buf[160];
    if (prop.multiProcessorCount != 256) { snprintf(buf, sizeof buf, "the device reports %d CUs: the persistent kernels need all 256 of an unpartitioned MI355X (SPX mode)", prop.multiProcessorCount); return buf; }
    if (m.head_size != 64) return "head size is not 64";
    if (m.arch_major == 5) return "RWKV-5 has no persistent kernel (per-op launches)";
    const int t = (int) m.header.data_type;
    if (t == T_F32 || t == T_F16) return "FP32 / FP16 files run the per-op launches (the persistent kernels stream quantised matrices)";
    return "no instantiation for this geometry (n_embed / ffn size / ranks / vocabulary)";
}
// the tag generation the next launch starts from (ctl[0]), through the pinned mirror
unsigned mega_v6_generation(void * h, hipStream_t st) {
    if (!mega_v6_ctl_fetch(h, st) || hipStreamSynchronize(st) != hipSuccess) return 0;
    if (is_p47(h)) return p47_generation_cached(h);
    return is_ring(h) ? ring_v6_generation_cached(h) : ((MegaV6 *) h)->h_ctl[0];
}
// Test hook: presets the rolling tag generation (ctl[0]; the kernel compares its low 16 bits), e.g. just below a 16-bit wrap.
bool mega_v6_set_tag(void * h, unsigned base, hipStream_t st) {
    if (is_ring(h)) return ring_v6_set_tag(h, base, st);
    if (is_p47(h)) return p47_set_tag(h, base, st);
    MegaV6 * mg = (MegaV6 *) h;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    return hipMemcpy(mg->ctl, &base, sizeof(unsigned), hipMemcpyHostToDevice) == hipSuccess;
}

}  // namespace rwkvmi
