// fused_v6.hip -- the RWKV-6 single-token (decode) layer as SEVEN launches instead of ~25 graph-op kernels:
//
//   A  k6_att_prep   LN1 + token shift + maa_x mix + Q8 quantise (every workgroup, redundantly) -> W1 rows + tanh
//   B  k6_mix2       five data-dependent mixes from the transposed F32 W2, quantised on the fly      (rwkv_graph.inc:313-346)
//   C  k6_rkvgw      R, K, V, G projections + decay W1 in one launch (silu / tanh epilogues)         (:349-363)
//   D  k6_wkv        per head: decay W2 + exp(-exp), WKV6 recurrence, GroupNorm * ln_x, gate, quantise (:357-382)
//   E  k6_att_out    output projection + residual add                                               (:384, :671)
//   F  k6_ffn_kr     LN2 + token shift + mixes + quantise -> key rows (relu^2, quantised per 32 rows) + receptance rows
//   G  k6_ffn_v      value projection, x += sigmoid(r) * (Wv k)                                     (:513-531, :672)
//
// Every launch boundary is a real all-to-all dependency (a full-vector LayerNorm or a projection that consumes a whole
// vector). Arithmetic and reduction orders are exactly those of the per-op kernels in kernels.hip,
// so this path is bit-identical to the generic path and to the CPU oracle; tests/test_gpu_* run through it.
//
// Quantised activations travel between these kernels in the "lohi" image: for a vector of nb blocks, bytes
// [b*16, b*16+16) hold elements 0..15 of block b and bytes [nb*16 + b*16, +16) hold elements 16..31, followed by
// f32 d[nb], f32 s[nb], i32 isum[nb]. Lane-linear 16-byte LDS reads of that image are bank-conflict free.
#include "fused_blocks.h"

#include <hip/hip_ext.h>

namespace rwkvmi {

// ---------------------------------------------------------------------------------------------------------------
// A: LN1 + shift + xxx + quantise  ->  W1 rows (+ tanh)
// ---------------------------------------------------------------------------------------------------------------

struct P6A {
    const float * x; const float * ln_w; const float * ln_b; const float * att_xx_in; const float * maa_x;
    float * att_xx_out; float * xn_out; float * sx_out;
    WPl w1; int64_t n_rows;  // 5 * r
    float * tl;
    int64_t D;
};

template <int FMT>
__global__ __launch_bounds__(1024) void k6_att_prep(P6A p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = 1024;
    const int64_t D = p.D;
    const int nb = (int) (D / 32);
    float * l_row = reinterpret_cast<float *>(smem);
    unsigned char * l_qv = smem + D * 4;
    double * red = reinterpret_cast<double *>(l_qv + ((qvec_bytes(D) + 15) / 16) * 16);
    const QVec lq = qvec_at(l_qv, D);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool pro = threadIdx.x < 256;   // the LayerNorm reduction tree is defined over 256 partials
    const int64_t row = (int64_t) blockIdx.x * 4 + wave;
    const bool has_row = wave < 4 && row < p.n_rows;
    Batch<FMT, 1, 2> bt;
    if (wave < 4) batch_issue<FMT, 1, 2>(bt, p.w1.qs, p.w1.qh, p.w1.sc, row < p.n_rows ? row : p.n_rows - 1, p.n_rows, nb, 0, lane);
    if (pro) fill_row(l_row, p.x, D);
    __syncthreads();
    double sacc = 0.0;
    if (pro) sacc = ln_partial_sum(l_row, D);
    const float mean = (float)(block_sum_d_1b(sacc, red) / (double) D);
    double s2 = 0.0;
    if (pro) s2 = ln_partial_var(l_row, D, mean);
    const float var = (float)(block_sum_d_1b(s2, red + 256) / (double) D);
    const float scale = 1.0f / sqrtf(var + 1e-5f);
    // elementwise + quantise: every thread, 4 independent element steps interleaved
    auto fin = [&](int64_t i, float lw, float lb, float pv, float mx) -> float {
        const float y = l_row[i] * scale;
        const float yw = y * lw;
        const float xn = yw + lb;
        const float sx = pv - xn;
        const float sm = sx * mx;
        if (blockIdx.x == 0) { p.xn_out[i] = xn; p.sx_out[i] = sx; p.att_xx_out[i] = xn; }
        return sm + xn;
    };
    int64_t i0 = threadIdx.x;
    for (; i0 + 3 * NT < D; i0 += 4 * NT) {
        float lw[4], lb[4], pv[4], mx[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int64_t i = i0 + u * NT; lw[u] = p.ln_w[i]; lb[u] = p.ln_b[i]; pv[u] = p.att_xx_in[i]; mx[u] = p.maa_x[i]; }
#pragma unroll
        for (int u = 0; u < 4; u++) xv[u] = fin(i0 + u * NT, lw[u], lb[u], pv[u], mx[u]);
        int qi[4], isum[4]; float d16[4], s16[4];
        quant_blocks<4>(xv, qi, d16, s16, isum);
#pragma unroll
        for (int u = 0; u < 4; u++) { const int64_t i = i0 + u * NT; qvec_store(lq, nb, (int) (i >> 5), (int) (i & 31), qi[u], d16[u], s16[u], isum[u]); }
    }
    for (; i0 < D; i0 += NT) {   // D % 64 == 0: whole waves in or out
        const float xxx = fin(i0, p.ln_w[i0], p.ln_b[i0], p.att_xx_in[i0], p.maa_x[i0]);
        int qi, isum; float d16, s16;
        quant_block32(xxx, qi, d16, s16, isum);
        qvec_store(lq, nb, (int) (i0 >> 5), (int) (i0 & 31), qi, d16, s16, isum);
    }
    __syncthreads();
    if (!has_row) return;
    float res[1];
    rows_finish<FMT, 1, 2>(bt, p.w1.qs, p.w1.qh, p.w1.sc, row, p.n_rows, nb, lq, lane, res);
    if (lane == 0) p.tl[row] = det_tanhf(res[0]);
}

// ---------------------------------------------------------------------------------------------------------------
// B: out_f[d] = (sum_m W2t[f][m][d] * tl[f*R+m] + maa_f[d]) * sx[d] + xn[d], quantised; f order w,k,v,r,g
// ---------------------------------------------------------------------------------------------------------------

struct P6B {
    const float * w2t;  // [5][R][D]
    const float * tl;   // [5R]
    const float * maa[5];
    const float * sx; const float * xn;
    void * out;         // 5 lohi images of D elements, qvec_bytes(D) apart (rounded up to 256)
    int64_t D, R, out_stride;
};

__global__ __launch_bounds__(256) void k6_mix2(P6B p) {
    __shared__ float l_tl[256];
    const int64_t D = p.D, R = p.R;
    const int64_t idx = (int64_t) blockIdx.x * 256 + threadIdx.x;  // D % 256 == 0: one f per workgroup
    const int f = (int) (idx / D);
    const int64_t d = idx - (int64_t) f * D;
    for (int m = threadIdx.x; m < R; m += 256) l_tl[m] = p.tl[f * R + m];
    __syncthreads();
    const float * col = p.w2t + (int64_t) f * R * D + d;
    float acc = 0.0f;
#pragma unroll 32
    for (int64_t m = 0; m < R; m++) acc += col[m * D] * l_tl[m];
    const float mm = (acc + p.maa[f][d]) * p.sx[d];
    const float o = mm + p.xn[d];
    int qi, isum; float d16, s16;
    quant_block32(o, qi, d16, s16, isum);
    const QVec ov = qvec_at((unsigned char *) p.out + (size_t) f * p.out_stride, D);
    qvec_store(ov, (int) (D / 32), (int) (d >> 5), (int) (d & 31), qi, d16, s16, isum);
}

// ---------------------------------------------------------------------------------------------------------------
// C: R, K, V, G (D x D) and decay W1 (DR x D) in one launch. 32-row groups, 4 waves x 8 rows.
// ---------------------------------------------------------------------------------------------------------------

struct P6C {
    WPl w[5];            // r, k, v, g, decay_w1
    const void * act;    // 5 lohi images (w, k, v, r, g)
    int64_t act_stride;
    float * out[5];      // r, k, v, g, dl
    int64_t D, DR;
};

template <int FMT>
__global__ __launch_bounds__(256) void k6_rkvgw(P6C p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t D = p.D;
    const int nb = (int) (D / 32);
    const int64_t G = D / 32;
    int mat = (int) (blockIdx.x / G);
    int64_t grp = blockIdx.x - (int64_t) mat * G;
    if (mat > 4) { mat = 4; grp = blockIdx.x - 4 * G; }
    const int act = (0x04213 >> (4 * mat)) & 0xF;  // projection r,k,v,g,decay -> mix output index (w,k,v,r,g order): 3,1,2,4,0
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t N = mat == 4 ? p.DR : D;
    const int64_t row0 = grp * 32 + wave * 8;
    Batch<FMT, 8, 2> bt;
    batch_issue<FMT, 8, 2>(bt, p.w[mat].qs, p.w[mat].qh, p.w[mat].sc, row0 < N ? row0 : N - 1, N, nb, 0, lane);
    qvec_stage((const unsigned char *) p.act + (size_t) act * p.act_stride, smem, D);
    __syncthreads();
    const QVec la = qvec_at(smem, D);
    if (row0 >= N) return;
    float res[8];
    rows_finish<FMT, 8, 2>(bt, p.w[mat].qs, p.w[mat].qh, p.w[mat].sc, row0, N, nb, la, lane, res);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (row0 + r >= N) break;
            float v = res[r];
            if (mat == 3) v = v / (1.0f + det_expf(-v));
            else if (mat == 4) v = det_tanhf(v);
            p.out[mat][row0 + r] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// D: per head -- decay (W2 rows, K = DR), WKV6, GroupNorm * ln_x, gate, quantise. One wave per head, S = 64.
// ---------------------------------------------------------------------------------------------------------------

struct P6D {
    const float * dl; WPl w2; const float * time_decay; const float * faaaa;
    const float * r; const float * k; const float * v; const float * g;
    const float * state_in; float * state_out;
    const float * lnx_w; const float * lnx_b;
    void * y_out;  // lohi image of D elements
    int64_t D, DR;
};

template <int FMT, int NBD>
__global__ __launch_bounds__(64) void k6_wkv(P6D p) {
    constexpr int S = 64;
    __shared__ __attribute__((aligned(16))) unsigned char l_dl[NBD * 32 + NBD * 12];
    const int lane = threadIdx.x;
    const int64_t h = blockIdx.x, c = h * S + lane;
    const int64_t D = p.D;
    // state column first: the long-latency loads fly while the decay is computed
    float s[S];
#pragma unroll
    for (int i = 0; i < S; i++) s[i] = p.state_in[h * S * S + i * S + lane];
    // 1. quantise dl (DR = 32 * NBD elements): half-wave = block
    const QVec ldl = qvec_at(l_dl, NBD * 32);
#pragma unroll
    for (int e0 = 0; e0 < NBD * 32; e0 += 64) {
        const int e = e0 + lane;
        const float val = e < NBD * 32 ? p.dl[e] : 0.0f;
        int qi, isum; float d16, s16;
        quant_block32(val, qi, d16, s16, isum);
        if (e < NBD * 32) qvec_store(ldl, NBD, e >> 5, e & 31, qi, d16, s16, isum);
    }
    __syncthreads();
    // 2. decay row of channel c: NBD partial sums, folded in the order of the 64-entry halving tree (zeros elsewhere)
    float P[NBD];
#pragma unroll
    for (int b = 0; b < NBD; b++) {
        WBlk<FMT> w;
        load_wblk<FMT>(w, p.w2.qs, p.w2.qh, p.w2.sc, c * NBD + b);
        const int4 alo = *reinterpret_cast<const int4 *>(ldl.q + b * 16);
        const int4 ahi = *reinterpret_cast<const int4 *>(ldl.q + NBD * 16 + b * 16);
        P[b] = blk_fma<FMT>(w, alo, ahi, ldl.d[b], ldl.s[b], ldl.isum[b], 0.0f);
    }
#pragma unroll
    for (int o = NBD / 2; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < o; i++) P[i] += P[i + o];
    const float wdec = det_expf(-det_expf(P[0] + p.time_decay[c]));
    // 3. WKV6 (ggml_rwkv_wkv6): lane j owns value column j; r_i, k_i, u_i, w_i are broadcast from lane i (v_readlane)
    const int rr_i = __float_as_int(p.r[c]), kk_i = __float_as_int(p.k[c]), uu_i = __float_as_int(p.faaaa[c]), ww_i = __float_as_int(wdec);
    const float vj = p.v[c];
    float o = 0.0f;
#pragma unroll
    for (int i = 0; i < S; i++) {
        const float ki = __int_as_float(__builtin_amdgcn_readlane(kk_i, i));
        const float ui = __int_as_float(__builtin_amdgcn_readlane(uu_i, i));
        const float ri = __int_as_float(__builtin_amdgcn_readlane(rr_i, i));
        const float wi = __int_as_float(__builtin_amdgcn_readlane(ww_i, i));
        const float kv = vj * ki;
        const float prev = s[i];
        const float temp = kv * ui + prev;
        o += temp * ri;
        s[i] = prev * wi + kv;
    }
#pragma unroll
    for (int i = 0; i < S; i++) p.state_out[h * S * S + i * S + lane] = s[i];
    // 4. GroupNorm over the head (64 partials = 64 lanes), * ln_x, gate
    const float mean = (float)(wave_sum_d((double) o) / (double) S);
    const float dv = o - mean;
    const float var = (float)(wave_sum_d((double)(dv * dv)) / (double) S);
    const float scale = 1.0f / sqrtf(var + 64e-5f);
    float y = dv * scale;
    y = y * p.lnx_w[c];
    y = y + p.lnx_b[c];
    y *= p.g[c];
    // 5. quantise for the output projection: this head is blocks 2h, 2h+1
    int qi, isum; float d16, s16;
    quant_block32(y, qi, d16, s16, isum);
    qvec_store(qvec_at(p.y_out, D), (int) (D / 32), (int) (2 * h + (lane >> 5)), lane & 31, qi, d16, s16, isum);
}

// ---------------------------------------------------------------------------------------------------------------
// E / G: projection with a residual epilogue.  E: x += Wo y.   G: x += sigmoid(r) * (Wv k)
// ---------------------------------------------------------------------------------------------------------------

struct P6E {
    WPl w; const void * act; float * x; const float * rgate;  // rgate == nullptr: plain residual add
    int64_t N, K;
};

template <int FMT, int R, int U, bool DB>
__global__ __launch_bounds__(256) void k6_proj_res(P6E p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t) blockIdx.x * 4 + wave) * R;
    const int nb = (int) (p.K / 32);
    Batch<FMT, R, U> bt;
    batch_issue<FMT, R, U>(bt, p.w.qs, p.w.qh, p.w.sc, row0 < p.N ? row0 : p.N - 1, p.N, nb, 0, lane);
    qvec_stage(p.act, smem, p.K);
    __syncthreads();
    const QVec la = qvec_at(smem, p.K);
    if (row0 >= p.N) return;
    float res[R];
    rows_finish<FMT, R, U, DB>(bt, p.w.qs, p.w.qh, p.w.sc, row0, p.N, nb, la, lane, res);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (row0 + r >= p.N) break;
            const int64_t n = row0 + r;
            if (p.rgate) { const float gte = sigmoid_f(p.rgate[n]) * res[r]; p.x[n] = p.x[n] + gte; }
            else p.x[n] = p.x[n] + res[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// F: LN2 + shift + two mixes + quantise -> key rows (relu^2, quantised per 32-row group) and receptance rows
// ---------------------------------------------------------------------------------------------------------------

struct P6F {
    const float * x; const float * ln_w; const float * ln_b; const float * ffn_xx_in; const float * maa_k; const float * maa_r;
    float * ffn_xx_out;
    WPl wk; WPl wr;
    void * k_out;   // lohi image of F elements (relu(Wk xk)^2 quantised)
    float * r_out;  // D floats (raw Wr xr)
    int64_t D, F;
    int groups_per_block;
    int mix_mode;   // 1: x_f = (x_prev - xn) * c_f + xn (v6, v7);  0: x_f = xn * c_f + (x_prev - x_prev * c_f) (v4, v5)
    int64_t r_rows; // rows of the receptance matrix handled here (D, or 0: RWKV-7 has none)
};

// Workgroup = 8 waves owning up to three 32-row groups (key groups first, then receptance groups): about one workgroup per
// CU, so the VALU-heavy prologue (two f32 divisions, DPP reductions, roundf per element and quantised vector) is replicated
// once per CU instead of once per 32 rows. Its statistics run on threads 0..255 (the reduction tree is defined over 256
// partials); the elementwise + quantise part is spread over all 512 threads with four independent chains interleaved per
// thread (measured: the serial 16-step version of this loop was 2/3 of k6_att_prep's 25k cycles).
template <int FMT>
__global__ __launch_bounds__(512) void k6_ffn_kr(P6F p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t D = p.D, F = p.F;
    const int nb = (int) (D / 32);
    const size_t qb = ((qvec_bytes(D) + 15) / 16) * 16;
    float * l_row = reinterpret_cast<float *>(smem);
    unsigned char * l_k = smem + D * 4;
    unsigned char * l_r = l_k + qb;
    double * red = reinterpret_cast<double *>(l_r + qb);
    float * l_out = reinterpret_cast<float *>(red + 512);
    const QVec qk = qvec_at(l_k, D), qr = qvec_at(l_r, D);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool pro = threadIdx.x < 256;

    const int64_t GK = F / 32, GT = GK + p.r_rows / 32;
    const int gpb = p.groups_per_block;
    const int64_t g0 = (int64_t) blockIdx.x * gpb;
    const int ng = (int) ((GT - g0) < gpb ? (GT - g0) : gpb);
    auto group_w = [&](int64_t g) -> const WPl & { return g < GK ? p.wk : p.wr; };
    auto group_row0 = [&](int64_t g) { return (g < GK ? g : g - GK) * 32 + wave * 4; };
    auto group_n = [&](int64_t g) { return g < GK ? F : D; };

    constexpr int MAXG = 3;   // groups per workgroup (host guarantees gpb <= MAXG)
    constexpr int NT = 512;
    Batch<FMT, 4, 2> bt[MAXG];
    auto issue_all = [&]() {
#pragma unroll
        for (int gi = 0; gi < MAXG; gi++) {
            if (gi < ng) { const WPl & w = group_w(g0 + gi); batch_issue<FMT, 4, 2>(bt[gi], w.qs, w.qh, w.sc, group_row0(g0 + gi), group_n(g0 + gi), nb, 0, lane); }
        }
    };

    // ---- prologue: statistics on threads 0..255 (the tree is defined over 256 partials), elementwise + quantise on all ----
    if (pro) fill_row(l_row, p.x, D);
    __syncthreads();
    double sacc = 0.0;
    if (pro) sacc = ln_partial_sum(l_row, D);
    const float mean = (float)(block_sum_d_1b(sacc, red) / (double) D);
    double s2 = 0.0;
    if (pro) s2 = ln_partial_var(l_row, D, mean);
    const float var = (float)(block_sum_d_1b(s2, red + 256) / (double) D);
    const float scale = 1.0f / sqrtf(var + 1e-5f);
    auto fin = [&](int64_t i, float lw, float lb, float pv, float mk, float mr, float & xk, float & xr) {
        const float y = l_row[i] * scale;
        const float yw = y * lw;
        const float xn = yw + lb;
        if (p.mix_mode == 1) {
            const float sx = pv - xn;
            const float sk = sx * mk;
            xk = sk + xn;
            const float sr = sx * mr;
            xr = sr + xn;
        } else {
            const float xck = xn * mk, pck = pv * mk;
            xk = xck + (pv - pck);
            const float xcr = xn * mr, pcr = pv * mr;
            xr = xcr + (pv - pcr);
        }
        if (blockIdx.x == 0) p.ffn_xx_out[i] = xn;
    };
    int64_t i0 = threadIdx.x;
    for (; i0 + 3 * NT < D; i0 += 4 * NT) {
        float lw[4], lb[4], pv[4], mk[4], mr[4], xk[4], xr[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int64_t i = i0 + u * NT; lw[u] = p.ln_w[i]; lb[u] = p.ln_b[i]; pv[u] = p.ffn_xx_in[i]; mk[u] = p.maa_k[i]; mr[u] = p.maa_r[i]; }
#pragma unroll
        for (int u = 0; u < 4; u++) fin(i0 + u * NT, lw[u], lb[u], pv[u], mk[u], mr[u], xk[u], xr[u]);
        int qi[4], isum[4]; float d16[4], s16[4];
        quant_blocks<4>(xk, qi, d16, s16, isum);
#pragma unroll
        for (int u = 0; u < 4; u++) { const int64_t i = i0 + u * NT; qvec_store(qk, nb, (int) (i >> 5), (int) (i & 31), qi[u], d16[u], s16[u], isum[u]); }
        quant_blocks<4>(xr, qi, d16, s16, isum);
#pragma unroll
        for (int u = 0; u < 4; u++) { const int64_t i = i0 + u * NT; qvec_store(qr, nb, (int) (i >> 5), (int) (i & 31), qi[u], d16[u], s16[u], isum[u]); }
    }
    for (; i0 < D; i0 += NT) {
        float xk, xr;
        fin(i0, p.ln_w[i0], p.ln_b[i0], p.ffn_xx_in[i0], p.maa_k[i0], p.maa_r[i0], xk, xr);
        int qi, isum; float d16, s16;
        quant_block32(xk, qi, d16, s16, isum);
        qvec_store(qk, nb, (int) (i0 >> 5), (int) (i0 & 31), qi, d16, s16, isum);
        quant_block32(xr, qi, d16, s16, isum);
        qvec_store(qr, nb, (int) (i0 >> 5), (int) (i0 & 31), qi, d16, s16, isum);
    }
    issue_all();
    __syncthreads();

    // ---- rows ----
#pragma unroll
    for (int gi = 0; gi < MAXG; gi++) {
        if (gi < ng) {
            const int64_t g = g0 + gi;
            const WPl & w = group_w(g);
            float res[4];
            rows_finish<FMT, 4, 2>(bt[gi], w.qs, w.qh, w.sc, group_row0(g), group_n(g), nb, g < GK ? qk : qr, lane, res);
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (g < GK) { const float t = res[r] > 0.0f ? res[r] : 0.0f; l_out[gi * 32 + wave * 4 + r] = t * t; }
                    else p.r_out[group_row0(g) + r] = res[r];
                }
            }
        }
    }
    __syncthreads();
    // ---- quantise the key groups (relu^2 outputs) for the value projection: wave 0, one group per pass ----
    if (threadIdx.x < 64) {
        for (int gi = 0; gi < ng; gi++) {
            const int64_t g = g0 + gi;
            if (g >= GK) break;
            const float v = l_out[gi * 32 + (threadIdx.x & 31)];
            int qi, isum; float d16, s16;
            quant_block32(v, qi, d16, s16, isum);
            if (threadIdx.x < 32) qvec_store(qvec_at(p.k_out, F), (int) (F / 32), (int) g, threadIdx.x, qi, d16, s16, isum);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

struct V6Scratch {
    float *xn, *sx, *tl, *r, *k, *v, *g, *dl, *rr;
    void *act5, *yq, *kq;
    int64_t act_stride;
};

bool fused_v6_supported(const Model & m) {
    if (m.arch_major != 6 || m.head_size != 64) return false;
    const int64_t D = m.n_embed();
    if (D % 256 != 0) return false;
    const int fmt = (int) m.header.data_type;
    if (!dtype_quantized(fmt)) return false;
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        const DevTensor * mats[] = {L.att_receptance, L.att_key, L.att_value, L.att_gate, L.att_output, L.att_time_maa_w1,
                                    L.att_time_decay_w1, L.att_time_decay_w2, L.ffn_key, L.ffn_value, L.ffn_receptance};
        for (const DevTensor * t : mats) if (!t || t->type != fmt) return false;
        const int64_t DR = L.att_time_decay_w1->ne[1], R5 = L.att_time_maa_w1->ne[1];
        if (!(DR == 64 || DR == 128) || R5 / 5 > 256 || L.ffn_key->ne[1] % 32 != 0) return false;
    }
    return true;
}

size_t fused_v6_scratch_bytes(const Model & m) {
    const size_t D = (size_t) m.n_embed(), F = (size_t) m.ffn_size;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    return 9 * up(D * 4) + up(2048 * 4) + up(256 * 4) + 5 * up(qvec_bytes(D)) + up(qvec_bytes(D)) + up(qvec_bytes(F)) + 4096;
}

// Launch, optionally bracketed by the kernel's own start/stop timestamps (hipExtLaunchKernelGGL events: the dispatch's
// begin/end as the profiler sees them, no host-side event overhead inside the interval).
template <typename Kern, typename Param>
static void launch6(rwkv_context::Prof * pf, uint64_t bytes, Kern kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const Param & prm) {
    if (pf && pf->on && bytes) {
        if (pf->used * 2 + 2 > pf->events.size()) {
            hipEvent_t a = nullptr, c = nullptr;
            (void) hipEventCreate(&a); (void) hipEventCreate(&c);
            pf->events.push_back(a); pf->events.push_back(c); pf->bytes.push_back(0);
        }
        pf->bytes[pf->used] = bytes;
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t) shmem, st, pf->events[pf->used * 2], pf->events[pf->used * 2 + 1], 0, prm);
        pf->used++;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, shmem, st, prm);
    }
}

template <int FMT>
static void fused_v6_layer_t(const Model & m, const LayerW & L, float * x, const float * sin, float * sout, void * scratch, hipStream_t st, rwkv_context::Prof * pf) {
    const int64_t D = m.n_embed(), F = L.ffn_key->ne[1], H = m.head_count;
    const int64_t R5 = L.att_time_maa_w1->ne[1], R = R5 / 5, DR = L.att_time_decay_w1->ne[1];
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    unsigned char * p = (unsigned char *) scratch;
    auto takef = [&](size_t n) { float * r = (float *) p; p += up(n * 4); return r; };
    V6Scratch s;
    s.xn = takef(D); s.sx = takef(D); s.r = takef(D); s.k = takef(D); s.v = takef(D); s.g = takef(D); s.rr = takef(D);
    takef(D); takef(D);
    s.tl = takef(2048); s.dl = takef(256);
    s.act_stride = (int64_t) up(qvec_bytes(D));
    s.act5 = p; p += 5 * s.act_stride;
    s.yq = p; p += up(qvec_bytes(D));
    s.kq = p; p += up(qvec_bytes(F));
    auto f = [](const DevTensor * t) { return (const float *) t->data; };
    const size_t qbD = ((qvec_bytes(D) + 15) / 16) * 16;

    const int gridA = (int) ((R5 + 3) / 4);
    P6A a{x, f(L.ln1_w), f(L.ln1_b), sin + D, f(L.att_time_maa_x), sout + D, s.xn, s.sx, planes(L.att_time_maa_w1), R5, s.tl, D};
    launch6(pf, 0, k6_att_prep<FMT>, dim3((unsigned) gridA), dim3(1024), (size_t) D * 4 + qbD + 512 * 8, st, a);

    P6B b{f(L.att_time_maa_w2), s.tl, {f(L.att_time_maa_w), f(L.att_time_maa_k), f(L.att_time_maa_v), f(L.att_time_maa_r), f(L.att_time_maa_g)},
          s.sx, s.xn, s.act5, D, R, s.act_stride};
    launch6(pf, 0, k6_mix2, dim3((unsigned) (5 * D / 256)), dim3(256), 0, st, b);

    P6C c{{planes(L.att_receptance), planes(L.att_key), planes(L.att_value), planes(L.att_gate), planes(L.att_time_decay_w1)},
          s.act5, s.act_stride, {s.r, s.k, s.v, s.g, s.dl}, D, DR};
    const uint64_t actD = qvec_bytes(D);
    launch6(pf, L.att_receptance->nbytes + L.att_key->nbytes + L.att_value->nbytes + L.att_gate->nbytes + L.att_time_decay_w1->nbytes + 5 * actD + (4 * D + DR) * 4,
            k6_rkvgw<FMT>, dim3((unsigned) (4 * (D / 32) + (DR + 31) / 32)), dim3(256), qbD, st, c);

    P6D d{s.dl, planes(L.att_time_decay_w2), f(L.att_time_decay), f(L.att_time_faaaa), s.r, s.k, s.v, s.g, sin + 2 * D, sout + 2 * D,
          f(L.att_ln_x_w), f(L.att_ln_x_b), s.yq, D, DR};
    if (DR == 128) launch6(pf, 0, k6_wkv<FMT, 4>, dim3((unsigned) H), dim3(64), 0, st, d);
    else launch6(pf, 0, k6_wkv<FMT, 2>, dim3((unsigned) H), dim3(64), 0, st, d);

    P6E e{planes(L.att_output), s.yq, x, nullptr, D, D};
    launch6(pf, L.att_output->nbytes + actD + D * 8, k6_proj_res<FMT, 4, 2, false>, dim3((unsigned) ((D + 15) / 16)), dim3(256), qbD, st, e);

    const int64_t groups = F / 32 + D / 32;
    const int gpb = (int) ((groups + 255) / 256) < 3 ? (int) ((groups + 255) / 256) : 3;   // ~one workgroup per CU: the prologue runs once per CU
    P6F ff{x, f(L.ln2_w), f(L.ln2_b), sin, f(L.ffn_time_maa_k), f(L.ffn_time_maa_r), sout, planes(L.ffn_key), planes(L.ffn_receptance), s.kq, s.rr, D, F, gpb, 1, D};
    launch6(pf, L.ffn_key->nbytes + L.ffn_receptance->nbytes + D * 12 + qvec_bytes(F) + D * 4, k6_ffn_kr<FMT>, dim3((unsigned) ((groups + gpb - 1) / gpb)), dim3(512),
            (size_t) D * 4 + 2 * qbD + 512 * 8 + (size_t) gpb * 32 * 4, st, ff);

    P6E g{planes(L.ffn_value), s.kq, x, s.rr, D, F};
    launch6(pf, L.ffn_value->nbytes + qvec_bytes(F) + D * 12, k6_proj_res<FMT, 4, 4, true>, dim3((unsigned) ((D + 15) / 16)), dim3(256), ((qvec_bytes(F) + 15) / 16) * 16, st, g);
}

// ---- the projection / channel-mixing launches shared with the other architectures' fused layers (fused_v7.hip, fused_v4.hip) ----
template <int FMT>
static void fused_proj_res_t(const DevTensor * W, const void * act, float * x, const float * rgate, int64_t N, int64_t K, bool long_rows, hipStream_t st, rwkv_context::Prof * pf) {
    P6E e{planes(W), act, x, rgate, N, K};
    const uint64_t bytes = W->nbytes + qvec_bytes(K) + (uint64_t) N * (rgate ? 12 : 8);
    const size_t shm = ((qvec_bytes(K) + 15) / 16) * 16;
    // rows per wave: 4 unless that leaves CUs without a workgroup (N / 16 workgroups), then 2
    const bool small = (N + 15) / 16 < 224;
    if (long_rows && small) launch6(pf, bytes, k6_proj_res<FMT, 2, 4, true>, dim3((unsigned) ((N + 7) / 8)), dim3(256), shm, st, e);
    else if (long_rows) launch6(pf, bytes, k6_proj_res<FMT, 4, 4, true>, dim3((unsigned) ((N + 15) / 16)), dim3(256), shm, st, e);
    else if (small) launch6(pf, bytes, k6_proj_res<FMT, 2, 2, false>, dim3((unsigned) ((N + 7) / 8)), dim3(256), shm, st, e);
    else launch6(pf, bytes, k6_proj_res<FMT, 4, 2, false>, dim3((unsigned) ((N + 15) / 16)), dim3(256), shm, st, e);
}
void fused_proj_res(int fmt, const DevTensor * W, const void * act, float * x, const float * rgate, int64_t N, int64_t K, bool long_rows, hipStream_t st, rwkv_context::Prof * pf) {
    switch (fmt) {
        case T_Q4_0: fused_proj_res_t<T_Q4_0>(W, act, x, rgate, N, K, long_rows, st, pf); break;
        case T_Q4_1: fused_proj_res_t<T_Q4_1>(W, act, x, rgate, N, K, long_rows, st, pf); break;
        case T_Q5_0: fused_proj_res_t<T_Q5_0>(W, act, x, rgate, N, K, long_rows, st, pf); break;
        case T_Q5_1: fused_proj_res_t<T_Q5_1>(W, act, x, rgate, N, K, long_rows, st, pf); break;
        case T_Q8_0: fused_proj_res_t<T_Q8_0>(W, act, x, rgate, N, K, long_rows, st, pf); break;
        default: break;
    }
}
template <int FMT>
static void fused_ffn_kr_t(const float * x, const float * ln_w, const float * ln_b, const float * xx_in, float * xx_out, const float * maa_k, const float * maa_r, int mix_mode,
                           const DevTensor * wk, const DevTensor * wr, void * k_out, float * r_out, int64_t D, int64_t F, hipStream_t st, rwkv_context::Prof * pf) {
    const int64_t r_rows = wr ? D : 0;
    const int64_t groups = F / 32 + r_rows / 32;
    const int gpb = (int) ((groups + 255) / 256) < 3 ? (int) ((groups + 255) / 256) : 3;   // (one group per workgroup with two workgroups per CU measured slower)
    const size_t qbD = ((qvec_bytes(D) + 15) / 16) * 16;
    P6F ff{x, ln_w, ln_b, xx_in, maa_k, maa_r, xx_out, planes(wk), wr ? planes(wr) : planes(wk), k_out, r_out, D, F, gpb, mix_mode, r_rows};
    launch6(pf, wk->nbytes + (wr ? wr->nbytes : 0) + D * 12 + qvec_bytes(F) + r_rows * 4, k6_ffn_kr<FMT>, dim3((unsigned) ((groups + gpb - 1) / gpb)), dim3(512),
            (size_t) D * 4 + 2 * qbD + 512 * 8 + (size_t) gpb * 32 * 4, st, ff);
}
void fused_ffn_kr(int fmt, const float * x, const float * ln_w, const float * ln_b, const float * xx_in, float * xx_out, const float * maa_k, const float * maa_r, int mix_mode,
                  const DevTensor * wk, const DevTensor * wr, void * k_out, float * r_out, int64_t D, int64_t F, hipStream_t st, rwkv_context::Prof * pf) {
    switch (fmt) {
        case T_Q4_0: fused_ffn_kr_t<T_Q4_0>(x, ln_w, ln_b, xx_in, xx_out, maa_k, maa_r, mix_mode, wk, wr, k_out, r_out, D, F, st, pf); break;
        case T_Q4_1: fused_ffn_kr_t<T_Q4_1>(x, ln_w, ln_b, xx_in, xx_out, maa_k, maa_r, mix_mode, wk, wr, k_out, r_out, D, F, st, pf); break;
        case T_Q5_0: fused_ffn_kr_t<T_Q5_0>(x, ln_w, ln_b, xx_in, xx_out, maa_k, maa_r, mix_mode, wk, wr, k_out, r_out, D, F, st, pf); break;
        case T_Q5_1: fused_ffn_kr_t<T_Q5_1>(x, ln_w, ln_b, xx_in, xx_out, maa_k, maa_r, mix_mode, wk, wr, k_out, r_out, D, F, st, pf); break;
        case T_Q8_0: fused_ffn_kr_t<T_Q8_0>(x, ln_w, ln_b, xx_in, xx_out, maa_k, maa_r, mix_mode, wk, wr, k_out, r_out, D, F, st, pf); break;
        default: break;
    }
}

void fused_v6_layer(const Model & m, const LayerW & L, float * x, const float * sin, float * sout, void * scratch, hipStream_t st, rwkv_context::Prof * pf) {
    switch ((int) m.header.data_type) {
        case T_Q4_0: fused_v6_layer_t<T_Q4_0>(m, L, x, sin, sout, scratch, st, pf); break;
        case T_Q4_1: fused_v6_layer_t<T_Q4_1>(m, L, x, sin, sout, scratch, st, pf); break;
        case T_Q5_0: fused_v6_layer_t<T_Q5_0>(m, L, x, sin, sout, scratch, st, pf); break;
        case T_Q5_1: fused_v6_layer_t<T_Q5_1>(m, L, x, sin, sout, scratch, st, pf); break;
        case T_Q8_0: fused_v6_layer_t<T_Q8_0>(m, L, x, sin, sout, scratch, st, pf); break;
        default: break;
    }
}

}  // namespace rwkvmi
