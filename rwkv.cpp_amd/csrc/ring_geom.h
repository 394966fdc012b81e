// ring_geom.h -- layout of the per-workgroup weight stream of the persistent RWKV-6 decode kernel on the LDS-DMA ring (ring_v6.hip).
//
// The file format fixes the bytes of a quantised block, not where they sit in HBM. The persistent kernel gives every workgroup (= CU) the
// same rows of every matrix in every layer and consumes them in a fixed order, so at context creation the rows of a workgroup are copied
// (k_ring_pack) into ONE contiguous stream per workgroup, in consumption order. A loader wave then moves that stream through a ring in
// LDS with 1-KiB LDS-DMA instructions (global_load_lds_dwordx4: 64 lanes x 16 bytes, no registers), and the consumer waves read whole
// "records" out of the ring. Host (stream builder, tests) and device (loader, consumers) compute every offset with the functions below.
//
// Record = R rows x U steps of one matrix (a step = 64 consecutive 32-weight blocks of a row; lane l of the consuming wave owns block
// 64 u + l of each row, the accumulation order of DESIGN.md section 4):
//     codes       [u][r][half][lane] 16 bytes      half = 0 (Q4 / Q5: the block's 16 code bytes), 0..1 (Q8_0: bytes 0..15, 16..31)
//     scales      [u][r][lane]       2 bytes (fp16 d) or 4 bytes (fp16 d, fp16 m)
//     fifth bits  [u][r][lane]       4 bytes (Q5 only)
// Every ds_read of a consumer is lane-linear (conflict-free). Blocks past the end of a row are zero bytes in the stream.
//
// Layer block of workgroup b (phases in consumption order; record j of a phase goes to consumer wave (j + rot) % 6):
//     W1    time_maa_w1 rows b, b + 256, ...              (R = 1, K = D)     -> tanh -> tl
//     C     two-row sets of ONE of receptance/key/value/gate (R = 2, K = D)  -> r, k, v, g
//     DW1   (empty: time_decay_w1 row b does not travel through the ring. On a consumer wave it cost the 128 workgroups that own one
//           0.6 - 1.2 us of their r/k/v/g phase -- in front of the sets it also held the loader's window back -- and the hand-over behind
//           that phase waits for the last workgroup; the comm wave of the workgroup, idle there, reads the row from the planes instead.)
//     E     output rows                                   (R = 1, K = D)     -> x += ...
//     FK    two-row sets of ffn.key                       (R = 2, K = D)     -> relu^2 -> k
//     FR    ffn.receptance rows                           (R = 1, K = D)
//     G     ffn.value rows                                (R = 1, K = F)     -> x += sigmoid(r) * ...
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define RG_HD __host__ __device__ inline
#else
#define RG_HD inline
#endif

namespace rwkvmi {

constexpr int RG_NBLK = 256;   // workgroups = CUs of an MI355X
constexpr int RG_NC = 6;       // consumer waves per workgroup
constexpr int RG_CHUNK = 1024; // bytes of one LDS-DMA instruction

enum { RG_W1 = 0, RG_C = 1, RG_DW1 = 2, RG_E = 3, RG_FK = 4, RG_FR = 5, RG_G = 6, RG_NPHASE = 7 };   // (= the order in the stream)

struct RingShape {
    int D, F, R5, DR;        // n_embed, ffn size, 5 x mix rank (rows of time_maa_w1), decay rank (rows of time_decay_w1)
    int qs, scb, qhb;        // bytes per block: codes (16 | 32), scales (2 | 4), fifth bits (0 | 4)
    int bal = 0;             // 1: the two-row sets of C and FK are dealt to the consumer waves by SIMD share (RG_BAL_*; D = 4096 only)
};

// Which consumer wave takes which record. Default: record j of a phase -> consumer (j + rot) % 6, a wave's records are six apart.
// Round 6 (profiles/r06_*_cp_*: real-time stamps per wave): the eight waves of a workgroup sit two per SIMD -- (loader, c2), (comm, c3),
// (c0, c4), (c1, c5) -- and a SIMD issues for its OLDER wave first: c4 and c5 finish every row phase last (C rows 3.4 us against 2.4), and
// the hand-over behind a phase waits for the last wave of 1536. A SIMD with two consumers retires a two-row record every ~800 cycles,
// a consumer beside the loader or the comm wave one every ~1000: 32 sets are dealt 5 5 7 7 4 4 (9 per shared SIMD, 7 per single), the
// 30 streamed key sets 5 5 6 6 4 4. Per PAIR of consumers (c >> 1) the record numbers of the even one; the odd one takes the next record.
constexpr int RG_BAL_C_N[3] = {5, 7, 4};
constexpr int RG_BAL_C_J[3][7] = {{2, 8, 14, 22, 28, 99, 99}, {0, 6, 10, 16, 20, 24, 30}, {4, 12, 18, 26, 99, 99, 99}};
constexpr int RG_BAL_K_N[3] = {5, 6, 4};
constexpr int RG_BAL_K_J[3][7] = {{2, 8, 14, 20, 26, 99, 99}, {0, 6, 12, 16, 22, 28, 99}, {4, 10, 18, 24, 99, 99, 99}};
constexpr int RG_BAL_C_SETS = 32, RG_BAL_K_SETS = 30, RG_BAL_TMAX = 7;

RG_HD int rg_steps(int K) { return (K / 32 + 63) / 64; }
RG_HD uint32_t rg_rec_bytes(const RingShape & s, int R, int K) { return (uint32_t) (rg_steps(K) * R * 64 * (s.qs + s.scb + s.qhb)); }
// offsets inside a record (lane 0)
RG_HD uint32_t rg_code_off(const RingShape & s, int R, int u, int r, int half) { return (uint32_t) (((u * R + r) * (s.qs / 16) + half) * 1024); }
RG_HD uint32_t rg_sc_off(const RingShape & s, int R, int U, int u, int r) { return (uint32_t) (U * R * 64 * s.qs + (u * R + r) * 64 * s.scb); }
RG_HD uint32_t rg_qh_off(const RingShape & s, int R, int U, int u, int r) { return (uint32_t) (U * R * 64 * (s.qs + s.scb) + (u * R + r) * 256); }

RG_HD int rg_rows_c(const RingShape & s) { return 4 * s.D / RG_NBLK; }                 // r/k/v/g rows per workgroup (all of one matrix)
RG_HD int rg_rows_e(const RingShape & s) { return s.D / RG_NBLK; }                     // output / receptance / value rows per workgroup
RG_HD int rg_gpb(const RingShape & s) { return (s.F / 32 + RG_NBLK - 1) / RG_NBLK; }   // 32-row groups of ffn.key per workgroup
// Two-row sets of ffn.key of workgroup b, and how many of them (the LAST ones) do not travel through the ring: 32 sets over six consumer
// waves are 6 + 6 + 5 + 5 + 5 + 5, and a wave that holds five records in registers (ring_v6.hip, NPK) never completes a phase of six
// during a hand-over wait -- so it never gets to the records behind it. The comm wave, idle through the key rows, takes the two odd sets
// (read from the planes like the decay row); 16 sets (2 + 2 + 3 + 3 + 3 + 3, a remainder of four) stay as they are.
RG_HD int rg_key_sets(const RingShape & s, int b) {
    const int gpb = rg_gpb(s), key0 = b * gpb * 32;
    return key0 < s.F ? ((s.F - key0 < gpb * 32 ? s.F - key0 : gpb * 32) / 2) : 0;
}
RG_HD int rg_key_comm(const RingShape & s, int b) { const int n = rg_key_sets(s, b); return (n >= RG_NC && n % RG_NC <= 2) ? n % RG_NC : 0; }

// rows of a record: phase-specific matrix (RG_C: the matrix index 0..3 = r, k, v, g), first row, rows per record, row length
struct RingRec { int mat, row0, R, K; };

struct RingCu {
    uint32_t n[RG_NPHASE];     // records per phase
    uint32_t rec[RG_NPHASE];   // bytes per record
    uint32_t off[RG_NPHASE];   // offset of the phase's first record in the layer block
    uint32_t rot[RG_NPHASE];   // record j -> consumer (j + rot) % RG_NC
    uint32_t bal[RG_NPHASE];   // 0: the rot mapping; 1: RG_BAL_C; 2: RG_BAL_K
    uint32_t layer_bytes;
};

RG_HD RingCu rg_cu(const RingShape & s, int b) {
    RingCu c;
    const int gpb = rg_gpb(s);
    const int key0 = b * gpb * 32;
    c.n[RG_W1] = (uint32_t) (s.R5 > b ? (s.R5 - b + RG_NBLK - 1) / RG_NBLK : 0);
    c.n[RG_DW1] = 0;   // (the decay row of workgroup b < DR is read from the planes by its comm wave, beside the consumers' r/k/v/g sets: ring_v6.hip)
    c.n[RG_C] = (uint32_t) (rg_rows_c(s) / 2);
    c.n[RG_E] = (uint32_t) rg_rows_e(s);
    (void) key0;
    c.n[RG_FK] = (uint32_t) (rg_key_sets(s, b) - rg_key_comm(s, b));
    c.n[RG_FR] = (uint32_t) rg_rows_e(s);
    c.n[RG_G] = (uint32_t) rg_rows_e(s);
    const uint32_t d1 = rg_rec_bytes(s, 1, s.D), d2 = rg_rec_bytes(s, 2, s.D), f1 = rg_rec_bytes(s, 1, s.F);
    c.rec[RG_W1] = d1; c.rec[RG_DW1] = d1; c.rec[RG_C] = d2; c.rec[RG_E] = d1; c.rec[RG_FK] = d2; c.rec[RG_FR] = d1; c.rec[RG_G] = f1;
    // E, FR and G share one mapping (a wave keeps the residual and the receptance of its rows in registers across the three phases).
    // The two-row sets of C and FK do not divide by six: the extra sets go to consumers 2 and 3 (waves 4 and 5, which share their SIMDs
    // with the loader and the comm wave, not with another consumer); the decay row goes to a wave with the fewest r/k/v/g sets
    c.rot[RG_W1] = 0; c.rot[RG_DW1] = RG_NC - 1; c.rot[RG_C] = 2; c.rot[RG_E] = 0; c.rot[RG_FK] = (rg_key_sets(s, b) - rg_key_comm(s, b)) % RG_NC ? 2 : 0; c.rot[RG_FR] = 0; c.rot[RG_G] = 0;
    uint32_t p = 0;
    for (int ph = 0; ph < RG_NPHASE; ph++) { c.off[ph] = p; p += c.n[ph] * c.rec[ph]; c.bal[ph] = 0; }
    if (s.bal && c.n[RG_C] == (uint32_t) RG_BAL_C_SETS) c.bal[RG_C] = 1;
    if (s.bal && c.n[RG_FK] == (uint32_t) RG_BAL_K_SETS) c.bal[RG_FK] = 2;
    c.layer_bytes = p;
    return c;
}

RG_HD RingRec rg_rec(const RingShape & s, int b, int phase, int j) {
    RingRec r;
    r.mat = 0; r.R = 1; r.K = s.D; r.row0 = 0;
    switch (phase) {
        case RG_W1:  r.row0 = b + RG_NBLK * j; break;
        case RG_DW1: r.row0 = b; break;
        case RG_C:   { const int rpb = rg_rows_c(s); r.mat = (b * rpb) / s.D; r.row0 = (b * rpb) % s.D + 2 * j; r.R = 2; } break;
        case RG_E:   r.row0 = b * rg_rows_e(s) + j; break;
        case RG_FK:  r.row0 = b * rg_gpb(s) * 32 + 2 * j; r.R = 2; break;
        case RG_FR:  r.row0 = b * rg_rows_e(s) + j; break;
        default:     r.row0 = b * rg_rows_e(s) + j; r.K = s.F; break;
    }
    return r;
}

// record number of consumer cons' t-th record of a phase (>= n[phase]: it has no t-th), and how many it has
RG_HD uint32_t rg_own_j(const RingCu & c, int phase, int cons, int t) {
    if (c.bal[phase]) {
        const int pr = cons >> 1;
        if (t >= RG_BAL_TMAX) return 0xFFFFu;
        const int j = c.bal[phase] == 1 ? RG_BAL_C_J[pr][t] : RG_BAL_K_J[pr][t];
        return j >= 99 ? 0xFFFFu : (uint32_t) (j + (cons & 1));
    }
    return (uint32_t) ((cons + RG_NC - (int) c.rot[phase] % RG_NC) % RG_NC) + (uint32_t) (RG_NC * t);
}
RG_HD uint32_t rg_own_count(const RingCu & c, int phase, int cons) {
    if (c.bal[phase]) return (uint32_t) (c.bal[phase] == 1 ? RG_BAL_C_N[cons >> 1] : RG_BAL_K_N[cons >> 1]);
    const uint32_t j0 = rg_own_j(c, phase, cons, 0);
    return j0 < c.n[phase] ? (c.n[phase] - j0 + RG_NC - 1) / RG_NC : 0u;
}
// first record of consumer c in a phase
RG_HD uint32_t rg_first_j(const RingCu & c, int phase, int cons) { return rg_own_j(c, phase, cons, 0); }

// Offset in the layer block of the first record consumer `cons` owns in phase `from` or later; RG_NONE when it owns none any more in
// this layer (the caller then continues with the next block of its stream: the next layer, or the head).
constexpr uint32_t RG_NONE = 0xFFFFFFFFu;
RG_HD uint32_t rg_next_own_in_layer(const RingCu & c, int cons, int from) {
    for (int p = from; p < RG_NPHASE; p++) {
        const uint32_t j0 = rg_first_j(c, p, cons);
        if (j0 < c.n[p]) return c.off[p] + j0 * c.rec[p];
    }
    return RG_NONE;
}

// ---- the head projection behind the last layer (F16 head.weight, ggml's F16 dot order: 32 partial sums k mod 32, four lanes per row) ----
// Workgroup b owns rows [b * hg * 16, (b + 1) * hg * 16) in hg row groups of 16 rows; a record = 16 rows x RG_HSTEPS steps of 32
// columns: [step][lane = 4 * row + q] 16 bytes = columns 32 step + 8 q .. + 7 of that row (lane-linear for the consumer, 1 KiB per step).
// Consumer c takes row groups c, c + 6, ... ("passes"); inside a pass the records of the participating consumers alternate, chunk by
// chunk, so that all of them work at once out of a ring that holds only a fraction of a row group.
constexpr int RG_HSTEPS = 8;
constexpr uint32_t RG_HREC = RG_HSTEPS * 1024;
struct RingHead { int hg, chunks, passes; uint32_t bytes; };      // row groups per workgroup, records per row group, passes, bytes per workgroup
RG_HD RingHead rg_head(int n_vocab, int K) {
    RingHead h;
    h.hg = n_vocab / (RG_NBLK * 16);
    h.chunks = K / 32 / RG_HSTEPS;
    h.passes = (h.hg + RG_NC - 1) / RG_NC;
    h.bytes = (uint32_t) h.hg * (uint32_t) h.chunks * RG_HREC;
    return h;
}
RG_HD int rg_head_npc(const RingHead & h, int pass) { const int r = h.hg - RG_NC * pass; return r < RG_NC ? r : RG_NC; }   // consumers in a pass
// offset (from the head's start in the workgroup's stream) of the record (pass, chunk, consumer)
RG_HD uint32_t rg_head_off(const RingHead & h, int pass, int chunk, int cons) {
    return ((uint32_t) (RG_NC * pass * h.chunks) + (uint32_t) (chunk * rg_head_npc(h, pass) + cons)) * RG_HREC;
}

// a consumer wave's rows of an x-like vector (E / FR / G mapping): unit (b, c) carries rows b * rows_e + c + RG_NC * t, t < 3
RG_HD int rg_x_rows(const RingShape & s, int cons) { const int n = rg_rows_e(s); return cons < n ? (n - cons + RG_NC - 1) / RG_NC : 0; }

}  // namespace rwkvmi
