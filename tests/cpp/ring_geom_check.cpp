// Host-side check of the per-workgroup weight stream layout (rwkv.cpp_amd/csrc/ring_geom.h), compiled and run by tests/test_ring_geom.py:
// every row of every matrix of a layer lands in exactly one record of exactly one workgroup, records tile the layer block without gaps,
// every record has exactly one consumer wave, a wave's "next own record" walks its records in stream order, and the head geometry
// covers every vocabulary row once. No GPU, no HIP headers.
#include "ring_geom.h"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <tuple>
#include <vector>

using namespace rwkvmi;

static int fails = 0;
#define CHECK(COND, ...) do { if (!(COND)) { fails++; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); if (fails > 20) exit(1); } } while (0)

static void check_shape(const RingShape & s, const char * name) {
    // rows expected per phase-matrix: (phase, mat) -> number of rows
    std::map<std::pair<int, int>, std::vector<int>> seen;   // (phase, mat) -> per row count
    auto rows_of = [&](int ph) { return ph == RG_W1 ? s.R5 : (ph == RG_DW1 ? s.DR : (ph == RG_FK ? s.F : s.D)); };
    for (int ph = 0; ph < RG_NPHASE; ph++)
        for (int m = 0; m < (ph == RG_C ? 4 : 1); m++) seen[{ph, m}] = std::vector<int>((size_t) rows_of(ph), 0);
    uint32_t layer_bytes0 = 0;
    for (int b = 0; b < RG_NBLK; b++) {
        const RingCu c = rg_cu(s, b);
        uint32_t off = 0;
        std::vector<std::tuple<uint32_t, int, int>> by_consumer[RG_NC];   // (offset, phase, j)
        for (int ph = 0; ph < RG_NPHASE; ph++) {
            CHECK(c.off[ph] == off, "%s wg %d phase %d: offset %u, expected %u", name, b, ph, c.off[ph], off);
            for (uint32_t j = 0; j < c.n[ph]; j++) {
                const RingRec r = rg_rec(s, b, ph, (int) j);
                CHECK(c.rec[ph] == rg_rec_bytes(s, r.R, r.K), "%s wg %d phase %d: record bytes", name, b, ph);
                for (int q = 0; q < r.R; q++) {
                    const int row = r.row0 + q;
                    auto & v = seen[{ph, ph == RG_C ? r.mat : 0}];
                    CHECK(row >= 0 && row < (int) v.size(), "%s wg %d phase %d record %u: row %d out of range", name, b, ph, j, row);
                    if (row >= 0 && row < (int) v.size()) v[(size_t) row]++;
                }
                int owner = -1, owners = 0;
                for (int cons = 0; cons < RG_NC; cons++)
                    for (uint32_t t = 0; t < rg_own_count(c, ph, cons); t++)
                        if (rg_own_j(c, ph, cons, (int) t) == j) { owner = cons; owners++; }
                CHECK(owners == 1, "%s wg %d phase %d record %u: %d owners", name, b, ph, j, owners);
                if (owner >= 0) by_consumer[owner].emplace_back(off + j * c.rec[ph], ph, (int) j);
            }
            off += c.n[ph] * c.rec[ph];
        }
        // the key sets the comm wave reads from the planes: the last ones of the workgroup, not in the stream
        for (int j = (int) c.n[RG_FK]; j < rg_key_sets(s, b); j++)
            for (int q = 0; q < 2; q++) { const int row = b * rg_gpb(s) * 32 + 2 * j + q; auto & v = seen[{RG_FK, 0}]; CHECK(row < (int) v.size(), "comm key row"); if (row < (int) v.size()) v[(size_t) row]++; }
        CHECK((int) c.n[RG_FK] + rg_key_comm(s, b) == rg_key_sets(s, b) && rg_key_comm(s, b) <= 2, "%s wg %d: key sets", name, b);
        CHECK(c.layer_bytes == off, "%s wg %d: layer bytes", name, b);
        for (int ph = 0; ph < RG_NPHASE; ph++) {   // a wave's records: as many as rg_own_count says, increasing, inside the phase, nothing behind them
            uint32_t total = 0;
            for (int cons = 0; cons < RG_NC; cons++) {
                const uint32_t k = rg_own_count(c, ph, cons);
                total += k;
                for (uint32_t t = 0; t < k; t++) CHECK(rg_own_j(c, ph, cons, (int) t) < c.n[ph] && (t == 0 || rg_own_j(c, ph, cons, (int) t) > rg_own_j(c, ph, cons, (int) t - 1)), "%s wg %d phase %d consumer %d: record %u", name, b, ph, cons, t);
                CHECK(rg_own_j(c, ph, cons, (int) k) >= c.n[ph], "%s wg %d phase %d consumer %d: a record behind the last", name, b, ph, cons);
            }
            CHECK(total == c.n[ph], "%s wg %d phase %d: %u of %u records owned", name, b, ph, total, c.n[ph]);
        }
        if (b == 0) layer_bytes0 = c.layer_bytes;
        // the cursor of a consumer wave: first own record of phase >= from, in stream order
        for (int cons = 0; cons < RG_NC; cons++) {
            for (int from = 0; from <= RG_NPHASE; from++) {
                uint32_t want = RG_NONE;
                for (auto & t : by_consumer[cons]) if (std::get<1>(t) >= from) { want = std::get<0>(t); break; }
                CHECK(rg_next_own_in_layer(c, cons, from) == want, "%s wg %d consumer %d from phase %d: next own record", name, b, cons, from);
            }
            for (size_t i = 1; i < by_consumer[cons].size(); i++)
                CHECK(std::get<0>(by_consumer[cons][i]) > std::get<0>(by_consumer[cons][i - 1]), "%s wg %d consumer %d: records out of stream order", name, b, cons);
        }
    }
    (void) layer_bytes0;
    for (auto & kv : seen)
        for (size_t r = 0; r < kv.second.size(); r++)
            CHECK(kv.second[r] == (kv.first.first == RG_DW1 ? 0 : 1) /* the decay rows are read from the planes, not streamed */, "%s phase %d matrix %d row %zu packed %d times", name, kv.first.first, kv.first.second, r, kv.second[r]);
    // E, FR and G share one row mapping (a wave keeps its rows' residual and receptance in registers across them)
    for (int b = 0; b < RG_NBLK; b += 37) {
        const RingCu c = rg_cu(s, b);
        CHECK(c.n[RG_E] == c.n[RG_FR] && c.n[RG_E] == c.n[RG_G] && c.rot[RG_E] == c.rot[RG_FR] && c.rot[RG_E] == c.rot[RG_G], "%s wg %d: E / FR / G mapping", name, b);
        for (uint32_t j = 0; j < c.n[RG_E]; j++)
            CHECK(rg_rec(s, b, RG_E, (int) j).row0 == rg_rec(s, b, RG_FR, (int) j).row0 && rg_rec(s, b, RG_E, (int) j).row0 == rg_rec(s, b, RG_G, (int) j).row0, "%s wg %d: E / FR / G rows", name, b);
    }
}

static void check_head(int n_vocab, int K) {
    const RingHead h = rg_head(n_vocab, K);
    CHECK(h.hg * RG_NBLK * 16 == n_vocab, "head: %d row groups do not cover %d rows", h.hg, n_vocab);
    CHECK(h.chunks * RG_HSTEPS * 32 == K, "head: chunks do not cover K = %d", K);
    std::set<uint32_t> offs;
    int groups = 0;
    for (int ps = 0; ps < h.passes; ps++) {
        const int npc = rg_head_npc(h, ps);
        CHECK(npc >= 1 && npc <= RG_NC, "head pass %d: %d consumers", ps, npc);
        groups += npc;
        for (int cons = 0; cons < npc; cons++)
            for (int ch = 0; ch < h.chunks; ch++) {
                const uint32_t o = rg_head_off(h, ps, ch, cons);
                CHECK(o % RG_HREC == 0 && o + RG_HREC <= h.bytes, "head record offset out of range");
                CHECK(offs.insert(o).second, "head record offset %u used twice", o);
            }
    }
    CHECK(groups == h.hg, "head: passes cover %d of %d row groups", groups, h.hg);
    CHECK(offs.size() * (size_t) RG_HREC == h.bytes, "head: records do not tile the head block");
}

int main() {
    struct Fmt { const char * name; int qs, scb, qhb; } fmts[] = {{"Q4_0", 16, 2, 0}, {"Q4_1", 16, 4, 0}, {"Q5_0", 16, 2, 4}, {"Q5_1", 16, 4, 4}, {"Q8_0", 32, 2, 0}};
    struct Geo { int D, F, R5, DR; } geos[] = {{2048, 7168, 160, 64}, {4096, 14336, 320, 128}, {4096, 14336, 160, 64}, {2560, 8960, 160, 64}};
    for (const Fmt & f : fmts)
        for (const Geo & g : geos) {
            for (int bal = 0; bal <= (g.D == 4096 ? 1 : 0); bal++) {
                RingShape s; s.D = g.D; s.F = g.F; s.R5 = g.R5; s.DR = g.DR; s.qs = f.qs; s.scb = f.scb; s.qhb = f.qhb; s.bal = bal;
                char name[64]; snprintf(name, sizeof name, "%s D=%d bal=%d", f.name, g.D, bal);
                check_shape(s, name);
            }
        }
    for (int v : {4096, 8192, 32768, 65536}) for (int K : {2048, 2560, 4096}) check_head(v, K);
    if (fails) { fprintf(stderr, "%d checks failed\n", fails); return 1; }
    printf("ring geometry OK\n");
    return 0;
}
