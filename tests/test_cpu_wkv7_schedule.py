"""The out chain of k_wkv7_seq (csrc/prefill.hip) as index algebra, no GPU: 16 lanes of a row each park the products of token t in a private
16-slot LDS ring and add them -- one step of lag, entry read a step ahead -- onto the running sum of the lane before; lane 15 emits
out[sigma - 16]. The model runs the kernel's schedule on symbols and checks that every emitted token is the ordered sum of its own sixteen
entries (the reference's j = 0 .. 63 order: lane q holds j = 4q .. 4q + 3), that no entry is read after its slot was overwritten, and that
the extra read at the start of a whole chunk returns the entry the lane already holds."""
import pytest

LANES, SLOTS, CH, LAG = 16, 16, 32, 16


def run_schedule(T):
    ring = [[None] * SLOTS for _ in range(LANES)]          # ring[q][slot] = token whose products lane q parked there
    o_run = [[] for _ in range(LANES)]                     # the chain a lane last produced: list of (token, lane) terms, in order
    pq_n = [None] * LANES                                  # token whose entry the lane adds at the next step (None = uninitialised LDS)
    out = {}
    n_chunks = (T + CH - 1) // CH

    def step(sigma, upd):
        nonlocal o_run, pq_n
        new = []
        for q in range(LANES):
            prev = o_run[q - 1] if q > 0 else []            # row_shr:1, lane 0 receives the reference's 0
            new.append(prev + [(pq_n[q], q)])
        o_run = new
        if sigma >= LAG:
            assert sigma - LAG < T
            out[sigma - LAG] = o_run[LANES - 1]
        if upd:
            for q in range(LANES):
                ring[q][sigma % SLOTS] = sigma
        pq_n = [ring[q][(sigma - q) % SLOTS] for q in range(LANES)]

    for c in range(n_chunks):
        n = min(CH, T - CH * c)
        if n == CH:
            again = [ring[q][(CH * c - 1 - q) % SLOTS] for q in range(LANES)]
            if c > 0:
                assert again == pq_n                        # the re-read that equalises the LDS queue returns what the lane holds
            pq_n = again
        for tt in range(n):
            step(CH * c + tt, True)
    for sigma in range(T, T + LAG):
        step(sigma, False)
    return out


@pytest.mark.parametrize("T", [32, 33, 47, 48, 64, 65, 97, 250, 1024])
def test_every_token_is_the_ordered_sum_of_its_own_entries(T):
    out = run_schedule(T)
    assert sorted(out) == list(range(T))
    for t, chain in out.items():
        assert chain == [(t, q) for q in range(LANES)], (T, t, chain)
