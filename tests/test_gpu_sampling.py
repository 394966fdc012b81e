"""Temperature / top-p sampling on the device (sampling.hip) against the reference's python/sampling.py statements restated in
float64 NumPy (the reference file itself only exists in the dev container; its algorithm is restated here line by line)."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import library, model, synth

pytestmark = pytest.mark.gpu


def ref_distribution(logits, temperature, top_p):
    """sample_probs() of the reference (python/sampling.py:19-52) up to the final np.random.choice: returns the probabilities."""
    x = logits.astype(np.float64)
    x = x - x.max()
    probs = np.exp(x) / np.exp(x).sum()
    if top_p == 0.0:
        top_p = 1.0
    if temperature == 0.0:
        out = np.zeros_like(probs); out[int(np.argmax(probs))] = 1.0
        return out
    if top_p < 1.0:
        sp = np.sort(probs)[::-1]
        cutoff = float(sp[np.argmax(np.cumsum(sp) > top_p)])
        probs = np.where(probs < cutoff, 0.0, probs)
    if temperature != 1.0:
        probs = np.power(probs, 1.0 / temperature)
    return probs / probs.sum()


@pytest.fixture(scope="module")
def ctx(tmp_path_factory):
    library()
    p = str(tmp_path_factory.mktemp("s") / "m.bin")
    synth.write_model(p, synth.CONFIGS["test-v6"], "Q8_0", seed=3)
    m = model(p)
    yield m
    m.free()


@pytest.mark.parametrize("temperature,top_p", [(1.0, 0.8), (0.7, 0.5), (1.5, 1.0), (1.0, 0.0), (0.0, 0.8), (0.3, 0.95)])
def test_sampler_follows_the_reference_distribution(ctx, temperature, top_p):
    m = ctx
    logits, _ = m.eval(7, None)
    logits = (logits * 6.0).astype(np.float32)      # a peaked distribution
    # evaluate again so that the device logits are the scaled ones? No: the sampler reads the context's own logits -> compare on those
    logits, _ = m.eval(7, None)
    pr = ref_distribution(logits, temperature, top_p)
    cdf = np.cumsum(pr)
    for u in np.linspace(0.001, 0.999, 41):
        tok = m.sample(temperature, top_p, u=float(u))
        assert pr[tok] > 0.0, (temperature, top_p, u, tok)
        lo = cdf[tok] - pr[tok]
        # the token's probability interval contains u (up to f32 rounding at the interval ends)
        assert lo - 1e-4 <= u <= cdf[tok] + 1e-4, (temperature, top_p, u, tok, lo, cdf[tok])


def test_sampling_decode_loop_is_reproducible_and_varied(ctx):
    m = ctx
    m.state_load(None)
    a, _ = m.decode_sample(5, 32, temperature=1.0, top_p=0.9, seed=1234)
    m.state_load(None)
    b, _ = m.decode_sample(5, 32, temperature=1.0, top_p=0.9, seed=1234)
    m.state_load(None)
    c, _ = m.decode_sample(5, 32, temperature=1.0, top_p=0.9, seed=99)
    assert list(a) == list(b) and list(a) != list(c)
    m.state_load(None)
    g, _ = m.decode_greedy(5, 32)
    m.state_load(None)
    z, _ = m.decode_sample(5, 32, temperature=0.0, top_p=0.9, seed=1)   # temperature 0 = argmax
    assert list(z) == list(g)
