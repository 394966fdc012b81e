"""The deterministic scalar routines (exp, tanh, sigmoid, ...) must agree bit for bit between kernels and oracle."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import gpu_unary

pytestmark = pytest.mark.gpu
NAMES = ["exp", "tanh", "sigmoid", "silu", "exp(-exp)", "v7 decay", "rsqrt(x+1e-5)"]


@pytest.mark.parametrize("op", range(7))
def test_unary_bit_exact(op):
    rng = np.random.default_rng(op)
    parts = [rng.standard_normal(1 << 20) * s for s in (1e-6, 1e-3, 0.1, 1.0, 5.0, 30.0)]
    parts.append(np.linspace(-110.0, 95.0, 1 << 18))
    parts.append(np.array([0.0, -0.0, 1e-30, -1e-30, 88.7, 88.73, -103.9, -104.1, 20.0, -20.0, 1e-4, 0.625, np.inf, -np.inf]))
    x = np.concatenate(parts).astype(np.float32)
    if op == 6:
        x = np.abs(x)
    g, c = gpu_unary(op, x), O.unary(op, x)
    bad = ~((g == c) | (np.isnan(g) & np.isnan(c)))
    assert not bad.any(), (NAMES[op], int(bad.sum()), x[bad][:5], g[bad][:5], c[bad][:5])
    if op == 0:  # and it is a faithful expf: within 1 ulp of the float64 value
        xs = x[(x > -80) & (x < 80)]
        ref = np.exp(xs.astype(np.float64))
        err = np.abs(O.unary(0, xs).astype(np.float64) - ref) / np.spacing(ref.astype(np.float32)).astype(np.float64)
        assert err.max() <= 0.5001


def test_activation_quantiser_bit_exact():
    from gpu_lib import gpu_quantize_act
    rng = np.random.default_rng(11)
    parts = [rng.standard_normal(32 * 4096) * s for s in (1e-6, 1e-3, 1.0, 50.0)]
    parts.append(np.maximum(rng.standard_normal(32 * 4096), 0) ** 2)          # relu^2-like (channel-mixing key)
    parts.append(1.0 / (1.0 + np.exp(-rng.standard_normal(32 * 4096) * 3)))   # sigmoid-like, all positive
    parts.append(np.zeros(64))
    ties = np.arange(32 * 64, dtype=np.float64).reshape(64, 32) % 255 - 127.0  # exact integers and halves: rounding ties
    ties[:, 0] = 127.0
    parts.append((ties * 0.5).reshape(-1))
    x = np.concatenate(parts).astype(np.float32)
    q, d, s, isum = gpu_quantize_act(x)
    oq, od, os_ = O.quantize_act(x)
    assert np.array_equal(q, oq), int((q != oq).sum())
    assert np.array_equal(d, od)
    assert np.array_equal(s, os_), (int((s != os_).sum()), s[s != os_][:4], os_[s != os_][:4])
    assert np.array_equal(isum, oq.reshape(-1, 32).astype(np.int32).sum(axis=1))


@pytest.mark.parametrize("op", [7, 8, 9])
def test_fast_wave_reductions_match_the_reference_butterfly(op):
    # op 7: float xor-butterfly, op 8: double, op 9: half-wave max / int sum; 1.0 where DPP/permlane form == ds_bpermute form
    rng = np.random.default_rng(op)
    x = (rng.standard_normal(64 * 4096) * np.repeat(10.0 ** rng.uniform(-6, 6, 4096), 64)).astype(np.float32)
    if op == 9:
        x = np.clip(x, -127, 127)
    ok = gpu_unary(op, x)
    assert np.all(ok == 1.0), int((ok != 1.0).sum())
