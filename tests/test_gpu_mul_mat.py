"""Projection kernels vs the oracle's ggml-semantics mul_mat on seeded random data at real model sizes
(K = 4096 and 14336 are RWKV-6 7B's row lengths; 2560 is RWKV-7 2.9B's; 64 exercises the short-row path)."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import gpu_mul_mat

pytestmark = pytest.mark.gpu

FORMATS = ["FP32", "FP16", "Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0"]


def _weights(rng, fmt, K, N):
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    t = O.TYPE_IDS[fmt]
    if fmt == "FP32":
        return t, w.view(np.uint8).reshape(-1)
    if fmt == "FP16":
        return t, w.astype(np.float16).view(np.uint8).reshape(-1)
    return t, np.concatenate([O.quantize_row(t, w[n]) for n in range(N)])


@pytest.mark.parametrize("fmt", FORMATS)
# K of every BASELINE configuration: RWKV-4 169M (768, 3072), RWKV-6 1.6B (2048, 7168), RWKV-7 2.9B (2560, 10240), RWKV-6 7B (4096, 14336)
@pytest.mark.parametrize("K,N", [(64, 96), (2560, 130), (4096, 515), (14336, 67), (768, 200), (3072, 70), (2048, 133), (7168, 66), (10240, 35)])
def test_single_token_matches_oracle(fmt, K, N):
    rng = np.random.default_rng(K * 7 + N)
    t, wb = _weights(rng, fmt, K, N)
    x = rng.standard_normal(K).astype(np.float32)
    x[:32] *= 30.0  # one block with a large scale
    y = gpu_mul_mat(t, wb, K, N, x)[0]
    ref = O.mul_mat(t, wb, K, N, x)[0]
    # same accumulation order as the oracle (F32/F16: ggml's AVX2 dot; quantised: 64 interleaved partials): bit-identical
    assert np.array_equal(y, ref), (fmt, K, N, float(np.abs(y - ref).max()))


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("K,N,T", [(128, 40, 3), (4096, 70, 9), (14336, 33, 17), (2560, 50, 8), (768, 40, 5), (2048, 70, 33), (7168, 33, 12), (10240, 20, 6), (3072, 24, 4)])
def test_token_tiled_is_bit_identical_to_single_token(fmt, K, N, T):
    rng = np.random.default_rng(K + N + T)
    t, wb = _weights(rng, fmt, K, N)
    x = rng.standard_normal((T, K)).astype(np.float32)
    y = gpu_mul_mat(t, wb, K, N, x)
    ref = O.mul_mat(t, wb, K, N, x)
    assert np.array_equal(y, ref)
    for i in range(T):
        assert np.array_equal(gpu_mul_mat(t, wb, K, N, x[i])[0], y[i]), (fmt, K, N, T, i)


@pytest.mark.parametrize("K,V,rows", [(768, 50277, 1200), (2048, 65536, 700)])
def test_head_slice_f16(K, V, rows):
    """The F16 head of RWKV-4 169M (50277 x 768: a vocabulary that is no multiple of the 64-row tile) and of the World models:
    the first / last rows of a full-size head against the oracle."""
    rng = np.random.default_rng(V)
    w = (rng.standard_normal((V, K)) * 0.03).astype(np.float16)
    x = rng.standard_normal(K).astype(np.float32)
    y = gpu_mul_mat(O.TYPE_IDS["FP16"], w.view(np.uint8).reshape(-1), K, V, x)[0]
    for sl in (slice(0, rows), slice(V - rows, V)):
        ws = np.ascontiguousarray(w[sl])
        ref = O.mul_mat(O.TYPE_IDS["FP16"], ws.view(np.uint8).reshape(-1), K, ws.shape[0], x)[0]
        assert np.array_equal(y[sl], ref)


def test_linearity_property_q8_0():
    # size-independent property: with weights exactly representable, W.(e_k * c) picks column k up to activation rounding
    K, N = 4096, 64
    rng = np.random.default_rng(3)
    w = rng.integers(-127, 128, size=(N, K)).astype(np.float32)
    w[:, ::32] = 127.0  # every block's amax is 127 -> d = 1, codes = values
    t = O.TYPE_IDS["Q8_0"]
    wb = np.concatenate([O.quantize_row(t, w[n]) for n in range(N)])
    for k in (0, 31, 32, 1000, 4095):
        x = np.zeros(K, dtype=np.float32)
        x[k] = 2.0
        y = gpu_mul_mat(t, wb, K, N, x)[0]
        assert np.allclose(y, 2.0 * w[:, k], rtol=1e-3, atol=1e-3)
