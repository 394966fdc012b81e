"""Sequence mode on F16 matrices on the matrix cores (k_mmf16_seq, csrc/kernels.hip): FP16 model files and the F16 low-rank stages of
RWKV-7 (rwkv_graph.inc:416-447) / the F16 emb of a quantised checkpoint, T >= 32 tokens per pass.

Contract (BASELINE north star; reference tests/test_eval_sequence_in_chunks.c:54, tests/test_tiny_rwkv.c:38-54): bit-exactness with the
CPU path is asked for FP32 files; FP16 files are validated against recorded thresholds. The matrix core keeps ggml's operand rounding
(activations -> fp16, exact products, f32 accumulation) but not the ORDER of the additions, so this arm is compared with the oracle
within a tolerance that is stated here:

  kernel level   |y - ref64| <= 2e-6 * sum_k |w_k x_k|    (ref64: the same fp16-rounded operands summed in float64; the oracle itself,
                 with ggml's order, sits inside the same bound): every product is right to rounding
  model level    logits: max |gpu - oracle| <= 1e-2 * (1 + max |oracle|), greedy token identical; state: same bound per layer slice.
                 Loose on purpose: these are RANDOM-weight networks, and RWKV-7's recurrence (decay = exp(-0.6065 sigmoid(.)), the
                 l2-normalised key, the in-context learning rate: all fed by the low-rank stages) amplifies a last-bit difference of one
                 product over the tokens of a pass -- measured 4e-4 on the three-layer stand-in at 32 tokens, 4e-3 on the two-layer
                 2.9B slice at 128 tokens (logits of magnitude 2 - 3), while the same kernel is inside 2e-6 relative per product. The
                 reference's own FP16 acceptance is of that order and looser (tests/test_tiny_rwkv.c:38-54: difference sums 0.006 - 0.46).

The exact arm (RWKV_MI_SEQ_F16=valu, k_mvf in ggml's order) stays bit-identical to the oracle -- every other test of the suite runs on
it (tests/conftest.py) -- and FP32 matrices never take the matrix-core kernel."""
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import gpu_mul_mat, hooks_library, library, model, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def mfma_arm():
    old = os.environ.get("RWKV_MI_SEQ_F16")
    os.environ["RWKV_MI_SEQ_F16"] = "mfma"
    yield
    if old is None:
        del os.environ["RWKV_MI_SEQ_F16"]
    else:
        os.environ["RWKV_MI_SEQ_F16"] = old


# K: the RWKV-7 2.9B ranks (96, 64, 320) and row length (2560), the 169M row (768), fewer steps than waves (32); N: below a
# 32-row wave tile, ragged, the ranks, more than one 128-row workgroup; T: one tile pair, ragged below / above 64, several token tiles
@pytest.mark.parametrize("K,N,T", [(96, 2560, 64), (64, 130, 33), (320, 96, 100), (2560, 96, 32), (2560, 320, 130), (96, 40, 65), (768, 257, 97), (32, 70, 64)])
def test_f16_gemm_against_float64_and_oracle(K, N, T):
    rng = np.random.default_rng(K + 7 * N + 13 * T)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float16)
    x = rng.standard_normal((T, K)).astype(np.float32)
    x[0, :16] *= 30.0
    x[T - 1] *= 0.01
    t = O.TYPE_IDS["FP16"]
    y = gpu_mul_mat(t, w.view(np.uint8).reshape(-1), K, N, x)
    xh = x.astype(np.float16).astype(np.float64)
    wd = w.astype(np.float64)
    ref64 = xh @ wd.T
    bound = 2e-6 * (np.abs(xh) @ np.abs(wd).T) + 1e-30
    assert (np.abs(y - ref64) <= bound).all(), float((np.abs(y - ref64) / bound).max())
    ref = O.mul_mat(t, w.view(np.uint8).reshape(-1), K, N, x)
    assert (np.abs(ref - ref64) <= bound).all()
    # the exact arm on the same operands IS the oracle
    os.environ["RWKV_MI_SEQ_F16"] = "valu"
    assert np.array_equal(gpu_mul_mat(t, w.view(np.uint8).reshape(-1), K, N, x), ref)
    # ... and FP32 matrices never leave it
    os.environ["RWKV_MI_SEQ_F16"] = "mfma"
    w32 = w.astype(np.float32)
    t32 = O.TYPE_IDS["FP32"]
    assert np.array_equal(gpu_mul_mat(t32, w32.view(np.uint8).reshape(-1), K, N, x), O.mul_mat(t32, w32.view(np.uint8).reshape(-1), K, N, x))


def _close(a, b, what):
    tol = 1e-2 * (1.0 + float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol, (what, err, tol)
    return err


@pytest.mark.parametrize("name,fmt", [("test-v6", "FP16"), ("test-v5.2", "FP16"), ("test-v4", "FP16"), ("test-v7", "FP16"),
                                      ("test-v7", "Q5_1"), ("slice-v7-2560", "Q5_1"), ("slice-v7-2560", "Q4_0")])
@pytest.mark.parametrize("T", [32, 97])
def test_sequence_pass_on_the_matrix_cores(tmp_path, name, fmt, T):
    """FP16 files (every matrix F16) and directly generated quantised RWKV-7 files (low-rank stages F16, as in a converted checkpoint)."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=59)
    toks = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(T)]
    om = O.OracleModel(p)
    ol, ost = om.eval_sequence(toks, om.init_state())
    m = model(p)
    gl, gst = m.eval_sequence(toks, None)
    err = _close(gl, ol, "logits")
    assert int(np.argmax(gl)) == int(np.argmax(ol))
    per = om.state_len // spec.n_layer
    for layer in range(spec.n_layer):
        _close(gst[layer * per:(layer + 1) * per], ost[layer * per:(layer + 1) * per], f"state of layer {layer}")
    # not bit-identical (or this arm is not the one that ran) -- except where nothing F16 is long enough to take it
    assert err > 0.0 or fmt != "FP16"
    # the exact arm on the same file: bit-identical
    os.environ["RWKV_MI_SEQ_F16"] = "valu"
    el, est = m.eval_sequence(toks, None)
    assert np.array_equal(el, ol) and np.array_equal(est, ost)
    # single tokens never take the matrix-core kernel: decode is bit-exact on either setting
    os.environ["RWKV_MI_SEQ_F16"] = "mfma"
    st, ost2 = None, om.init_state()
    for t in toks[:4]:
        lg, st = m.eval(t, st)
        olg, ost2 = om.eval(t, ost2)
    assert np.array_equal(lg, olg) and np.array_equal(st, ost2)
    m.free()
    om.free()


def _mmf16_launches():
    import ctypes
    L = hooks_library().library
    L.rwkv_mi_test_mmf16_launches.restype = ctypes.c_uint64
    return int(L.rwkv_mi_test_mmf16_launches())


@pytest.mark.parametrize("name,fmt", [("slice-v7-2560", "Q5_1"), ("test-v7", "Q5_1"), ("test-v6", "FP16"), ("test-v4", "FP16")])
@pytest.mark.parametrize("T", [32, 33, 64, 97, 130])
def test_one_layer_slices_on_the_matrix_cores_tight(tmp_path, name, fmt, T):
    """The product's DEFAULT sequence arm for F16 matrices at ragged lengths, held two orders tighter than the multi-layer cases above:
    ONE layer (no depth to amplify a last-bit difference through further recurrences), logits and state within 1e-4 * (1 + max |oracle|).
    A tile-edge or split-K wiring error is worth >= 1e-3 here. The launch counter -- not err > 0 -- proves the matrix-core kernel ran."""
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=61, limit_layers=1)
    toks = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(T)]
    om = O.OracleModel(p)
    ol, ost = om.eval_sequence(toks, om.init_state())
    m = model(p, hooks=True)
    before = _mmf16_launches()
    gl, gst = m.eval_sequence(toks, None)
    assert _mmf16_launches() > before, "the F16 matrix-core kernel did not run"
    for a, b, what in ((gl, ol, "logits"), (gst, ost, "state")):
        tol = 1e-4 * (1.0 + float(np.abs(b).max()))
        err = float(np.abs(a - b).max())
        assert err <= tol, (name, fmt, T, what, err, tol)
    # chunked: passes of 40 tokens take the kernel again at other tile shapes (ragged last pass below 32 tokens: the exact kernels)
    cl, cst = m.eval_sequence_in_chunks(toks, None, chunk_size=40)
    for a, b, what in ((cl, ol, "chunked logits"), (cst, ost, "chunked state")):
        tol = 1e-4 * (1.0 + float(np.abs(b).max()))
        assert float(np.abs(a - b).max()) <= tol, (name, fmt, T, what)
    os.environ["RWKV_MI_SEQ_F16"] = "valu"
    before = _mmf16_launches()
    el, est = m.eval_sequence(toks, None)
    assert _mmf16_launches() == before and np.array_equal(el, ol) and np.array_equal(est, ost)
    os.environ["RWKV_MI_SEQ_F16"] = "mfma"
    m.free()
    om.free()
