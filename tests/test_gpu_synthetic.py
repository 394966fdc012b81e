"""GPU vs CPU oracle on synthetic models with real head geometry (S = 64) and every weight format,
plus end-to-end properties at sizes the oracle finishes in seconds."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import library, model, synth

pytestmark = pytest.mark.gpu

TOKENS = [1, 2, 3, 400, 5, 77, 300, 9]
# The kernels implement the oracle's arithmetic exactly (DESIGN.md "Numerics"): results must be bit-identical.


@pytest.mark.parametrize("name", ["test-v4", "test-v5.1", "test-v5.2", "test-v6", "test-v7"])
@pytest.mark.parametrize("fmt", ["FP32", "FP16", "Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0"])
def test_gpu_matches_oracle(tmp_path, name, fmt):
    lib = library()
    lib.rwkv_set_print_errors(None, False)
    src = str(tmp_path / "src.bin")
    synth.write_model(src, synth.CONFIGS[name], "FP32" if fmt == "FP32" else "FP16", seed=7)
    path = src
    if fmt not in ("FP32", "FP16"):
        path = str(tmp_path / "q.bin")
        lib.rwkv_quantize_model_file(src, path, fmt)
    lib.rwkv_set_print_errors(None, True)
    om = O.OracleModel(path)
    m = model(path)
    assert m.state_len == om.state_len and m.bytes_per_token() == om.bytes_per_token
    ost = om.init_state()
    st = None
    for t in TOKENS:
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)
    assert np.isfinite(lg).all()
    assert np.array_equal(lg, ol), (name, fmt, float(np.abs(lg - ol).max()))
    assert np.array_equal(st, ost), (name, fmt, float(np.abs(st - ost).max()))
    # sequence == serial (bit-exact), also across an odd split
    lg2, st2 = m.eval_sequence(TOKENS, None)
    assert np.array_equal(lg2, lg) and np.array_equal(st2, st)
    lg3, st3 = m.eval_sequence_in_chunks(TOKENS, None, chunk_size=3)
    assert np.array_equal(lg3, lg) and np.array_equal(st3, st)
    m.free()
    om.free()


def test_direct_quantised_synthetic_file_loads(tmp_path):
    # the bench generates quantised blocks directly; make sure such files go through the same path
    p = str(tmp_path / "d.bin")
    synth.write_model(p, synth.CONFIGS["test-v6"], "Q4_0", seed=3)
    om = O.OracleModel(p)
    m = model(p)
    ost, st = om.init_state(), None
    for t in TOKENS:
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)
    assert np.array_equal(lg, ol) and np.array_equal(st, ost)
    m.free()
