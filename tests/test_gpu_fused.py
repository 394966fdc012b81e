"""The fused single-token layers of RWKV-7 (six launches, fused_v7.hip) and RWKV-4 (four launches) against the CPU oracle and
against the one-kernel-per-graph-op path: logits and state bit for bit, quantised-from-FP32 files (low-rank matrices stay
F32) and directly generated quantised files (low-rank matrices / head in F16, as in a converted checkpoint)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import library, model, synth

pytestmark = pytest.mark.gpu

TOKENS = [1, 2, 3, 400, 5, 77, 300, 9, 11, 12]


def _file(tmp_path, name, fmt, direct):
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    if direct:
        synth.write_model(p, spec, fmt, seed=13)
    else:
        src = str(tmp_path / "f.bin")
        synth.write_model(src, spec, "FP32", seed=13)
        O.quantize_file(src, p, fmt)
    return p


@pytest.mark.parametrize("name,fmt,direct", [("test-v7", "Q4_0", False), ("test-v7", "Q5_1", True), ("test-v7", "Q8_0", True), ("test-v7", "Q4_1", False),
                                             ("test-v7", "Q5_0", True), ("test-v4", "Q4_0", False), ("test-v4", "Q5_1", False), ("test-v4", "Q8_0", True)])
def test_fused_layer_matches_oracle_and_generic_path(tmp_path, name, fmt, direct):
    library()
    p = _file(tmp_path, name, fmt, direct)
    om = O.OracleModel(p)
    os.environ["RWKV_MI_NO_MEGA"] = "1"     # (the persistent launch of persist_v47.hip takes these models by default: tests/test_gpu_persist_v47.py)
    try:
        m = model(p)
    finally:
        del os.environ["RWKV_MI_NO_MEGA"]
    assert m.decode_path() == 1, "fused path not selected"
    os.environ["RWKV_MI_NO_FUSED"] = "1"
    try:
        g = model(p)
    finally:
        del os.environ["RWKV_MI_NO_FUSED"]
    assert g.decode_path() == 0
    ost, st, gst = om.init_state(), None, None
    for i, t in enumerate(TOKENS):
        t %= synth.CONFIGS[name].n_vocab
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)
        gl, gst = g.eval(t, gst)
        assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, fmt, i)
        assert np.array_equal(gl, ol) and np.array_equal(gst, ost), (name, fmt, i)
    # graph-replayed greedy loop on the fused path == serial evaluation
    m.state_load(None)
    toks, _ = m.decode_greedy(5, 6)
    st2, tok, ref = None, 5, []
    for _ in range(6):
        lg, st2 = g.eval(tok, st2)
        tok = int(np.argmax(lg))
        ref.append(tok)
    assert list(toks) == ref
    m.free(); g.free(); om.free()
