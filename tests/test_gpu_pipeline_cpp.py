"""The layer pipeline behind the unmodified rwkv.h ABI (csrc/pipeline.cpp): RWKV_MI_DEVICES lists one device per stage;
rwkv_init_from_file builds the chain and rwkv_eval / rwkv_eval_sequence(_in_chunks) / rwkv_clone_context walk it. On a one-GPU
box the stages share device 0 ("0,0", "0,0,0"): same code path, peer copies degenerate to device-to-device copies. Results must
equal the single-context ones bit for bit (and hence the oracle's)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import library, model, synth

pytestmark = pytest.mark.gpu

TOKENS = [1, 2, 3, 400, 5, 77, 300, 9]


def _pipeline_model(p, devices):
    os.environ["RWKV_MI_DEVICES"] = devices
    try:
        return model(p)
    finally:
        del os.environ["RWKV_MI_DEVICES"]


@pytest.mark.parametrize("name,fmt,devices", [("mega-v6-2048", "Q4_0", "0,0"), ("test-v6", "Q5_1", "0,0"), ("test-v7", "Q8_0", "0,0,0"),
                                              ("test-v4", "Q4_0", "0,0"), ("test-v5.2", "FP16", "0-0"), ("test-v7", "Q4_1", "0,0")])
def test_stage_chain_equals_single_context(tmp_path, name, fmt, devices):
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    if fmt in ("FP16", "FP32") or name.startswith("mega"):
        synth.write_model(p, spec, fmt, seed=19)
    else:
        src = str(tmp_path / "f.bin")
        synth.write_model(src, spec, "FP32", seed=19)
        O.quantize_file(src, p, fmt)
    om = O.OracleModel(p)
    pm = _pipeline_model(p, devices)
    assert pm.n_layer == spec.n_layer and pm.state_len == om.state_len
    toks = [t % spec.n_vocab for t in TOKENS]
    ost, st = om.init_state(), None
    for i, t in enumerate(toks):
        ol, ost = om.eval(t, ost)
        lg, st = pm.eval(t, st)
        assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, devices, i)
    # sequence mode through the chain (one pass, and chunks)
    seq = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(45)]
    ol, ost = om.eval_sequence(seq, om.init_state())
    lg, st = pm.eval_sequence(seq, None)
    assert np.array_equal(lg, ol) and np.array_equal(st, ost)
    lg, st = pm.eval_sequence_in_chunks(seq, None, chunk_size=7)
    assert np.array_equal(lg, ol) and np.array_equal(st, ost)
    # clones walk their own chain
    c = pm.clone()
    lg2, st2 = c.eval_sequence(seq, None)
    assert np.array_equal(lg2, ol) and np.array_equal(st2, ost)
    c.free()
    pm.free()
    om.free()


@pytest.mark.parametrize("name,fmt,devices", [("test-v7", "Q8_0", "0,0,0"), ("test-v6", "Q4_0", "0,0,0")])
def test_many_passes_through_a_three_stage_chain(tmp_path, name, fmt, devices):
    """A middle stage's x is input, running residual stream and the source of its outgoing copy: the stage before it may overwrite it
    for the next pass only behind that copy (round 2 recorded the reuse event behind the layers instead: a rare mismatch of
    eval_sequence_in_chunks, reproduced once in 30 runs of the test above). Many short passes, repeated."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    src = str(tmp_path / "f.bin")
    synth.write_model(src, spec, "FP32", seed=31)
    O.quantize_file(src, p, fmt)
    om = O.OracleModel(p)
    pm = _pipeline_model(p, devices)
    seq = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(96)]
    ol, ost = om.eval_sequence(seq, om.init_state())
    for rep in range(25):
        for chunk in (3, 7, 32):
            lg, st = pm.eval_sequence_in_chunks(seq, None, chunk_size=chunk)
            assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, rep, chunk)
    pm.free()
    om.free()


def test_bad_device_list_is_an_argument_error(tmp_path):
    lib = library()
    p = str(tmp_path / "m.bin")
    synth.write_model(p, synth.CONFIGS["test-v4"], "FP32", seed=1)
    for bad in ("x", "0,,1", "3-1", "99"):
        os.environ["RWKV_MI_DEVICES"] = bad
        try:
            with pytest.raises(ValueError):
                model(p)
        finally:
            del os.environ["RWKV_MI_DEVICES"]
