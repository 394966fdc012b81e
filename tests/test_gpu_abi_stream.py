"""rwkv_eval with the caller's state streamed in layer groups under the layers (engine.hip, forward_streamed): the unmodified ABI
(rwkv_eval.inc:2-22,38-76; rwkv.h:106-108 -- state_in / state_out are host buffers on every call) on every decode path, against the CPU
oracle bit for bit and against the serial upload -> token -> download form. RWKV_MI_ABI_STREAM=1 forces the streamed form for the small
test states (the default takes it from 4 MB of state on)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import library, model, synth

pytestmark = pytest.mark.gpu

TOKENS = [1, 2, 3, 400, 5, 77, 300, 9]


@pytest.fixture(autouse=True)
def streamed():
    os.environ["RWKV_MI_ABI_STREAM"] = "1"
    os.environ["RWKV_MI_NO_AUTOTUNE"] = "1"
    yield
    del os.environ["RWKV_MI_ABI_STREAM"]
    del os.environ["RWKV_MI_NO_AUTOTUNE"]


# the ring kernel on layer ranges (3 layers -> 3 groups; the folded head in the last one), the seven-launch RWKV-6 layer, RWKV-7 / RWKV-4
# fused layers, the one-kernel-per-op path (FP32 / RWKV-5), a 32-layer model (eight groups of four)
@pytest.mark.parametrize("name,fmt,env", [("mega-v6-2048", "Q4_0", {}), ("mega-v6-2048-v8k", "Q5_1", {}), ("mega-v6-4096", "Q8_0", {}),
                                          ("mega-v6-2048", "Q4_0", {"RWKV_MI_NO_MEGA": "1"}), ("test-v7", "Q5_1", {}), ("test-v4", "Q4_0", {}),
                                          ("test-v5.2", "FP32", {}), ("test-v6", "FP16", {}), ("chain-v6-32x256", "Q4_0", {})])
def test_streamed_eval_matches_oracle(tmp_path, name, fmt, env):
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=83)
    om = O.OracleModel(p)
    os.environ.update(env)
    try:
        m = model(p)
    finally:
        for k in env:
            del os.environ[k]
    toks = [t % spec.n_vocab for t in TOKENS]
    # (1) state_in = None, then separate in / out buffers
    ost = om.init_state()
    st = None
    for i, t in enumerate(toks[:4]):
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)
        assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, fmt, "separate", i)
    # (2) in place: state_in == state_out, caller-provided logits buffer
    state = st.copy()
    logits = np.empty(m.n_vocab, dtype=np.float32)
    for i, t in enumerate(toks[4:]):
        ol, ost = om.eval(t, ost)
        m.eval(t, state, state, logits)
        assert np.array_equal(logits, ol) and np.array_equal(state, ost), (name, fmt, "in place", i)
    # (3) the same token through the serial form (upload, token, download) from the same state
    before = state.copy()
    a_l, a_s = m.eval(5, before)
    os.environ["RWKV_MI_ABI_STREAM"] = "0"
    b_l, b_s = m.eval(5, before)
    os.environ["RWKV_MI_ABI_STREAM"] = "1"
    assert np.array_equal(a_l, b_l) and np.array_equal(a_s, b_s)
    # (4) the resident-state extensions continue from what the streamed call left on the device
    m.state_load(before)
    toks_r, _ = m.decode_greedy(5, 3)
    st2, tok, ref = before, 5, []
    for _ in range(3):
        lg, st2 = m.eval(tok, st2)
        tok = int(np.argmax(lg))
        ref.append(tok)
    assert list(toks_r) == ref
    assert m.healthy()
    m.free()
    om.free()


def test_streamed_eval_without_logits_or_without_state_out(tmp_path):
    """rwkv_eval accepts logits_out == NULL (rwkv_eval.inc:38-76: logits skipped) and the binding passes state_out always; the C ABI
    also takes state_out == NULL in this library (nothing to download)."""
    import ctypes
    lib = library()
    L = lib.library
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS["mega-v6-2048"]
    synth.write_model(p, spec, "Q4_0", seed=89)
    om = O.OracleModel(p)
    m = model(p)
    ost = om.init_state()
    ol, ost1 = om.eval(7, ost)
    state = m.init_state()
    F = ctypes.POINTER(ctypes.c_float)
    assert L.rwkv_eval(m._ctx.ptr, 7, state.ctypes.data_as(F), state.ctypes.data_as(F), None)     # no logits
    assert np.array_equal(state, ost1)
    ol2, ost2 = om.eval(8, ost1)
    logits = np.empty(m.n_vocab, dtype=np.float32)
    assert L.rwkv_eval(m._ctx.ptr, 8, state.ctypes.data_as(F), None, logits.ctypes.data_as(F))    # no state out
    assert np.array_equal(logits, ol2)
    assert np.array_equal(m.state_store(), ost2)                                                   # ... but it is on the device
    m.free()
    om.free()


def test_streamed_eval_error_before_the_download_job_is_armed(tmp_path):
    """forward_streamed used to publish the download job (caller's state_out pointer included) BEFORE a fallible set-up step and return
    early on its failure: the download thread then finished the NEXT call's groups inside the stale job and ran the new job a second time
    after that call had returned -- a late write into caller memory. Inject the failure (test hook), then: the failed call leaves the
    caller's buffers untouched, the next call is right, and a buffer handed to the failed call is never written afterwards."""
    import ctypes
    import time
    from gpu_lib import hooks_library
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS["mega-v6-2048"]
    synth.write_model(p, spec, "Q4_0", seed=87)
    om = O.OracleModel(p)
    m = model(p, hooks=True)
    H = hooks_library().library
    H.rwkv_mi_test_fail_state_init.argtypes = [ctypes.c_int]
    H.rwkv_mi_test_fail_state_init.restype = None
    sentinel = np.float32(-12345.5)
    victim = np.full(m.state_len, sentinel, dtype=np.float32)
    logits = np.full(m.n_vocab, sentinel, dtype=np.float32)
    H.rwkv_mi_test_fail_state_init(1)
    with pytest.raises(ValueError):
        m.eval(5, None, victim, logits)           # state_in NULL: the fresh state is initialised on the device -> injected failure
    assert (victim == sentinel).all() and (logits == sentinel).all()
    ost = om.init_state()
    st = None
    for t in TOKENS[:3]:
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)                    # other output buffers
        assert np.array_equal(lg, ol) and np.array_equal(st, ost)
    time.sleep(0.2)                               # (a stale download job would have fired by now)
    assert (victim == sentinel).all(), "late write into the buffer of the failed call"
    assert m.healthy()
    m.free(); om.free()
