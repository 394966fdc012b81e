"""Which persistent decode kernel a context got and why (rwkv_mi_persist_info, PERSISTENT_DECODE in rwkv_get_system_info_string), and the
fall-back when its polls time out: the abort word is set from the host (what a timed-out poll leaves behind -- a second process on the GPU,
a partitioned device), the next steps must still equal the oracle bit for bit, on the per-layer launches, and the context must say so."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import hooks_library, library, model, synth

pytestmark = pytest.mark.gpu

TOKENS = [5, 77, 130, 9, 201, 44, 3, 250]


def test_system_info_names_the_device_and_the_fast_path():
    info = library().rwkv_get_system_info_string()
    assert "HIP=1" in info and "CU=" in info
    assert ("PERSISTENT_DECODE=available" in info) == ("CU=256" in info)
    assert "PERSISTENT_DECODE=" in info


@pytest.mark.parametrize("name,fmt,kinds", [("mega-v6-2048", "Q4_0", ("ring", "regs")), ("test-v7", "Q5_1", ("k47",))])
def test_fallback_after_a_forced_abort_keeps_parity_and_says_why(tmp_path, name, fmt, kinds):
    hooks_library()
    p = str(tmp_path / "m.bin")
    synth.write_model(p, synth.CONFIGS[name], fmt, seed=29)
    om = O.OracleModel(p)
    m = model(p, hooks=True)
    info = m.persist_info()
    if m.decode_path() != 2:
        pytest.skip(f"the persistent path is not active on this box: {info}")
    assert info.split(";")[0] in tuple("persist: " + k for k in kinds), info
    if os.environ.get("RWKV_MI_NO_AUTOTUNE") != "1":     # (an earlier test of the session may have switched the timing at context creation off)
        assert "calibration" in info and "ms / token" in info, info
    ost, st = om.init_state(), None
    for t in TOKENS[:3]:
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)
        assert np.array_equal(lg, ol) and np.array_equal(st, ost)
    L = hooks_library().library
    L.rwkv_mi_test_force_abort.argtypes = [ctypes.c_void_p]
    L.rwkv_mi_test_force_abort.restype = ctypes.c_bool
    assert L.rwkv_mi_test_force_abort(m._ctx.ptr)
    for i, t in enumerate(TOKENS[3:]):
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)      # the first of these finds the abort word, drops the persistent kernel and repeats the step
        assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, i, float(np.abs(lg - ol).max()))
    assert m.decode_path() != 2 and m.persist_kind() == 0
    info = m.persist_info()
    assert info.startswith("persist: none") and "timed out" in info and "fell back" in info, info
    # the greedy loop continues on the per-layer launches too
    m.state_load(None)
    toks, _ = m.decode_greedy(5, 6)
    os2, tok, ref = om.init_state(), 5, []
    for _ in range(6):
        ol, os2 = om.eval(tok, os2)
        tok = int(np.argmax(ol))
        ref.append(tok)
    assert list(toks) == ref
    m.free()
    om.free()


def test_models_without_a_persistent_kernel_say_why(tmp_path):
    library()
    p = str(tmp_path / "m.bin")
    synth.write_model(p, synth.CONFIGS["test-v6"], "FP16", seed=5)
    m = model(p)
    info = m.persist_info()
    assert info.startswith("persist: none;") and len(info) > len("persist: none; "), info
    m.free()
