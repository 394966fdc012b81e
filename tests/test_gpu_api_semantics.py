"""The reference's self-consistency tests (memcmp-level) and ABI semantics, on the GPU through the C ABI:
tests/test_eval_sequence_in_chunks.c, tests/test_logit_calculation_skipping.c, tests/test_context_cloning.c."""
import ctypes

import numpy as np
import pytest

import reference_constants as R
from gpu_lib import library, model

pytestmark = pytest.mark.gpu

PROMPT_70 = "This is a port of [BlinkDL/RWKV-LM](https://github.com/BlinkDL/RWKV-LM"
CASES = [("5v2-730K", "FP32"), ("4v0-660K", "FP32"), ("5v1-730K", "FP16"), ("6v0-3m", "Q5_0"), ("7v0-834K", "FP32"), ("7v0-834K", "Q5_1"), ("4v0-660K", "Q5_1")]


@pytest.mark.parametrize("version,fmt", CASES)
def test_eval_sequence_in_chunks(golden_dir, version, fmt):
    # test_eval_sequence_in_chunks.c:10-76 (reference runs it on 5v2 FP32 only)
    m = model(R.fixture_path(golden_dir, version, fmt))
    for prompt in (PROMPT_70, "T"):
        toks = [ord(c) for c in prompt]
        exp_logits, exp_state = m.eval(toks[0], None)
        for t in toks[1:]:
            exp_logits, exp_state = m.eval(t, exp_state, exp_state, exp_logits)
        for chunk in (1, 2, 8, 10, 64, 128):
            logits, state = m.eval_sequence_in_chunks(toks, None, chunk_size=chunk)
            assert np.array_equal(state, exp_state), (version, fmt, chunk)
            assert np.array_equal(logits, exp_logits), (version, fmt, chunk)
    m.free()


@pytest.mark.parametrize("version,fmt", CASES[:4])
def test_logit_calculation_skipping(golden_dir, version, fmt):
    # test_logit_calculation_skipping.c:14-90
    lib = library()
    m = model(R.fixture_path(golden_dir, version, fmt))
    toks = [ord(c) for c in "hello world"]
    _, exp_state = m.eval(toks[0], None)
    for t in toks[1:]:
        _, exp_state = m.eval(t, exp_state)
    state = np.zeros(m.state_len, dtype=np.float32)
    lib.rwkv_eval(m._ctx, toks[0], 0, state.ctypes.data, 0)
    for t in toks[1:]:
        lib.rwkv_eval(m._ctx, t, state.ctypes.data, state.ctypes.data, 0)
    assert np.array_equal(state, exp_state)
    _, seq_state = m.eval_sequence(toks, None)
    state2 = np.zeros(m.state_len, dtype=np.float32)
    lib.rwkv_eval_sequence(m._ctx, toks, 0, state2.ctypes.data, 0)
    assert np.array_equal(state2, seq_state) and np.array_equal(seq_state, exp_state)
    m.free()


def test_context_cloning(golden_dir):
    # test_context_cloning.c:10-57: the clone keeps working after the original is freed; logits identical
    m = model(R.fixture_path(golden_dir, "5v2-730K", "FP32"))
    toks = [ord(c) for c in "hello world"]
    exp_logits, st = m.eval(toks[0], None)
    for t in toks[1:]:
        exp_logits, st = m.eval(t, st)
    m2 = m.clone(2)
    assert m2._ctx.ptr != m._ctx.ptr
    m.free()
    logits, st2 = m2.eval(toks[0], None)
    for t in toks[1:]:
        logits, st2 = m2.eval(t, st2)
    assert np.array_equal(logits, exp_logits)
    m2.free()


def test_init_state_equals_null_state(golden_dir):
    for version, fmt in (("4v0-660K", "FP32"), ("6v0-3m", "Q5_1")):
        m = model(R.fixture_path(golden_dir, version, fmt))
        s0 = m.init_state()
        if version.startswith("4"):
            s = s0.reshape(m.n_layer, 5, m.n_embed)
            assert np.all(s[:, :4] == 0) and np.all(s[:, 4] == np.float32(-1e30))
        else:
            assert np.all(s0 == 0)
        a = m.eval(65, None)
        b = m.eval(65, s0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        m.free()


def test_error_behaviour(golden_dir):
    lib = library()
    m = model(R.fixture_path(golden_dir, "5v2-730K", "FP32"))
    L = lib.library
    st = np.zeros(m.state_len, dtype=np.float32)
    lg = np.zeros(m.n_vocab, dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    null = ctypes.cast(0, fp)
    # invalid token -> false, RWKV_ERROR_ARGS on the context, cleared by the getter (rwkv_eval.inc:43, rwkv.cpp:229-234)
    assert not L.rwkv_eval(m._ctx.ptr, 256, null, ctypes.cast(st.ctypes.data, fp), ctypes.cast(lg.ctypes.data, fp))
    assert lib.rwkv_get_last_error(m._ctx) == 1 << 8
    assert lib.rwkv_get_last_error(m._ctx) == 0
    arr = (ctypes.c_uint32 * 3)(1, 2, 999)
    assert not L.rwkv_eval_sequence(m._ctx.ptr, arr, 3, null, null, null)
    assert lib.rwkv_get_last_error(m._ctx) == 1 << 8
    # length 0 / chunk 0 (rwkv_eval.inc:89,167-168)
    assert not L.rwkv_eval_sequence(m._ctx.ptr, arr, 0, null, null, null)
    assert lib.rwkv_get_last_error(m._ctx) == 1 << 8
    arr2 = (ctypes.c_uint32 * 3)(1, 2, 3)
    assert not L.rwkv_eval_sequence_in_chunks(m._ctx.ptr, arr2, 3, 0, null, null, null)
    assert lib.rwkv_get_last_error(m._ctx) == 1 << 8
    # tokens == NULL: prepare only, returns true, writes nothing (rwkv_eval.inc:122,152-154)
    lg[:] = 7.0
    assert L.rwkv_eval_sequence(m._ctx.ptr, ctypes.cast(0, ctypes.POINTER(ctypes.c_uint32)), 5, null, null, ctypes.cast(lg.ctypes.data, fp))
    assert np.all(lg == 7.0)
    # a context starts silent, the global default is printing (rwkv.cpp:74, rwkv_error_handling.inc:1-2)
    assert L.rwkv_get_print_errors(m._ctx.ptr) is False
    assert L.rwkv_get_print_errors(None) is True
    # geometry getters incl. the legacy pair (rwkv.cpp:145-179)
    assert (L.rwkv_get_n_vocab(m._ctx.ptr), L.rwkv_get_n_embed(m._ctx.ptr), L.rwkv_get_n_layer(m._ctx.ptr)) == (256, 64, 12)
    assert L.rwkv_get_state_len(m._ctx.ptr) == 64 * (2 + 8) * 12 == L.rwkv_get_state_buffer_element_count(m._ctx.ptr)
    assert L.rwkv_get_logits_len(m._ctx.ptr) == 256 == L.rwkv_get_logits_buffer_element_count(m._ctx.ptr)
    m.free()


def test_resident_state_and_greedy_match_the_abi_path(golden_dir):
    m = model(R.fixture_path(golden_dir, "6v0-3m", "Q5_0"))
    toks = [ord(c) for c in "hello world, hello"]
    exp_logits, exp_state = m.eval_sequence(toks, None)
    m.state_load(None)
    logits = m.eval_resident(toks[:5])
    logits = m.eval_resident(toks[5:])
    assert np.array_equal(logits, exp_logits)
    assert np.array_equal(m.state_store(), exp_state)
    # greedy decode on the device == greedy decode driven from the host through rwkv_eval
    first = int(np.argmax(exp_logits))
    st, tok, host = exp_state, first, []
    for _ in range(12):
        lg, st = m.eval(tok, st)
        tok = int(np.argmax(lg))
        host.append(tok)
    m.state_load(exp_state)
    dev, ms = m.decode_greedy(first, 12)
    assert list(dev) == host and ms > 0
    m.free()
