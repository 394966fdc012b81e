"""The persistent decode launch of RWKV-4 and RWKV-7 (csrc/persist_v47.hip: one launch per token, four / five tagged hand-overs per
layer) against the CPU oracle -- logits AND state with np.array_equal -- and against the two paths it supersedes (the fused per-layer
launches of fused_v7.hip, one kernel per graph op), on the stand-in geometry (D 256: GPB 1, 32 row workgroups, two first-stage jobs per
polling wave) and on the two BASELINE geometries it is built for: RWKV-4-Pile-169M (D 768, 96 row workgroups, V 50277) and
RWKV-7-World-2.9B (D 2560: GPB 2, 160 row workgroups + 40 head workgroups, ranks 96 / 96 / 64 / 320).
Reference: rwkv_graph.inc:84-197 (rwkv_att_v4), :387-482 (rwkv_att_v7), :484-543 (ffn), rwkv_operators_wkv_v7.inc:37-107;
tests/test_tiny_rwkv.c:136-173 runs every architecture x format."""
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import library, model, synth

pytestmark = pytest.mark.gpu

TOKENS = [1, 2, 3, 400, 5, 77, 300, 9, 11, 12]
CASES = [("test-v4", f) for f in ("Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0")] + [("test-v7", f) for f in ("Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0")] + \
        [("slice-v4-768", "Q5_1"), ("slice-v4-768", "Q4_0"), ("slice-v4-768", "Q8_0"), ("slice-v7-2560", "Q5_1"), ("slice-v7-2560", "Q4_0"), ("slice-v7-2560", "Q4_1")]


def _with_env(var, fn):
    os.environ[var] = "1"
    try:
        return fn()
    finally:
        del os.environ[var]


@pytest.fixture(autouse=True)
def _no_autotune():
    # contexts time the persistent launch against the fused launches at creation and keep the faster; these tests are about the persistent one
    os.environ["RWKV_MI_NO_AUTOTUNE"] = "1"
    yield
    del os.environ["RWKV_MI_NO_AUTOTUNE"]


@pytest.mark.parametrize("name,fmt", CASES)
def test_persistent_v4_v7_matches_oracle_and_the_other_paths(tmp_path, name, fmt):
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=29)
    om = O.OracleModel(p)
    m = model(p)
    assert m.decode_path() == 2 and m.persist_kind() == 3, (name, fmt, m.decode_path(), m.persist_kind())
    f = _with_env("RWKV_MI_NO_MEGA", lambda: model(p))
    assert f.decode_path() == 1
    ost, st, fst = om.init_state(), None, None
    for i, t in enumerate(TOKENS):
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)
        fl, fst = f.eval(t, fst)
        assert np.array_equal(st, ost), (name, fmt, i, "state", float(np.abs(st - ost).max()), int((st != ost).argmax()))
        assert np.array_equal(lg, ol), (name, fmt, i, float(np.abs(lg - ol).max()))
        assert np.array_equal(fl, ol) and np.array_equal(fst, ost), (name, fmt, i, "fused")
    # device-resident greedy loop (graph replay of the launch + argmax + embedding) == the oracle's greedy continuation
    m.state_load(None)
    toks, _ = m.decode_greedy(5, 12)
    os2, tok, ref = om.init_state(), 5, []
    for _ in range(12):
        ol, os2 = om.eval(tok, os2)
        tok = int(np.argmax(ol))
        ref.append(tok)
    assert list(toks) == ref
    assert np.array_equal(m.state_store(), os2)
    assert m.healthy()
    # sequence mode is not this kernel's, but it must leave a context able to continue on it: sequence -> single tokens from that state
    seq = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(40)]
    ol, ost = om.eval_sequence(seq, om.init_state())
    gl, gst = m.eval_sequence(seq, None)
    assert np.array_equal(gl, ol) and np.array_equal(gst, ost)
    ol, ost = om.eval(7, ost)
    gl, gst = m.eval(7, gst)
    assert np.array_equal(gl, ol) and np.array_equal(gst, ost)
    m.free(); f.free(); om.free()


@pytest.mark.parametrize("name,fmt", [("test-v4", "Q5_1"), ("test-v7", "Q5_1"), ("slice-v7-2560", "Q5_1")])
def test_persistent_v4_v7_across_the_16_bit_tag_wrap(tmp_path, name, fmt):
    """The hand-over tag is a rolling 16-bit generation advancing 8 per layer; a launch also publishes its input under the tag before its
    first. Preset just below the 16-bit wrap and below the 32-bit wrap of the counter: logits and state stay the oracle's."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=31)
    om = O.OracleModel(p)
    m = model(p, hooks=True)
    assert m.decode_path() == 2 and m.persist_kind() == 3
    per_token = 8 * spec.n_layer
    for base in (0x10000 - 2 * per_token - 8, 0x10000 - per_token, 0xFFFFFFF8 - 3 * per_token):
        assert m.test_set_tag(base)
        ost, st = om.init_state(), None
        for i, t in enumerate(TOKENS[:6]):
            ol, ost = om.eval(t, ost)
            lg, st = m.eval(t, st)
            assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, hex(base), i)
    assert m.test_set_tag(0x10000 - 3 * per_token)
    m.state_load(None)
    toks, _ = m.decode_greedy(5, 8)
    ost, tok, ref = om.init_state(), 5, []
    for _ in range(8):
        ol, ost = om.eval(tok, ost)
        tok = int(np.argmax(ol))
        ref.append(tok)
    assert list(toks) == ref and np.array_equal(m.state_store(), ost)
    assert m.healthy()
    m.free(); om.free()


@pytest.mark.parametrize("name,fmt,cuts", [("test-v7", "Q5_1", [(0, 1), (1, 3)]), ("test-v7", "Q4_0", [(0, 2), (2, 3)]), ("test-v4", "Q5_1", [(0, 1), (1, 2)])])
def test_persistent_v4_v7_stages_reproduce_the_full_model(tmp_path, name, fmt, cuts):
    """Layer ranges (pipeline stages): the residual stream -- and RWKV-7's v_first, which a later stage's head workgroups read from plain
    memory instead of keeping it from layer 0 -- handed over in plain device memory."""
    import torch
    from rwkv_cpp_amd import pipeline
    lib = library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=37)
    om = O.OracleModel(p)
    ost, tok, exp = om.init_state(), 9, []
    for _ in range(6):
        ol, ost = om.eval(tok, ost)
        tok = int(np.argmax(ol))
        exp.append(tok)
    stages = [pipeline.LibStageExecutor(lib, p, b, e, spec.n_layer) for b, e in cuts]
    handles = [s.new_stream() for s in stages]
    assert all(lib.library.rwkv_mi_decode_path(h) == 2 and lib.library.rwkv_mi_persist_kind(h) == 3 for h in handles)
    tok = torch.tensor([9], dtype=torch.int32, device="cuda")
    nxt = torch.zeros(1, dtype=torch.int32, device="cuda")
    xs = [torch.zeros(stages[0].handoff_len, dtype=torch.float32, device="cuda") for _ in range(len(stages) + 1)]
    got = []
    for _ in range(6):
        for i, s in enumerate(stages):
            s.step(handles[i], tok, xs[i], xs[i + 1], nxt)
            torch.cuda.synchronize()
        got.append(int(nxt.item()))
        tok.copy_(nxt)
        torch.cuda.synchronize()
    assert got == exp
    for s in stages:
        s.close()
    om.free()


@pytest.mark.parametrize("name", ["test-v4", "test-v7"])
def test_persistent_v4_v7_through_the_streamed_abi(tmp_path, name):
    """rwkv_eval with the caller's state streamed in layer groups: one launch per group over a layer RANGE of the persistent kernel."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, "Q5_1", seed=41)
    om = O.OracleModel(p)
    m = _with_env("RWKV_MI_ABI_STREAM", lambda: model(p))
    os.environ["RWKV_MI_ABI_STREAM"] = "1"
    try:
        assert m.decode_path() == 2 and m.persist_kind() == 3
        ost, st = om.init_state(), None
        for i, t in enumerate(TOKENS):
            ol, ost = om.eval(t, ost)
            lg, st = m.eval(t, st)
            assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, i)
    finally:
        del os.environ["RWKV_MI_ABI_STREAM"]
    m.free(); om.free()


def test_persistent_v47_trace_and_clones(tmp_path):
    import ctypes
    import threading
    lib = library()
    p = str(tmp_path / "m.bin")
    synth.write_model(p, synth.CONFIGS["slice-v4-768"], "Q5_1", seed=43)
    om = O.OracleModel(p)
    ost, tok, ref = om.init_state(), 7, []
    for _ in range(16):
        ol, ost = om.eval(tok, ost)
        tok = int(np.argmax(ol))
        ref.append(tok)
    a = model(p)
    b = a.clone()
    assert a.persist_kind() == 3 and b.persist_kind() == 3
    out, errs = {}, []

    def greedy(nm, mm):
        try:
            mm.state_load(None)
            toks, _ = mm.decode_greedy(7, 16)
            out[nm] = list(toks)
        except Exception as e:   # noqa: BLE001
            errs.append((nm, repr(e)))

    th = [threading.Thread(target=greedy, args=(n, mm)) for n, mm in (("a", a), ("b", b))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs and out["a"] == ref and out["b"] == ref
    L = lib.library
    L.rwkv_mi_trace_phases.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.rwkv_mi_trace_phases.restype = ctypes.c_bool
    buf = np.zeros(256 * 8 * 32, dtype=np.int64)
    assert L.rwkv_mi_trace_phases(a._ctx.ptr, 5, 1, 2, buf.ctypes.data)
    t = buf[:256 * 9 * 16].reshape(256, 9, 16)
    assert (np.diff(t[:96, :8, :11], axis=2) >= 0).all() and (t[:96, :8, 10] > t[:96, :8, 0]).all()   # worker waves of the 96 row workgroups
    assert (np.diff(t[:96, 8, :9], axis=1) >= 0).all()                                                    # their polling waves
    assert a.healthy() and b.healthy()
    a.free(); b.free(); om.free()


@pytest.mark.parametrize("name,env", [("test-v4", "RWKV_MI_P47_NOFOLD"), ("test-v7", "RWKV_MI_P47_NOFOLD"), ("slice-v4-768", "RWKV_MI_P47_CALM0"), ("test-v7", "RWKV_MI_P47_CALM0")])
def test_persistent_v47_measurement_switches_keep_the_result(tmp_path, name, env):
    """RWKV_MI_P47_NOFOLD=1 (embedding + ln0 and ln_out + head + argmax as separate launches) and RWKV_MI_P47_CALM=0 (spare / head workgroups
    poll the last layer's x at full width) are A/B aids of DESIGN.md 6.3c: other schedules of the same arithmetic -- logits, state and the
    greedy continuation stay the oracle's."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, "Q5_1", seed=47)
    om = O.OracleModel(p)
    var, val = (env, "1") if env.endswith("NOFOLD") else ("RWKV_MI_P47_CALM", "0")
    os.environ[var] = val
    try:
        m = model(p)
    finally:
        del os.environ[var]
    assert m.decode_path() == 2 and m.persist_kind() == 3
    ost, st = om.init_state(), None
    for i, t in enumerate(TOKENS[:6]):
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)
        assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, env, i)
    m.state_load(None)
    toks, _ = m.decode_greedy(5, 8)
    os2, tok, ref = om.init_state(), 5, []
    for _ in range(8):
        ol, os2 = om.eval(tok, os2)
        tok = int(np.argmax(ol))
        ref.append(tok)
    assert list(toks) == ref and np.array_equal(m.state_store(), os2)
    assert m.healthy()
    m.free(); om.free()
