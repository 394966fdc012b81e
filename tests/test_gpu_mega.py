"""The persistent whole-stage RWKV-6 decode kernel (mega_v6.hip) against the CPU oracle: bit-exact logits and state over
several tokens, on the two geometries it is instantiated for, and identical to the seven-launch fused path."""
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import library, model, synth

pytestmark = pytest.mark.gpu

# contexts normally time both single-token paths at creation and keep the faster one; these tests are about the persistent kernel
os.environ["RWKV_MI_NO_AUTOTUNE"] = "1"

TOKENS = [1, 2, 3, 400, 5, 77, 300, 9, 11, 12]


@pytest.mark.parametrize("name,fmt", [("mega-v6-2048", "Q4_0"), ("mega-v6-4096", "Q4_0"), ("mega-v6-2048", "Q4_1"), ("mega-v6-2048", "Q5_0"),
                                      ("mega-v6-4096", "Q5_1"), ("mega-v6-4096", "Q8_0")])
def test_mega_matches_oracle(tmp_path, name, fmt):
    library()
    p = str(tmp_path / "m.bin")
    synth.write_model(p, synth.CONFIGS[name], fmt, seed=11)
    om = O.OracleModel(p)
    m = model(p)
    assert m.decode_path() == 2, "persistent kernel not selected for a geometry it is built for"
    ost, st = om.init_state(), None
    for i, t in enumerate(TOKENS):
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)
        assert np.array_equal(lg, ol), (name, i, float(np.abs(lg - ol).max()))
        assert np.array_equal(st, ost), (name, i, float(np.abs(st - ost).max()))
    # device-resident greedy decode (graph replay of the persistent kernel) == serial evaluation
    m.state_load(None)
    toks, _ = m.decode_greedy(5, 6)
    st2, tok, ref = None, 5, []
    for _ in range(6):
        lg, st2 = m.eval(tok, st2)
        tok = int(np.argmax(lg))
        ref.append(tok)
    assert list(toks) == ref
    m.free()
    om.free()


def test_mega_equals_fused_path(tmp_path):
    library()
    p = str(tmp_path / "m.bin")
    synth.write_model(p, synth.CONFIGS["mega-v6-2048"], "Q4_0", seed=5)
    m = model(p)
    assert m.decode_path() == 2
    os.environ["RWKV_MI_NO_MEGA"] = "1"
    try:
        f = model(p)
    finally:
        del os.environ["RWKV_MI_NO_MEGA"]
    assert f.decode_path() == 1
    sa = sb = None
    for t in TOKENS:
        la, sa = m.eval(t, sa)
        lb, sb = f.eval(t, sb)
    assert np.array_equal(la, lb) and np.array_equal(sa, sb)
    m.free()
    f.free()


def test_mega_stages_reproduce_full_model(tmp_path):
    """Layer ranges (pipeline stages) through the persistent kernel: residual stream handed over in plain device memory."""
    import ctypes
    import torch
    from rwkv_cpp_amd import pipeline
    lib = library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS["mega-v6-2048"]
    synth.write_model(p, spec, "Q4_0", seed=21)
    full = model(p)
    full.state_load(None)
    exp, _ = full.decode_greedy(9, 5)
    stages = [pipeline.LibStageExecutor(lib, p, b, e, spec.n_layer) for b, e in [(0, 1), (1, 3)]]
    handles = [s.new_stream() for s in stages]
    assert all(lib.library.rwkv_mi_decode_path(h) == 2 for h in handles)
    tok = torch.tensor([9], dtype=torch.int32, device="cuda")
    nxt = torch.zeros(1, dtype=torch.int32, device="cuda")
    xs = [torch.zeros(stages[0].handoff_len, dtype=torch.float32, device="cuda") for _ in range(3)]
    got = []
    for _ in range(5):
        for i, s in enumerate(stages):
            s.step(handles[i], tok, xs[i], xs[i + 1], nxt)
            torch.cuda.synchronize()
        got.append(int(nxt.item()))
        tok.copy_(nxt)
        torch.cuda.synchronize()
    assert got == list(exp)
    for s in stages:
        s.close()
    full.free()


def test_mega_health_and_trace(tmp_path):
    """rwkv_mi_decode_healthy stays true over ordinary use; rwkv_mi_trace_phases returns monotonic stamps per wave."""
    import ctypes
    lib = library()
    p = str(tmp_path / "m.bin")
    synth.write_model(p, synth.CONFIGS["mega-v6-2048"], "Q4_0", seed=3)
    m = model(p)
    assert m.decode_path() == 2
    m.state_load(None)
    m.decode_greedy(3, 8)
    L = lib.library
    assert L.rwkv_mi_decode_healthy(m._ctx.ptr)
    L.rwkv_mi_trace_phases.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.rwkv_mi_trace_phases.restype = ctypes.c_bool
    out = np.zeros(256 * 8 * 32, dtype=np.int64)
    assert L.rwkv_mi_trace_phases(m._ctx.ptr, 5, 1, 2, out.ctypes.data)
    t = out.reshape(256, 8, 32)
    assert (np.diff(t[:, 1:, :17], axis=2) >= 0).all() and (t[:, 1:, 16] > t[:, 1:, 0]).all()   # worker shader-clock stamps
    assert (np.diff(t[:, 0, :13], axis=1) >= 0).all()                                              # comm wave
    assert L.rwkv_mi_decode_healthy(m._ctx.ptr)
    m.free()
