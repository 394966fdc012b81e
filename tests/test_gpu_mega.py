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


@pytest.fixture(params=["ring", "regs"], autouse=True)
def persist(request):
    """Every test runs on both persistent kernels: the LDS-DMA weight ring (ring_v6.hip, the default) and the register prefetch (mega_v6.hip)."""
    os.environ["RWKV_MI_PERSIST"] = request.param
    yield {"ring": 2, "regs": 1}[request.param]
    del os.environ["RWKV_MI_PERSIST"]

TOKENS = [1, 2, 3, 400, 5, 77, 300, 9, 11, 12]   # (every test vocabulary has at least 512 entries)


@pytest.mark.parametrize("name,fmt", [("mega-v6-2048", "Q4_0"), ("mega-v6-4096", "Q4_0"), ("mega-v6-2048", "Q4_1"), ("mega-v6-2048", "Q5_0"),
                                      ("mega-v6-4096", "Q5_1"), ("mega-v6-4096", "Q8_0"),
                                      # vocabularies of 1, 2 and 8 sixteen-row groups per workgroup: the ring kernel runs ln_out + head inside the launch
                                      ("mega-v6-4096-v4k", "Q4_0"), ("mega-v6-2048-v8k", "Q5_1"), ("mega-v6-2048-v32k", "Q4_0"), ("mega-v6-2048-v8k", "Q8_0")])
def test_mega_matches_oracle(tmp_path, name, fmt, persist):
    library()
    p = str(tmp_path / "m.bin")
    synth.write_model(p, synth.CONFIGS[name], fmt, seed=11)
    om = O.OracleModel(p)
    m = model(p)
    assert m.decode_path() == 2, "persistent kernel not selected for a geometry it is built for"
    assert m.persist_kind() == persist
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


def test_mega_health_and_trace(tmp_path, persist):
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
    if persist == 1:
        assert (np.diff(t[:, 1:, :17], axis=2) >= 0).all() and (t[:, 1:, 16] > t[:, 1:, 0]).all()   # worker shader-clock stamps
        assert (np.diff(t[:, 0, :13], axis=1) >= 0).all()                                              # comm wave
    else:
        assert (np.diff(t[:, 2:, :14], axis=2) >= 0).all() and (t[:, 2:, 13] > t[:, 2:, 0]).all()   # consumer waves
        assert (np.diff(t[:, 1, :12], axis=1) >= 0).all()                                              # comm wave
        assert (t[:, 0, 1] > t[:, 0, 0]).all()                                                         # loader: start / end
    assert L.rwkv_mi_decode_healthy(m._ctx.ptr)
    m.free()


@pytest.mark.parametrize("name,fmt", [("mega-v6-2048", "Q4_0"), ("mega-v6-4096", "Q5_1")])
def test_mega_across_the_16_bit_tag_wrap(tmp_path, name, fmt):
    """The hand-over tag is a rolling 16-bit generation advancing 8 per layer (wraps every 256 tokens at 32 layers). Preset a few
    layers below the wrap, then logits and state must stay bit-identical to the oracle while the tag rolls over -- serially
    (rwkv_eval) and through the graph-replayed greedy loop."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=17)
    om = O.OracleModel(p)
    m = model(p, hooks=True)      # (the tag preset is a test entry point: librwkv_testhooks.so, same objects as the product library)
    assert m.decode_path() == 2
    per_token = 8 * spec.n_layer
    for base in (0x10000 - 2 * per_token - 8, 0xFFFFFFF8 - 3 * per_token):      # 16-bit wrap inside token 2 / 32-bit wrap of the counter itself
        assert m.test_set_tag(base)
        ost, st = om.init_state(), None
        for i, t in enumerate(TOKENS[:6]):
            ol, ost = om.eval(t, ost)
            lg, st = m.eval(t, st)
            assert np.array_equal(lg, ol), (name, hex(base), i)
            assert np.array_equal(st, ost), (name, hex(base), i)
    assert m.test_set_tag(0x10000 - 3 * per_token)
    m.state_load(None)
    toks, _ = m.decode_greedy(5, 8)
    ost, tok, ref = om.init_state(), 5, []
    for _ in range(8):
        ol, ost = om.eval(tok, ost)
        tok = int(np.argmax(ol))
        ref.append(tok)
    assert list(toks) == ref
    assert np.array_equal(m.state_store(), ost)
    assert m.healthy()
    m.free()
    om.free()


def test_concurrent_clones_on_the_persistent_path(tmp_path):
    """rwkv.h:64-68: parallel inference = one clone per thread. Every clone owns a persistent kernel on its own stream; their
    launches are chained per device (engine.hip), so two threads decoding at once neither dead-lock nor time out."""
    import threading
    library()
    p = str(tmp_path / "m.bin")
    synth.write_model(p, synth.CONFIGS["mega-v6-2048"], "Q4_0", seed=23)
    om = O.OracleModel(p)
    ost, tok, ref = om.init_state(), 7, []
    for _ in range(24):
        ol, ost = om.eval(tok, ost)
        tok = int(np.argmax(ol))
        ref.append(tok)
    a = model(p)
    b = a.clone()
    assert a.decode_path() == 2 and b.decode_path() == 2
    out, errs = {}, []

    def greedy(name, m):
        try:
            m.state_load(None)
            toks, _ = m.decode_greedy(7, 24)
            out[name] = list(toks)
        except Exception as e:   # noqa: BLE001
            errs.append((name, repr(e)))

    def serial(name, m):
        try:
            st, tok, got = None, 7, []
            for _ in range(24):
                lg, st = m.eval(tok, st)
                tok = int(np.argmax(lg))
                got.append(tok)
            out[name] = got
        except Exception as e:   # noqa: BLE001
            errs.append((name, repr(e)))

    for rnd in range(2):
        th = [threading.Thread(target=greedy if rnd == 0 else serial, args=("a", a)), threading.Thread(target=serial, args=("b", b))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        assert out["a"] == ref and out["b"] == ref
    assert a.healthy() and b.healthy()
    b.free()
    a.free()
    om.free()


def test_new_context_starts_from_the_fresh_state(tmp_path):
    """A context (or clone) used through the resident-state extensions without an explicit load starts from the reference's fresh
    state, not from what the creation-time calibration left in the buffers."""
    library()
    p = str(tmp_path / "m.bin")
    synth.write_model(p, synth.CONFIGS["mega-v6-2048"], "Q4_0", seed=29)
    os.environ.pop("RWKV_MI_NO_AUTOTUNE", None)
    try:
        m = model(p)
        c = m.clone()
    finally:
        os.environ["RWKV_MI_NO_AUTOTUNE"] = "1"
    toks_c, _ = c.decode_greedy(3, 6)
    toks_m, _ = m.decode_greedy(3, 6)
    m.state_load(None)
    toks_r, _ = m.decode_greedy(3, 6)
    assert list(toks_c) == list(toks_r) == list(toks_m)
    c.free()
    m.free()


@pytest.mark.parametrize("name", ["mega-v6-2048-v8k", "test-v6"])
def test_device_side_tokens_stay_inside_the_vocabulary(tmp_path, name, persist):
    """The greedy loop feeds the device's own argmax into the next embedding lookup. With a state of NaNs every logit is NaN and no
    element compares greater than anything: the chosen token must still be a row of the embedding table (0, like numpy's argmax of
    all-NaN), not the reduction's initial index -- that read far outside the table and ended the process with a GPU memory fault
    (seen when two processes shared the GPU and a persistent kernel timed out). The context stays usable afterwards."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, "Q4_0", seed=47)
    m = model(p)
    m.state_load(np.full(m.state_len, np.nan, dtype=np.float32))
    toks, _ = m.decode_greedy(3, 12)
    assert all(int(t) < spec.n_vocab for t in toks)
    assert int(m.sample(temperature=1.0, top_p=0.5, u=0.3)) < spec.n_vocab
    m.state_load(None)
    om = O.OracleModel(p)
    ol, _ = om.eval(5, om.init_state())
    lg, _ = m.eval(5, None)
    assert np.array_equal(lg, ol)
    m.free()
    om.free()
