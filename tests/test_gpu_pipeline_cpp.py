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


def test_eight_stages_on_one_device(tmp_path):
    """An eight-GPU node's chain (RWKV_MI_DEVICES=0-7) with every stage on device 0: a 32-layer model in eight stages of four layers.
    rwkv.h entry points, the resident-state extensions and the C++ greedy loop (runner.cpp) must all equal the oracle / the one-device
    context bit for bit -- single stream and several clones interleaved."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS["chain-v6-32x256"]
    synth.write_model(p, spec, "Q4_0", seed=37)
    om = O.OracleModel(p)
    pm = _pipeline_model(p, "0,0,0,0,0,0,0,0")
    one = model(p)
    ost, st = om.init_state(), None
    for i, t in enumerate(TOKENS):
        ol, ost = om.eval(t, ost)
        lg, st = pm.eval(t, st)
        assert np.array_equal(lg, ol) and np.array_equal(st, ost), i
    # resident state on the chain: load, greedy loop in C++, store
    one.state_load(ost)
    pm.state_load(ost)
    ref, _ = one.decode_greedy(7, 40)
    got, ms = pm.decode_greedy(7, 40)
    assert list(got) == list(ref) and ms > 0
    assert np.array_equal(pm.state_store(), one.state_store())
    # three decode streams through the same chain, interleaved by the library
    clones = [pm, pm.clone(), pm.clone()]
    firsts = [7, 8, 300]
    for m in clones:
        m.state_load(None)
    toks, _ = type(pm).decode_greedy_streams(clones, firsts, 24)
    for j, f in enumerate(firsts):
        one.state_load(None)
        ref, _ = one.decode_greedy(f, 24)
        assert list(toks[j]) == list(ref), j
        assert np.array_equal(clones[j].state_store(), one.state_store()), j
    # the same entry point on plain one-device contexts (chains of one stage)
    pair = [one, one.clone()]
    for m in pair:
        m.state_load(None)
    toks2, _ = type(pm).decode_greedy_streams(pair, firsts[:2], 24)
    assert np.array_equal(toks2, toks[:2])
    pair[1].free()
    for m in clones[1:]:
        m.free()
    pm.free()
    one.free()
    om.free()


def test_native_stage_runner_on_one_rank(tmp_path):
    """rwkv_mi_stage_run (the loop one process per GPU runs, ncclSend / ncclRecv on the stage's stream) with a world of one: no hop,
    same iteration. And librccl.so can be bound at run time: an id, a communicator of one rank."""
    import ctypes
    lib = library()
    L = lib.library
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS["test-v6"]
    synth.write_model(p, spec, "Q5_1", seed=41)
    one = model(p)
    one.state_load(None)
    ref, _ = one.decode_greedy(9, 16)
    ctx = L.rwkv_mi_init_stage(p.encode(), 1, 0, spec.n_layer)
    assert ctx
    assert L.rwkv_mi_state_load(ctx, None)
    arr = (ctypes.c_void_p * 1)(ctx)
    first = (ctypes.c_uint32 * 1)(9)
    out = np.zeros((1, 16), dtype=np.uint32)
    ms = ctypes.c_float(0.0)
    assert L.rwkv_mi_stage_run(arr, 1, first, 16, 0, 1, None, None, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.byref(ms))
    assert list(out[0]) == list(ref)
    # a stage that is not the whole model refuses a world of one
    half = L.rwkv_mi_init_stage(p.encode(), 1, 0, 1)
    assert half and not L.rwkv_mi_stage_run((ctypes.c_void_p * 1)(half), 1, first, 4, 0, 1, None, None, None, None)
    L.rwkv_free(half)
    L.rwkv_free(ctx)
    assert L.rwkv_mi_comm_available()
    uid = np.zeros(128, dtype=np.uint8)
    assert L.rwkv_mi_comm_unique_id(ctypes.c_void_p(uid.ctypes.data), 128) and uid.any()
    comm = L.rwkv_mi_comm_init(ctypes.c_void_p(uid.ctypes.data), 0, 1)
    assert comm
    L.rwkv_mi_comm_free(ctypes.c_void_p(comm))
    one.free()


@pytest.mark.parametrize("devices", ["0,0,0", "0,0"])
def test_rwkv7_greedy_loop_through_a_chain(tmp_path, devices):
    """An RWKV-7 stage hands over TWO vectors per iteration, x and v_first. Round 3 sent both through one mailbox of the hop: the second
    send overwrote the first before the receiver had run, and every later stage computed with x == v_first -- silently wrong tokens
    from rwkv_mi_decode_greedy / rwkv_mi_decode_greedy_streams on a RWKV_MI_DEVICES chain. Every message has its own mailbox now
    (runner.cpp LocalHop): single stream and two interleaved streams against the one-device context and the oracle."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS["test-v7"]
    synth.write_model(p, spec, "Q5_1", seed=53)
    om = O.OracleModel(p)
    ost, tok, ref = om.init_state(), 7, []
    for _ in range(20):
        ol, ost = om.eval(tok, ost)
        tok = int(np.argmax(ol))
        ref.append(tok)
    pm = _pipeline_model(p, devices)
    pm.state_load(None)
    got, _ = pm.decode_greedy(7, 20)
    assert list(got) == ref
    assert np.array_equal(pm.state_store(), ost)
    one = model(p)
    clones = [pm, pm.clone()]
    firsts = [7, 300]
    for m in clones:
        m.state_load(None)
    toks, _ = type(pm).decode_greedy_streams(clones, firsts, 20)
    for j, f in enumerate(firsts):
        one.state_load(None)
        r, _ = one.decode_greedy(f, 20)
        assert list(toks[j]) == list(r), j
        assert np.array_equal(clones[j].state_store(), one.state_store()), j
    assert list(toks[0]) == ref
    # the same context twice is an argument error, not two streams interleaved on one state
    with pytest.raises(ValueError):
        type(pm).decode_greedy_streams([pm, pm], firsts, 4)
    clones[1].free()
    one.free()
    pm.free()
    om.free()


@pytest.mark.parametrize("name,fmt,devices,kind", [("mega-v6-2048-v8k", "Q4_0", "0,0,0", 2), ("slice-v7-2560", "Q5_1", "0,0", 3), ("chain-v6-32x256", "Q4_0", "0,0,0,0", 0)])
def test_hop_arms_of_the_greedy_loop(tmp_path, name, fmt, devices, kind):
    """The forms of a hop of the C++ greedy loop (runner.cpp LocalHop) give the same tokens and state as the CPU oracle: the stage's last
    layer storing the residual stream in the next stage's buffer and the last stage's launch the token in the first stage's token word, one
    event record and one wait per hop (default where the stage is one persistent launch: kinds 2 / 3), one peer copy into that buffer
    (RWKV_MI_HOP=copy; also what stages on the per-layer launches get), round 5's mailbox with a copy on each side (RWKV_MI_HOP=mailbox), and
    the default with its `taken` / own events back ("events") -- single stream and two interleaved streams, tokens appended to the history
    inside the launch where the last stage folds its argmax."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=61)
    om = O.OracleModel(p)
    n = 16
    refs = []
    for first in (7, 300):
        ost, tok, ref = om.init_state(), first, []
        for _ in range(n):
            ol, ost = om.eval(tok, ost)
            tok = int(np.argmax(ol))
            ref.append(tok)
        refs.append((ref, ost))
    os.environ["RWKV_MI_NO_AUTOTUNE"] = "1"
    try:
        pm = _pipeline_model(p, devices)
    finally:
        del os.environ["RWKV_MI_NO_AUTOTUNE"]
    if kind:
        assert pm.persist_kind() == kind
    clone = pm.clone()
    try:
        for arm in (None, "copy", "mailbox", "events"):
            if arm == "events":       # the default hop with its `taken` events and its own "there" event back (RWKV_MI_HOP_TAKEN / _OWN_EVENT)
                os.environ["RWKV_MI_HOP_TAKEN"] = "1"
                os.environ["RWKV_MI_HOP_OWN_EVENT"] = "1"
            elif arm:
                os.environ["RWKV_MI_HOP"] = arm
            try:
                for _ in range(2):      # (twice: the second call finds the buffers of the first released)
                    pm.state_load(None)
                    got, _ = pm.decode_greedy(7, n)
                    assert list(got) == refs[0][0], arm
                    assert np.array_equal(pm.state_store(), refs[0][1]), arm
                for m in (pm, clone):
                    m.state_load(None)
                toks, _ = type(pm).decode_greedy_streams([pm, clone], [7, 300], n)
                for j, m in enumerate((pm, clone)):
                    assert list(toks[j]) == refs[j][0], (arm, j)
                    assert np.array_equal(m.state_store(), refs[j][1]), (arm, j)
            finally:
                for k in ("RWKV_MI_HOP", "RWKV_MI_HOP_TAKEN", "RWKV_MI_HOP_OWN_EVENT"):
                    os.environ.pop(k, None)
        # the plain ABI on the same chain afterwards: the stages' own residual buffers are in use again (x_out was reset)
        ost, st = om.init_state(), None
        for t in TOKENS:
            ol, ost = om.eval(t % spec.n_vocab, ost)
            lg, st = pm.eval(t % spec.n_vocab, st)
            assert np.array_equal(lg, ol) and np.array_equal(st, ost)
    finally:
        clone.free()
        pm.free()
        om.free()
