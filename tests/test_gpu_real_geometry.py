"""BASELINE configurations C4 (RWKV-7-World-2.9B) and C2 (RWKV-4-Pile-169M) at their REAL row lengths, low-rank widths and vocabulary,
as two-layer slices, and the World vocabulary (65536 rows: three head passes per workgroup) on the ring kernel's folded head.

The small stand-ins of the other GPU tests (D = 256) do not reach what the decode kernels are tuned to: 2560-long rows (80 blocks: a
partial second 64-block step), low-rank widths 96 / 96 / 64 / 320 (k7_att_in keeps 48 of a row's 80 steps in flight), a vocabulary of
50277 rows (not a multiple of any tile) over D = 768. Every case: ten single tokens, the device-resident greedy loop and a 97-token
sequence pass against the CPU oracle -- logits AND state with np.array_equal -- on the fused path and on the one-kernel-per-op path
(reference: tests/test_tiny_rwkv.c:136-173 runs every architecture x format; rwkv_graph.inc:416-447 are the low-rank stages)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import FAST_ARM_REL_TOL, SEQ_ARM_ENV
from gpu_lib import library, model, synth

pytestmark = pytest.mark.gpu

TOKENS = [1, 2, 3, 400, 5, 77, 300, 9, 11, 12]


def _check(tmp_path, name, fmt, seed, want_path, env_off, env_on=None):
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=seed)
    om = O.OracleModel(p)
    if env_on:
        os.environ[env_on] = "1"
    try:
        m = model(p)
    finally:
        if env_on:
            del os.environ[env_on]
    assert m.decode_path() == want_path, (name, m.decode_path())
    os.environ[env_off] = "1"
    try:
        g = model(p)
    finally:
        del os.environ[env_off]
    assert g.decode_path() != want_path
    ost, st, gst = om.init_state(), None, None
    for i, t in enumerate(TOKENS):
        ol, ost = om.eval(t, ost)
        lg, st = m.eval(t, st)
        gl, gst = g.eval(t, gst)
        assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, fmt, i, float(np.abs(lg - ol).max()))
        assert np.array_equal(gl, ol) and np.array_equal(gst, ost), (name, fmt, i, float(np.abs(gl - ol).max()))
    # device-resident greedy loop (graph replay + on-device argmax feeding the next embedding row) == the oracle's greedy continuation
    m.state_load(None)
    toks, _ = m.decode_greedy(5, 8)
    os2, tok, ref = om.init_state(), 5, []
    for _ in range(8):
        ol, os2 = om.eval(tok, os2)
        tok = int(np.argmax(ol))
        ref.append(tok)
    assert list(toks) == ref
    assert np.array_equal(m.state_store(), os2)
    # a 97-token sequence pass (GEMM path for the quantised matrices, token tiles for the F16 ones) and its chunked form
    seq = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(97)]
    ol, ost = om.eval_sequence(seq, om.init_state())
    gl, gst = m.eval_sequence(seq, None)
    assert np.array_equal(gl, ol), (name, fmt, "sequence", float(np.abs(gl - ol).max()))
    assert np.array_equal(gst, ost), (name, fmt, "sequence state", float(np.abs(gst - ost).max()))
    cl, cst = g.eval_sequence_in_chunks(seq, None, chunk_size=40)
    assert np.array_equal(cl, ol) and np.array_equal(cst, ost)
    # ... and the same pass on the opt-in arms (plain-order quantised GEMM, F16 matrices on the matrix cores): the stated tolerance
    prev = {k: os.environ.get(k) for k in SEQ_ARM_ENV["fast"]}
    os.environ.update(SEQ_ARM_ENV["fast"])
    try:
        fl, fst = m.eval_sequence(seq, None)
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for got, want, what in ((fl, ol, "logits"), (fst, ost, "state")):
        err, tol = float(np.abs(got - want).max()), FAST_ARM_REL_TOL * (1.0 + float(np.abs(want).max()))
        assert np.isfinite(got).all() and err <= tol, (name, fmt, "fast arms", what, err, tol)
    m.free(); g.free(); om.free()


@pytest.mark.parametrize("fmt", ["Q5_1", "Q4_0"])
def test_rwkv7_2b9_slice(tmp_path, fmt):
    """fused launches (fused_v7.hip) + one kernel per op"""
    _check(tmp_path, "slice-v7-2560", fmt, 61, 1, "RWKV_MI_NO_FUSED", "RWKV_MI_NO_MEGA")


@pytest.mark.parametrize("fmt", ["Q5_1"])
def test_rwkv4_169m_slice(tmp_path, fmt):
    _check(tmp_path, "slice-v4-768", fmt, 67, 1, "RWKV_MI_NO_FUSED", "RWKV_MI_NO_MEGA")


@pytest.mark.parametrize("name,fmt", [("slice-v7-2560", "Q5_1"), ("slice-v7-2560", "Q4_0"), ("slice-v4-768", "Q5_1")])
def test_real_geometry_on_the_persistent_launch(tmp_path, name, fmt):
    """the default path of these geometries: one persistent launch per token (persist_v47.hip) + the fused launches beside it"""
    os.environ["RWKV_MI_NO_AUTOTUNE"] = "1"
    try:
        _check(tmp_path, name, fmt, 79, 2, "RWKV_MI_NO_MEGA")
    finally:
        del os.environ["RWKV_MI_NO_AUTOTUNE"]


@pytest.mark.parametrize("kind", ["ring", "regs"])
def test_rwkv6_world_vocabulary_head(tmp_path, kind):
    """V = 65536: sixteen 16-row groups per workgroup = three passes of the six consumer waves through the folded head (ring); the
    register-prefetch kernel leaves the head to k_mvf."""
    os.environ["RWKV_MI_NO_AUTOTUNE"] = "1"
    os.environ["RWKV_MI_PERSIST"] = kind
    try:
        library()
        p = str(tmp_path / "m.bin")
        spec = synth.CONFIGS["mega-v6-2048-v64k"]
        synth.write_model(p, spec, "Q4_0", seed=71)
        om = O.OracleModel(p)
        m = model(p)
        assert m.decode_path() == 2 and m.persist_kind() == {"ring": 2, "regs": 1}[kind]
        ost, st = om.init_state(), None
        for i, t in enumerate(TOKENS[:6] + [65535, 40000]):
            ol, ost = om.eval(t, ost)
            lg, st = m.eval(t, st)
            assert np.array_equal(lg, ol) and np.array_equal(st, ost), (i, float(np.abs(lg - ol).max()))
        m.state_load(None)
        toks, _ = m.decode_greedy(5, 8)
        os2, tok, ref = om.init_state(), 5, []
        for _ in range(8):
            ol, os2 = om.eval(tok, os2)
            tok = int(np.argmax(ol))
            ref.append(tok)
        assert list(toks) == ref and np.array_equal(m.state_store(), os2)
        seq = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(97)]
        ol, ost = om.eval_sequence(seq, om.init_state())
        gl, gst = m.eval_sequence(seq, None)
        assert np.array_equal(gl, ol) and np.array_equal(gst, ost)
        m.free(); om.free()
    finally:
        del os.environ["RWKV_MI_PERSIST"]
        del os.environ["RWKV_MI_NO_AUTOTUNE"]


@pytest.mark.parametrize("fmt", ["Q4_0", "Q5_1", "Q8_0"])
def test_rwkv6_3b_geometry_on_the_ring_kernel(tmp_path, fmt):
    """RWKV-6-World-3B: D = 2560 (80 blocks per row: a short second 64-block step in every D-long record), F = 8960 (280 blocks: key groups
    on 140 of the 256 workgroups), 40 heads, ten output rows per workgroup (two records on four of the six consumer waves, one on two).
    Round 3's ring kernel was instantiated for D = 4096 and 2048 only and this geometry fell to the seven launches (ring_v6.hip,
    variant table). Decode, greedy loop and the folded head against the oracle; the register-prefetch kernel has no such variant."""
    os.environ["RWKV_MI_NO_AUTOTUNE"] = "1"
    os.environ["RWKV_MI_PERSIST"] = "ring"
    try:
        library()
        p = str(tmp_path / "m.bin")
        spec = synth.CONFIGS["mega-v6-2560"]
        synth.write_model(p, spec, fmt, seed=73)
        om = O.OracleModel(p)
        m = model(p)
        assert m.decode_path() == 2 and m.persist_kind() == 2
        ost, st = om.init_state(), None
        for i, t in enumerate(TOKENS):
            ol, ost = om.eval(t, ost)
            lg, st = m.eval(t, st)
            assert np.array_equal(lg, ol) and np.array_equal(st, ost), (fmt, i, float(np.abs(lg - ol).max()))
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
    finally:
        del os.environ["RWKV_MI_PERSIST"]
        del os.environ["RWKV_MI_NO_AUTOTUNE"]
