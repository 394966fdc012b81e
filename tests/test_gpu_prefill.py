"""Sequence mode on the matrix cores (prefill.hip): the int8 MFMA GEMM, the tile-major activation quantiser and the lane-pipelined
WKV-5/6 kernel against the CPU oracle -- bit for bit, like the single-token path (the reference guarantees and tests
serial == sequence with memcmp, tests/test_eval_sequence_in_chunks.c:54; here it holds for every weight format)."""
import numpy as np
import pytest

import oracle_lib as O
import ctypes

from gpu_lib import gpu_mul_mat, hooks_library, library, model, synth

pytestmark = pytest.mark.gpu

QFORMATS = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0"]


def L_fast_launches():
    """launches of the plain-order GEMM (k_mmq_fast) by contexts of the test-hooks library so far"""
    L = hooks_library().library
    L.rwkv_mi_test_mmq_fast_launches.restype = ctypes.c_uint64
    return int(L.rwkv_mi_test_mmq_fast_launches())


def _weights(rng, fmt, K, N):
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    t = O.TYPE_IDS[fmt]
    return t, np.concatenate([O.quantize_row(t, w[n]) for n in range(N)])


# K: 64 = 2 blocks (RWKV-6 decay_w2), 768 / 2560 = 24 / 80 blocks (partial leaf set), 2048 = exactly 64 leaves, 4096 / 7168 / 14336 = 2 / 3.5 / 7
# blocks per leaf. N: below a row tile, ragged row tiles, more than one 128-row panel. T: one MFMA tile, ragged, several 64-token tiles.
@pytest.mark.parametrize("fmt", QFORMATS)
@pytest.mark.parametrize("K,N,T", [(64, 40, 32), (768, 70, 33), (2560, 130, 64), (2048, 160, 100), (4096, 129, 65), (7168, 33, 130), (14336, 64, 40)])
def test_mfma_gemm_matches_oracle(fmt, K, N, T):
    rng = np.random.default_rng(K * 3 + N + T)
    t, wb = _weights(rng, fmt, K, N)
    x = rng.standard_normal((T, K)).astype(np.float32)
    x[0, :32] *= 40.0
    x[T - 1] *= 0.01
    x[T // 2, 32:64] = 0.0          # an all-zero block (d = 0)
    y = gpu_mul_mat(t, wb, K, N, x)
    ref = O.mul_mat(t, wb, K, N, x)
    assert np.array_equal(y, ref), (fmt, K, N, T, float(np.abs(y - ref).max()))
    for i in (0, T // 2, T - 1):     # ... and identical to the single-token kernel
        assert np.array_equal(gpu_mul_mat(t, wb, K, N, x[i])[0], y[i])


@pytest.mark.parametrize("name,fmt", [("test-v6", "Q4_0"), ("test-v6", "Q5_1"), ("test-v6", "Q8_0"), ("test-v5.2", "Q4_1"), ("test-v5.1", "Q5_0"),
                                      ("test-v4", "Q4_0"), ("test-v7", "Q8_0"), ("test-v6", "FP16")])
@pytest.mark.parametrize("T", [32, 97])
def test_sequence_pass_matches_oracle_and_serial(tmp_path, name, fmt, T, seq_arm):
    library()
    src = str(tmp_path / "f.bin")
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    if fmt == "FP16":
        synth.write_model(p, spec, "FP16", seed=31)
    else:
        synth.write_model(src, spec, "FP32", seed=31)
        O.quantize_file(src, p, fmt)
    toks = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(T)]
    om = O.OracleModel(p)
    ol, ost = om.eval_sequence(toks, om.init_state())
    m = model(p)
    gl, gst = m.eval_sequence(toks, None)
    seq_arm(gl, ol, (name, fmt, T, "logits")); seq_arm(gst, ost, (name, fmt, T, "state"))
    # chunked: a GEMM pass of 40 tokens, then single tokens / small tiles on the other kernels
    cl, cst = m.eval_sequence_in_chunks(toks, None, chunk_size=40)
    seq_arm(cl, ol, (name, fmt, T, "chunked logits")); seq_arm(cst, ost, (name, fmt, T, "chunked state"))
    if seq_arm.arm == "exact":   # the shipped default: chunking does not change a bit (the reference's property, tests/test_eval_sequence_in_chunks.c:54)
        assert np.array_equal(cl, gl) and np.array_equal(cst, gst)
    m.free()
    om.free()


@pytest.mark.parametrize("fmt", ["FP32", "Q5_1"])
def test_wkv7_sequence_kernel_at_chunk_edges(tmp_path, fmt):
    """k_wkv7_seq stages 32 tokens per chunk and drains its out chain 15 steps behind the last token: lengths around the chunk size,
    one past it, a multiple, and a long ragged one. Logits only see the last token, so the whole state (every row's recurrence over
    every token) is compared too, and a second pass continues from the first one's state."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS["test-v7"]
    if fmt == "FP32":
        synth.write_model(p, spec, "FP32", seed=43)
    else:
        src = str(tmp_path / "f.bin")
        synth.write_model(src, spec, "FP32", seed=43)
        O.quantize_file(src, p, fmt)
    om = O.OracleModel(p)
    m = model(p)
    for T in (32, 33, 47, 64, 65, 250):
        toks = [int((1103515245 * (i + T) + 12345) % spec.n_vocab) for i in range(T)]
        ol, ost = om.eval_sequence(toks, om.init_state())
        gl, gst = m.eval_sequence(toks, None)
        assert np.array_equal(gl, ol) and np.array_equal(gst, ost), (fmt, T, float(np.abs(gst - ost).max()))
        ol2, ost2 = om.eval_sequence(toks[:40], ost)
        gl2, gst2 = m.eval_sequence(toks[:40], gst)
        assert np.array_equal(gl2, ol2) and np.array_equal(gst2, ost2), (fmt, T, "continued")
    m.free()
    om.free()


def test_sequence_pass_on_the_real_head_geometry(tmp_path):
    """RWKV-6 with D = 2048 (32 heads of 64, 64 leaves per row, ffn rows of 3.5 blocks per leaf): 3 layers, 200 tokens."""
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS["mega-v6-2048"]
    synth.write_model(p, spec, "Q4_0", seed=41)
    toks = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(200)]
    om = O.OracleModel(p)
    ol, ost = om.eval_sequence(toks, om.init_state())
    m = model(p)
    gl, gst = m.eval_sequence(toks, None)
    assert np.array_equal(gl, ol) and np.array_equal(gst, ost)
    m.free()
    om.free()


@pytest.mark.parametrize("fmt", ["Q4_0", "Q5_1"])
def test_full_length_pass_matches_the_oracle(tmp_path, fmt, seq_arm):
    """The pass that bench.py --mode prefill times is 1024 tokens: 16 token tiles per 128-row panel, the XCD-aware block map and the split
    walks + k_mmq_combine at full occupancy -- none of which a 200-token pass exercises. Same geometry (D = 2048, F = 7168, 3 layers),
    T = 1024, logits and state bit for bit against the oracle, and again through rwkv_eval_sequence_in_chunks (reference
    tests/test_eval_sequence_in_chunks.c:54 asks for memcmp equality of the two)."""
    library()
    O.lib().orc_set_fast(1)
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS["mega-v6-2048"]
    synth.write_model(p, spec, fmt, seed=43)
    toks = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(1024)]
    om = O.OracleModel(p)
    ol, ost = om.eval_sequence(toks, om.init_state())
    m = model(p, hooks=True)      # (the launch counter is a test entry point: librwkv_testhooks.so, same objects as the product library)
    n0 = L_fast_launches()
    gl, gst = m.eval_sequence(toks, None)
    seq_arm(gl, ol, "logits"); seq_arm(gst, ost, "state")
    cl, cst = m.eval_sequence_in_chunks(toks, None, chunk_size=300)
    seq_arm(cl, ol, "chunked logits"); seq_arm(cst, ost, "chunked state")
    # (the arm that was asked for is the arm that ran: the plain-order GEMM counts its launches)
    assert (L_fast_launches() > n0) == (seq_arm.arm == "fast")
    m.free()
    om.free()


def test_full_length_rwkv7_pass_matches_the_oracle(tmp_path, seq_arm):
    """RWKV-7 at the length the prefill bench times (1024 tokens = 32 staged chunks of k_wkv7_seq, the F16 low-rank stages on the token
    tiles, the quantised projections on the matrix cores): logits and the whole state bit for bit, in one pass and in chunks of 300."""
    library()
    O.lib().orc_set_fast(1)
    src, p = str(tmp_path / "f.bin"), str(tmp_path / "m.bin")
    spec = synth.CONFIGS["test-v7"]
    synth.write_model(src, spec, "FP32", seed=53)
    O.quantize_file(src, p, "Q5_1")
    toks = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(1024)]
    om = O.OracleModel(p)
    ol, ost = om.eval_sequence(toks, om.init_state())
    m = model(p)
    gl, gst = m.eval_sequence(toks, None)
    seq_arm(gl, ol, "logits"); seq_arm(gst, ost, "state")
    cl, cst = m.eval_sequence_in_chunks(toks, None, chunk_size=300)
    seq_arm(cl, ol, "chunked logits"); seq_arm(cst, ost, "chunked state")
    m.free()
    om.free()


@pytest.mark.parametrize("name,fmt", [("mega-v6-2048", "Q4_0"), ("mega-v6-2048", "Q5_1"), ("mega-v6-2048", "Q5_0"), ("slice-v7-2560", "Q8_0")])
@pytest.mark.parametrize("T", [64, 130])
def test_folded_quantiser_gives_the_same_bits_as_the_separate_launches(tmp_path, name, fmt, T):
    """Round 6 folded the activation quantiser into three producers of sequence mode (prefill.hip: the channel-mixing key product's epilogue,
    group norm + gate, the first RWKV-6 mix / RWKV-7's channel-mixing mix). Each has a switch that restores round 5's separate launches:
    logits and state of a sequence pass at real row lengths must be the oracle's bit for bit with every switch setting -- ragged last
    token tile included (T = 130), and for the formats whose images carry extra arrays (Q5_1: s, Q5_0: o)."""
    import os
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=67)
    om = O.OracleModel(p)
    seq = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(T)]
    ol, ost = om.eval_sequence(seq, om.init_state())
    m = model(p)
    switches = ["RWKV_MI_NO_EPI_QUANT", "RWKV_MI_NO_GN_QUANT", "RWKV_MI_NO_MIX_QUANT"]
    try:
        for off in ([], [switches[0]], [switches[1]], [switches[2]], switches):
            for k in off:
                os.environ[k] = "1"
            try:
                lg, st = m.eval_sequence(seq, None)
                assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, fmt, T, off)
            finally:
                for k in off:
                    os.environ.pop(k, None)
    finally:
        m.free()
        om.free()
