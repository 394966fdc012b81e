"""Sequence mode of F16 / F32 matrices on the matrix cores in ggml's own addition order (k_mmfx_seq, csrc/kernels.hip; round 6): FP16 / FP32
model files and the F16 low-rank stages of RWKV-7 (rwkv_graph.inc:416-447, :744-866), T >= 32 tokens per pass.

v_mfma_f32_16x16x4_f32 adds its four k slices onto the accumulator as a chain of single fused multiply-adds (tools/mfma_f32_chain.hip), so
with the slices p, p + 32, p + 64, p + 96 of a 128-element step one instruction is four links of ggml's partial sum p. The kernel must give
the SAME BITS as the VALU kernel in ggml's order (k_mvf: RWKV_MI_SEQ_F=valu, and what a single token always runs) and as the CPU oracle --
np.array_equal, no tolerance -- at kernel level for every tail (K not a multiple of 128, ragged N and T) and at model level."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import gpu_mul_mat, hooks_library, library, model, synth

pytestmark = pytest.mark.gpu


def _launches():
    L = hooks_library().library
    L.rwkv_mi_test_mmfx_launches.restype = ctypes.c_uint64
    return int(L.rwkv_mi_test_mmfx_launches())


# K: one slice, two, three (the tails of a 128-element step), exactly one step, the RWKV-7 2.9B ranks and row length, the 169M row, 7B;
# N: below a 16-row tile, ragged, more than one workgroup; T: the gate, ragged around the 16- and 32-token tiles
@pytest.mark.parametrize("K,N,T", [(32, 70, 64), (64, 130, 33), (96, 2560, 64), (128, 16, 32), (320, 96, 100), (2560, 96, 47), (2560, 320, 130),
                                   (768, 257, 97), (4096, 40, 65)])
@pytest.mark.parametrize("fmt", ["FP16", "FP32"])
def test_exact_gemm_is_the_valu_kernel_and_the_oracle(K, N, T, fmt):
    rng = np.random.default_rng(K + 7 * N + 13 * T)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float16 if fmt == "FP16" else np.float32)
    x = rng.standard_normal((T, K)).astype(np.float32)
    x[0, :16] *= 30.0
    x[T - 1] *= 0.01
    x[T // 2, :32] = 0.0
    w[N // 2, -32:] = 0
    t = O.TYPE_IDS[fmt]
    wb = w.view(np.uint8).reshape(-1)
    n0 = _launches()
    y = gpu_mul_mat(t, wb, K, N, x)
    assert _launches() == n0 + 1, "the matrix-core arm did not run"
    ref = O.mul_mat(t, wb, K, N, x)
    assert np.array_equal(y, ref), (fmt, K, N, T, float(np.abs(y - ref).max()))
    os.environ["RWKV_MI_SEQ_F"] = "valu"
    try:
        n1 = _launches()
        yv = gpu_mul_mat(t, wb, K, N, x)
        assert _launches() == n1
    finally:
        del os.environ["RWKV_MI_SEQ_F"]
    assert np.array_equal(y, yv)
    for i in (0, T // 2, T - 1):     # ... and a single token (always k_mvf) gives the row of the pass
        assert np.array_equal(gpu_mul_mat(t, wb, K, N, x[i])[0], y[i])


@pytest.mark.parametrize("name,fmt", [("test-v6", "FP16"), ("test-v6", "FP32"), ("test-v5.2", "FP16"), ("test-v4", "FP32"), ("test-v7", "FP16"),
                                      ("test-v7", "FP32"), ("slice-v7-2560", "Q5_1")])
@pytest.mark.parametrize("T", [32, 97])
def test_sequence_pass_on_the_exact_matrix_core_arm(tmp_path, name, fmt, T):
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=71)
    om = O.OracleModel(p)
    seq = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(T)]
    ol, ost = om.eval_sequence(seq, om.init_state())
    m = model(p)
    try:
        lg, st = m.eval_sequence(seq, None)
        assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, fmt, T)
        lg, st = m.eval_sequence_in_chunks(seq, None, chunk_size=40)      # (40 >= 32: chunks on the matrix cores; the last one may fall below the gate)
        assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, fmt, T, "chunks")
        os.environ["RWKV_MI_SEQ_F"] = "valu"
        try:
            lg, st = m.eval_sequence(seq, None)
        finally:
            del os.environ["RWKV_MI_SEQ_F"]
        assert np.array_equal(lg, ol) and np.array_equal(st, ost), (name, fmt, T, "valu")
    finally:
        m.free()
        om.free()
