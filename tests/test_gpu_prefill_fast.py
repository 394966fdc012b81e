"""The DEFAULT sequence-mode arm for quantised matrices (csrc/prefill_fast.hip, k_mmq_fast): the int8 MFMA GEMM with the block sums
accumulated in plain K order, where the shape fills the chip (rows of >= 16 blocks in multiples of 8, >= 128 output tiles of 128 x 64).

Contract (BASELINE north star; reference tests/test_eval_sequence_in_chunks.c:54, tests/test_tiny_rwkv.c:70-134): serial == sequence with
memcmp is promised and tested for FP32 files; quantised formats are validated against recorded thresholds. This arm keeps ggml's quantised
operands and exact integer block sums and changes only the order (and one association) of the f32 additions, so it is compared with the
oracle within bounds that are stated here:

  kernel level   |y - oracle| <= 1.5 * (2 nb + 8) * 2^-24 * A,  A = |x_q|_2 |w_deq|_2 >= sum_k |x_q[k]| |w_deq[k]|: the rounding-error bound of two f32 sums of nb
                 block terms (the oracle's tree and this kernel's chain) around the same real number. A wrong tile edge, a swapped nibble
                 or a missing block is worth >= 1e-2 A.
  model level    ONE-layer slices of the BASELINE geometries at full and ragged lengths: logits and state within 1e-4 * (1 + max |oracle|).

The exact arm (RWKV_MI_SEQ_Q=exact, what every other test of the suite runs on) stays bit-identical on the same operands; a launch
counter -- not err > 0 -- proves which kernel ran."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import gpu_mul_mat, hooks_library, library, model, synth

pytestmark = pytest.mark.gpu

QFORMATS = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0"]


@pytest.fixture(autouse=True)
def fast_arm():
    # the oracle on its AVX row kernels (bit-identical to its scalar form) and on a bounded thread count: with one OpenMP thread per hardware
    # thread of the GPU box a 512 x 256 x 128 product took 16 s inside orc_mul_mat
    O.lib().orc_set_fast(1)
    O.lib().orc_set_threads(min(16, os.cpu_count() or 1))
    old = os.environ.get("RWKV_MI_SEQ_Q")
    os.environ["RWKV_MI_SEQ_Q"] = "fast"
    yield
    if old is None:
        del os.environ["RWKV_MI_SEQ_Q"]
    else:
        os.environ["RWKV_MI_SEQ_Q"] = old


def _launches():
    L = hooks_library().library
    L.rwkv_mi_test_mmq_fast_launches.restype = ctypes.c_uint64
    return int(L.rwkv_mi_test_mmq_fast_launches())


def _set_arm(v):
    os.environ["RWKV_MI_SEQ_Q"] = v


# K: 16 blocks (the shortest rows the kernel takes), 64 (RWKV-6 1.6B), 80 (2560: a chunk count that is not a power of two), 224 (7168).
# N: whole panels, a ragged last panel / row tile. T: full 64-token tiles and a ragged last one. Small shapes run under RWKV_MI_SEQ_Q=force
# (the product only takes this kernel from 128 output tiles on); one full-size product of the 1.6B model on the default setting.
@pytest.mark.parametrize("fmt", QFORMATS)
@pytest.mark.parametrize("K,N,T,arm", [(512, 256, 128, "force"), (2048, 300, 200, "force"), (2560, 136, 100, "force"), (7168, 128, 64, "force"), (1024, 1024, 1024, "fast")])
def test_plain_order_gemm_against_the_oracle(fmt, K, N, T, arm):
    if arm == "fast" and fmt != "Q4_0":
        pytest.skip("the product-size launch (128 tiles, default setting) runs on the headline format")
    _set_arm(arm)
    rng = np.random.default_rng(K + 3 * N + 7 * T)
    t = O.TYPE_IDS[fmt]
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    wb = np.concatenate([O.quantize_row(t, w[n]) for n in range(N)])
    x = rng.standard_normal((T, K)).astype(np.float32)
    x[0, :32] *= 40.0
    x[T - 1] *= 0.01
    x[T // 2, 32:64] = 0.0
    before = _launches()
    y = gpu_mul_mat(t, wb, K, N, x)
    assert _launches() == before + 1, "k_mmq_fast did not run"
    ref = O.mul_mat(t, wb, K, N, x)
    row_bytes = wb.size // N
    wd = np.stack([O.dequantize_row(t, wb[n * row_bytes:(n + 1) * row_bytes], K) for n in range(N)]).astype(np.float64)
    xq = np.empty((T, K), dtype=np.float64)
    for i in range(T):
        q, d, _ = O.quantize_act(x[i])
        xq[i] = q.astype(np.float64) * np.repeat(d.astype(np.float64), 32)
    # A = sum_k |x_q[k]| |w_deq[k]| is bounded by the product of the two 2-norms (Cauchy-Schwarz; 1.6 x looser on these operands): an outer
    # product of norms instead of a float64 GEMM
    A = np.sqrt((xq * xq).sum(axis=1))[:, None] * np.sqrt((wd * wd).sum(axis=1))[None, :]
    nb = K // 32
    bound = 1.5 * (2 * nb + 8) * 2.0 ** -24 * A + 1e-30
    ratio = float((np.abs(y.astype(np.float64) - ref.astype(np.float64)) / bound).max())
    assert ratio <= 1.0, (fmt, K, N, T, ratio)
    # the exact arm on the same operands IS the oracle
    _set_arm("exact")
    before = _launches()
    assert np.array_equal(gpu_mul_mat(t, wb, K, N, x), ref)
    assert _launches() == before


@pytest.mark.parametrize("name,fmt,T,arm", [("rwkv6-1b6", "Q4_0", 130, "force"), ("rwkv6-1b6", "Q5_1", 97, "force"), ("rwkv6-1b6", "Q8_0", 64, "force"),
                                            ("rwkv7-2b9", "Q5_1", 100, "force"), ("rwkv7-2b9", "Q4_1", 65, "force"), ("rwkv4-169m", "Q5_0", 200, "force")])
def test_one_layer_slices_on_the_plain_order_arm(tmp_path, name, fmt, T, arm):
    """One layer of a BASELINE geometry (its real row lengths, a vocabulary of 4096): the plain-order arm within 1e-4 * (1 + max |oracle|)
    on logits and state, the exact arm bit for bit, the chunked form (passes of 3/5 of the length + a tail) inside the bound."""
    import dataclasses
    _set_arm(arm)
    p = str(tmp_path / "m.bin")
    spec = dataclasses.replace(synth.CONFIGS[name], n_vocab=4096)
    synth.write_model(p, spec, fmt, seed=67, limit_layers=1)
    toks = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(T)]
    O.lib().orc_set_fast(1)                     # (AVX row kernels of the oracle: bit-identical to its scalar form, ten times faster)
    om = O.OracleModel(p)
    ol, ost = om.eval_sequence(toks, om.init_state())
    os.environ["RWKV_MI_SEQ_F16"] = "valu"      # (only the quantised arm differs from the oracle here: RWKV-7's F16 stages stay on the exact kernel)
    m = model(p, hooks=True)
    before = _launches()
    gl, gst = m.eval_sequence(toks, None)
    assert _launches() > before, "k_mmq_fast did not run"
    for a, b, what in ((gl, ol, "logits"), (gst, ost, "state")):
        tol = 1e-4 * (1.0 + float(np.abs(b).max()))
        err = float(np.abs(a - b).max())
        assert err <= tol, (name, fmt, T, what, err, tol)
    cl, cst = m.eval_sequence_in_chunks(toks, None, chunk_size=max(64, 3 * T // 5))   # (a second pass shape; the tail below 32 tokens runs the exact kernels)
    for a, b, what in ((cl, ol, "chunked logits"), (cst, ost, "chunked state")):
        tol = 1e-4 * (1.0 + float(np.abs(b).max()))
        assert float(np.abs(a - b).max()) <= tol, (name, fmt, T, what)
    _set_arm("exact")
    before = _launches()
    el, est = m.eval_sequence(toks, None)
    assert _launches() == before and np.array_equal(el, ol) and np.array_equal(est, ost)
    m.free()
    om.free()
