"""Pins the CPU oracle against every golden vector the reference's tests hold for this path (CPU-only)."""
import filecmp
import os

import numpy as np
import pytest

import oracle_lib as O
import reference_constants as R


def _diff_sum(model_path, golden_dir, version, sequence=False):
    exp = R.expected_logits(golden_dir, version)
    m = O.OracleModel(model_path)
    st = m.init_state()
    if sequence:
        logits, st = m.eval_sequence(R.PROMPT, st)
    else:
        for t in R.PROMPT:
            logits, st = m.eval(t, st)
    m.free()
    d = logits - exp
    return float(d.sum(dtype=np.float32)), float(np.abs(d).max())


@pytest.mark.parametrize("version", R.HAVE_FP32_FP16)
def test_fp32_golden_logits(golden_dir, version):
    # tests/test_tiny_rwkv.c:40-53: |sum| <= 1.05 * 0.001; plus a much tighter element-wise bound.
    s, mx = _diff_sum(R.fixture_path(golden_dir, version, "FP32"), golden_dir, version)
    assert abs(s) <= 0.001 * R.TOLERANCE_FACTOR
    assert mx <= 1e-5


@pytest.mark.parametrize("version", R.HAVE_FP32_FP16)
def test_fp16_recorded_threshold(golden_dir, version):
    s, _ = _diff_sum(R.fixture_path(golden_dir, version, "FP16"), golden_dir, version)
    assert abs(s) <= abs(R.FULL[version]["FP16"]) * R.TOLERANCE_FACTOR


@pytest.mark.parametrize("version", list(R.SHIPPED_Q5))
@pytest.mark.parametrize("fmt", ["Q5_0", "Q5_1"])
def test_shipped_q5_fixtures(golden_dir, version, fmt):
    # tests/test_quantization_format_compatibility.c
    for seq in (False, True):
        s, _ = _diff_sum(R.fixture_path(golden_dir, version, fmt), golden_dir, version, sequence=seq)
        assert abs(s) <= abs(R.SHIPPED_Q5[version][fmt]) * R.TOLERANCE_FACTOR


def test_v6_q5_0_matches_recorded_value(golden_dir):
    # The only v6 known-answer with fixtures present: recorded -21.151785.
    s, _ = _diff_sum(R.fixture_path(golden_dir, "6v0-3m", "Q5_0"), golden_dir, "6v0-3m")
    assert s == pytest.approx(-21.151785, abs=2e-3)


@pytest.mark.parametrize("version", R.HAVE_FP32_FP16)
@pytest.mark.parametrize("fmt", ["Q5_0", "Q5_1"])
def test_quantizer_byte_exact_vs_shipped(golden_dir, tmp_path, version, fmt):
    out = str(tmp_path / "q.bin")
    O.quantize_file(R.fixture_path(golden_dir, version, "FP32"), out, fmt)
    assert filecmp.cmp(out, R.fixture_path(golden_dir, version, fmt), shallow=False)


@pytest.mark.parametrize("version", R.HAVE_FP32_FP16)
@pytest.mark.parametrize("source", ["FP32", "FP16"])
def test_quantized_recorded_thresholds(golden_dir, tmp_path, version, source):
    # tests/test_tiny_rwkv.c:136-173
    table = R.FROM_FP32 if source == "FP32" else R.FROM_FP16
    for i, fmt in enumerate(R.QUANT_FORMATS):
        out = str(tmp_path / f"{fmt}.bin")
        O.quantize_file(R.fixture_path(golden_dir, version, source), out, fmt)
        s, _ = _diff_sum(out, golden_dir, version)
        assert abs(s) <= abs(table[version][i]) * R.TOLERANCE_FACTOR, (version, source, fmt, s)
        key = (version, source, fmt)
        if key in R.TIGHT_KAT:
            assert s == pytest.approx(R.TIGHT_KAT[key], abs=2e-5), key


def test_fp16_conversion_exhaustive():
    L = O.lib()
    h = np.arange(65536, dtype=np.uint16)
    ref = h.view(np.float16).astype(np.float32)
    got = np.array([L.orc_f16_to_f32(int(v)) for v in h[::7]], dtype=np.float32)
    np.testing.assert_array_equal(got.view(np.uint32)[~np.isnan(ref[::7])], ref[::7].view(np.uint32)[~np.isnan(ref[::7])])
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 1e3, 7e4)])
    x = np.concatenate([x, np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e10, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5], dtype=np.float32)])
    want = x.astype(np.float16).view(np.uint16)
    got = np.array([L.orc_f32_to_f16(float(v)) for v in x], dtype=np.uint16)
    np.testing.assert_array_equal(got, want)


def test_quantize_dequantize_roundtrip_error_bounds():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(32 * 64).astype(np.float32)
    for name, bits in (("Q4_0", 4), ("Q4_1", 4), ("Q5_0", 5), ("Q5_1", 5), ("Q8_0", 8)):
        t = O.TYPE_IDS[name]
        y = O.dequantize_row(t, O.quantize_row(t, x), x.size)
        err = np.abs(y - x).reshape(-1, 32).max(axis=1)
        span = np.abs(x).reshape(-1, 32).max(axis=1) * 2
        assert np.all(err <= span / (2 ** bits - 1) * 1.01 + 1e-3), name


def test_state_layout_and_init(golden_dir):
    # rwkv.cpp:171-179, rwkv_eval.inc:224-241
    m = O.OracleModel(R.fixture_path(golden_dir, "4v0-660K", "FP32"))
    assert m.state_len == m.n_embed * 5 * m.n_layer
    s = m.init_state().reshape(m.n_layer, 5, m.n_embed)
    assert np.all(s[:, :4] == 0) and np.all(s[:, 4] == np.float32(-1e30))
    m5 = O.OracleModel(R.fixture_path(golden_dir, "5v2-730K", "FP32"))
    assert (m5.head_count, m5.head_size) == (8, 8)
    assert m5.state_len == m5.n_embed * (2 + m5.head_size) * m5.n_layer
    assert np.all(m5.init_state() == 0)
    m7 = O.OracleModel(R.fixture_path(golden_dir, "7v0-834K", "FP32"))
    assert (m7.arch_major, m7.head_count, m7.head_size) == (7, 1, 64)
    m6 = O.OracleModel(R.fixture_path(golden_dir, "6v0-3m", "Q5_0"))
    assert (m6.arch_major, m6.head_count, m6.head_size, m6.ffn_size) == (6, 16, 8, 448)


def test_simd_row_kernels_are_bit_identical_to_the_scalar_oracle(golden_dir):
    """oracle/rwkv_oracle_fast.c (AVX2 / AVX-512-VNNI row kernels, used only by bench.py's cpu_baseline leg) against the scalar
    loops: random rows of every weight type, and whole evaluations of the shipped fixtures."""
    import glob
    L = O.lib()
    rng = np.random.default_rng(5)
    try:
        for fmt, t in O.TYPE_IDS.items():
            for K, N, T in ((64, 7, 2), (2560, 33, 3), (4096, 17, 1)):
                w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
                if fmt == "FP32":
                    wb = w.view(np.uint8).reshape(-1)
                elif fmt == "FP16":
                    wb = w.astype(np.float16).view(np.uint8).reshape(-1)
                else:
                    wb = np.concatenate([O.quantize_row(t, w[n]) for n in range(N)])
                x = rng.standard_normal((T, K)).astype(np.float32)
                x[0, :32] *= 50.0
                L.orc_set_fast(0)
                ref = O.mul_mat(t, wb, K, N, x)
                L.orc_set_fast(1)
                got = O.mul_mat(t, wb, K, N, x)
                assert np.array_equal(ref, got), (fmt, K, N)
        for path in sorted(glob.glob(os.path.join(golden_dir, "tiny-rwkv-*.bin"))):
            outs = []
            for fast in (0, 1):
                L.orc_set_fast(fast)
                m = O.OracleModel(path)
                st = m.init_state()
                for tok in (34, 105, 110):
                    lg, st = m.eval(tok, st)
                outs.append((lg, st))
                m.free()
            assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), path
    finally:
        L.orc_set_fast(0)
