"""GPU parity against the reference's own known answers: the Python rendering of tests/test_tiny_rwkv.c and
tests/test_quantization_format_compatibility.c, run through the C ABI of librwkv.so on an MI355X."""
import filecmp

import numpy as np
import pytest

import oracle_lib as O
import reference_constants as R
from gpu_lib import library, model, run_prompt

pytestmark = pytest.mark.gpu


def _check(path, golden_dir, version, recorded, extra_abs=None):
    """logit_difference_validator.inc:29-97: serial then sequence; |sum| <= 1.05 |recorded| both times."""
    exp = R.expected_logits(golden_dir, version)
    m = model(path)
    assert m.n_vocab == 256
    out = {}
    for seq in (False, True):
        logits, state = run_prompt(m, R.PROMPT, sequence=seq)
        d = logits - exp
        s = float(d.sum(dtype=np.float32))
        assert abs(s) <= abs(recorded) * R.TOLERANCE_FACTOR, (path, "sequence" if seq else "serial", s, recorded)
        out[seq] = (logits, state)
        if extra_abs is not None:
            assert float(np.abs(d).max()) <= extra_abs
    # serial == sequence, bit for bit (stronger than the reference asks)
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])
    m.free()
    return out[False][0]


def _oracle_logits(path):
    om = O.OracleModel(path)
    st = om.init_state()
    for t in R.PROMPT:
        lg, st = om.eval(t, st)
    om.free()
    return lg


@pytest.mark.parametrize("version", R.HAVE_FP32_FP16)
def test_fp32(golden_dir, version):
    p = R.fixture_path(golden_dir, version, "FP32")
    logits = _check(p, golden_dir, version, R.FULL[version]["FP32"], extra_abs=1e-5)
    assert np.array_equal(logits, _oracle_logits(p)), "GPU logits are not bit-identical to the CPU oracle"


@pytest.mark.parametrize("version", R.HAVE_FP32_FP16)
def test_fp16(golden_dir, version):
    p = R.fixture_path(golden_dir, version, "FP16")
    logits = _check(p, golden_dir, version, R.FULL[version]["FP16"])
    assert np.array_equal(logits, _oracle_logits(p)), "GPU logits are not bit-identical to the CPU oracle"


@pytest.mark.parametrize("version", list(R.SHIPPED_Q5))
@pytest.mark.parametrize("fmt", ["Q5_0", "Q5_1"])
def test_shipped_q5(golden_dir, version, fmt):
    p = R.fixture_path(golden_dir, version, fmt)
    logits = _check(p, golden_dir, version, R.SHIPPED_Q5[version][fmt])
    assert np.array_equal(logits, _oracle_logits(p)), "GPU logits are not bit-identical to the CPU oracle"


@pytest.mark.parametrize("version", R.HAVE_FP32_FP16)
@pytest.mark.parametrize("source", ["FP32", "FP16"])
def test_quantize_then_eval(golden_dir, tmp_path, version, source):
    # tests/test_tiny_rwkv.c:136-173: quantise on the fly with rwkv_quantize_model_file, then validate
    lib = library()
    lib.rwkv_set_print_errors(None, False)
    table = R.FROM_FP32 if source == "FP32" else R.FROM_FP16
    for i, fmt in enumerate(R.QUANT_FORMATS):
        out = str(tmp_path / f"{version}-{source}-to-{fmt}.bin")
        lib.rwkv_quantize_model_file(R.fixture_path(golden_dir, version, source), out, fmt)
        ref = str(tmp_path / "oracle.bin")
        O.quantize_file(R.fixture_path(golden_dir, version, source), ref, fmt)
        assert filecmp.cmp(out, ref, shallow=False), "product quantiser differs from the oracle's"
        if source == "FP32" and fmt in ("Q5_0", "Q5_1"):
            assert filecmp.cmp(out, R.fixture_path(golden_dir, version, fmt), shallow=False), "differs from the shipped fixture"
        logits = _check(out, golden_dir, version, table[version][i])
        assert np.array_equal(logits, _oracle_logits(out)), (version, source, fmt, "GPU != oracle")
    lib.rwkv_set_print_errors(None, True)
