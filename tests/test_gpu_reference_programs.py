"""The reference's own ggml-free test programs (reference tests/CMakeLists.txt:51-57), compiled UNMODIFIED from
/root/reference/tests/*.c against include/rwkv.h + librwkv.so (recipe: oracle/Makefile `ref_tests`, outputs oracle/_ref/) and
run here on the GPU in a directory that holds tests/golden/* -- the reference-side callers of the drop-in boundary.

  test_quantization_format_compatibility.c  shipped Q5_0/Q5_1 files of all five architectures against the recorded thresholds
  test_eval_sequence_in_chunks.c            memcmp(serial state, chunked-sequence state), chunk sizes 1/2/8/10 (:54)
  test_logit_calculation_skipping.c         memcmp with / without logits, serial and sequence (:45,:83)
  test_context_cloning.c                    memcmp(original context, clone) (:48)
  test_tiny_rwkv.c                          every architecture x {FP32, FP16, FP32->Qx, FP16->Qx} incl. rwkv_quantize_model_file;
                                            the 6v0 FP32/FP16 fixtures are missing from the reference mount (.MISSING_LARGE_BLOBS),
                                            so the program is expected to pass 4v0, 5v1, 5v2 (42 models) and stop at that file.

The binaries are built where /root/reference exists (the dev container, by __graft_entry__.build()) and travel to the GPU box
like the product .so; nothing here reads /root/reference at run time.
"""
import os
import shutil
import subprocess

import pytest

from conftest import SEQ_ARM_ENV

from gpu_lib import library

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _binary(name):
    p = os.path.join(REF_BIN, name)
    if not os.path.exists(p) and os.path.isdir("/root/reference/tests"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_tests"], stdout=subprocess.DEVNULL)
    if not os.path.exists(p):
        pytest.fail(f"{p} missing: build it in the dev container with `make -C oracle ref_tests` (needs /root/reference)")
    return p


def _run(name, tmp_path, arm="exact"):
    library()  # makes sure librwkv.so is built
    for f in os.listdir(GOLDEN):
        if f.endswith(".bin"):
            shutil.copy(os.path.join(GOLDEN, f), tmp_path / f)
    env = dict(os.environ)
    env.pop("RWKV_MI_NO_MEGA", None)
    env.update(SEQ_ARM_ENV[arm])   # the product's default arms (exact) / the opt-in arms: the reference's own thresholds hold on both
    return subprocess.run([_binary(name)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("arm", ["exact", "fast"])
@pytest.mark.parametrize("name", ["test_quantization_format_compatibility", "test_eval_sequence_in_chunks",
                                  "test_logit_calculation_skipping", "test_context_cloning"])
def test_reference_program_passes(name, arm, tmp_path):
    r = _run(name, tmp_path, arm)
    assert r.returncode == 0, f"{name} exited with {r.returncode}\n{r.stderr[-4000:]}"
    assert "Assertion failed" not in r.stderr


@pytest.mark.parametrize("arm", ["exact", "fast"])
def test_reference_tiny_rwkv_with_the_missing_6v0_entry_skipped(arm, tmp_path):
    """The reference's test_tiny_rwkv.c, its 6v0 entry skipped at build time (oracle/Makefile: the mount has no 6v0 FP32 / FP16 fixture): RWKV-4,
    5.1, 5.2 and RWKV-7 in FP32, FP16 and every quantised format from both sources -- 48 models, serial and sequence mode, every recorded
    threshold of test_tiny_rwkv.c:38-134 for them -- must pass, i.e. the program exits 0 (it aborts at the end when any check failed)."""
    r = _run("test_tiny_rwkv_no6v0", tmp_path, arm)
    err = r.stderr
    assert r.returncode == 0, f"exit code {r.returncode}\n{err[-4000:]}"
    assert "Assertion failed" not in err and "6v0" not in err
    tested = [l for l in err.splitlines() if l.startswith("Testing tiny-rwkv-")]
    assert len(tested) == 48, tested
    for arch in ("4v0-660K", "5v1-730K", "5v2-730K", "7v0-834K"):
        assert sum(arch in l for l in tested) == 12
    assert err.count("Serial difference sum") == 48 and err.count("Sequence difference sum") == 48


def test_reference_tiny_rwkv_passes_until_the_missing_fixture(tmp_path):
    r = _run("test_tiny_rwkv", tmp_path)
    err = r.stderr
    stop = "Testing tiny-rwkv-6v0-3m-FP32.bin"
    assert stop in err, err[-3000:]
    before = err[:err.index(stop)]
    assert "Assertion failed" not in before, before[-3000:]
    # 3 architectures x (FP32, FP16, 5 formats x {from FP32, from FP16}) = 36 models, each serial + sequence
    tested = [l for l in before.splitlines() if l.startswith("Testing tiny-rwkv-")]
    assert len(tested) == 36, tested
    for arch in ("4v0-660K", "5v1-730K", "5v2-730K"):
        assert sum(arch in l for l in tested) == 12
    assert before.count("Serial difference sum") == 36 and before.count("Sequence difference sum") == 36
    # ... and it stops exactly where the reference mount has no file (rwkv_init_from_file fails -> the program dereferences NULL / aborts)
    assert r.returncode != 0
