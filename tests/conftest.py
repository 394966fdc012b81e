import os
import sys

# The CPU oracle is OpenMP code; its default (active spinning at barriers, one thread per visible core) turns every parallel region into
# scheduler quanta when the host is shared with other jobs -- seen on the GPU pool as a 15 s test file taking 20 minutes. Passive waiting,
# set before libgomp initialises.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# Sequence mode has two arms per matrix kind (csrc/kernels.hip launch_matvec_f, csrc/prefill_fast.hip launch_mmq_fast):
#   exact (the product's default since round 6): F16 matrices on k_mvf in ggml's addition order, quantised matrices on the walk of k_mmq_mfma --
#         rwkv_eval_sequence is bit-identical to repeated rwkv_eval and to the CPU oracle for every format, whatever the chunking;
#   fast  (opt-in: RWKV_MI_SEQ_Q=fast RWKV_MI_SEQ_F16=mfma): k_mmq_fast / k_mmf16_seq -- the same operands, the f32 additions in another order.
# The model-level sequence tests run on BOTH (fixture `seq_arm` below): exact => np.array_equal, fast => the stated tolerance.
SEQ_ARM_ENV = {"exact": {"RWKV_MI_SEQ_Q": "exact", "RWKV_MI_SEQ_F16": "valu"}, "fast": {"RWKV_MI_SEQ_Q": "fast", "RWKV_MI_SEQ_F16": "mfma"}}
# what the fast arms are held to on model slices of one or two layers (logits and state): 1e-2 * (1 + max |oracle|). A product is inside 2e-6 relative
# (tests/test_gpu_prefill_fast.py, test_gpu_seq_f16.py); a random-weight network amplifies that through exp / WKV accumulation (1.4e-3 measured on a
# two-layer 1.6B slice over 1024 tokens, 4e-3 on RWKV-7's recurrence over 128)
FAST_ARM_REL_TOL = 1e-2


@pytest.fixture(params=["exact", "fast"])
def seq_arm(request):
    """Runs the test once per sequence arm; yields a checker: check(got, want, what) -> asserts equality (exact) or the tolerance (fast)."""
    import numpy as np
    arm = request.param
    prev = {k: os.environ.get(k) for k in SEQ_ARM_ENV[arm]}
    os.environ.update(SEQ_ARM_ENV[arm])

    def check(got, want, what=""):
        got, want = np.asarray(got), np.asarray(want)
        if arm == "exact":
            assert np.array_equal(got, want), (arm, what, float(np.abs(got - want).max()))
        else:
            tol = FAST_ARM_REL_TOL * (1.0 + float(np.abs(want).max()))
            err = float(np.abs(got - want).max())
            assert np.isfinite(got).all() and err <= tol, (arm, what, err, tol)
    check.arm = arm
    try:
        yield check
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
