import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# Sequence mode on F16 matrices has two arms (csrc/kernels.hip, launch_matvec_f): the matrix-core kernel k_mmf16_seq (the product's default:
# agrees with the CPU oracle to rounding, the order of the f32 additions differs) and the VALU kernel k_mvf in ggml's exact addition
# order (bit-identical to the oracle). The suite's np.array_equal checks run on the exact arm; tests/test_gpu_seq_f16.py switches to
# the matrix-core arm and checks it against the oracle within a stated tolerance.
os.environ.setdefault("RWKV_MI_SEQ_F16", "valu")
# The same for quantised matrices (csrc/prefill_fast.hip, launch_mmq_fast): the default accumulates the block sums in plain K order
# (k_mmq_fast), the exact arm (k_mmq_mfma) walks K in the single-token kernel's order. tests/test_gpu_prefill_fast.py runs the default arm.
os.environ.setdefault("RWKV_MI_SEQ_Q", "exact")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
