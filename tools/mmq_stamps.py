"""Timeline of one k_mmq_mfma workgroup (timing-only build -DPF_EXP_STAMP of csrc/prefill.hip, see the kernel): where a launch's time goes.
Run on the GPU box: RWKV_LIB_DIR=lib_stamp python tools/mmq_stamps.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpu_lib import gpu_mul_mat, hooks_library
import oracle_lib as O

def main():
    rng = np.random.default_rng(5)
    t = O.TYPE_IDS["Q4_0"]
    for (K, N, T) in [(64, 2048, 1024), (2048, 2048, 1024), (7168, 2048, 1024), (2048, 7168, 1024), (2048, 2048, 64)]:
        w = rng.standard_normal((N, K), dtype=np.float32) * 0.05
        wb = np.concatenate([O.quantize_row(t, w[n]) for n in range(N)])
        x = rng.standard_normal((T, K), dtype=np.float32)
        for rep in range(3):
            t0 = time.perf_counter()
            y = gpu_mul_mat(t, wb, K, N, x)
            dt = (time.perf_counter() - t0) * 1e6
        print(f"K {K:5d} N {N:5d} T {T:5d}: prologue issued {y[0][0]:7.2f} us, first chunk landed {y[0][1]:7.2f}, walk done {y[0][2]:7.2f}, "
              f"epilogue done {y[0][3]:7.2f}; steps {int(y[0][4])}; host call {dt:9.1f} us", flush=True)

main()
