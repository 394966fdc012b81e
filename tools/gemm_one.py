"""One shape of the sequence-mode GEMM, a few launches (for counter collection): gemm_one.py K N T [fmt]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ.setdefault("RWKV_MI_TIME_MM", "3")
import numpy as np
import oracle_lib as O
from gpu_lib import gpu_mul_mat
K, N, T = (int(a) for a in sys.argv[1:4])
fmt = sys.argv[4] if len(sys.argv) > 4 else "Q4_0"
t = O.TYPE_IDS[fmt]
rng = np.random.default_rng(0)
wb = rng.integers(0, 255, size=N * (K // 32) * O.TYPE_SIZE[t], dtype=np.uint8)
wv = wb.reshape(N * (K // 32), O.TYPE_SIZE[t])
wv[:, 0:2] = np.frombuffer(np.float16(0.01).tobytes(), dtype=np.uint8)
x = rng.standard_normal((T, K)).astype(np.float32)
gpu_mul_mat(t, wb, K, N, x)
