"""Times the sequence-mode GEMM (k_mmq_mfma) alone: RWKV_MI_TIME_MM=n makes the mul_mat test hook repeat the launch n times
between two HIP events and print the average. Shapes: the projections of RWKV-6 1.6B / 7B at T = 1024."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ["RWKV_MI_TIME_MM"] = "20"
import numpy as np
import oracle_lib as O
from gpu_lib import gpu_mul_mat
rng = np.random.default_rng(0)
fmt = sys.argv[1] if len(sys.argv) > 1 else "Q4_0"
t = O.TYPE_IDS[fmt]
for (K, N, T) in [(2048, 2048, 1024), (2048, 7168, 1024), (7168, 2048, 1024), (2048, 160, 1024), (64, 2048, 1024), (4096, 4096, 1024), (2048, 2048, 64)]:
    nbytes = N * (K // 32) * O.TYPE_SIZE[t]
    wb = rng.integers(0, 255, size=nbytes, dtype=np.uint8)
    wv = wb.reshape(N * (K // 32), O.TYPE_SIZE[t])
    wv[:, 0:2] = np.frombuffer(np.float16(0.01).tobytes(), dtype=np.uint8)     # sane fp16 scales
    if fmt in ("Q4_1", "Q5_1"):
        wv[:, 2:4] = np.frombuffer(np.float16(-0.08).tobytes(), dtype=np.uint8)
    x = rng.standard_normal((T, K)).astype(np.float32)
    gpu_mul_mat(t, wb, K, N, x)
