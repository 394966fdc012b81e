import sys, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import oracle_lib as O
from gpu_lib import gpu_mul_mat, library
library()
rng = np.random.default_rng(1)
shapes = [(64, 40, 32), (768, 70, 33), (2560, 130, 64), (2048, 160, 100), (4096, 129, 65), (7168, 33, 130), (14336, 64, 40)]
for rep in range(int(sys.argv[1])):
    for fmt in ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0"]:
        t = O.TYPE_IDS[fmt]
        for (K, N, T) in shapes:
            w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
            wb = np.concatenate([O.quantize_row(t, w[n]) for n in range(N)])
            x = rng.standard_normal((T, K)).astype(np.float32)
            print(rep, fmt, K, N, T, 'gemm...', end=' ', flush=True)
            y = gpu_mul_mat(t, wb, K, N, x)
            print('single...', end=' ', flush=True)
            y1 = gpu_mul_mat(t, wb, K, N, x[0])
            print('oracle...', end=' ', flush=True)
            ref = O.mul_mat(t, wb, K, N, x)
            print('ok' if np.array_equal(y, ref) and np.array_equal(y1[0], y[0]) else 'MISMATCH', flush=True)
