#!/usr/bin/env python3
"""tokens/s through the UNMODIFIED rwkv.h ABI (rwkv_eval: state in and out over PCIe on every call) next to the
device-resident greedy decode, same model file as bench.py. DESIGN.md section 7.3 quotes the first number; it is never
bench.py's `value`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch; torch.cuda.init()
from gpu_lib import library, model, synth

lib = library()
p = "/tmp/synthetic-rwkv6-7b-Q4_0-seed42.bin"
if not os.path.exists(p):
    synth.write_model(p, synth.CONFIGS["rwkv6-7b"], "Q4_0", seed=42)
m = model(p)
tok, st = 5, None
for _ in range(4):
    lg, st = m.eval(tok, st); tok = int(np.argmax(lg))
n = 48
t0 = time.perf_counter()
for _ in range(n):
    lg, st = m.eval(tok, st); tok = int(np.argmax(lg))
dt = time.perf_counter() - t0
print("rwkv_eval (host state in/out each call, %.1f MB each way): %.1f tokens/s, %.2f ms/token" % (st.nbytes / 1e6, n / dt, dt / n * 1e3))
m.state_load(None)
m.decode_greedy(5, 8)
toks, ms = m.decode_greedy(5, 64)
print("rwkv_mi_decode_greedy (state resident): %.1f tokens/s" % (64 / (ms * 1e-3)))
m.free()
