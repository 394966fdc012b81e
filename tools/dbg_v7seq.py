import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import oracle_lib as O
from gpu_lib import library, model, synth
library()
fmt = sys.argv[1] if len(sys.argv) > 1 else "FP32"
spec = synth.CONFIGS["test-v7"]
p = '/tmp/t7_%s.bin' % fmt
if fmt == "FP32":
    synth.write_model(p, spec, "FP32", seed=43)
else:
    synth.write_model(p + '.f32', spec, "FP32", seed=43); O.quantize_file(p + '.f32', p, fmt)
om = O.OracleModel(p)
m = model(p)
for T in (32, 33, 47, 64, 65, 250):
    toks = [int((1103515245 * (i + T) + 12345) % spec.n_vocab) for i in range(T)]
    t0 = time.time()
    ol, ost = om.eval_sequence(toks, om.init_state())
    t1 = time.time()
    print('T', T, 'oracle', round(t1 - t0, 2), flush=True)
    gl, gst = m.eval_sequence(toks, None)
    print('  gpu', round(time.time() - t1, 2), np.array_equal(gl, ol), np.array_equal(gst, ost), flush=True)
    gl2, gst2 = m.eval_sequence(toks[:40], gst)
    ol2, ost2 = om.eval_sequence(toks[:40], ost)
    print('  cont', np.array_equal(gl2, ol2), np.array_equal(gst2, ost2), flush=True)
