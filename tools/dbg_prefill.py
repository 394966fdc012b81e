import sys, time, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import oracle_lib as O
from gpu_lib import library, model, synth
T = int(sys.argv[1]); name = sys.argv[2]
library()
p='/tmp/dbg_%s.bin' % name
spec = synth.CONFIGS[name]
synth.write_model(p, spec, "Q4_0", seed=41)
toks = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(T)]
print('create', flush=True)
m = model(p)
print('created path', m.decode_path(), flush=True)
t=time.time(); gl, gst = m.eval_sequence(toks, None); print('gpu seq done', time.time()-t, flush=True)
om = O.OracleModel(p)
ol, ost = om.eval_sequence(toks, om.init_state())
print('equal', np.array_equal(gl, ol), np.array_equal(gst, ost), flush=True)
