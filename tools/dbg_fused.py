"""Quick GPU check of a fused single-token path against the oracle: python tools/dbg_fused.py <config> <fmt> [direct]"""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import oracle_lib as O
from gpu_lib import library, model, synth
name, fmt = sys.argv[1], sys.argv[2]
direct = len(sys.argv) > 3
library()
spec = synth.CONFIGS[name]
p = f'/tmp/dbg_{name}_{fmt}.bin'
if direct:
    synth.write_model(p, spec, fmt, seed=7)            # low-rank / head in F16
else:
    synth.write_model(p + '.f32', spec, "FP32", seed=7)
    O.quantize_file(p + '.f32', p, fmt)                 # low-rank stay F32
om = O.OracleModel(p)
m = model(p)
print(name, fmt, 'path', m.decode_path(), 'persist', m.persist_kind(), flush=True)
ost, st, ok = om.init_state(), None, True
for i, t in enumerate([1, 2, 3, 400 % spec.n_vocab, 5, 77, 300 % spec.n_vocab, 9]):
    ol, ost = om.eval(t, ost)
    gl, st = m.eval(t, st)
    e1, e2 = np.array_equal(gl, ol), np.array_equal(st, ost)
    ok = ok and e1 and e2
    if not (e1 and e2):
        print('  token', i, 'logits equal', e1, 'state equal', e2, 'max logit diff', float(np.abs(gl - ol).max()), 'state diff', float(np.abs(st - ost).max()))
        d = np.abs(st - ost)
        per = om.state_len // spec.n_layer
        for l in range(spec.n_layer):
            seg = d[l * per:(l + 1) * per]
            D = spec.n_embed
            print('   layer', l, 'ffn_xx', float(seg[:D].max()), 'att_xx', float(seg[D:2 * D].max()), 'rest', float(seg[2 * D:].max()))
        break
print('RESULT', name, fmt, 'direct' if direct else 'quantised-from-f32', 'OK' if ok else 'MISMATCH', flush=True)
