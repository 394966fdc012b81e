"""Stamps of the head phase (ln_out + head projection inside the ring kernel): python tools/trace_head.py [config]"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
os.environ.setdefault('RWKV_MI_PERSIST', 'ring')
os.environ['RWKV_MI_NO_AUTOTUNE'] = '1'
import torch; torch.cuda.init()
from gpu_lib import library, model, synth
lib = library()
cfg = sys.argv[1] if len(sys.argv) > 1 else 'rwkv6-7b'
spec = synth.CONFIGS[cfg]
p = '/tmp/synthetic-%s-Q4_0-seed42.bin' % cfg
if not os.path.exists(p): synth.write_model(p, spec, 'Q4_0', seed=42)
m = model(p); m.state_load(None)
assert m.persist_kind() == 2, 'ring kernel not active'
L = lib.library
L.rwkv_mi_trace_phases.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; L.rwkv_mi_trace_phases.restype = ctypes.c_bool
out = np.zeros(256 * 8 * 32, dtype=np.int64)
assert L.rwkv_mi_trace_phases(m._ctx.ptr, 5, spec.n_layer, 3, out.ctypes.data)
t = out.reshape(256, 8, 32).astype(float)
c = t[:, 2:, :]
print('HEAD phase, consumer waves, cycles mean/min/max (us at 2.4 GHz)')
for nm, a, b in (('gather x', 0, 1), ('ln_out', 1, 2), ('pass 0', 2, 3), ('pass 1', 3, 4), ('pass 2', 4, 5)):
    d = c[:, :, b] - c[:, :, a]; d = d[c[:, :, b] > 0]
    if d.size: print('%-10s %9.0f %9.0f %9.0f   %.2f' % (nm, d.mean(), d.min(), d.max(), d.mean() / 2400))
print('total (real time, 100 MHz): mean %.2f us max %.2f' % (((c[:, :, 17] - c[:, :, 16]) / 100).mean(), ((c[:, :, 17] - c[:, :, 16]) / 100).max()))
print('whole phase wall: %.2f us' % ((c[:, :, 17].max() - c[:, :, 16].min()) / 100))
print('waiting for the loader: cycles mean %.0f (%.2f us)  by consumer: %s' % (c[:, :, 21].mean(), c[:, :, 21].mean() / 2400, ' '.join('%.0f' % c[:, k, 21].mean() for k in range(6))))
print('loader ahead at the start of the rows (KiB): mean %.0f min %.0f max %.0f' % ((c[:, :, 20] / 1024).mean(), (c[:, :, 20] / 1024).min(), (c[:, :, 20] / 1024).max()))
ld = t[:, 0, :4]
print('LOADER: ring-full rounds mean %.0f, rounds mean %.0f' % (ld[:, 2].mean(), ld[:, 3].mean()))
sys.stdout.flush(); os._exit(0)
