"""Raw real-time stamps of one layer of the ring kernel (a build with -DR6_RT_STAMPS=1): dumps gpurun_out/<name>.npy [256][8][32] for offline
critical-path analysis. python tools/trace_ring_cp.py [config] [layer] [out.npy]"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
os.environ.setdefault('RWKV_MI_PERSIST', 'ring')
os.environ['RWKV_MI_NO_AUTOTUNE'] = '1'
import torch; torch.cuda.init()
from gpu_lib import library, model, synth
lib = library()
cfg = sys.argv[1] if len(sys.argv) > 1 else 'rwkv6-7b'
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 5
outp = sys.argv[3] if len(sys.argv) > 3 else 'gpurun_out/trace_cp.npy'
p = '/tmp/synthetic-%s-Q4_0-seed42.bin' % cfg
if not os.path.exists(p): synth.write_model(p, synth.CONFIGS[cfg], 'Q4_0', seed=42)
m = model(p); m.state_load(None)
assert m.persist_kind() == 2, 'ring kernel not active'
L = lib.library
L.rwkv_mi_trace_phases.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; L.rwkv_mi_trace_phases.restype = ctypes.c_bool
NB = 256
reps = []
for rep in range(int(os.environ.get('TRACE_REPS', '3'))):
    out = np.zeros(NB * 8 * 32, dtype=np.int64)
    assert L.rwkv_mi_trace_phases(m._ctx.ptr, 5, layer, 3, out.ctypes.data)
    reps.append(out.reshape(NB, 8, 32).copy())
np.save(outp, np.stack(reps))
print('saved', outp, np.stack(reps).shape)
sys.stdout.flush(); os._exit(0)
