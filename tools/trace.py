import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'tests'))
import torch; torch.cuda.init()
from gpu_lib import library, model, synth
lib=library()
p='/tmp/synthetic-rwkv6-7b-Q4_0-seed42.bin'
if not os.path.exists(p): synth.write_model(p, synth.CONFIGS['rwkv6-7b'], 'Q4_0', seed=42)
m=model(p); m.state_load(None)
L=lib.library; L.rwkv_mi_trace_phases.argtypes=[ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; L.rwkv_mi_trace_phases.restype=ctypes.c_bool
NB=256
out=np.zeros(NB*8*32,dtype=np.int64)
assert L.rwkv_mi_trace_phases(m._ctx.ptr, 5, 5, 3, out.ctypes.data)
t=out.reshape(NB,8,32)
cn=['A.poll','A.bar+prol','B.load+poll','B.comp+C.poll','D','E.poll','F.poll(+bar)','F.prol','F.bar','F.quant','G.poll','G.bar']
c=np.diff(t[:,0,:13],axis=1).astype(float)
print('COMM wave: mean / min / max cycles')
for i,n in enumerate(cn): print('%-14s %8.0f %8.0f %8.0f'%(n,c[:,i].mean(),c[:,i].min(),c[:,i].max()))
print('comm layer total', (t[:,0,12]-t[:,0,0]).mean())
wn=['-','A.barwait','A.prol','A.w1','C.issue','-','C.barwait','C.comp+issueEF','E.barwait','E.comp+pf','F.barwait','F.prol','F.rows','F.bar+G1.issue','G.barwait','G.comp']
w=np.diff(t[:,1:,:17],axis=2).astype(float)
print('WORKER waves: mean / min / max cycles')
for i,n in enumerate(wn): print('%-16s %8.0f %8.0f %8.0f'%(n,w[:,:,i].mean(),w[:,:,i].min(),w[:,:,i].max()))
print('worker layer total', (t[:,1:,16]-t[:,1:,0]).mean())
sys.stdout.flush(); os._exit(0)
