import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'tests'))
import torch; torch.cuda.init()
from gpu_lib import library, model, synth
lib=library()
cfg = sys.argv[1] if len(sys.argv) > 1 else 'rwkv6-7b'
p='/tmp/synthetic-%s-Q4_0-seed42.bin' % cfg
if not os.path.exists(p): synth.write_model(p, synth.CONFIGS[cfg], 'Q4_0', seed=42)
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
# hand-over latencies from the 100 MHz real-time counter (consistent across XCDs): ready = last producer stamp anywhere
R=t.astype(float)
def ho(name, prod, cons):
    ready=max(R[:,w,k].max() for (w,k) in prod)
    seen=R[:,0,cons]
    print('%-22s last producer -> consumer staged: mean %.2f us  min %.2f  max %.2f   (producer spread %.2f us)'%(name,(seen.mean()-ready)/100,(seen.min()-ready)/100,(seen.max()-ready)/100,(ready-min(R[:,w,k].min() for (w,k) in prod))/100))
W=range(1,8)
ho('tl (A->B)', [(w,17) for w in W], 18)
ho('act5 (B->C)', [(0,19)], 20)
ho('yq (D->E)', [(0,22)], 23)
ho('x_att (E->F)', [(0,24)]+[(w,24) for w in W], 25)
ho('kq (F->G)', [(0,26)], 27)
# who is late? producer stamps relative to the earliest, by XCD (workgroup index mod 8) and the five latest workgroups
for name,(ws,k) in (('tl stored',(W,17)),('act5 stored',([0],19)),('rkvg stored',(range(8),21)),('x_att stored',(range(8),24)),('kq stored',([0],26)),('x_ffn stored',(range(8),28))):
    tt=np.stack([R[:,w,k] for w in ws],axis=1).max(axis=1)
    rel=(tt-tt.min())/100
    byx=[rel[x::8].mean() for x in range(8)]
    late=np.argsort(-rel)[:5]
    print('%-13s spread %.2f us; mean lateness by XCD %s; latest workgroups %s'%(name,rel.max(),' '.join('%.2f'%v for v in byx),' '.join('%d(%.2f)'%(b,rel[b]) for b in late)))
# is the comm wave (wave 0: also the eighth row owner) the straggler of the row phases?
for name,k in (('rkvg stored',21),('x_att stored',24),('x_ffn stored',28)):
    rel=(R[:,:,k]-R[:,:,k].min())/100
    print('%-13s mean lateness by wave: %s'%(name,' '.join('%.2f'%rel[:,w].mean() for w in range(8))))
print('layer wall (xffn stored, max over all) - (x staged A, min): %.2f us'%((max(R[:,w,28].max() for w in range(8))-R[:,0,17].min())/100))
sys.stdout.flush(); os._exit(0)
