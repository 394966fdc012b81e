"""Phase stamps of the persistent RWKV-6 decode kernel on the LDS-DMA ring (ring_v6.hip), one layer of one token:
python tools/trace_ring.py [config] [layer]. Prints cycles per phase of the consumer waves and of the comm wave, and the loader's stalls."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
os.environ.setdefault('RWKV_MI_PERSIST', 'ring')
os.environ['RWKV_MI_NO_AUTOTUNE'] = '1'
import torch; torch.cuda.init()
from gpu_lib import library, model, synth
lib = library()
cfg = sys.argv[1] if len(sys.argv) > 1 else 'rwkv6-7b'
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 5
p = '/tmp/synthetic-%s-Q4_0-seed42.bin' % cfg
if not os.path.exists(p): synth.write_model(p, synth.CONFIGS[cfg], 'Q4_0', seed=42)
m = model(p); m.state_load(None)
assert m.persist_kind() == 2, 'ring kernel not active'
L = lib.library
L.rwkv_mi_trace_phases.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; L.rwkv_mi_trace_phases.restype = ctypes.c_bool
NB = 256
out = np.zeros(NB * 8 * 32, dtype=np.int64)
assert L.rwkv_mi_trace_phases(m._ctx.ptr, 5, layer, 3, out.ctypes.data)
t = out.reshape(NB, 8, 32).astype(float)
cons = t[:, 2:, :14]
names = ['A.gather x', 'A.prologue', 'A.W1 rows', 'C.gather act', 'C.rows', 'E.gather yq', 'E.rows', 'F.gather x', 'F.prologue', 'F.key rows', 'F.rec rows', 'G.gather kq', 'G.rows']
d = np.diff(cons, axis=2)
print('CONSUMER waves: mean / min / max cycles (mean us at 2.4 GHz)')
for i, n in enumerate(names): print('%-14s %8.0f %8.0f %8.0f   %.2f' % (n, d[:, :, i].mean(), d[:, :, i].min(), d[:, :, i].max(), d[:, :, i].mean() / 2400))
print('consumer layer total cycles', (cons[:, :, 13] - cons[:, :, 0]).mean())
cr = d[:, :, 4]
# per matrix group
print('C.rows by workgroup quarter (r, k, v, g matrices) mean cycles:', ' '.join('%.0f' % cr[q * 64:(q + 1) * 64].mean() for q in range(4)), ' by consumer:', ' '.join('%.0f' % cr[:, c].mean() for c in range(6)))
for nm, i in (('F.key rows', 9), ('G.rows', 12), ('E.rows', 6)):
    print(nm, 'by consumer:', ' '.join('%.0f' % d[:, c, i].mean() for c in range(6)))
comm = t[:, 1, :12]
cn = ['A.gather x', 'A.W2+pro wait', 'B.poll tl', 'B.mix', 'C.gather act', 'D.head', 'E.gather yq', 'F.gather x', 'F.keys wait', 'F.quant k', 'G.gather kq']
dc = np.diff(comm, axis=1)
print('COMM wave: mean / min / max cycles')
for i, n in enumerate(cn): print('%-14s %8.0f %8.0f %8.0f   %.2f' % (n, dc[:, i].mean(), dc[:, i].min(), dc[:, i].max(), dc[:, i].mean() / 2400))
hw = np.argsort(-dc[:, 5])[:int((dc[:, 5] > 4 * np.median(dc[:, 5])).sum())]
print('comm D.head on the head workgroups (%d of them, first %d):' % (len(hw), hw.min() if len(hw) else -1), dc[hw, 5].mean() if len(hw) else 0, ' others:', np.delete(dc[:, 5], hw).mean())
if os.environ.get('RWKV_MI_RING_DBG', '0') != '0' and int(os.environ['RWKV_MI_RING_DBG']) & 16:
    pr = t[:, 2:, 16:20]
    print('C.rows split (cycles, mean over consumer waves): load+wait %.0f  arithmetic+butterfly %.0f  epilogue %.0f' % (pr[:, :, 0].mean(), pr[:, :, 2].mean(), pr[:, :, 3].mean()))
hv = t[:, 2:, 15].astype(np.int64)
print('records already in registers at the start of a row phase (mean over workgroups, by consumer):')
for nm, sh in (('W1', 0), ('C', 4), ('E', 8), ('FK', 12), ('FR', 16), ('G', 20)):
    print('   %-3s' % nm, ' '.join('%.2f' % ((hv[:, c] >> sh) & 15).mean() for c in range(6)))
ah = t[:, 2:, 20:23] / 1024
print('loader ahead of the phase start when the consumers begin (KiB, mean / min / max): C %.0f %.0f %.0f   keys %.0f %.0f %.0f   G %.0f %.0f %.0f' % (ah[:, :, 0].mean(), ah[:, :, 0].min(), ah[:, :, 0].max(), ah[:224, :, 1].mean(), ah[:224, :, 1].min(), ah[:224, :, 1].max(), ah[:, :, 2].mean(), ah[:, :, 2].min(), ah[:, :, 2].max()))
ld = t[:, 0, :4]
print('LOADER: active cycles mean %.0f (whole token), ring-full rounds mean %.0f max %.0f, rounds mean %.0f' % ((ld[:, 1] - ld[:, 0]).mean(), ld[:, 2].mean(), ld[:, 2].max(), ld[:, 3].mean()))
R = t
rt = lambda w, k: R[:, w, k]
print('hand-over (100 MHz real time): x staged (comm 17) spread %.2f us' % ((rt(1, 17).max() - rt(1, 17).min()) / 100))
for nm, a, b in (('tl polled -> act5 stored', 18, 19), ('act5 stored(max) -> act gathered', 19, 20), ('act gathered -> yq stored', 20, 21), ('yq stored(max) -> yq gathered', 21, 22),
                 ('yq gathered -> x_att gathered', 22, 23), ('x_att gathered -> kq stored', 23, 24), ('kq stored(max) -> kq gathered', 24, 25)):
    if 'max' in nm: v = (rt(1, b) - rt(1, a).max()) / 100
    else: v = (rt(1, b) - rt(1, a)) / 100
    print('%-36s mean %.2f us  min %.2f  max %.2f' % (nm, v.mean(), v.min(), v.max()))
end = R[:, 2:, 14].max(); start = rt(1, 17).min()
print('layer wall: (x_ffn stored, max over consumers) - (x staged, min over comm): %.2f us' % ((end - start) / 100))
by = (R[:, 2:, 14].max(axis=1) - R[:, 2:, 14].min()) / 100
print('x_ffn stored lateness by XCD:', ' '.join('%.2f' % by[x::8].mean() for x in range(8)), ' by consumer:', ' '.join('%.2f' % ((R[:, 2 + c, 14] - R[:, 2:, 14].min()) / 100).mean() for c in range(6)))
lt = os.environ.get('RWKV_MI_RING_LTRACE')
if lt and os.path.exists(lt):
    smp = np.fromfile(lt, dtype=np.int64).reshape(2, 512, 4)
    for wi, b in ((0, 0), (1, 131)):
        sm = smp[wi]; sm = sm[sm[:, 0] > 0]
        if not len(sm): continue
        t0 = sm[0, 0]
        ph = out.reshape(NB, 8, 32)[b]
        print('LOADER rounds of workgroup %d around layer %d (us since the first sample; KiB relative to the layer block): %d samples' % (b, layer, len(sm)))
        marks = {'C.rows start': ph[2:, 26], 'C.rows end': ph[2:, 27], 'keys start': ph[2:, 28], 'rec end': ph[2:, 29], 'G start': ph[2:, 30], 'G end': ph[2:, 14]}
        for k, v in marks.items(): print('   consumers %-13s %s' % (k, ' '.join('%.2f' % ((x - t0) / 100) for x in v)))
        cu_off = None
        step = max(1, len(sm) // 70)
        for r in sm[::step]: print('   t %6.2f  issued %7.1f  landed %7.1f  min_done %7.1f' % ((r[0] - t0) / 100, r[1] / 1024, r[2] / 1024, r[3] / 1024))
wt = t[:, 2:, 23:26]
print('waiting for the loader inside the layer (cycles, mean over consumer waves): W1 + C rows %.0f   E + keys + rec + G rows %.0f   by consumer (whole layer): %s' % (
    (wt[:, :, 1] - wt[:, :, 0]).mean(), (wt[:, :, 2] - wt[:, :, 1]).mean(), ' '.join('%.0f' % (wt[:, c, 2] - wt[:, c, 0]).mean() for c in range(6))))
raw = out.reshape(NB, 8, 32).astype(float)
for nm, st_, en_ in (('C.rows', 26, 27), ('keys+rec', 28, 29), ('G.rows', 30, 14)):
    st = raw[:, 2:, st_]; en = raw[:, 2:, en_]
    dur = (en.max(axis=1) - st.min(axis=1)) / 100          # per workgroup: first consumer in -> last consumer out
    late = (en.max(axis=1) - en.max(axis=1).min()) / 100
    order = np.argsort(-late)[:12]
    print('%s per workgroup (us): duration mean %.2f min %.2f max %.2f; start spread %.2f; end spread %.2f' % (nm, dur.mean(), dur.min(), dur.max(), (st.min(axis=1).max() - st.min(axis=1).min()) / 100, late.max()))
    print('   by XCD (wg %% 8): ' + ' '.join('%.2f' % dur[x::8].mean() for x in range(8)) + '   by quarter: ' + ' '.join('%.2f' % dur[q * 64:(q + 1) * 64].mean() for q in range(4)))
    print('   latest workgroups (wg: lateness, duration): ' + ' '.join('%d: %.2f %.2f |' % (b, late[b], dur[b]) for b in order))
hw = out.reshape(NB, 8, 32)[:, :, 31]
simd = (hw >> 4) & 3
print('SIMD of waves 0..7 (workgroups 0..3):', ' | '.join(' '.join(str(int(v)) for v in simd[b]) for b in range(4)))
import collections
print('wave -> SIMD patterns over all workgroups:', collections.Counter(tuple(int(v) for v in simd[b]) for b in range(NB)).most_common(4))
sys.stdout.flush(); os._exit(0)
