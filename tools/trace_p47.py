"""Phase trace of the persistent RWKV-4 / RWKV-7 decode launch (csrc/persist_v47.hip): stamps of the 100 MHz real-time counter (consistent
across XCDs) per wave in one layer, printed as microseconds since the earliest stamp of the layer.
usage: python tools/trace_p47.py <config> <dtype> [layer]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import torch  # noqa: E402
torch.cuda.init()
from gpu_lib import library, model, synth  # noqa: E402

lib = library()
cfg = sys.argv[1] if len(sys.argv) > 1 else 'rwkv4-169m'
dt = sys.argv[2] if len(sys.argv) > 2 else 'Q5_1'
layer = int(sys.argv[3]) if len(sys.argv) > 3 else 5
p = '/tmp/synthetic-%s-%s-seed42.bin' % (cfg, dt)
spec = synth.CONFIGS[cfg]
if not os.path.exists(p):
    synth.write_model(p, spec, dt, seed=42)
os.environ['RWKV_MI_NO_AUTOTUNE'] = '1'
m = model(p)
assert m.persist_kind() == 3, m.persist_kind()
m.state_load(None)
m.decode_greedy(5, 8)
L = lib.library
L.rwkv_mi_trace_phases.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L.rwkv_mi_trace_phases.restype = ctypes.c_bool
out = np.zeros(256 * 8 * 32, dtype=np.int64)
assert L.rwkv_mi_trace_phases(m._ctx.ptr, 5, layer, 3, out.ctypes.data)
t = out[:256 * 9 * 16].reshape(256, 9, 16).astype(float) / 100.0      # microseconds
if os.environ.get('TRACE_DUMP'):   # raw stamps for offline analysis (tools/late_p47.py)
    np.save(os.environ['TRACE_DUMP'], out[:256 * 9 * 16].reshape(256, 9, 16))
D = spec.n_embed
v7 = spec.arch == '7'
GK = D // 8
GPB = 1 if GK + (D // 64 if v7 else 0) <= 256 else 2
NR = GK // GPB
H = D // 64 if v7 else 0
rows = t[:NR]
t0 = rows[:, :8, 0].min()
print(f'{cfg} {dt} layer {layer}: NR {NR} row workgroups, {H} head workgroups, GPB {GPB}; microseconds since the first worker entered the layer')


def show(name, a):
    print('  %-46s mean %7.2f   min %7.2f   max %7.2f' % (name, a.mean() - t0, a.min() - t0, a.max() - t0))


print('row workgroups, WORKER waves (stamp = reached)')
wn = ['0 layer top', '1 B1 passed (x - mean, scale in LDS)', '2 prologue A done', '3 B2 passed + A rows + epilogue stored', '4 B3 passed (yq staged)', '5 output rows, x_att stored',
      '6 B4 passed (stats of x_att)', '7 prologue F done + B5', '8 key (+ receptance) rows done', '9 B6 passed (kq staged)', '10 value rows, x stored']
for k, n in enumerate(wn):
    show(n, rows[:, :8, k])
print('row workgroups, COMM wave')
cn = ['0 layer top', '1 x gathered', '2 stats done (+ lr1 jobs issued)', '3 B1 B2 passed (+ lr1 rows stored)', '4 y gathered / quantised', '5 B3 passed + x_att gathered', '6 stats done',
      '7 B4 B5 passed + key flag seen', '8 kq gathered (all groups staged)', '9 own key groups quantised + stored']
for k, n in ((0, cn[0]), (1, cn[1]), (2, cn[2]), (3, cn[3]), (4, cn[4]), (5, cn[5]), (6, cn[6]), (7, cn[7]), (9, cn[9]), (8, cn[8])):
    show(n, rows[:, 8, k])
if H:
    hd = t[NR:NR + H]
    print('head workgroups, COMM wave')
    for k, n in enumerate(['0 layer top', '1 lr1 gathered (r k v: under the second stages)', '2 H1 H2 passed (second stages done)', '3 WKV-7 .. yq stored']):
        show(n, hd[:, 8, k])
print('layer wall (first worker in -> last worker out): %.2f us' % (rows[:, :8, 10].max() - t0))
# who is late: the stores that feed a hand-over (worker stamps 3, 5, 8, 10), latest wave per workgroup -- by XCD (workgroup id mod 8) and the six latest
for k in (0, 3, 5, 8, 10):
    w = rows[:, :8, k].max(axis=1) - t0
    by = ' '.join('%6.2f' % w[x::8].mean() for x in range(8))
    late = np.argsort(-w)[:6]
    print('  late @%-2d  mean by XCD: %s | latest: %s' % (k, by, ', '.join('wg %d (%+.2f)' % (b, w[b] - w.mean()) for b in late)))
for k in (1, 4, 5, 8):
    w = rows[:, 8, k] - t0
    by = ' '.join('%6.2f' % w[x::8].mean() for x in range(8))
    late = np.argsort(-w)[:4]
    print('  comm @%-2d  mean by XCD: %s | latest: %s' % (k, by, ', '.join('wg %d (%+.2f)' % (b, w[b] - w.mean()) for b in late)))
# the waves of the latest workgroup at stamp 10
b = int(np.argmax(rows[:, :8, 10].max(axis=1)))
print('  waves of wg %d @10: %s ; @8: %s' % (b, ' '.join('%.2f' % (v - t0) for v in rows[b, :8, 10]), ' '.join('%.2f' % (v - t0) for v in rows[b, :8, 8])))
# the tail (ln_out + head + argmax inside the launch): stamps 11..15 of every wave of the whole grid, relative to the first wave entering it
ta = t[:, :, 11:16]
if ta[:NR].max() > 0:
    live = ta[:, :, 0] > 0
    z = ta[:, :, 0][live].min()
    def show2(name, a, mask):
        a = a[mask]
        print('  %-46s mean %7.2f   min %7.2f   max %7.2f' % (name, a.mean() - z, a.min() - z, a.max() - z))
    nblk = NR + H
    for nm, sel in (('row workgroups', slice(0, NR)), ('head workgroups', slice(NR, nblk)), ('spare workgroups', slice(nblk, 256))):
        if sel.stop <= sel.start or not live[sel].any():
            continue
        print('tail,', nm)
        for k, n in enumerate(['11 entered', '12 T1 passed (ln_out statistics)', '13 T2 passed (normalised x in LDS)', '14 head rows done']):
            show2(n, ta[sel, :, k], live[sel])
    print('  argmax written by workgroup 0: %.2f' % (ta[0, 0, 4] - z))
m.free()
