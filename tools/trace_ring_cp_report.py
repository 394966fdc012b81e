"""Critical path of one layer of the ring kernel from the real-time stamps dumped by tools/trace_ring_cp.py (a build with -DR6_RT_STAMPS=1):
python tools/trace_ring_cp_report.py <dump.npy> [rep]. Prints first / mean / last of every phase boundary over all waves, one clock for the chip."""
import numpy as np, sys
a = np.load(sys.argv[1]).astype(np.int64)
rep = int(sys.argv[2]) if len(sys.argv) > 2 else -1
t = a[rep].astype(float) / 100.0   # us
cons = t[:, 2:, :]; comm = t[:, 1, :]
t0 = cons[:, :, 0].min()
def st(x): return '%6.2f %6.2f %6.2f' % (x.min() - t0, x.mean() - t0, x.max() - t0)
names = ['layer start', 'x gathered', 'prologue A done', 'W1 rows done', 'act gathered', 'C rows done (rkvg stored)', 'yq gathered', 'E rows done (xatt stored)', 'xatt gathered', 'prologue F done',
         'key rows done (+xr quant)', 'rec rows done', 'kq gathered', 'G rows done (xffn stored)']
print('CONSUMERS  (first / mean / last over 1536 waves, us since the first wave entered the layer)')
for k, n in enumerate(names): print('  %-30s %s' % (n, st(cons[:, :, k])))
cn = ['layer start', 'x gathered', 'FL_PRO waited', 'tl polled', 'act5 stored', 'act gathered', 'head done / yq stored', 'yq gathered', 'xatt gathered', 'keys waited', 'kq stored', 'kq gathered']
print('COMM')
for k, n in enumerate(cn): print('  %-30s %s' % (n, st(comm[:, k])))
# per-consumer-index breakdown of the phase ends
for k in (5, 10, 13):
    print(names[k], 'by consumer (mean, max):', ' '.join('%.2f/%.2f' % (cons[:, c, k].mean() - t0, cons[:, c, k].max() - t0) for c in range(6)))
# which workgroups are last
for k in (3, 5, 7, 10, 13):
    e = cons[:, :, k].max(axis=1); o = np.argsort(-e)[:8]
    print('last WGs at', names[k], ':', ' '.join('%d(%.2f)' % (b, e[b] - t0) for b in o))
e = comm[:, 4]; o = np.argsort(-e)[:8]; print('last WGs at act5 stored:', ' '.join('%d(%.2f)' % (b, e[b] - t0) for b in o))
e = comm[:, 6]; o = np.argsort(-e)[:8]; print('last WGs at yq stored:', ' '.join('%d(%.2f)' % (b, e[b] - t0) for b in o))
e = comm[:, 10]; o = np.argsort(-e)[:8]; print('last WGs at kq stored:', ' '.join('%d(%.2f)' % (b, e[b] - t0) for b in o))
