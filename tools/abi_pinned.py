"""What the plain rwkv_eval ABI costs by the kind of host memory the caller hands in (round 6): pageable numpy arrays (what a caller that knows
nothing does) against page-locked ones (torch pin_memory). python tools/abi_pinned.py [config] [dtype] [tokens]"""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
torch.cuda.init()
from gpu_lib import library, model, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else 'rwkv6-7b'
dt = sys.argv[2] if len(sys.argv) > 2 else 'Q4_0'
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
p = '/tmp/synthetic-%s-%s-seed42.bin' % (cfg, dt)
if not os.path.exists(p): synth.write_model(p, synth.CONFIGS[cfg], dt, seed=42)
lib = library(); m = model(p)
print(m.persist_info())
def run(state, logits, label):
    tok = 5
    m.eval(tok, state, state, logits); tok = int(np.argmax(logits))
    t0 = time.perf_counter()
    for _ in range(n):
        m.eval(tok, state, state, logits); tok = int(np.argmax(logits))
    d = time.perf_counter() - t0
    print('%-34s %.1f tokens/s  %.3f ms/token' % (label, n / d, d * 1e3 / n), flush=True)
s0 = m.init_state()
run(s0.copy(), np.empty(m.n_vocab, dtype=np.float32), 'pageable state + logits')
ps = torch.empty(s0.size, dtype=torch.float32).pin_memory(); ps.numpy()[:] = s0
pl = torch.empty(m.n_vocab, dtype=torch.float32).pin_memory()
run(ps.numpy(), pl.numpy(), 'page-locked state + logits')
run(s0.copy(), np.empty(m.n_vocab, dtype=np.float32), 'pageable again')
# raw copy rates of a state-sized buffer
x = torch.empty(s0.size, dtype=torch.float32, device='cuda')
for label, h in (('pageable', torch.from_numpy(s0.copy())), ('page-locked', ps)):
    for direction in ('H2D', 'D2H'):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            if direction == 'H2D': x.copy_(h, non_blocking=True)
            else: h.copy_(x, non_blocking=True)
        torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 10
        print('%s %s: %.1f GB/s' % (label, direction, s0.nbytes / d / 1e9))
os._exit(0)
