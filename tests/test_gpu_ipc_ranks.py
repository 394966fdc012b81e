"""rwkv_mi_stage_run -- the loop one process per GPU runs (forward hops on one communicator, the token feedback on a second one, several
decode streams in flight, runner.cpp) -- with MORE THAN ONE RANK. No multi-GPU node was available to any round and RCCL refuses two ranks
on one device, so the ranks here are processes sharing GPU 0 over HIP-IPC mailboxes (rwkv_mi_comm_init_ipc): same iteration, same order
of sends and receives, another transport. Tokens must equal the one-device greedy decode of the same streams."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gpu_lib import library, model, synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name,fmt,ranges,streams", [("test-v6", "Q5_1", [(0, 1), (1, 2)], 1), ("test-v7", "Q5_1", [(0, 1), (1, 3)], 2),
                                                     ("test-v7", "Q4_0", [(0, 1), (1, 2), (2, 3)], 3), ("test-v4", "Q4_0", [(0, 1), (1, 2)], 2)])
def test_stage_run_with_several_ranks_on_one_gpu(tmp_path, name, fmt, ranges, streams):
    library()
    p = str(tmp_path / "m.bin")
    spec = synth.CONFIGS[name]
    synth.write_model(p, spec, fmt, seed=97)
    n_tokens = 10
    firsts = [(7 + 293 * j) % 500 for j in range(streams)]
    one = model(p)
    ref = []
    for f in firsts:
        one.state_load(None)
        toks, _ = one.decode_greedy(f, n_tokens)
        ref.append(list(toks))
    one.free()
    world = len(ranges)
    shm = f"/rwkvmi_test_{os.getpid()}_{name.replace('.', '')}_{fmt}_{world}"
    out = str(tmp_path / "tokens.npy")
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ipc_rank_worker.py"), p, str(r), str(world), str(lb), str(le), str(spec.n_layer),
                               str(streams), str(n_tokens), shm, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r, (lb, le) in enumerate(ranges)]
    logs = []
    try:
        for pr in procs:
            o, _ = pr.communicate(timeout=240)
            logs.append(o)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    assert all(pr.returncode == 0 for pr in procs), "\n".join(logs)
    got = np.load(out)
    assert got.shape == (streams, n_tokens)
    for j in range(streams):
        assert list(got[j]) == ref[j], (j, list(got[j]), ref[j], "\n".join(logs))
