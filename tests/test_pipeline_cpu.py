"""The N > 1 path on CPU: world_size-2 (and 3) gloo runs of rwkv.cpp_amd/pipeline.py with the CPU oracle as the stage
executor -- partitioning, hand-off protocol (x forward, token back), multi-stream scheduling, v7's v_first hand-off."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import reference_constants as R
from gpu_lib import pkg  # noqa: F401  (loads the package so that rwkv_cpp_amd.pipeline is importable)
from rwkv_cpp_amd import pipeline


def test_partition_layers_balances_bytes():
    assert pipeline.partition_layers([1.0] * 32, 1) == [(0, 32)]
    r = pipeline.partition_layers([1.0] * 32, 8)
    assert r == [(i * 4, i * 4 + 4) for i in range(8)]
    # a head worth ~4.7 layers on the last stage (RWKV-6 7B Q4_0: 537 MB vs 113 MB per layer)
    r = pipeline.partition_layers([113.0] * 32, 8, head_cost=537.0)
    assert r[0][0] == 0 and r[-1][1] == 32 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    cost = [113.0 * (e - b) for b, e in r]
    cost[-1] += 537.0
    assert r[-1] == (31, 32) and max(cost) == pytest.approx(113.0 + 537.0)   # the head stage keeps the minimum of one layer
    with pytest.raises(ValueError):
        pipeline.partition_layers([1.0] * 3, 4)


class OracleStageExecutor(pipeline.StageExecutor):
    def __init__(self, path, lb, le):
        self.m = O.OracleModel(path)
        self.lb, self.le = lb, le
        self.is_first, self.is_last = lb == 0, le == self.m.n_layer
        self.handoff_len = self.m.n_embed * (2 if self.m.arch_major == 7 else 1)

    def new_stream(self):
        return self.m.init_state()

    def new_buffers(self):
        return torch.zeros(self.handoff_len, dtype=torch.float32), torch.zeros(1, dtype=torch.int32)

    def step(self, handle, token_buf, x_in, x_out, next_token_buf):
        xio = np.ascontiguousarray(x_in.numpy()).copy()
        logits = self.m.eval_stage(self.lb, self.le, int(token_buf[0]) if self.is_first else 0, xio, handle, True)
        if self.is_last:
            next_token_buf[0] = int(np.argmax(logits))
        else:
            x_out.copy_(torch.from_numpy(xio))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, path, n_layer, firsts, n_tokens, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ranges = pipeline.partition_layers([1.0] * n_layer, world, head_cost=0.5)
    ex = OracleStageExecutor(path, *ranges[rank])
    fb = dist.new_group(list(range(world)))
    hist, _ = pipeline.run_pipeline(ex, dist, rank, world, firsts, n_tokens, fb_group=fb)
    if rank == world - 1:
        np.save(out_path, np.array(hist, dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("version,fmt,world", [("5v2-730K", "FP32", 2), ("7v0-834K", "Q5_1", 2), ("6v0-3m", "Q5_0", 3)])
def test_pipeline_matches_full_model(golden_dir, tmp_path, version, fmt, world):
    path = R.fixture_path(golden_dir, version, fmt)
    full = O.OracleModel(path)
    firsts, n_tokens = [34, 105, 110], 6
    expect = []
    for f in firsts:
        st, tok, seq = full.init_state(), f, []
        for _ in range(n_tokens):
            lg, st = full.eval(tok, st)
            tok = int(np.argmax(lg))
            seq.append(tok)
        expect.append(seq)
    out = str(tmp_path / "hist.npy")
    mp.spawn(_worker, args=(world, _free_port(), path, full.n_layer, firsts, n_tokens, out), nprocs=world, join=True)
    assert np.load(out).tolist() == expect
