"""One rank of the two-/three-process pipeline test (tests/test_gpu_ipc_ranks.py): python ipc_rank_worker.py <model> <rank> <world> <lb> <le>
<n_layer> <n_streams> <n_tokens> <shm name> <out.npy>. Runs rwkv_mi_stage_run over HIP-IPC communicators (rwkv_mi_comm_init_ipc)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    path, rank, world, lb, le, n_layer, S, n_tokens, name, out = sys.argv[1:11]
    rank, world, lb, le, n_layer, S, n_tokens = int(rank), int(world), int(lb), int(le), int(n_layer), int(S), int(n_tokens)
    os.environ["RWKV_MI_NO_MEGA"] = "1"          # (two processes cannot both hold every CU of the one GPU)
    pkg = graft.load_package()
    lib = pkg.load_rwkv_shared_library()
    L = lib.library
    ctx = L.rwkv_mi_init_stage(path.encode(), 1, lb, le)
    assert ctx, "rwkv_mi_init_stage failed"
    handles = [ctx] + [L.rwkv_clone_context(ctx, 1) for _ in range(S - 1)]
    for h in handles:
        assert h and L.rwkv_mi_state_load(h, None)
    fwd = L.rwkv_mi_comm_init_ipc((name + "_fwd").encode(), rank, world)
    fb = L.rwkv_mi_comm_init_ipc((name + "_fb").encode(), rank, world)
    assert fwd and fb, "rwkv_mi_comm_init_ipc failed"
    arr = (ctypes.c_void_p * S)(*handles)
    first = (ctypes.c_uint32 * S)(*[(7 + 293 * j) % 500 for j in range(S)])
    toks = np.zeros((S, n_tokens), dtype=np.uint32)
    ms = ctypes.c_float(0.0)
    ok = L.rwkv_mi_stage_run(arr, S, first, n_tokens, rank, world, ctypes.c_void_p(fwd), ctypes.c_void_p(fb),
                             toks.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.byref(ms))
    assert ok, "rwkv_mi_stage_run failed"
    if rank == world - 1:
        np.save(out, toks)
    L.rwkv_mi_comm_free(ctypes.c_void_p(fwd))
    L.rwkv_mi_comm_free(ctypes.c_void_p(fb))
    for h in handles[1:]:
        L.rwkv_free(h)
    L.rwkv_free(ctx)
    print(f"rank {rank} done in {ms.value:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
