"""Stage API on one MI355X: a model cut into stages (separate contexts, hand-off through device buffers) must reproduce the
full model bit for bit; this is the per-rank code path of the multi-GPU layer pipeline (rwkv.cpp_amd/pipeline.py)."""
import numpy as np
import pytest

import oracle_lib as O
import reference_constants as R
from gpu_lib import library, model, synth
from rwkv_cpp_amd import pipeline

pytestmark = pytest.mark.gpu


class _NoDist:
    pass


def _chain_decode(lib, path, n_layer, cuts, first, n_tokens):
    import torch
    stages = [pipeline.LibStageExecutor(lib, path, b, e, n_layer) for b, e in cuts]
    handles = [s.new_stream() for s in stages]
    tok = torch.tensor([first], dtype=torch.int32, device="cuda")
    nxt = torch.zeros(1, dtype=torch.int32, device="cuda")
    xs = [torch.zeros(stages[0].handoff_len, dtype=torch.float32, device="cuda") for _ in range(len(stages) + 1)]
    out = []
    for _ in range(n_tokens):
        for i, s in enumerate(stages):
            s.step(handles[i], tok, xs[i], xs[i + 1], nxt)   # each stage runs on its own stream: synchronise between them
            torch.cuda.synchronize()
        out.append(int(nxt.item()))
        tok.copy_(nxt)
        torch.cuda.synchronize()
    import ctypes
    n_vocab = int(lib.library.rwkv_get_n_vocab(handles[-1]))
    logits = np.empty(n_vocab, dtype=np.float32)
    assert lib.library.rwkv_mi_logits_store(handles[-1], logits.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    for s in stages:
        s.close()
    return out, logits


@pytest.mark.parametrize("name,fmt,cuts", [
    ("test-v6", "Q4_0", [(0, 1), (1, 2)]),
    ("test-v7", "Q5_1", [(0, 1), (1, 2), (2, 3)]),
    ("test-v4", "FP16", [(0, 1), (1, 2)]),
])
def test_stages_reproduce_the_full_model(tmp_path, name, fmt, cuts):
    lib = library()
    lib.rwkv_set_print_errors(None, False)
    src = str(tmp_path / "src.bin")
    synth.write_model(src, synth.CONFIGS[name], "FP16", seed=11)
    path = src
    if fmt not in ("FP32", "FP16"):
        path = str(tmp_path / "q.bin")
        lib.rwkv_quantize_model_file(src, path, fmt)
    lib.rwkv_set_print_errors(None, True)
    n_layer = synth.CONFIGS[name].n_layer
    full = model(path)
    full.state_load(None)
    exp_tokens, _ = full.decode_greedy(7, 6)
    om = O.OracleModel(path)
    st, tok = om.init_state(), 7
    for _ in range(6):
        ol, st = om.eval(tok, st)
        tok = int(np.argmax(ol))
    got, logits = _chain_decode(lib, path, n_layer, cuts, 7, 6)
    assert got == list(exp_tokens)
    assert np.array_equal(logits, ol)
    full.free()


def test_fixture_stages(golden_dir):
    lib = library()
    path = R.fixture_path(golden_dir, "7v0-834K", "FP32")
    got, _ = _chain_decode(lib, path, 12, [(0, 5), (5, 12)], 34, 5)
    full = model(path)
    full.state_load(None)
    exp, _ = full.decode_greedy(34, 5)
    assert got == list(exp)
    full.free()
