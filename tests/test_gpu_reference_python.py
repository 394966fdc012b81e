"""The reference's own Python consumers on the GPU: python/rwkv_cpp/rwkv_cpp_shared_library.py and rwkv_cpp_model.py,
UNMODIFIED (staged by `make -C oracle ref_py` into the git-ignored oracle/_ref/py/, like the reference's C test programs), bound to
librwkv.so. RWKVModel.eval / eval_sequence / eval_sequence_in_chunks (rwkv_cpp_model.py:85-299) with NumPy arrays and with PyTorch CPU
tensors -- including the in == out aliasing the reference's scripts use -- and the evaluation loop of python/measure_pexplexity.py:64-109;
logits and state must equal the CPU oracle's bit for bit."""
import importlib.util
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from gpu_lib import library, pkg

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PY = os.path.join(ROOT, "oracle", "_ref", "py")


def _ref_modules():
    if not os.path.isfile(os.path.join(REF_PY, "rwkv_cpp_model.py")):
        pytest.skip("oracle/_ref/py is not staged (run __graft_entry__.build() where /root/reference exists)")
    mods = {}
    for name in ("rwkv_cpp_shared_library", "rwkv_cpp_model"):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_PY, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod          # rwkv_cpp_model.py imports rwkv_cpp_shared_library as a top-level module
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods["rwkv_cpp_shared_library"], mods["rwkv_cpp_model"]


TOKENS = [34, 105, 110, 32, 97, 32, 115, 104, 111, 99, 107, 105, 110, 103, 32, 102, 105, 110, 100, 105, 110, 103]


@pytest.mark.parametrize("fixture", ["tiny-rwkv-6v0-3m-Q5_1.bin", "tiny-rwkv-7v0-834K-FP32.bin", "tiny-rwkv-4v0-660K-Q5_0.bin", "tiny-rwkv-5v2-730K-FP16.bin"])
def test_reference_rwkvmodel_numpy(golden_dir, fixture):
    library()   # (torch initialises HIP before librwkv.so is loaded: see gpu_lib)
    shl, mdl = _ref_modules()
    path = os.path.join(golden_dir, fixture)
    lib = shl.RWKVSharedLibrary(pkg.LIB_PATH)
    model = mdl.RWKVModel(lib, path, thread_count=2, gpu_layer_count=0)
    om = O.OracleModel(path)
    assert model.n_vocab == om.n_vocab
    logits, state, ost = None, None, om.init_state()
    for t in TOKENS:
        logits, state = model.eval(t, state, use_numpy=True)
        ol, ost = om.eval(t, ost)
        assert np.array_equal(logits, ol) and np.array_equal(state, ost)
    # sequence mode and chunked sequence mode
    sl, sst = model.eval_sequence(TOKENS, None, use_numpy=True)
    assert np.array_equal(sl, logits) and np.array_equal(sst, state)
    cl, cst = model.eval_sequence_in_chunks(TOKENS, None, chunk_size=5, use_numpy=True)
    assert np.array_equal(cl, logits) and np.array_equal(cst, state)
    model.free()
    om.free()


def test_reference_rwkvmodel_torch_tensors_and_the_perplexity_loop(golden_dir):
    """measure_pexplexity.py:64-109: `logits, state = model.eval(token, state, state, logits)` with PyTorch CPU tensors, state_in == state_out."""
    import torch
    library()
    shl, mdl = _ref_modules()
    path = os.path.join(golden_dir, "tiny-rwkv-6v0-3m-Q5_0.bin")
    lib = shl.RWKVSharedLibrary(pkg.LIB_PATH)
    model = mdl.RWKVModel(lib, path, thread_count=2, gpu_layer_count=0)
    om = O.OracleModel(path)
    tokens = [int((1103515245 * i + 12345) % 256) for i in range(201)]
    logits, state = None, None
    ost = om.init_state()
    loss_sum, oloss_sum = torch.tensor([0.0]), torch.tensor([0.0])
    for i in range(len(tokens) - 1):
        logits, state = model.eval(tokens[i], state, state, logits)
        assert isinstance(logits, torch.Tensor) and isinstance(state, torch.Tensor)
        ol, ost = om.eval(tokens[i], ost)
        target = torch.tensor(tokens[i + 1], dtype=torch.long)
        loss_sum += torch.nn.functional.cross_entropy(logits, target, reduction="none").item()
        oloss_sum += torch.nn.functional.cross_entropy(torch.from_numpy(ol), target, reduction="none").item()
    assert np.array_equal(logits.numpy(), ol) and np.array_equal(state.numpy(), ost)
    assert loss_sum.item() == oloss_sum.item()
    # chunked sequence evaluation on torch tensors, output buffers given by the caller
    st_out = torch.zeros(model._state_buffer_element_count, dtype=torch.float32)
    lg_out = torch.zeros(model._logits_buffer_element_count, dtype=torch.float32)
    l2, s2 = model.eval_sequence_in_chunks(tokens[:-1], None, st_out, lg_out, chunk_size=64)
    assert l2 is lg_out and s2 is st_out
    assert np.array_equal(lg_out.numpy(), ol) and np.array_equal(st_out.numpy(), ost)
    model.free()
    om.free()
