"""Shared helpers of the GPU parity tests: the product library through its C ABI + the CPU oracle as checker."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from rwkv_cpp_amd import synth  # noqa: E402,F401

_lib = None
_hooks = None


def _torch_first():
    # When torch is going to touch the GPU in this process it must initialise HIP BEFORE librwkv.so is loaded: torch wheels
    # bundle their own libamdhip64 and a second runtime loaded afterwards finds "No HIP GPUs" (same soname, first one wins).
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def library():
    """The PRODUCT library (rwkv.cpp_amd/lib/librwkv.so): what every model-level test runs on."""
    global _lib
    if _lib is None:
        _torch_first()
        pkg.build_library()
        _lib = pkg.load_rwkv_shared_library()
    return _lib


def hooks_library():
    """lib/librwkv_testhooks.so: the same objects + the kernel-level test entry points of include/rwkv_testhooks.h (one projection through
    the production kernels, the activation quantiser, the scalar routines, the persistent kernel's tag preset). Tests only."""
    global _hooks
    if _hooks is None:
        _torch_first()
        pkg.build_library()
        _hooks = pkg.RWKVSharedLibrary(pkg.HOOKS_LIB_PATH)
        L = _hooks.library
        L.rwkv_mi_test_mul_mat.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.rwkv_mi_test_mul_mat.restype = ctypes.c_bool
        L.rwkv_mi_test_quantize_act.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 4
        L.rwkv_mi_test_quantize_act.restype = ctypes.c_bool
        L.rwkv_mi_test_unary.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        L.rwkv_mi_test_unary.restype = ctypes.c_bool
    return _hooks


def model(path, hooks=False, **kw):
    return pkg.RWKVModel(hooks_library() if hooks else library(), path, thread_count=2, gpu_layer_count=0, **kw)


def gpu_mul_mat(type_id, w_bytes, K, N, x):
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, K)
    T = x.shape[0]
    w = np.ascontiguousarray(w_bytes).view(np.uint8)
    y = np.empty((T, N), dtype=np.float32)
    ok = hooks_library().library.rwkv_mi_test_mul_mat(type_id, w.ctypes.data, K, N, x.ctypes.data, T, y.ctypes.data)
    assert ok, "rwkv_mi_test_mul_mat failed"
    return y


def gpu_quantize_act(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.size
    q = np.empty(n, dtype=np.int8)
    d = np.empty(n // 32, dtype=np.float32)
    s = np.empty(n // 32, dtype=np.float32)
    isum = np.empty(n // 32, dtype=np.int32)
    assert hooks_library().library.rwkv_mi_test_quantize_act(x.ctypes.data, n, q.ctypes.data, d.ctypes.data, s.ctypes.data, isum.ctypes.data)
    return q, d, s, isum


def gpu_unary(op, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    assert hooks_library().library.rwkv_mi_test_unary(op, x.ctypes.data, y.ctypes.data, x.size)
    return y


def run_prompt(m, tokens, sequence=False):
    if sequence:
        return m.eval_sequence(tokens, None)
    state, logits = None, None
    for t in tokens:
        logits, state = m.eval(t, state)
    return logits, state
