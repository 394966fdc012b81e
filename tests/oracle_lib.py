"""ctypes binding of the CPU oracle (oracle/librwkv_oracle.so). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "librwkv_oracle.so")

TYPE_IDS = {"FP32": 0, "FP16": 1, "Q4_0": 2, "Q4_1": 3, "Q5_0": 7, "Q5_1": 8, "Q8_0": 9}
TYPE_SIZE = {0: 4, 1: 2, 2: 18, 3: 20, 7: 22, 8: 24, 9: 34}
BLOCK_SIZE = {0: 1, 1: 1, 2: 32, 3: 32, 7: 32, 8: 32, 9: 32}


def build_oracle(force: bool = False) -> str:
    src = [os.path.join(ORACLE_DIR, n) for n in ("rwkv_oracle.c", "rwkv_oracle_fast.c", "rwkv_oracle.h", "Makefile")]
    stale = (not os.path.exists(ORACLE_SO)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "librwkv_oracle.so"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build_oracle())
        P = ctypes.c_void_p
        L.orc_load.restype = P
        L.orc_load.argtypes = [ctypes.c_char_p]
        L.orc_free.argtypes = [P]
        L.orc_info.argtypes = [P, P]
        L.orc_state_len.restype = ctypes.c_size_t
        L.orc_state_len.argtypes = [P]
        L.orc_bytes_per_token.restype = ctypes.c_uint64
        L.orc_bytes_per_token.argtypes = [P]
        L.orc_init_state.argtypes = [P, P]
        L.orc_set_threads.argtypes = [ctypes.c_int]
        # (tests: at most 16 threads -- the models are small, and a shared host with more threads than free cores crawls; bench.py sets its own count)
        try:
            L.orc_set_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
        except Exception:
            pass
        L.orc_set_fast.argtypes = [ctypes.c_int]
        L.orc_fast_uses_vnni.restype = ctypes.c_int
        L.orc_eval.restype = ctypes.c_int
        L.orc_eval.argtypes = [P, ctypes.c_uint32, P, P, P]
        L.orc_eval_sequence.restype = ctypes.c_int
        L.orc_eval_sequence.argtypes = [P, P, ctypes.c_size_t, P, P, P]
        L.orc_eval_stage.restype = ctypes.c_int
        L.orc_eval_stage.argtypes = [P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, P, P, P]
        L.orc_quantize_row.argtypes = [ctypes.c_int, P, P, ctypes.c_int64]
        L.orc_dequantize_row.argtypes = [ctypes.c_int, P, P, ctypes.c_int64]
        L.orc_quantize_act.argtypes = [P, ctypes.c_int64, P, P, P]
        L.orc_mul_mat.argtypes = [ctypes.c_int, P, ctypes.c_int64, ctypes.c_int64, P, ctypes.c_int64, P]
        L.orc_quantize_file.restype = ctypes.c_int
        L.orc_quantize_file.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        L.orc_unary.argtypes = [ctypes.c_int, P, P, ctypes.c_int64]
        L.orc_f32_to_f16.restype = ctypes.c_uint16
        L.orc_f32_to_f16.argtypes = [ctypes.c_float]
        L.orc_f16_to_f32.restype = ctypes.c_float
        L.orc_f16_to_f32.argtypes = [ctypes.c_uint16]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class OracleModel:
    """Mirror of the reference's RWKVModel surface (python/rwkv_cpp/rwkv_cpp_model.py) on the CPU oracle."""

    def __init__(self, path: str):
        self._l = lib()
        self._m = self._l.orc_load(path.encode())
        if not self._m:
            raise ValueError(f"oracle failed to load {path}")
        info = np.zeros(10, dtype=np.int64)
        self._l.orc_info(self._m, _p(info))
        (self.arch_major, self.arch_minor, self.n_vocab, self.n_embed, self.n_layer,
         self.head_count, self.head_size, self.data_type, self.version, self.ffn_size) = [int(v) for v in info]
        self.state_len = int(self._l.orc_state_len(self._m))
        self.bytes_per_token = int(self._l.orc_bytes_per_token(self._m))

    def init_state(self) -> np.ndarray:
        s = np.empty(self.state_len, dtype=np.float32)
        self._l.orc_init_state(self._m, _p(s))
        return s

    def eval(self, token, state_in, want_logits=True):
        state_out = np.empty(self.state_len, dtype=np.float32)
        logits = np.empty(self.n_vocab, dtype=np.float32) if want_logits else None
        rc = self._l.orc_eval(self._m, int(token), _p(state_in), _p(state_out), _p(logits))
        if rc != 0:
            raise ValueError("orc_eval failed")
        return logits, state_out

    def eval_sequence(self, tokens, state_in, want_logits=True):
        toks = np.ascontiguousarray(np.asarray(tokens, dtype=np.uint32))
        state_out = np.empty(self.state_len, dtype=np.float32)
        logits = np.empty(self.n_vocab, dtype=np.float32) if want_logits else None
        rc = self._l.orc_eval_sequence(self._m, _p(toks), len(toks), _p(state_in), _p(state_out), _p(logits))
        if rc != 0:
            raise ValueError("orc_eval_sequence failed")
        return logits, state_out

    def eval_stage(self, layer_begin, layer_end, token, xio, state, want_logits):
        """In-place stage step on `state`; returns logits (last stage, when wanted) or None. xio: np.float32 hand-off buffer."""
        logits = np.empty(self.n_vocab, dtype=np.float32) if (want_logits and layer_end == self.n_layer) else None
        rc = self._l.orc_eval_stage(self._m, layer_begin, layer_end, int(token), _p(xio), _p(state), _p(logits))
        if rc != 0:
            raise ValueError("orc_eval_stage failed")
        return logits

    def free(self):
        if self._m:
            self._l.orc_free(self._m)
            self._m = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def quantize_row(type_id: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.size
    out = np.empty(TYPE_SIZE[type_id] * n // BLOCK_SIZE[type_id], dtype=np.uint8)
    lib().orc_quantize_row(type_id, _p(x), _p(out), n)
    return out


def dequantize_row(type_id: int, q: np.ndarray, n: int) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.uint8)
    out = np.empty(n, dtype=np.float32)
    lib().orc_dequantize_row(type_id, _p(q), _p(out), n)
    return out


def quantize_act(x: np.ndarray):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.size
    q = np.empty(n, dtype=np.int8)
    d = np.empty(n // 32, dtype=np.float32)
    s = np.empty(n // 32, dtype=np.float32)
    lib().orc_quantize_act(_p(x), n, _p(q), _p(d), _p(s))
    return q, d, s


def mul_mat(type_id: int, w_bytes: np.ndarray, K: int, N: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, K)
    T = x.shape[0]
    w_bytes = np.ascontiguousarray(w_bytes).view(np.uint8)
    y = np.empty((T, N), dtype=np.float32)
    lib().orc_mul_mat(type_id, _p(w_bytes), K, N, _p(x), T, _p(y))
    return y


def unary(op: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().orc_unary(op, _p(x), _p(y), x.size)
    return y


def quantize_file(src: str, dst: str, fmt: str) -> None:
    rc = lib().orc_quantize_file(src.encode(), dst.encode(), fmt.encode())
    if rc != 0:
        raise ValueError(f"orc_quantize_file failed rc={rc}")
