"""Host-side mirror of the reference's Python binding on top of the C ABI of librwkv.so.

Mirrors python/rwkv_cpp/rwkv_cpp_shared_library.py (RWKVSharedLibrary: one method per rwkv.h entry point, same
names and argument meaning, ValueError on failure) and python/rwkv_cpp/rwkv_cpp_model.py (RWKVModel: eval /
eval_sequence / eval_sequence_in_chunks returning (logits, state)) of RWKV/rwkv.cpp @ 2025-02-19, plus the rwkv_mi_*
extensions. The reference's own wrapper also works unchanged against this library.

There is no fallback: if librwkv.so is missing or no MI355X is visible, loading / init raises.
"""
import ctypes
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# RWKV_LIB_DIR selects an alternative in-tree build directory (A/B variants built with `make LIBDIR=... OBJDIR=... EXTRA=-D...`)
LIB_PATH = os.path.join(PKG_DIR, os.environ.get("RWKV_LIB_DIR", "lib"), "librwkv.so")
# tests/ only: the same objects + csrc/testhooks.cpp (include/rwkv_testhooks.h); the product library does not export test entry points
HOOKS_LIB_PATH = os.path.join(PKG_DIR, os.environ.get("RWKV_LIB_DIR", "lib"), "librwkv_testhooks.so")

QUANTIZED_FORMAT_NAMES = ("Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0")
P_FLOAT = ctypes.POINTER(ctypes.c_float)
P_UINT32 = ctypes.POINTER(ctypes.c_uint32)


def build_library(force: bool = False) -> str:
    """Compiles every HIP/C++ source for gfx950 into rwkv.cpp_amd/lib/librwkv.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", PKG_DIR, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


class RWKVContext:
    def __init__(self, ptr) -> None:
        self.ptr = ptr


class RWKVSharedLibrary:
    """ctypes declarations of every symbol include/rwkv.h and include/rwkv_mi355x.h export."""

    def __init__(self, shared_library_path: str = LIB_PATH) -> None:
        if not os.path.isfile(shared_library_path):
            raise FileNotFoundError(f"{shared_library_path} not found: build it with __graft_entry__.build() (no CPU fallback exists)")
        self.library = L = ctypes.cdll.LoadLibrary(shared_library_path)
        c_ctx = ctypes.c_void_p

        L.rwkv_init_from_file.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32]
        L.rwkv_init_from_file.restype = c_ctx
        L.rwkv_clone_context.argtypes = [c_ctx, ctypes.c_uint32]
        L.rwkv_clone_context.restype = c_ctx
        L.rwkv_eval.argtypes = [c_ctx, ctypes.c_int32, P_FLOAT, P_FLOAT, P_FLOAT]
        L.rwkv_eval.restype = ctypes.c_bool
        L.rwkv_eval_sequence.argtypes = [c_ctx, P_UINT32, ctypes.c_size_t, P_FLOAT, P_FLOAT, P_FLOAT]
        L.rwkv_eval_sequence.restype = ctypes.c_bool
        L.rwkv_eval_sequence_in_chunks.argtypes = [c_ctx, P_UINT32, ctypes.c_size_t, ctypes.c_size_t, P_FLOAT, P_FLOAT, P_FLOAT]
        L.rwkv_eval_sequence_in_chunks.restype = ctypes.c_bool
        for name in ("rwkv_get_n_vocab", "rwkv_get_n_embed", "rwkv_get_n_layer", "rwkv_get_state_len", "rwkv_get_logits_len"):
            getattr(L, name).argtypes = [c_ctx]
            getattr(L, name).restype = ctypes.c_size_t
        for name in ("rwkv_get_state_buffer_element_count", "rwkv_get_logits_buffer_element_count"):
            getattr(L, name).argtypes = [c_ctx]
            getattr(L, name).restype = ctypes.c_uint32
        L.rwkv_init_state.argtypes = [c_ctx, P_FLOAT]
        L.rwkv_init_state.restype = None
        L.rwkv_free.argtypes = [c_ctx]
        L.rwkv_free.restype = None
        L.rwkv_quantize_model_file.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        L.rwkv_quantize_model_file.restype = ctypes.c_bool
        L.rwkv_get_system_info_string.argtypes = []
        L.rwkv_get_system_info_string.restype = ctypes.c_char_p
        L.rwkv_set_print_errors.argtypes = [c_ctx, ctypes.c_bool]
        L.rwkv_set_print_errors.restype = None
        L.rwkv_get_print_errors.argtypes = [c_ctx]
        L.rwkv_get_print_errors.restype = ctypes.c_bool
        L.rwkv_get_last_error.argtypes = [c_ctx]
        L.rwkv_get_last_error.restype = ctypes.c_int
        # extensions
        L.rwkv_mi_state_load.argtypes = [c_ctx, P_FLOAT]
        L.rwkv_mi_state_load.restype = ctypes.c_bool
        L.rwkv_mi_state_store.argtypes = [c_ctx, P_FLOAT]
        L.rwkv_mi_state_store.restype = ctypes.c_bool
        L.rwkv_mi_eval_resident.argtypes = [c_ctx, P_UINT32, ctypes.c_size_t, P_FLOAT]
        L.rwkv_mi_eval_resident.restype = ctypes.c_bool
        L.rwkv_mi_decode_greedy.argtypes = [c_ctx, ctypes.c_uint32, ctypes.c_size_t, P_UINT32, P_FLOAT]
        L.rwkv_mi_decode_greedy.restype = ctypes.c_bool
        L.rwkv_mi_profile_decode.argtypes = [c_ctx, ctypes.c_uint32, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double)]
        L.rwkv_mi_profile_decode.restype = ctypes.c_bool
        L.rwkv_mi_profile_prefill.argtypes = [c_ctx, P_UINT32, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double)]
        L.rwkv_mi_profile_prefill.restype = ctypes.c_bool
        L.rwkv_mi_bytes_per_token.argtypes = [c_ctx]
        L.rwkv_mi_bytes_per_token.restype = ctypes.c_uint64
        L.rwkv_mi_weight_bytes.argtypes = [c_ctx]
        L.rwkv_mi_weight_bytes.restype = ctypes.c_uint64
        L.rwkv_mi_prefill_flops.argtypes = [c_ctx, ctypes.c_size_t]
        L.rwkv_mi_prefill_flops.restype = ctypes.c_uint64
        L.rwkv_mi_get_arch.argtypes = [c_ctx, P_UINT32, P_UINT32, P_UINT32, P_UINT32]
        L.rwkv_mi_get_arch.restype = None
        L.rwkv_mi_init_stage.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.rwkv_mi_init_stage.restype = c_ctx
        L.rwkv_mi_set_stream.argtypes = [c_ctx, ctypes.c_void_p]
        L.rwkv_mi_set_stream.restype = ctypes.c_bool
        L.rwkv_mi_handoff_len.argtypes = [c_ctx]
        L.rwkv_mi_handoff_len.restype = ctypes.c_size_t
        L.rwkv_mi_stage_step.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.rwkv_mi_stage_step.restype = ctypes.c_bool
        L.rwkv_mi_logits_store.argtypes = [c_ctx, P_FLOAT]
        L.rwkv_mi_logits_store.restype = ctypes.c_bool
        L.rwkv_mi_logits_device_ptr.argtypes = [c_ctx]
        L.rwkv_mi_logits_device_ptr.restype = ctypes.c_void_p
        L.rwkv_mi_set_graph_enabled.argtypes = [c_ctx, ctypes.c_bool]
        L.rwkv_mi_set_graph_enabled.restype = None
        L.rwkv_mi_decode_path.argtypes = [c_ctx]
        L.rwkv_mi_decode_path.restype = ctypes.c_int
        L.rwkv_mi_persist_kind.argtypes = [c_ctx]
        L.rwkv_mi_persist_kind.restype = ctypes.c_int
        if hasattr(L, "rwkv_mi_persist_info"):   # (absent from older A/B builds loaded through RWKV_LIB_DIR)
            L.rwkv_mi_persist_info.argtypes = [c_ctx]
            L.rwkv_mi_persist_info.restype = ctypes.c_char_p
        L.rwkv_mi_load_stats.argtypes = [c_ctx, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
        L.rwkv_mi_load_stats.restype = None
        L.rwkv_mi_decode_healthy.argtypes = [c_ctx]
        L.rwkv_mi_decode_healthy.restype = ctypes.c_bool
        L.rwkv_mi_sample.argtypes = [c_ctx, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, P_UINT32]
        L.rwkv_mi_sample.restype = ctypes.c_bool
        L.rwkv_mi_decode_sample.argtypes = [c_ctx, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, P_UINT32, P_FLOAT]
        L.rwkv_mi_decode_sample.restype = ctypes.c_bool
        L.rwkv_mi_decode_generation.argtypes = [c_ctx]
        L.rwkv_mi_decode_generation.restype = ctypes.c_uint32
        if hasattr(L, "rwkv_mi_test_set_tag"):   # (librwkv_testhooks.so only)
            L.rwkv_mi_test_set_tag.argtypes = [c_ctx, ctypes.c_uint32]
            L.rwkv_mi_test_set_tag.restype = ctypes.c_bool
        # the C++ decode loop of a pipeline (runner.cpp)
        L.rwkv_mi_decode_greedy_streams.argtypes = [ctypes.POINTER(c_ctx), ctypes.c_size_t, P_UINT32, ctypes.c_size_t, P_UINT32, P_FLOAT]
        L.rwkv_mi_decode_greedy_streams.restype = ctypes.c_bool
        L.rwkv_mi_comm_available.argtypes = []
        L.rwkv_mi_comm_available.restype = ctypes.c_bool
        L.rwkv_mi_comm_unique_id.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.rwkv_mi_comm_unique_id.restype = ctypes.c_bool
        L.rwkv_mi_comm_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.rwkv_mi_comm_init.restype = ctypes.c_void_p
        if hasattr(L, "rwkv_mi_comm_init_ipc"):   # (absent from older A/B builds loaded through RWKV_LIB_DIR)
            L.rwkv_mi_comm_init_ipc.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
            L.rwkv_mi_comm_init_ipc.restype = ctypes.c_void_p
        L.rwkv_mi_comm_free.argtypes = [ctypes.c_void_p]
        L.rwkv_mi_comm_free.restype = None
        L.rwkv_mi_stage_run.argtypes = [ctypes.POINTER(c_ctx), ctypes.c_size_t, P_UINT32, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        P_UINT32, P_FLOAT]
        L.rwkv_mi_stage_run.restype = ctypes.c_bool

    # --- rwkv.h ---------------------------------------------------------------------------------------------

    def rwkv_init_from_file(self, model_file_path: str, thread_count: int, offload_layers: int) -> RWKVContext:
        ptr = self.library.rwkv_init_from_file(model_file_path.encode("utf-8"), ctypes.c_uint32(thread_count), ctypes.c_uint32(offload_layers))
        if not ptr:
            raise ValueError("rwkv_init_from_file failed, check stderr")
        return RWKVContext(ptr)

    def rwkv_clone_context(self, ctx: RWKVContext, thread_count: int) -> RWKVContext:
        ptr = self.library.rwkv_clone_context(ctx.ptr, ctypes.c_uint32(thread_count))
        if not ptr:
            raise ValueError("rwkv_clone_context failed, check stderr")
        return RWKVContext(ptr)

    def rwkv_eval(self, ctx: RWKVContext, token: int, state_in_address: Optional[int], state_out_address: int, logits_out_address: int) -> None:
        if not self.library.rwkv_eval(ctx.ptr, ctypes.c_int32(token), ctypes.cast(state_in_address or 0, P_FLOAT),
                                      ctypes.cast(state_out_address or 0, P_FLOAT), ctypes.cast(logits_out_address or 0, P_FLOAT)):
            raise ValueError("rwkv_eval failed, check stderr")

    def rwkv_eval_sequence(self, ctx: RWKVContext, tokens: List[int], state_in_address: Optional[int], state_out_address: int, logits_out_address: int) -> None:
        arr = (ctypes.c_uint32 * len(tokens))(*tokens)
        if not self.library.rwkv_eval_sequence(ctx.ptr, arr, ctypes.c_size_t(len(tokens)), ctypes.cast(state_in_address or 0, P_FLOAT),
                                               ctypes.cast(state_out_address or 0, P_FLOAT), ctypes.cast(logits_out_address or 0, P_FLOAT)):
            raise ValueError("rwkv_eval_sequence failed, check stderr")

    def rwkv_eval_sequence_in_chunks(self, ctx: RWKVContext, tokens: List[int], chunk_size: int, state_in_address: Optional[int],
                                     state_out_address: int, logits_out_address: int) -> None:
        arr = (ctypes.c_uint32 * len(tokens))(*tokens)
        if not self.library.rwkv_eval_sequence_in_chunks(ctx.ptr, arr, ctypes.c_size_t(len(tokens)), ctypes.c_size_t(chunk_size),
                                                         ctypes.cast(state_in_address or 0, P_FLOAT), ctypes.cast(state_out_address or 0, P_FLOAT),
                                                         ctypes.cast(logits_out_address or 0, P_FLOAT)):
            raise ValueError("rwkv_eval_sequence_in_chunks failed, check stderr")

    def rwkv_get_n_vocab(self, ctx: RWKVContext) -> int:
        return self.library.rwkv_get_n_vocab(ctx.ptr)

    def rwkv_get_n_embed(self, ctx: RWKVContext) -> int:
        return self.library.rwkv_get_n_embed(ctx.ptr)

    def rwkv_get_n_layer(self, ctx: RWKVContext) -> int:
        return self.library.rwkv_get_n_layer(ctx.ptr)

    def rwkv_get_state_buffer_element_count(self, ctx: RWKVContext) -> int:
        return self.library.rwkv_get_state_buffer_element_count(ctx.ptr)

    def rwkv_get_logits_buffer_element_count(self, ctx: RWKVContext) -> int:
        return self.library.rwkv_get_logits_buffer_element_count(ctx.ptr)

    def rwkv_init_state(self, ctx: RWKVContext, state_address: int) -> None:
        self.library.rwkv_init_state(ctx.ptr, ctypes.cast(state_address, P_FLOAT))

    def rwkv_free(self, ctx: RWKVContext) -> None:
        self.library.rwkv_free(ctx.ptr)
        ctx.ptr = ctypes.cast(0, ctypes.c_void_p)

    def rwkv_quantize_model_file(self, model_file_path_in: str, model_file_path_out: str, format_name: str) -> None:
        if format_name not in QUANTIZED_FORMAT_NAMES:
            raise ValueError(f"Unknown format name {format_name}, use one of {QUANTIZED_FORMAT_NAMES}")
        if not self.library.rwkv_quantize_model_file(model_file_path_in.encode("utf-8"), model_file_path_out.encode("utf-8"), format_name.encode("utf-8")):
            raise ValueError("rwkv_quantize_model_file failed, check stderr")

    def rwkv_get_system_info_string(self) -> str:
        return self.library.rwkv_get_system_info_string().decode("utf-8")

    def rwkv_set_print_errors(self, ctx: Optional[RWKVContext], print_errors: bool) -> None:
        self.library.rwkv_set_print_errors(ctx.ptr if ctx else None, print_errors)

    def rwkv_get_last_error(self, ctx: Optional[RWKVContext]) -> int:
        return int(self.library.rwkv_get_last_error(ctx.ptr if ctx else None))


def load_rwkv_shared_library() -> RWKVSharedLibrary:
    return RWKVSharedLibrary(LIB_PATH)


def _ptr(a: Optional[np.ndarray]) -> int:
    return 0 if a is None else a.ctypes.data


class RWKVModel:
    """numpy flavour of the reference's RWKVModel (python/rwkv_cpp/rwkv_cpp_model.py:22-364)."""

    def __init__(self, shared_library: RWKVSharedLibrary, model_path: str, thread_count: int = 1, gpu_layer_count: int = 0, **kwargs) -> None:
        if "gpu_layers_count" in kwargs:
            gpu_layer_count = kwargs["gpu_layers_count"]
        if not os.path.isfile(model_path):
            raise ValueError(f"{model_path} is not a file")
        if thread_count <= 0:
            raise ValueError("Thread count must be > 0")
        self._library = shared_library
        self._ctx = shared_library.rwkv_init_from_file(model_path, thread_count, gpu_layer_count)
        self._state_buffer_element_count = shared_library.rwkv_get_state_buffer_element_count(self._ctx)
        self._logits_buffer_element_count = shared_library.rwkv_get_logits_buffer_element_count(self._ctx)
        self._valid = True

    @property
    def n_vocab(self) -> int:
        return self._library.rwkv_get_n_vocab(self._ctx)

    @property
    def n_embed(self) -> int:
        return self._library.rwkv_get_n_embed(self._ctx)

    @property
    def n_layer(self) -> int:
        return self._library.rwkv_get_n_layer(self._ctx)

    @property
    def state_len(self) -> int:
        return self._state_buffer_element_count

    def arch(self) -> Tuple[int, int, int, int]:
        v = [ctypes.c_uint32() for _ in range(4)]
        self._library.library.rwkv_mi_get_arch(self._ctx.ptr, *[ctypes.byref(x) for x in v])
        return tuple(int(x.value) for x in v)

    def _check(self, a: Optional[np.ndarray], name: str, size: int) -> None:
        if a is None:
            return
        if a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"] or a.shape != (size,):
            raise ValueError(f"{name} must be a contiguous float32 array of shape ({size},)")

    def _outputs(self, state_out, logits_out):
        self._check(state_out, "state_out", self._state_buffer_element_count)
        self._check(logits_out, "logits_out", self._logits_buffer_element_count)
        if state_out is None:
            state_out = np.zeros(self._state_buffer_element_count, dtype=np.float32)
        if logits_out is None:
            logits_out = np.zeros(self._logits_buffer_element_count, dtype=np.float32)
        return state_out, logits_out

    def init_state(self) -> np.ndarray:
        s = np.empty(self._state_buffer_element_count, dtype=np.float32)
        self._library.rwkv_init_state(self._ctx, s.ctypes.data)
        return s

    def eval(self, token: int, state_in: Optional[np.ndarray], state_out: Optional[np.ndarray] = None,
             logits_out: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        if not self._valid:
            raise ValueError("Model was freed")
        self._check(state_in, "state_in", self._state_buffer_element_count)
        state_out, logits_out = self._outputs(state_out, logits_out)
        self._library.rwkv_eval(self._ctx, token, _ptr(state_in), _ptr(state_out), _ptr(logits_out))
        return logits_out, state_out

    def eval_sequence(self, tokens: List[int], state_in: Optional[np.ndarray], state_out: Optional[np.ndarray] = None,
                      logits_out: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        if not self._valid:
            raise ValueError("Model was freed")
        self._check(state_in, "state_in", self._state_buffer_element_count)
        state_out, logits_out = self._outputs(state_out, logits_out)
        self._library.rwkv_eval_sequence(self._ctx, list(tokens), _ptr(state_in), _ptr(state_out), _ptr(logits_out))
        return logits_out, state_out

    def eval_sequence_in_chunks(self, tokens: List[int], state_in: Optional[np.ndarray], state_out: Optional[np.ndarray] = None,
                                logits_out: Optional[np.ndarray] = None, chunk_size: int = 16) -> Tuple[np.ndarray, np.ndarray]:
        if not self._valid:
            raise ValueError("Model was freed")
        self._check(state_in, "state_in", self._state_buffer_element_count)
        state_out, logits_out = self._outputs(state_out, logits_out)
        self._library.rwkv_eval_sequence_in_chunks(self._ctx, list(tokens), chunk_size, _ptr(state_in), _ptr(state_out), _ptr(logits_out))
        return logits_out, state_out

    # --- rwkv_mi_* extensions: state resident in HBM ------------------------------------------------------

    def state_load(self, state_in: Optional[np.ndarray]) -> None:
        self._check(state_in, "state_in", self._state_buffer_element_count)
        if not self._library.library.rwkv_mi_state_load(self._ctx.ptr, ctypes.cast(_ptr(state_in), P_FLOAT)):
            raise ValueError("rwkv_mi_state_load failed")

    def state_store(self) -> np.ndarray:
        s = np.empty(self._state_buffer_element_count, dtype=np.float32)
        if not self._library.library.rwkv_mi_state_store(self._ctx.ptr, ctypes.cast(s.ctypes.data, P_FLOAT)):
            raise ValueError("rwkv_mi_state_store failed")
        return s

    def logits_store(self) -> np.ndarray:
        """Logits of the last step that produced any (device -> host; synchronises the context's stream)."""
        lg = np.empty(self._logits_buffer_element_count, dtype=np.float32)
        if not self._library.library.rwkv_mi_logits_store(self._ctx.ptr, ctypes.cast(_ptr(lg), P_FLOAT)):
            raise ValueError("rwkv_mi_logits_store failed")
        return lg

    def eval_resident(self, tokens: List[int], want_logits: bool = True) -> Optional[np.ndarray]:
        arr = (ctypes.c_uint32 * len(tokens))(*tokens)
        logits = np.empty(self._logits_buffer_element_count, dtype=np.float32) if want_logits else None
        if not self._library.library.rwkv_mi_eval_resident(self._ctx.ptr, arr, len(tokens), ctypes.cast(_ptr(logits), P_FLOAT)):
            raise ValueError("rwkv_mi_eval_resident failed")
        return logits

    def decode_greedy(self, first_token: int, n_tokens: int) -> Tuple[np.ndarray, float]:
        out = np.empty(n_tokens, dtype=np.uint32)
        ms = ctypes.c_float(0.0)
        if not self._library.library.rwkv_mi_decode_greedy(self._ctx.ptr, first_token, n_tokens, ctypes.cast(out.ctypes.data, P_UINT32), ctypes.byref(ms)):
            raise ValueError("rwkv_mi_decode_greedy failed")
        return out, float(ms.value)

    @staticmethod
    def decode_greedy_streams(models: List["RWKVModel"], first_tokens: List[int], n_tokens: int) -> Tuple[np.ndarray, float]:
        """Greedy decode of several resident-state contexts (a model and its clones; RWKV_MI_DEVICES chains included) interleaved by the
        library's C++ loop: tokens [n_streams][n_tokens], wall milliseconds."""
        n = len(models)
        L = models[0]._library.library
        arr = (ctypes.c_void_p * n)(*[m._ctx.ptr for m in models])
        first = (ctypes.c_uint32 * n)(*first_tokens)
        out = np.empty((n, n_tokens), dtype=np.uint32)
        ms = ctypes.c_float(0.0)
        if not L.rwkv_mi_decode_greedy_streams(arr, n, first, n_tokens, ctypes.cast(out.ctypes.data, P_UINT32), ctypes.byref(ms)):
            raise ValueError("rwkv_mi_decode_greedy_streams failed")
        return out, float(ms.value)

    def sample(self, temperature: float = 1.0, top_p: float = 0.8, u: float = -1.0, seed: int = 0) -> int:
        """Samples from the logits of the last evaluation on the device (mirror of the reference's sampling.sample_logits)."""
        tok = ctypes.c_uint32(0)
        if not self._library.library.rwkv_mi_sample(self._ctx.ptr, temperature, top_p, u, seed, ctypes.byref(tok)):
            raise ValueError("rwkv_mi_sample failed")
        return int(tok.value)

    def decode_sample(self, first_token: int, n_tokens: int, temperature: float = 1.0, top_p: float = 0.8, seed: int = 0) -> Tuple[np.ndarray, float]:
        out = np.empty(n_tokens, dtype=np.uint32)
        ms = ctypes.c_float(0.0)
        if not self._library.library.rwkv_mi_decode_sample(self._ctx.ptr, first_token, n_tokens, temperature, top_p, seed,
                                                           ctypes.cast(out.ctypes.data, P_UINT32), ctypes.byref(ms)):
            raise ValueError("rwkv_mi_decode_sample failed")
        return out, float(ms.value)

    def profile_decode(self, first_token: int, n_tokens: int) -> dict:
        out = (ctypes.c_double * 4)()
        if not self._library.library.rwkv_mi_profile_decode(self._ctx.ptr, first_token, n_tokens, out):
            raise ValueError("rwkv_mi_profile_decode failed")
        return {"kernel_ms": out[0], "launches": int(out[1]), "bytes": int(out[2]), "wall_ms": out[3]}

    def profile_prefill(self, tokens: List[int]) -> dict:
        arr = (ctypes.c_uint32 * len(tokens))(*tokens)
        out = (ctypes.c_double * 4)()
        if not self._library.library.rwkv_mi_profile_prefill(self._ctx.ptr, arr, len(tokens), out):
            raise ValueError("rwkv_mi_profile_prefill failed")
        return {"kernel_ms": out[0], "launches": int(out[1]), "ops": int(out[2]), "wall_ms": out[3]}

    def bytes_per_token(self) -> int:
        return int(self._library.library.rwkv_mi_bytes_per_token(self._ctx.ptr))

    def prefill_flops(self, n_tokens: int) -> int:
        return int(self._library.library.rwkv_mi_prefill_flops(self._ctx.ptr, n_tokens))

    def set_graph_enabled(self, enabled: bool) -> None:
        self._library.library.rwkv_mi_set_graph_enabled(self._ctx.ptr, enabled)

    def decode_path(self) -> int:
        """0 = per-op kernels, 1 = fused RWKV-6 layer, 2 = persistent whole-stage kernel."""
        return int(self._library.library.rwkv_mi_decode_path(self._ctx.ptr))

    def persist_kind(self) -> int:
        """Persistent kernel behind decode path 2: 2 = LDS-DMA weight ring, 1 = register prefetch, 0 = none."""
        return int(self._library.library.rwkv_mi_persist_kind(self._ctx.ptr))

    def load_stats(self) -> Tuple[float, int]:
        """(seconds, bytes) of the model file's payload on its way to HBM at creation."""
        sec, b = ctypes.c_double(0.0), ctypes.c_uint64(0)
        self._library.library.rwkv_mi_load_stats(self._ctx.ptr, ctypes.byref(sec), ctypes.byref(b))
        return float(sec.value), int(b.value)

    def persist_info(self) -> str:
        """"persist: ring|regs|k47|none; <what decided it>" (rwkv_mi_persist_info)."""
        return self._library.library.rwkv_mi_persist_info(self._ctx.ptr).decode("utf-8")

    def healthy(self) -> bool:
        return bool(self._library.library.rwkv_mi_decode_healthy(self._ctx.ptr))

    def decode_generation(self) -> int:
        """Hand-over generation the persistent kernel's next launch starts from (advances 8 per layer and launch; 0: path 2 is off)."""
        return int(self._library.library.rwkv_mi_decode_generation(self._ctx.ptr))

    def test_set_tag(self, base: int) -> bool:
        """Test hook (a model opened through librwkv_testhooks.so only): presets the persistent kernel's rolling hand-over tag."""
        return bool(self._library.library.rwkv_mi_test_set_tag(self._ctx.ptr, base & 0xFFFFFFFF))

    def clone(self, thread_count: int = 1) -> "RWKVModel":
        other = object.__new__(RWKVModel)
        other._library = self._library
        other._ctx = self._library.rwkv_clone_context(self._ctx, thread_count)
        other._state_buffer_element_count = self._state_buffer_element_count
        other._logits_buffer_element_count = self._logits_buffer_element_count
        other._valid = True
        return other

    def free(self) -> None:
        if not self._valid:
            raise ValueError("Already freed")
        self._valid = False
        self._library.rwkv_free(self._ctx)

    def __del__(self) -> None:
        if hasattr(self, "_valid") and self._valid:
            self.free()
