"""Synthetic rwkv.cpp-format model files (no real checkpoints are available offline).

Writes files with exactly the tensor names / shapes / dtypes a converted + quantised checkpoint has
(names: reference rwkv_model_loading.inc:132-282; stored-tensor conventions: python/convert_pytorch_to_ggml.py:83-135;
which tensors the quantiser touches: rwkv_quantize.inc:1-13,137-140), filled with seeded random values
(SURVEY.md 8d). Used by the parity tests (small sizes, through rwkv_quantize_model_file) and by bench.py
(BASELINE.json configurations, quantised blocks generated directly because 7.6e9 weights do not fit a float pass).
"""
import struct
from dataclasses import dataclass
from typing import Optional

import numpy as np

TYPE_IDS = {"FP32": 0, "FP16": 1, "Q4_0": 2, "Q4_1": 3, "Q5_0": 7, "Q5_1": 8, "Q8_0": 9}
BLOCK_BYTES = {"Q4_0": 18, "Q4_1": 20, "Q5_0": 22, "Q5_1": 24, "Q8_0": 34}
NEVER_QUANT = ("att.v1", "att.v2", "att.g1", "att.g2", "att.a1", "att.a2", "att.w1", "att.w2", "att.r_k")


@dataclass
class ModelSpec:
    arch: str                 # "4", "5.1", "5.2", "6", "7"
    n_layer: int
    n_embed: int
    n_vocab: int
    ffn: int
    head_size: int = 64
    mix_rank: int = 32        # v6 time_maa_w1/w2 rank
    decay_rank: int = 64      # v6 time_decay_w1/w2 rank
    v7_rank_w: int = 64
    v7_rank_a: int = 64
    v7_rank_v: int = 32
    v7_rank_g: int = 128
    name: str = "synthetic"


# BASELINE.json configurations
CONFIGS = {
    "rwkv4-169m": ModelSpec("4", 12, 768, 50277, 3072, name="RWKV-4-Pile-169M"),
    "rwkv6-1b6": ModelSpec("6", 24, 2048, 65536, 7168, 64, 32, 64, name="RWKV-6-World-1.6B"),
    "rwkv7-2b9": ModelSpec("7", 32, 2560, 65536, 10240, 64, v7_rank_w=96, v7_rank_a=96, v7_rank_v=64, v7_rank_g=320, name="RWKV-7-World-2.9B"),
    "rwkv6-7b": ModelSpec("6", 32, 4096, 65536, 14336, 64, 64, 128, name="RWKV-6-World-7B"),
    # small stand-ins with real head geometry for the parity tests
    "test-v4": ModelSpec("4", 2, 256, 512, 1024, name="test-v4"),
    "test-v5.1": ModelSpec("5.1", 2, 256, 512, 896, 64, name="test-v5.1"),
    "test-v5.2": ModelSpec("5.2", 2, 256, 512, 896, 64, name="test-v5.2"),
    "test-v6": ModelSpec("6", 2, 256, 512, 896, 64, 32, 64, name="test-v6"),
    # many thin layers: the layer chain with one stage per GPU of an eight-GPU node (tests: eight stages on one device)
    "chain-v6-32x256": ModelSpec("6", 32, 256, 512, 896, 64, 32, 64, name="chain-v6-32x256"),
    # geometries the persistent RWKV-6 decode kernel is built for (mega_v6.hip), with few layers and a small vocabulary
    "mega-v6-2048": ModelSpec("6", 3, 2048, 512, 7168, 64, 32, 64, name="mega-v6-2048"),
    "mega-v6-4096": ModelSpec("6", 2, 4096, 512, 14336, 64, 64, 128, name="mega-v6-4096"),
    # ... with vocabularies the ring kernel folds the head projection for (multiples of 4096: 1, 2 and 8 row groups of 16 rows per workgroup)
    "mega-v6-4096-v4k": ModelSpec("6", 2, 4096, 4096, 14336, 64, 64, 128, name="mega-v6-4096-v4k"),
    "mega-v6-2048-v8k": ModelSpec("6", 3, 2048, 8192, 7168, 64, 32, 64, name="mega-v6-2048-v8k"),
    "mega-v6-2048-v32k": ModelSpec("6", 2, 2048, 32768, 7168, 64, 32, 64, name="mega-v6-2048-v32k"),
    "test-v7": ModelSpec("7", 3, 256, 512, 1024, 64, v7_rank_w=64, v7_rank_a=64, v7_rank_v=32, v7_rank_g=96, name="test-v7"),
    # two-layer slices of the BASELINE configurations C4 / C2 at their REAL row lengths, ranks and vocabulary (tests/test_gpu_real_geometry.py):
    # RWKV-7-World-2.9B (D 2560: 80 blocks per row, ranks 96 / 96 / 64 / 320) and RWKV-4-Pile-169M (D 768, V 50277: a multiple of nothing)
    "slice-v7-2560": ModelSpec("7", 2, 2560, 4096, 10240, 64, v7_rank_w=96, v7_rank_a=96, v7_rank_v=64, v7_rank_g=320, name="slice-v7-2560"),
    "slice-v4-768": ModelSpec("4", 2, 768, 50277, 3072, name="slice-v4-768"),
    # the World vocabulary on the ring kernel's folded head: 65536 rows = 16 sixteen-row groups per workgroup = three passes of the six consumers
    "mega-v6-2048-v64k": ModelSpec("6", 1, 2048, 65536, 7168, 64, 32, 64, name="mega-v6-2048-v64k"),
    # RWKV-6-World-3B geometry (D 2560, F 8960, mix rank 32, decay rank 64)
    "mega-v6-2560": ModelSpec("6", 2, 2560, 4096, 8960, 64, 32, 64, name="mega-v6-2560"),
}


def _tensor_list(s: ModelSpec):
    """[(key, ggml-order dims, kind)]; kind: 'mat' (2-D matrix), 'ln_w', 'ln_b', 'mix', or a named special."""
    D, F, V, S = s.n_embed, s.ffn, s.n_vocab, s.head_size
    H = D // S if s.arch != "4" else 0
    out = [("emb.weight", (D, V), "emb")]
    for i in range(s.n_layer):
        p = f"blocks.{i}."
        out += [(p + "ln1.weight", (D,), "ln_w"), (p + "ln1.bias", (D,), "ln_b"), (p + "ln2.weight", (D,), "ln_w"), (p + "ln2.bias", (D,), "ln_b")]
        if i == 0:
            out += [(p + "ln0.weight", (D,), "ln_w"), (p + "ln0.bias", (D,), "ln_b")]
        if s.arch == "4":
            out += [(p + "att.time_decay", (D,), "v4_decay"), (p + "att.time_first", (D,), "v4_first")]
            out += [(p + f"att.time_mix_{c}", (D,), "mix") for c in "kvr"]
        elif s.arch in ("5.1", "5.2"):
            if s.arch == "5.1":
                out += [(p + "att.time_decay", (1, 1, H), "v5_decay"), (p + "att.time_first", (1, 1, H), "v5_first")]
            else:
                out += [(p + "att.time_decay", (1, S, H), "v5_decay"), (p + "att.time_faaaa", (1, S, H), "small")]
            out += [(p + f"att.time_mix_{c}", (D,), "mix") for c in ("kvr" if s.arch == "5.1" else "kvrg")]
            out += [(p + "att.ln_x.weight", (D,), "ln_w"), (p + "att.ln_x.bias", (D,), "ln_b")]
        elif s.arch == "6":
            out += [(p + f"att.time_maa_{c}", (D,), "mix") for c in "xwkvrg"]
            out += [(p + "att.time_maa_w1", (D, 5 * s.mix_rank), "mat_f32src"), (p + "att.time_maa_w2", (s.mix_rank, D, 5), "w2")]
            out += [(p + "att.time_decay", (1, S, H), "v6_decay"), (p + "att.time_decay_w1", (D, s.decay_rank), "mat_f32src"),
                    (p + "att.time_decay_w2", (s.decay_rank, D), "mat_f32src"), (p + "att.time_faaaa", (1, S, H), "small")]
            out += [(p + "att.ln_x.weight", (D,), "ln_w"), (p + "att.ln_x.bias", (D,), "ln_b")]
        elif s.arch == "7":
            out += [(p + "att.x_rwkvag", (D, 1, 6), "mix")]
            out += [(p + "att.w1", (D, s.v7_rank_w), "mat"), (p + "att.w2", (s.v7_rank_w, D), "mat"), (p + "att.w0", (D, 1, 1), "v7_w0")]
            out += [(p + "att.a1", (D, s.v7_rank_a), "mat"), (p + "att.a2", (s.v7_rank_a, D), "mat"), (p + "att.a0", (D, 1, 1), "small")]
            if i != 0:
                out += [(p + "att.v1", (D, s.v7_rank_v), "mat"), (p + "att.v2", (s.v7_rank_v, D), "mat"), (p + "att.v0", (D, 1, 1), "small")]
            out += [(p + "att.g1", (D, s.v7_rank_g), "mat"), (p + "att.g2", (s.v7_rank_g, D), "mat")]
            out += [(p + "att.k_k", (D, 1, 1), "mix"), (p + "att.k_a", (D, 1, 1), "mix"), (p + "att.r_k", (S, H), "small")]
            out += [(p + "att.ln_x.weight", (D,), "ln_w"), (p + "att.ln_x.bias", (D,), "ln_b")]
        out += [(p + f"att.{n}.weight", (D, D), "mat") for n in ("key", "value", "receptance", "output")]
        if s.arch in ("5.2", "6"):
            out += [(p + "att.gate.weight", (D, D), "mat")]
        if s.arch in ("4", "5.1", "5.2"):
            out += [(p + "ffn.time_mix_k", (D,), "mix"), (p + "ffn.time_mix_r", (D,), "mix")]
        elif s.arch == "6":
            out += [(p + "ffn.time_maa_k", (D,), "mix"), (p + "ffn.time_maa_r", (D,), "mix")]
        else:
            out += [(p + "ffn.x_k", (D, 1, 1), "mix")]
        out += [(p + "ffn.key.weight", (D, F), "mat"), (p + "ffn.value.weight", (F, D), "mat")]
        if s.arch != "7":
            out += [(p + "ffn.receptance.weight", (D, D), "mat")]
    out += [("ln_out.weight", (D,), "ln_w"), ("ln_out.bias", (D,), "ln_b"), ("head.weight", (D, V), "head")]
    return out


def _f32_values(rng, kind, n, D):
    sd = 0.02 * np.sqrt(768.0 / D)
    if kind in ("mat", "mat_f32src", "emb", "head", "w2"):
        scale = sd * (4.0 if kind == "emb" else 1.0)
        return (rng.standard_normal(n, dtype=np.float32) * np.float32(scale)).astype(np.float32)
    if kind == "ln_w":
        return (1.0 + 0.01 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)
    if kind == "ln_b":
        return (0.01 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)
    if kind == "mix":
        return rng.random(n, dtype=np.float32)
    if kind == "v4_decay":
        return (-np.exp(rng.uniform(-5.0, 1.0, n))).astype(np.float32)
    if kind == "v4_first":
        return rng.uniform(-1.0, 1.0, n).astype(np.float32)
    if kind == "v5_decay":
        return rng.uniform(0.9, 0.999, n).astype(np.float32)       # stored already as exp(-exp(w))
    if kind == "v5_first":
        return np.exp(rng.uniform(-1.0, 1.0, n)).astype(np.float32)  # stored already exp'ed
    if kind in ("v6_decay", "v7_w0"):
        return rng.uniform(-6.0, -1.0, n).astype(np.float32)
    return (0.1 * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)  # "small"


def _direct_blocks(rng, fmt, n_blocks, D):
    """Random blocks of a quantised format, generated directly (uniform codes, fp16 scale matched to N(0, sd^2))."""
    sd = 0.02 * np.sqrt(768.0 / D)
    bb = BLOCK_BYTES[fmt]
    out = np.empty((n_blocks, bb), dtype=np.uint8)
    levels = {"Q4_0": 16, "Q4_1": 16, "Q5_0": 32, "Q5_1": 32, "Q8_0": 256}[fmt]
    d = np.float16(sd * np.sqrt(12.0) / levels)  # uniform codes over `levels` steps have std = d*levels/sqrt(12)
    dv = (d * (1.0 + 0.25 * rng.random(n_blocks, dtype=np.float32))).astype(np.float16)
    out[:, 0:2] = dv.view(np.uint8).reshape(n_blocks, 2)
    pos = 2
    if fmt in ("Q4_1", "Q5_1"):
        m = (-(dv.astype(np.float32)) * (levels / 2)).astype(np.float16)
        out[:, 2:4] = m.view(np.uint8).reshape(n_blocks, 2)
        pos = 4
    payload = bb - pos
    chunk = 1 << 22
    for i in range(0, n_blocks, chunk):
        j = min(n_blocks, i + chunk)
        out[i:j, pos:] = np.frombuffer(rng.bytes((j - i) * payload), dtype=np.uint8).reshape(j - i, payload)
    return out


def write_model(path: str, spec: ModelSpec, dtype: str = "FP32", seed: int = 42, limit_layers: Optional[int] = None) -> dict:
    """Writes a model file. dtype: FP32 | FP16 | Q4_0 | Q4_1 | Q5_0 | Q5_1 | Q8_0.

    FP32: everything f32. FP16: matrices f16 except the ones the converter keeps f32 (keys containing '.time_' and the
    v7 vectors). Qx_y: as a file quantised from FP16 -- quantisable 2-D matrices as random blocks, emb/head/v7 low-rank F16.
    Returns {"bytes": file size, "params": parameter count}.
    """
    rng = np.random.default_rng(seed)
    D = spec.n_embed
    n_layer = spec.n_layer if limit_layers is None else limit_layers
    spec = ModelSpec(**{**spec.__dict__, "n_layer": n_layer})
    quant = dtype in BLOCK_BYTES
    params = 0
    with open(path, "wb") as f:
        f.write(struct.pack("<6I", 0x67676D66, 101, spec.n_vocab, D, n_layer, TYPE_IDS[dtype]))
        for key, dims, kind in _tensor_list(spec):
            n = int(np.prod(dims))
            params += n
            is_matrix = len(dims) == 2 and kind in ("mat", "mat_f32src", "emb", "head")
            quantisable = is_matrix and key not in ("emb.weight", "head.weight") and not any(t in key for t in NEVER_QUANT)
            if quant and quantisable:
                ttype, payload = dtype, _direct_blocks(rng, dtype, n // 32, D)
            elif dtype != "FP32" and is_matrix and kind != "mat_f32src":
                ttype = "FP16"
                if n >= (1 << 26):  # huge emb/head: cheap integer-valued f16 instead of a 268M-sample normal draw
                    sd = 0.02 * np.sqrt(768.0 / D) * (4.0 if kind == "emb" else 1.0)
                    payload = (rng.integers(-2047, 2048, size=n, dtype=np.int16).astype(np.float16) * np.float16(sd * np.sqrt(3.0) / 2047.0))
                else:
                    payload = _f32_values(rng, kind, n, D).astype(np.float16)
            else:
                ttype, payload = "FP32", _f32_values(rng, kind, n, D)
            kb = key.encode()
            f.write(struct.pack("<3I", len(dims), len(kb), TYPE_IDS[ttype]))
            f.write(struct.pack(f"<{len(dims)}I", *dims))
            f.write(kb)
            f.write(payload.tobytes())
        size = f.tell()
    return {"bytes": size, "params": params}
