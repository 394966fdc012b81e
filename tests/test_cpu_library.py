"""CPU-only checks of the product library: it builds for gfx950, exports every symbol the headers declare, its host-side
pieces (file parsing errors, the quantiser) behave like the reference, and it fails loudly without a GPU."""
import ctypes
import filecmp
import os
import re
import struct

import numpy as np
import pytest

import oracle_lib as O
import reference_constants as R
from gpu_lib import ROOT, library, pkg


def _declared_symbols():
    names = []
    for h in ("rwkv.h", "rwkv_mi355x.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"RWKV_API[^;(]*?\b(rwkv_\w+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = library().library
    names = _declared_symbols()
    assert len(names) >= 19 + 8
    for required in ("rwkv_init_from_file", "rwkv_eval", "rwkv_eval_sequence", "rwkv_eval_sequence_in_chunks", "rwkv_clone_context",
                     "rwkv_get_state_buffer_element_count", "rwkv_get_logits_buffer_element_count", "rwkv_quantize_model_file"):
        assert required in names
    for n in names:
        assert hasattr(lib, n), n


def _dynamic_symbols(path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_product_library_exports_the_declared_surface_and_nothing_else():
    """A drop-in exports the reference's rwkv.h symbols + the declared rwkv_mi_* extensions: no kernel host stubs (_ZN6rwkvmi...), no
    engine internals, no test entry points (csrc/rwkv.map; round-3 review). The test entry points live in librwkv_testhooks.so."""
    library()
    declared = set(_declared_symbols())
    exported = _dynamic_symbols(pkg.LIB_PATH)
    assert exported, "nm found no dynamic symbols"
    assert set(exported) == declared, (sorted(set(exported) - declared), sorted(declared - set(exported)))
    assert not any(n.startswith("rwkv_mi_test_") for n in exported)
    text = open(os.path.join(ROOT, "include", "rwkv_testhooks.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    hooks = set(re.findall(r"RWKV_API[^;(]*?\b(rwkv_\w+)\s*\(", text))
    assert len(hooks) == 9
    assert set(_dynamic_symbols(pkg.HOOKS_LIB_PATH)) == declared | hooks


def test_library_contains_gfx950_code_objects():
    blob = open(pkg.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_mvq_t1" in blob


def test_system_info_string_has_reference_keys():
    s = library().rwkv_get_system_info_string()
    for key in ("AVX=", "AVX2=", "AVX512=", "FMA=", "NEON=", "ARM_FMA=", "F16C=", "FP16_VA=", "WASM_SIMD=", "SSE3=", "VSX="):
        assert key in s


@pytest.mark.parametrize("version", R.HAVE_FP32_FP16)
def test_product_quantiser_is_byte_exact(golden_dir, tmp_path, version):
    lib = library()
    lib.rwkv_set_print_errors(None, False)
    for fmt in R.QUANT_FORMATS:
        for source in ("FP32", "FP16"):
            out, ref = str(tmp_path / "o.bin"), str(tmp_path / "r.bin")
            lib.rwkv_quantize_model_file(R.fixture_path(golden_dir, version, source), out, fmt)
            O.quantize_file(R.fixture_path(golden_dir, version, source), ref, fmt)
            assert filecmp.cmp(out, ref, shallow=False), (version, source, fmt)
            if source == "FP32" and fmt in ("Q5_0", "Q5_1"):
                assert filecmp.cmp(out, R.fixture_path(golden_dir, version, fmt), shallow=False)
    lib.rwkv_set_print_errors(None, True)


def test_quantiser_rejects_bad_arguments(golden_dir, tmp_path):
    lib = library()
    L = lib.library
    lib.rwkv_set_print_errors(None, False)
    src = R.fixture_path(golden_dir, "5v2-730K", "FP32").encode()
    assert not L.rwkv_quantize_model_file(src, str(tmp_path / "x.bin").encode(), b"Q4_3")
    assert lib.rwkv_get_last_error(None) == (1 << 8) | 8          # ARGS | DATA_TYPE (rwkv_quantize.inc:21-26)
    assert not L.rwkv_quantize_model_file(b"/nonexistent.bin", str(tmp_path / "x.bin").encode(), b"Q4_0")
    assert lib.rwkv_get_last_error(None) == (2 << 8) | 2          # FILE | FILE_OPEN
    q = R.fixture_path(golden_dir, "5v2-730K", "Q5_0").encode()
    assert not L.rwkv_quantize_model_file(q, str(tmp_path / "x.bin").encode(), b"Q4_0")
    assert lib.rwkv_get_last_error(None) & (2 << 8)               # already quantised input: FILE
    lib.rwkv_set_print_errors(None, True)


def test_load_errors_are_reported_like_the_reference(golden_dir, tmp_path):
    lib = library()
    L = lib.library
    lib.rwkv_set_print_errors(None, False)
    assert not L.rwkv_init_from_file(b"/nonexistent/model.bin", 1, 0)
    assert lib.rwkv_get_last_error(None) == (2 << 8) | 2          # FILE | FILE_OPEN (rwkv_model_loading.inc:295)
    assert lib.rwkv_get_last_error(None) == 0                     # cleared by the read
    bad = tmp_path / "bad.bin"
    bad.write_bytes(struct.pack("<6I", 0x12345678, 101, 256, 64, 12, 0))
    assert not L.rwkv_init_from_file(str(bad).encode(), 1, 0)
    assert lib.rwkv_get_last_error(None) == (2 << 8) | 6          # FILE | FILE_MAGIC
    bad.write_bytes(struct.pack("<6I", 0x67676D66, 99, 256, 64, 12, 0))
    assert not L.rwkv_init_from_file(str(bad).encode(), 1, 0)
    assert lib.rwkv_get_last_error(None) == (2 << 8) | 7          # FILE | FILE_VERSION
    bad.write_bytes(struct.pack("<6I", 0x67676D66, 101, 256, 64, 12, 4))
    assert not L.rwkv_init_from_file(str(bad).encode(), 1, 0)
    assert lib.rwkv_get_last_error(None) == (2 << 8) | 8          # FILE | DATA_TYPE (removed format Q4_1_O)
    bad.write_bytes(struct.pack("<6I", 0x67676D66, 100, 256, 64, 12, 2))
    assert not L.rwkv_init_from_file(str(bad).encode(), 1, 0)
    assert lib.rwkv_get_last_error(None) == (2 << 8) | 8          # quantised file with version 100
    lib.rwkv_set_print_errors(None, True)


def test_no_cpu_fallback_without_a_gpu(golden_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    lib = library()
    lib.rwkv_set_print_errors(None, False)
    ptr = lib.library.rwkv_init_from_file(R.fixture_path(golden_dir, "5v2-730K", "FP32").encode(), 1, 0)
    err = lib.rwkv_get_last_error(None)
    lib.rwkv_set_print_errors(None, True)
    assert not ptr, "the product must not run without its HIP device"
    assert err & (6 << 8) and (err & 0xFF) == 9                   # CTX | UNSUPPORTED


def test_synthetic_writer_matches_reference_converter_layout(tmp_path):
    # python/convert_pytorch_to_ggml.test.py:21-44 spells the container out byte by byte; same contract here
    from rwkv_cpp_amd import synth
    p = str(tmp_path / "s.bin")
    synth.write_model(p, synth.CONFIGS["test-v6"], "FP16", seed=0)
    raw = open(p, "rb").read()
    magic, ver, nv, ne, nl, dt = struct.unpack("<6I", raw[:24])
    assert (magic, ver, nv, ne, nl, dt) == (0x67676D66, 101, 512, 256, 2, 1)
    dc, kl, ty, d0, d1 = struct.unpack("<5I", raw[24:44])
    assert (dc, kl, ty, d0, d1) == (2, len("emb.weight"), 1, 256, 512) and raw[44:54] == b"emb.weight"
    om = O.OracleModel(p)
    assert (om.arch_major, om.head_count, om.head_size, om.ffn_size) == (6, 4, 64, 896)
