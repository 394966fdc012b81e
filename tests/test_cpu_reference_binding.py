"""The reference's own ctypes wrapper (python/rwkv_cpp/rwkv_cpp_shared_library.py:49-107) bound, unmodified, to librwkv.so.

It is imported BY PATH from /root/reference (never copied); that mount only exists in the dev container, where there is no GPU,
so this covers what can be checked without one: every symbol / signature the wrapper declares resolves in librwkv.so, the
non-compute entry points answer, and failures surface the way the wrapper expects (NULL -> ValueError). On the GPU box the
compute side of the boundary is exercised by the reference's C programs (tests/test_gpu_reference_programs.py) and by
rwkv.cpp_amd/rwkv_cpp.py, which mirrors this wrapper method for method."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PY = "/root/reference/python/rwkv_cpp"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_PY), reason="reference mount absent (GPU box): nothing to import")


def _ref_module(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF_PY, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def ref_lib():
    import __graft_entry__ as graft
    pkg = graft.load_package()
    pkg.build_library()
    return _ref_module("rwkv_cpp_shared_library").RWKVSharedLibrary(pkg.LIB_PATH)


def test_reference_wrapper_binds_every_symbol(ref_lib):
    # the constructor touched every rwkv_* function it declares argtypes for; a missing export raises AttributeError there
    info = ref_lib.rwkv_get_system_info_string()
    assert "AVX=" in info and "VSX=" in info


def test_reference_wrapper_error_paths(ref_lib, tmp_path):
    with pytest.raises(ValueError):
        ref_lib.rwkv_init_from_file(str(tmp_path / "missing.bin"), 1, 0)
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\0" * 64)
    with pytest.raises(ValueError):
        ref_lib.rwkv_init_from_file(str(bad), 1, 0)
    with pytest.raises(ValueError):
        ref_lib.rwkv_quantize_model_file(str(bad), str(tmp_path / "out.bin"), "Q5_1")


def test_reference_wrapper_quantizes_a_fixture(ref_lib, tmp_path, golden_dir):
    src = os.path.join(golden_dir, "tiny-rwkv-5v2-730K-FP32.bin")
    dst = tmp_path / "q.bin"
    ref_lib.rwkv_quantize_model_file(src, str(dst), "Q5_1")
    assert dst.read_bytes() == open(os.path.join(golden_dir, "tiny-rwkv-5v2-730K-Q5_1.bin"), "rb").read()
