"""bench.py quotes rocprofv3 PMC passes (HBM traffic of the decode kernel, matrix-pipe busy cycles of the sequence GEMM) only for the build
they were taken on: the quote files under profiles/ carry a hash of the kernel's sources and bench.py drops a quote whose hash differs.
These tests pin what the hashes cover (no GPU: bench.py is imported, nothing is launched)."""
import importlib.util
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(root=None):
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    if root is not None:
        b.ROOT = root
    return b


GEMM_SOURCES = ("prefill_fast.hip", "prefill.hip", "prefill_mm.h", "kdev.h")


def _copy_sources(tmp_path):
    dst = tmp_path / "rwkv.cpp_amd" / "csrc"
    dst.mkdir(parents=True)
    for f in GEMM_SOURCES:
        shutil.copy(os.path.join(ROOT, "rwkv.cpp_amd", "csrc", f), dst / f)
    return dst


def test_prefill_stamp_covers_the_complete_sources_of_both_gemm_arms(tmp_path):
    """Round 4 hashed prefill.hip with the WKV-7 section cut out, so an edit there -- a shared helper, the LDS budget -- was invisible to the
    staleness check (advisor, round 4). The stamp now covers every byte of the four translation-unit sources of the sequence GEMMs."""
    dst = _copy_sources(tmp_path)
    b = _bench(str(tmp_path))
    base = b.prefill_source_stamp()
    assert base == _bench().prefill_source_stamp()
    for f in GEMM_SOURCES:
        text = (dst / f).read_text()
        (dst / f).write_text(text + "\n// edited\n")
        assert b.prefill_source_stamp() != base, f
        (dst / f).write_text(text)
        assert b.prefill_source_stamp() == base
    text = (dst / "prefill.hip").read_text()
    a = text.index("// WKV-7 over a sequence")
    (dst / "prefill.hip").write_text(text[:a + 30] + " (edited)" + text[a + 30:])
    assert b.prefill_source_stamp() != base          # (the section round 4 left out)


def test_a_quote_is_either_of_this_build_or_reported_stale():
    """bench.py never repeats a quote from another build: for the sources in the tree each committed quote either carries their hash
    (then it is quoted) or is reported as stale (then the line says so). Both are honest; quoting with a foreign hash is not."""
    import types
    b = _bench()
    mf = json.load(open(os.path.join(ROOT, "profiles", "pmc_mfma.json")))["rwkv6-1b6:Q4_0:prefill"]
    got = b.mfma_busy(types.SimpleNamespace(config="rwkv6-1b6", dtype="Q4_0"))
    if mf["prefill_source_stamp"] == b.prefill_source_stamp():
        assert got == mf
    else:
        assert set(got) == {"stale"}
    traffic, source = b.pmc_traffic(2, types.SimpleNamespace(config="rwkv6-7b", dtype="Q4_0"), kind=2)
    tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    fresh = any(isinstance(e, dict) and e.get("kernel_source_stamp") == b.kernel_source_stamp(2) for e in tr.values())
    if fresh:
        assert traffic is not None and traffic > 1e9
    else:
        assert traffic is None and source and source.startswith("stale")
