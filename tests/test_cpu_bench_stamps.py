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


def _copy_sources(tmp_path):
    dst = tmp_path / "rwkv.cpp_amd" / "csrc"
    dst.mkdir(parents=True)
    for f in ("prefill.hip", "kdev.h"):
        shutil.copy(os.path.join(ROOT, "rwkv.cpp_amd", "csrc", f), dst / f)
    return dst


def test_prefill_stamp_covers_the_gemm_and_not_the_wkv7_section(tmp_path):
    dst = _copy_sources(tmp_path)
    b = _bench(str(tmp_path))
    base = b.prefill_source_stamp()
    assert base == _bench().prefill_source_stamp()
    text = (dst / "prefill.hip").read_text()
    a = text.index("// WKV-7 over a sequence")
    z = text.index("bool launch_wkv7_seq")
    assert a < z
    # a change inside the WKV-7 section leaves the stamp alone ...
    (dst / "prefill.hip").write_text(text[:a + 30] + " (edited)" + text[a + 30:])
    assert b.prefill_source_stamp() == base
    # ... a change anywhere else (here: inside k_mmq_mfma) or in kdev.h does not
    g = text.index("void k_mmq_mfma(MmqArgs A)")
    (dst / "prefill.hip").write_text(text[:g] + "/* edited */ " + text[g:])
    assert b.prefill_source_stamp() != base
    (dst / "prefill.hip").write_text(text)
    assert b.prefill_source_stamp() == base
    with open(dst / "kdev.h", "a") as f:
        f.write("\n// edited\n")
    assert b.prefill_source_stamp() != base


def test_committed_quotes_belong_to_the_committed_sources():
    """The quotes in profiles/ were taken on the sources in the tree: a commit that edits a kernel re-takes its PMC pass (or bench.py prints
    the quote as stale, which the docs then have to say)."""
    b = _bench()
    mf = json.load(open(os.path.join(ROOT, "profiles", "pmc_mfma.json")))
    assert mf["rwkv6-1b6:Q4_0:prefill"]["prefill_source_stamp"] == b.prefill_source_stamp()
    tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    stamps = {e.get("kernel_source_stamp") for e in tr.values() if isinstance(e, dict)}
    assert b.kernel_source_stamp(2) in stamps, (stamps, b.kernel_source_stamp(2))
