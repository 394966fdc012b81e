"""The four-deep watch of persist_v47.hip (persist.h watch4) leaves up to three reads in flight into ONE register when it ends; the register must
stay untouched until they have landed (the sweep behind the watch waits for its own, younger reads; watch_drain waits itself). That is a
property of the generated code, not of the source: this test compiles the kernel's device assembly (no GPU needed, ~30 s) and runs the
static check of tools/check_watch_regs.py over it -- a compiler that copies the register, or an END placed straight behind the watch, fails
here instead of corrupting a register once in a while on the GPU."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_nothing_writes_the_watch_register_while_its_reads_are_in_flight(tmp_path):
    src = os.path.join(ROOT, "rwkv.cpp_amd", "csrc", "persist_v47.hip")
    out = str(tmp_path / "persist_v47.s")
    cmd = [HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DRWKV_SHARED", "-DRWKV_BUILD",
           "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rwkv.cpp_amd", "csrc"), "-Wno-unused-function", "-Wno-unused-variable",
           src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    spec = importlib.util.spec_from_file_location("check_watch_regs", os.path.join(ROOT, "tools", "check_watch_regs.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sites, bad = m.check(out)
    assert sites >= 40, sites            # (five watches per RWKV-4 kernel, six per RWKV-7 kernel, twenty kernels; fewer = the markers are gone)
    assert not bad, bad[:5]
    # no register spills in any variant of the kernel: a spilled watch register is the same hazard, and a spill in a phase is a microsecond
    text = open(out).read()
    assert "scratch_store" not in text and "scratch_load" not in text
