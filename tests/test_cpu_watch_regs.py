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


def _check_text(tmp_path, text):
    spec = importlib.util.spec_from_file_location("check_watch_regs", os.path.join(ROOT, "tools", "check_watch_regs.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    f = tmp_path / "k.s"
    f.write_text(text)
    return m.check(str(f))


def test_the_checker_itself_sees_the_three_ways_the_register_can_be_lost(tmp_path):
    """the hazards the static check exists for, as hand-written assembly: (a) clean, (b) the register written while reads are in flight,
    (c) the value copied (END names another register), (d) END straight behind the watch without a full wait (what the first version of the
    worker-side y sweep did: found on the GPU as garbage trace stamps, not by the parity tests)"""
    clean = """_Zk:
	s_waitcnt vmcnt(0)
	buffer_load_dword v7, v3, s[4:7], 0 offen sc1
	; WATCH4_BEGIN v7
	buffer_load_dwordx4 v[10:13], v3, s[4:7], 0 offen sc1
	s_waitcnt vmcnt(0)
	v_add_u32_e32 v9, v10, v11
	; WATCH4_END v7
	s_endpgm
"""
    sites, bad = _check_text(tmp_path, clean)
    assert sites == 1 and not bad
    written = clean.replace("\tv_add_u32_e32 v9, v10, v11", "\tv_add_u32_e32 v7, v10, v11")
    assert _check_text(tmp_path, written)[1]
    copied = clean.replace("; WATCH4_END v7", "; WATCH4_END v8")
    assert _check_text(tmp_path, copied)[1]
    unwaited = clean.replace("\ts_waitcnt vmcnt(0)\n\tv_add", "\tv_add")
    assert _check_text(tmp_path, unwaited)[1]
