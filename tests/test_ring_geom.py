"""The per-workgroup weight stream layout of the ring kernel (csrc/ring_geom.h) is host-and-device code: its invariants are checked
on the CPU by a small C++ program (tests/cpp/ring_geom_check.cpp) -- every row packed exactly once, records tile a layer block,
one consumer wave per record, cursors in stream order, head records cover the vocabulary."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ring_geometry_invariants(tmp_path):
    exe = str(tmp_path / "ring_geom_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "rwkv.cpp_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "ring_geom_check.cpp"), "-o", exe],
                   check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ring geometry OK" in r.stdout
