"""No instantiation of the sequence-mode kernels may use scratch memory: on this stack a kernel with a private segment costs ~45 us per
LAUNCH (measured when a first version of k_mmq_mfma spilled), more than the kernel itself. The check compiles prefill.hip for gfx950
(device side only, no GPU needed) and reads the kernel metadata hipcc emits."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_sequence_kernels_have_no_private_segment(tmp_path):
    src = os.path.join(ROOT, "rwkv.cpp_amd", "csrc", "prefill.hip")
    out = str(tmp_path / "prefill.s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DRWKV_SHARED", "-DRWKV_BUILD",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rwkv.cpp_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    meta = text[text.index("amdhsa.kernels"):]
    seen = 0
    for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", meta, re.S):
        name, private, vgprs, spills = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
        if any(k in name for k in ("k_mmq_mfma", "k_mmq_combine", "k_wkv6_seq", "k_quant_act_tiles", "k_v6_mix2_seq", "k_mix_seq_q")):
            seen += 1
            assert private == 0 and spills == 0, (name, private, vgprs, spills)
            if "k_mmq_mfma" in name:
                assert vgprs <= 256, (name, vgprs)      # two waves per SIMD
    assert seen >= 12, seen
