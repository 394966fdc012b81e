"""Known-answer constants recorded by the reference's own tests (RWKV/rwkv.cpp @ 2025-02-19).

Signed sums of (logits - expected_logits) after the prompt '"in' (tokens 34, 105, 110); a run passes when
abs(sum) <= 1.05 * abs(recorded)  (tests/logit_difference_validator.inc:68,83).
"""
PROMPT = [34, 105, 110]  # '"', 'i', 'n'   (tests/logit_difference_validator.inc:49-51)
TOLERANCE_FACTOR = 1.05

VERSIONS = ["4v0-660K", "5v1-730K", "5v2-730K", "6v0-3m", "7v0-834K"]
QUANT_FORMATS = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0"]

# tests/test_tiny_rwkv.c:38-54
FULL = {
    "4v0-660K": {"FP32": +0.001000, "FP16": -0.013652},
    "5v1-730K": {"FP32": +0.001000, "FP16": -0.289921},
    "5v2-730K": {"FP32": +0.001000, "FP16": +0.455912},
    "6v0-3m": {"FP32": +0.001000, "FP16": -0.416620},
    "7v0-834K": {"FP32": +0.001000, "FP16": +0.005766},
}

# tests/test_tiny_rwkv.c:70-101  (quantised from the FP32 file)
FROM_FP32 = {
    "4v0-660K": [-0.160030, -0.547409, -0.170404, +0.278034, +0.076282],
    "5v1-730K": [+117.932594, -26.712271, -163.439407, -18.017435, +0.585238],
    "5v2-730K": [+35.271305, +67.015076, +25.273308, +48.068733, -9.441034],
    "6v0-3m": [-7.588121, +21.939022, -27.332073, +3.576909, -9.539596],
    "7v0-834K": [+0.136785, +0.002614, -0.063645, -0.064663, +0.011924],
}

# tests/test_tiny_rwkv.c:103-134  (quantised from the FP16 file)
FROM_FP16 = {
    "4v0-660K": [+0.154614, -0.539827, -0.180142, +0.294953, +0.077226],
    "5v1-730K": [+119.471931, -28.245888, -159.870956, -39.708530, -0.962695],
    "5v2-730K": [+34.135971, +65.573822, +21.588751, +29.726818, -7.242277],
    "6v0-3m": [-7.660988, +21.797060, -27.269241, +3.405264, -9.734720],
    "7v0-834K": [+0.136678, -0.005140, -0.064447, -0.063531, +0.010921],
}

# tests/test_quantization_format_compatibility.c:22-35  (shipped Q5_0 / Q5_1 fixtures)
SHIPPED_Q5 = {
    "4v0-660K": {"Q5_0": -0.170404, "Q5_1": +0.278034},
    "5v1-730K": {"Q5_0": -163.439407, "Q5_1": -18.017435},
    "5v2-730K": {"Q5_0": +25.273308, "Q5_1": +48.068733},
    "6v0-3m": {"Q5_0": -21.151785, "Q5_1": +3.576909},
}

# Fixtures absent from the reference mount (.MISSING_LARGE_BLOBS): 6v0 FP32 / FP16.
HAVE_FP32_FP16 = ["4v0-660K", "5v1-730K", "5v2-730K", "7v0-834K"]

# Recorded values that the ggml-emulating arithmetic reproduces to ~1e-5 (recent records); used as tight
# known-answer tests for the CPU oracle: (version, source, format) -> recorded
TIGHT_KAT = {
    ("7v0-834K", "FP32", "Q4_0"): +0.136785, ("7v0-834K", "FP32", "Q4_1"): +0.002614,
    ("7v0-834K", "FP32", "Q5_0"): -0.063645, ("7v0-834K", "FP32", "Q8_0"): +0.011924,
    ("7v0-834K", "FP16", "Q4_0"): +0.136678, ("7v0-834K", "FP16", "Q4_1"): -0.005140,
    ("7v0-834K", "FP16", "Q5_0"): -0.064447, ("7v0-834K", "FP16", "Q5_1"): -0.063531,
    ("7v0-834K", "FP16", "Q8_0"): +0.010921,
    ("4v0-660K", "FP32", "Q8_0"): +0.076282,
}


def fixture_path(golden_dir, version, fmt):
    import os
    return os.path.join(golden_dir, f"tiny-rwkv-{version}-{fmt}.bin")


def expected_logits(golden_dir, version):
    import os
    import numpy as np
    return np.fromfile(os.path.join(golden_dir, f"expected-logits-{version}.bin"), dtype=np.float32)


# The opt-in sequence arms (RWKV_MI_SEQ_Q=fast RWKV_MI_SEQ_F16=mfma) against the default ones on the WHOLE benchmarked model, recorded on an MI355X
# by `bench.py --mode prefill` (round 6; synthetic weights of the named geometry, seed 42, the bench's prompt). The analogue of the reference's
# criterion for quantised formats (tests/logit_difference_validator.inc:68,83): (config, dtype, prompt tokens) ->
# sum over the vocabulary of (logits_fast - logits_default) after the pass; a later run must stay inside 1.05 x |recorded|.
FAST_ARM_LOGIT_DIFFERENCE_SUM = {
    ("rwkv6-1b6", "Q4_0", 1024): -0.7023004796355963,
    ("rwkv7-2b9", "Q5_1", 1024): +0.7769884578883648,
}
# ... and whether the 64 greedy tokens behind the pass were the default arms' when that was recorded. RWKV-6 1.6B: yes. RWKV-7 2.9B: NO -- with
# random weights its logits are nearly flat (the arms differ by 0.04 at most) and the continuation parts ways: the opt-in arms are not offered as
# equivalent there, which is one reason they are not the default.
FAST_ARM_GREEDY_CONTINUATION_EQUAL = {
    ("rwkv6-1b6", "Q4_0", 1024): True,
    ("rwkv7-2b9", "Q5_1", 1024): False,
}
