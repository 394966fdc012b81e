#!/usr/bin/env bash
# Imports the reference's known-answer fixtures (binary data, no source code) into tests/golden/.
# Provenance: /root/reference/tests/{tiny-rwkv-*.bin, expected-logits-*.bin} (RWKV/rwkv.cpp @ 2025-02-19).
# These are the models + golden logits the reference's own tests use
# (tests/test_tiny_rwkv.c, tests/test_quantization_format_compatibility.c); /root/reference does not
# exist on the GPU box, so the fixtures travel with the repo.
set -euo pipefail
SRC="${1:-/root/reference/tests}"
DST="$(cd "$(dirname "$0")" && pwd)"
cp -v "$SRC"/tiny-rwkv-*.bin "$SRC"/expected-logits-*.bin "$DST"/
( cd "$DST" && sha256sum tiny-rwkv-*.bin expected-logits-*.bin > SHA256SUMS )
