#!/bin/bash
# TEST INFRASTRUCTURE.  Stages the reference's UNMODIFIED Python model code where a gpurun snapshot can carry it to the GPU
# box: /root/reference/{model,lora,generator}.py -> oracle/_ref/refpy/ (git-ignored build output, like the rest of
# oracle/_ref/; never committed, never imported by exllama_amd/).  tests/test_reference_dropin_gpu.py then runs the
# reference's own ExLlama class against this repository's `cuda_ext` shim on an MI355X -- the drop-in claim of the north star
# ("model.py is a drop-in"), executed rather than asserted.  The test skips when the staged copy is absent (round-end boxes).
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")/.." && pwd)
DST=$HERE/oracle/_ref/refpy
[ -f "$REF/model.py" ] || { echo "no reference checkout at $REF"; exit 0; }
mkdir -p "$DST"
for f in model.py lora.py generator.py; do cp "$REF/$f" "$DST/$f"; done
echo "staged $(ls "$DST" | tr '\n' ' ')-> $DST"
