#!/bin/bash
# TEST INFRASTRUCTURE.  Stages the reference's UNMODIFIED Python model code where a gpurun snapshot can carry it to the GPU
# box: /root/reference/{model,lora,generator}.py -> the archive oracle/_ref/refpy.tgz (git-ignored build output, like the rest
# of oracle/_ref/; never committed, never imported by exllama_amd/; no reference source file lies in the tree, the test unpacks
# the archive into a temporary directory).  tests/test_reference_dropin_gpu.py then runs the
# reference's own ExLlama class against this repository's `cuda_ext` shim on an MI355X -- the drop-in claim of the north star
# ("model.py is a drop-in"), executed rather than asserted.  The test skips when the staged copy is absent (round-end boxes).
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")/.." && pwd)
DST=$HERE/oracle/_ref/refpy.tgz
[ -f "$REF/model.py" ] || { echo "no reference checkout at $REF"; exit 0; }
mkdir -p "$HERE/oracle/_ref"
rm -rf "$HERE/oracle/_ref/refpy"                   # unpacked copies of earlier versions of this script
tar -czf "$DST" -C "$REF" model.py lora.py generator.py
echo "staged model.py lora.py generator.py -> $DST"
