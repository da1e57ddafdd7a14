#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- builds the REFERENCE's own GPU kernels as a comparator for the oracle.
#
# The reference's exllama_ext/cuda_func/*.cu are CUDA sources.  This recipe translates them with ROCm's stock
# hipify-perl FROM WHERE THEY LIE under /root/reference into a temporary directory OUTSIDE the repository (removed when the
# script ends: no reference source, translated or not, is ever left in the tree), applies the one fix they need on ROCm >= 5.6 (hip_compat.cuh:4-15 re-defines hrcp / h2rcp with
# a __half -> __fp16 conversion that no longer exists; the native hrcp / h2rcp of hip_fp16.h are used instead), and
# compiles them together with oracle/ref_shim.cpp (a plain C ABI over the reference's *_cuda entry points) into
# oracle/_ref/libexl_ref_kernels.so for gfx950.  hipcc cross-compiles: this runs in the build container, the .so travels
# to the GPU box with the gpurun snapshot, and oracle/make_ref_golden.py runs THERE to produce tests/golden/ref_*.npz.
#
# Nothing under exllama_amd/ links, loads or imports this library: it is the checker of the checker (oracle/exl_oracle.py).
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
SRC=$(mktemp -d /tmp/exl_ref_kernels.XXXXXX)
trap 'rm -rf "$SRC"' EXIT
EXT=$REF/exllama_ext
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
HIPIFY=${HIPIFY:-/opt/rocm/bin/hipify-perl}

[ -d "$EXT" ] || { echo "reference sources not found at $EXT (GPU box: use the prebuilt oracle/_ref/*.so)"; exit 0; }
rm -rf "$OUT/src"                                  # translated sources of earlier versions of this recipe
mkdir -p "$OUT" "$SRC/cuda_func" "$SRC/stub/ATen/cuda" "$SRC/obj"

for f in cuda_buffers.cu cuda_buffers.cuh cuda_compat.cuh hip_compat.cuh matrix.cuh tuning.h util.cuh \
         cuda_func/column_remap.cu cuda_func/column_remap.cuh cuda_func/half_matmul.cu cuda_func/half_matmul.cuh \
         cuda_func/q4_attn.cu cuda_func/q4_attn.cuh cuda_func/q4_matmul.cu cuda_func/q4_matmul.cuh \
         cuda_func/q4_matrix.cu cuda_func/q4_matrix.cuh cuda_func/q4_mlp.cu cuda_func/q4_mlp.cuh \
         cuda_func/rms_norm.cu cuda_func/rms_norm.cuh cuda_func/rope.cu cuda_func/rope.cuh; do
    dst=$SRC/${f%.cu}
    case $f in *.cu) dst=$dst.hip ;; *) dst=$SRC/$f ;; esac
    "$HIPIFY" "$EXT/$f" > "$dst" 2>/dev/null
done

# the one fix: drop the ROCm <= 5.5 reciprocal shim (hip_compat.cuh:4-15); everything else in that header stays
python3 - "$SRC/hip_compat.cuh" <<'EOF'
import re, sys
p = sys.argv[1]
s = open(p).read()
s = re.sub(r"// Workaround for a bug in hipamd.*?#define h2rcp __compat_h2rcp\n", "", s, flags=re.S)
open(p, "w").write(s)
EOF

# the reference takes its BLAS handle type from torch's ATen header; the comparator has no torch dependency
# (hipify-perl renames the include to ATen/cuda/HIPContext.h)
cat > "$SRC/stub/ATen/cuda/HIPContext.h" <<'EOF'
#pragma once
#include <hipblas/hipblas.h>
#include <cstdio>
#include <cstdlib>
#define TORCH_CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "reference TORCH_CHECK failed: %s\n", #cond); abort(); } } while (0)
EOF
cp "$SRC/stub/ATen/cuda/HIPContext.h" "$SRC/stub/ATen/cuda/CUDAContext.h"

FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -DUSE_ROCM -D__HIP_PLATFORM_AMD__ -I$SRC/stub -I$SRC -I$SRC/cuda_func -Wno-unused-result -Wno-deprecated-declarations -w"
OBJS=""
for f in cuda_buffers cuda_func/column_remap cuda_func/half_matmul cuda_func/q4_attn cuda_func/q4_matmul \
         cuda_func/q4_matrix cuda_func/q4_mlp cuda_func/rms_norm cuda_func/rope; do
    o=$SRC/obj/$(basename $f).o
    "$HIPCC" $FLAGS -c "$SRC/$f.hip" -o "$o" &
    OBJS="$OBJS $o"
done
"$HIPCC" $FLAGS -x hip -c "$HERE/ref_shim.cpp" -o "$SRC/obj/ref_shim.o" &
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $OBJS "$SRC/obj/ref_shim.o" -L/opt/rocm/lib -lhipblas -o "$OUT/libexl_ref_kernels.so"
echo "built $OUT/libexl_ref_kernels.so"
