#!/bin/bash
# Builds measurement variants of the library next to the product one (cross-compiles without a GPU) and the C-ABI decoder driver:
#   build/ring_probe  -DEXL_RING_PROBE   phase stamps inside dec_ring_kernel (printed by bench_decoder)
#   build/ring_ablate -DEXL_RING_ABLATE  the ring's loads without the dequantisation + MFMA
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -fno-gpu-rdc -DNDEBUG"
for v in probe:-DEXL_RING_PROBE ablate:-DEXL_RING_ABLATE; do
    name=${v%%:*}; def=${v#*:}
    mkdir -p build/ring_$name
    make -C exllama_amd/csrc -j8 OBJDIR=../../build/ring_$name/obj TARGET=../../build/ring_$name/libexl_amd.so CXXFLAGS="$FLAGS $def" > /dev/null
    /opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 $def scripts/bench_decoder.cpp -Iinclude -Lbuild/ring_$name -lexl_amd -Wl,-rpath,'$ORIGIN' -o build/ring_$name/bench_decoder
done
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 scripts/bench_decoder.cpp -Iinclude -Lexllama_amd -lexl_amd -Wl,-rpath,'$ORIGIN/../exllama_amd' -o build/bench_decoder
