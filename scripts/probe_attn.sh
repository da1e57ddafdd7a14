# Builds a probe variant of the library (EXL_ATTN_PROBE) next to the product one and runs the C-ABI decoder driver on it.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/probe
make -C exllama_amd/csrc OBJDIR=../../build/probe/obj TARGET=../../build/probe/libexl_amd.so CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -fno-gpu-rdc -DNDEBUG -DEXL_ATTN_PROBE" > /dev/null
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -DEXL_ATTN_PROBE scripts/bench_decoder.cpp -Iinclude -Lbuild/probe -lexl_amd -Wl,-rpath,'$ORIGIN' -o build/probe/bench_decoder
