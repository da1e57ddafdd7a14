# A/B of the prefill GEMM kernels through the C-ABI driver (build/bench_gemm): default (loader waves) vs the mid-step kernel.
mkdir -p gpurun_out/mb
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "gemm or dual or threshold or lora or qkv" --timeout=300 > gpurun_out/mb/pytest_gemm.log 2>&1
tail -3 gpurun_out/mb/pytest_gemm.log
echo "--- default (8 MFMA waves + 4 loader waves)"; timeout 120 build/bench_gemm 2048 20
echo "--- EXL_GEMM_NO_LOADER_WAVES (mid-step barrier kernel, every wave loads)"; EXL_GEMM_NO_LOADER_WAVES=1 timeout 120 build/bench_gemm 2048 20
echo "--- M=512 (mid-step kernel, 128 x 128 tile)"; timeout 120 build/bench_gemm 512 20
