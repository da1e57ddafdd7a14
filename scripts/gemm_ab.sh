mkdir -p gpurun_out/mb
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "gemm or dual or threshold or lora" --timeout=300 > gpurun_out/mb/pytest_gemm.log 2>&1
tail -5 gpurun_out/mb/pytest_gemm.log
echo "--- default (mid-step barrier)"; timeout 120 build/bench_gemm 2048 20
echo "--- variant 4 (mid-step)"; EXL_GEMM_VARIANT=4 timeout 120 build/bench_gemm 2048 20
echo "--- variant 1 (previous pipelined kernel)"; EXL_GEMM_VARIANT=1 timeout 120 build/bench_gemm 2048 20
echo "--- M=512 default / variant 1"; timeout 120 build/bench_gemm 512 20; EXL_GEMM_VARIANT=1 timeout 120 build/bench_gemm 512 20
