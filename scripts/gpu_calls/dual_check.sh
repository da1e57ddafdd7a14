mkdir -p gpurun_out/r01r
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "dual or gemm" --timeout=300 > gpurun_out/r01r/pytest_dual.log 2>&1
echo "exit $?" >> gpurun_out/r01r/pytest_dual.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gen 8 --no-roofline-probe > gpurun_out/r01r/bench_dual.json 2> gpurun_out/r01r/bench_dual.err
EXL_GEMM_NO_DUAL=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gen 8 --no-roofline-probe > gpurun_out/r01r/bench_nodual.json 2> gpurun_out/r01r/bench_nodual.err
tail -15 gpurun_out/r01r/pytest_dual.log
cat gpurun_out/r01r/bench_dual.json gpurun_out/r01r/bench_nodual.json
