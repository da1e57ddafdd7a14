mkdir -p gpurun_out/r01l
for bpc in 2 3 4; do
  EXL_DEC_BLOCKS_PER_CU=$bpc python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline-probe > gpurun_out/r01l/bench_bpc$bpc.json 2> gpurun_out/r01l/bench_bpc$bpc.err
done
cd /tmp && export TMPDIR=/tmp
EXL_DEC_BLOCKS_PER_CU=2 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01l/prof2 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline-probe > /dev/null 2>&1
EXL_DEC_BLOCKS_PER_CU=4 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01l/prof4 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline-probe > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r01l -type f ! -name "*stats*" -size +4M -delete
