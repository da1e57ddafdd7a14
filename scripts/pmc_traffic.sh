# HBM traffic of the decode kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (guide: MI355X_MICROARCH.md section HBM)
mkdir -p gpurun_out/mb
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mb/pmc_fetch -o p -- $GRAFT_REPO_ROOT/build/bench_decoder 8 2048 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mb/pmc_write -o p -- $GRAFT_REPO_ROOT/build/bench_decoder 8 2048 > /dev/null 2>&1
