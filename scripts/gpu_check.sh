#!/bin/bash
# Runs on the GPU box through gpurun: tests, smoke, bench, rocprof summary. Everything lands in gpurun_out/.
# usage: scripts/gpu_check.sh [tag] [what...]   what in: tests smoke bench prof pmc
set -u
TAG=${1:-r01}; shift || true
WHAT=${*:-tests smoke bench prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/gpu.txt
nproc >> $OUT/gpu.txt
for w in $WHAT; do
case $w in
tests)
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout=600 > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu.log ;;
smoke)
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log ;;
bench)
  timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err ;;
prof)
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- \
      python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
  echo "prof exit $?" >> $OUT/prof.err
  find $OUT/prof -name "*kernel_stats*" | head -3 >> $OUT/prof.err
  # keep only the small summaries (traces can be huge)
  find $OUT/prof -type f ! -name "*stats*" -size +8M -delete ;;
pmc)
  ( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --gen 16 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc.err )
  echo "pmc exit $?" >> $OUT/pmc.err ;;
esac
done
tail -5 $OUT/*.log 2>/dev/null
cat $OUT/bench.json 2>/dev/null | head -c 3000
