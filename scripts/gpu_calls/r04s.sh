#!/bin/bash
# round 4, last call: every GPU test (4 worker processes: most of the time is the CPU oracle), smoke(), the default bench line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04s
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
EXL_TOL_STATS=$OUT/tol_stats.jsonl timeout 420 python -m pytest tests -q -m gpu -n 4 > $OUT/full_tests.log 2>&1; echo "suite rc $?" >> $OUT/full_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
timeout 200 python bench.py > $OUT/bench_7b_default.json 2> $OUT/bench_7b_default.err
tail -n 8 $OUT/full_tests.log; tail -n 2 $OUT/smoke.log; cut -c1-400 $OUT/bench_7b_default.json
