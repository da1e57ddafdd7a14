#!/bin/bash
# round 5, call 9: the full-depth perplexity case through the committed oracle log-likelihoods (the sampled text must be the golden one),
# and once more with the oracle forced (EXL_PPL_ORACLE=1) is NOT repeated here: call 8 ran that path (302 s)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
EXL_TOL_STATS=$OUT/tol_stats.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "perplexity or adapter_on_an_act_order or path_report" --durations=5 > $OUT/t.log 2>&1; echo "rc $?" >> $OUT/t.log
tail -n 12 $OUT/t.log; grep -o '"oracle_source": "[^"]*"' $OUT/tol_stats.jsonl
