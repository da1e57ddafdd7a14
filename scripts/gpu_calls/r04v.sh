#!/bin/bash
# round 4: L2 <-> fabric traffic of the 8-wave prompt attention kernel (PMC, one counter per pass), stand-alone driver
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04v
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 60 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- $GRAFT_REPO_ROOT/build/bench_flash 2048 32 32 0 4 > /dev/null 2> $OUT/pmc_$c.err
done
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(lambda: [0.0, set()])
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            tot[k][0] += float(r["Counter_Value"]); tot[k][1].add(r["Dispatch_Id"])
    for k, (v, d) in tot.items():
        print(c, k, "sum", v, "dispatches", len(d), "per dispatch", v / max(1, len(d)))
PY
find $OUT -type f -size +2M -delete
