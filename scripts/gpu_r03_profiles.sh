#!/bin/bash
# round-3 profiles: rocprofv3 kernel stats of the bench command + HBM traffic (PMC) of the decode kernels at ONE context
# (round 2 averaged the attention kernel's counters over the context-2048 and the context-4 pass of bench_decoder)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03prof
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
echo "prof exit $?" >> $OUT/prof.err
find $OUT/prof -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -type f ! -name "*stats*" -size +4M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- $GRAFT_REPO_ROOT/build/bench_decoder 8 2048 128 1 > /dev/null 2> $OUT/pmc_$c.err
done
python - <<PY
import csv, json, collections, glob
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "dec_" not in n or r["Counter_Name"] != c: continue
            res[n][c].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over build/bench_decoder 8 2048 128 1 (context 2048 only; scripts/gpu_r03_profiles.sh), MI355X, Llama-7B shapes g128, round 3",
       "units": "counters are KiB per dispatch; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
       "kernels": {}}
for n, v in res.items():
    f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])); w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    out["kernels"][n[:100]] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024), "dispatches": len(v["FETCH_SIZE"])}
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3500])
PY
find $OUT -name "*.csv" -size +2M -delete
head -14 $OUT/kernel_stats.csv | cut -c1-170
